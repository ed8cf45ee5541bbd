"""Tolerance-mode GEMMs of the prompt pass (kr_moe_set_gemm_mode / KR_GEMM_FAST, krasis_amd/csrc/kr_prefill_h.hip): f16 activation rows x
INT4 / INT8 weights de-quantized in registers on the f16 matrix cores, f32 accumulation over the whole k range -- the dataflow of the reference's
GPU prompt pass (gpu_prefill.py:64-239) instead of the CPU engine's INT16-digit arithmetic.  The checker is the EXACT kernel (bit-identical to
the oracle, tests/test_prefill_gpu.py); stated tolerances below.  Routing ids, the sort, skipped slots and the weighted combine are shared code."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import make_experts, rand_bf16, upload

pytestmark = pytest.mark.gpu
F = np.float32


def _setup(H, I, E, k, n_shared=0, rsf=1.0, seed=0, bits=4, w2_bits=None):
    import torch
    from krasis_amd import KrasisEngine, ModelConfig
    rng = np.random.default_rng(seed)
    experts = make_experts(rng, E, H, I, bits, w2_bits)
    shared = make_experts(rng, 1, H, n_shared * I, bits, w2_bits)[0] if n_shared else None
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, 1, n_shared, rsf))
    upload(eng, 0, experts, shared)
    return eng, experts, shared, rng, torch


def _prefill_f32(eng, torch, x, ids, w, fast, routed_only=False):
    from krasis_amd import _lib
    from krasis_amd._lib import check
    M, H = x.shape
    xt = torch.from_numpy(x.view(np.int16)).cuda(); it = torch.from_numpy(ids).cuda(); wt = torch.from_numpy(w).cuda()
    out = torch.empty((M, H), dtype=torch.float32, device="cuda")
    check(eng._lib.kr_moe_set_gemm_mode(eng._h, 1 if fast else 0))
    check(eng._lib.kr_moe_prefill(eng._h, 0, xt.data_ptr(), it.data_ptr(), wt.data_ptr(), out.data_ptr(), M, ids.shape[1], _lib.KR_OUT_F32, int(routed_only), 1))
    torch.cuda.synchronize()
    check(eng._lib.kr_moe_set_gemm_mode(eng._h, 0))
    return out.cpu().numpy()


def _deq_t(w, sc, bits, gs=128):
    """the CPU transposed layout (weights/mod.rs:287) -> float64 [K, N]: int4 words [K/8, N] hold k = 8 row .. 8 row + 7 in nibble order"""
    s = O.bf16_to_f32(sc).astype(np.float64)
    if bits == 8:
        q = w.astype(np.float64)
    else:
        q = np.stack([((w >> (4 * j)) & 15).astype(np.float64) - 8.0 for j in range(8)], axis=1).reshape(w.shape[0] * 8, w.shape[1])
    return q * np.repeat(s, gs, axis=0)[: q.shape[0]]


@pytest.mark.parametrize("H,I,E,k,M,n_shared,bits,w2_bits", [
    (256, 128, 8, 2, 200, 0, 4, 4),
    (512, 384, 16, 4, 333, 1, 4, 4),       # odd group count in w2 (half-empty last stage), shared expert, ragged tiles
    (2048, 512, 32, 10, 700, 1, 4, 4),     # QCN expert shape; ~220 rows per expert: full and ragged tiles, the tiles of one expert share an XCD run
    (512, 384, 16, 4, 333, 1, 8, 8),       # INT8-g128 experts (Q8 configuration)
    (256, 128, 8, 2, 200, 0, 4, 8),        # mixed: INT4 gate/up, INT8 down
    (256, 768, 8, 2, 4096, 1, 4, 4),       # many full 64-row tiles (unguarded store path), rows of 768 intermediates (two chunks per lane in the activation kernel)
    (256, 2304, 4, 2, 150, 0, 4, 4),       # intermediate rows longer than 2048: the block-per-row activation kernel
])
def test_fast_expert_gemm_vs_exact(H, I, E, k, M, n_shared, bits, w2_bits):
    """STATED TOLERANCE: relative RMS error of the f32 outputs <= 1e-3 against the exact kernel, worst element <= 5e-3 of the largest output
    (expected ~2e-4: the activations are rounded to 11 significant bits instead of 15 per group; the weights are de-quantized exactly).  Tokens
    with every slot skipped stay exactly zero / exactly the shared expert."""
    eng, experts, shared, rng, torch = _setup(H, I, E, k, n_shared, 2.0 if n_shared else 1.0, bits=bits, w2_bits=w2_bits)
    x = rand_bf16(rng, (M, H)); ids = np.stack([rng.choice(E, k, replace=False) for _ in range(M)]).astype(np.int32)
    ids[3, 1] = -1; ids[7, :] = -1
    if E >= 16:
        ids[:, 0] = 5                                                # a hot expert: multi-tile path
    x[11] = 0                                                        # an all-zero row: the row multiplier must not produce NaN
    x[12] = O.f32_to_bf16(O.bf16_to_f32(x[12]) * 3000.0)             # a row far outside the f16 range before its power-of-two scaling
    w = rng.random((M, k)).astype(F)
    ex = _prefill_f32(eng, torch, x, ids, w, False)
    fa = _prefill_f32(eng, torch, x, ids, w, True)
    assert np.isfinite(fa).all()
    rel = float(np.sqrt(np.mean((fa - ex) ** 2)) / np.sqrt(np.mean(ex ** 2)))
    # per row: the worst element against that row's own scale (row 12 is 3000 x larger than the others)
    rowmax = np.abs(ex).max(axis=1); ok = rowmax > 0
    worst = float((np.abs(fa - ex).max(axis=1)[ok] / rowmax[ok]).max())
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/gemm_fast_err.txt", "a") as f:
            f.write(f"experts H={H} I={I} E={E} k={k} M={M} shared={n_shared} bits={bits}/{w2_bits}: rel_rms={rel:.3e} worst_row_rel={worst:.3e}\n")
    assert rel <= 1e-3, rel
    assert worst <= 5e-3, worst
    if not n_shared:
        assert not fa[7].any() and not ex[7].any()
    # routed-only form (gpu_prefill.py:4467) takes the same path
    ex_r = _prefill_f32(eng, torch, x[:70], ids[:70], w[:70], False, routed_only=True)
    fa_r = _prefill_f32(eng, torch, x[:70], ids[:70], w[:70], True, routed_only=True)
    assert float(np.sqrt(np.mean((fa_r - ex_r) ** 2)) / np.sqrt(np.mean(ex_r ** 2))) <= 1e-3


def test_fast_gemm_is_closer_to_real_arithmetic_than_its_tolerance():
    """both forms against float64 arithmetic on the de-quantized weights (bf16 inputs are exact in f64; SiLU with the same degree-5 sigmoid is
    replaced by the true one, so a common ~1e-4 floor remains): the tolerance form's error is of the order of the exact form's, not of its 1e-3 budget"""
    H, I, E, k, M = 512, 256, 8, 2, 150
    eng, experts, shared, rng, torch = _setup(H, I, E, k, seed=5)
    x = rand_bf16(rng, (M, H)); ids = np.stack([rng.choice(E, k, replace=False) for _ in range(M)]).astype(np.int32)
    w = rng.random((M, k)).astype(F)
    ex = _prefill_f32(eng, torch, x, ids, w, False).astype(np.float64)
    fa = _prefill_f32(eng, torch, x, ids, w, True).astype(np.float64)
    xf = O.bf16_to_f32(x).astype(np.float64)
    W13 = [_deq_t(e.w13, e.w13_scales, 4) for e in experts]; W2 = [_deq_t(e.w2, e.w2_scales, 4) for e in experts]
    ref = np.zeros((M, H))
    for t in range(M):
        for s in range(k):
            e = ids[t, s]
            gu = xf[t] @ W13[e]
            h = gu[:I] / (1.0 + np.exp(-gu[:I])) * gu[I:]
            ref[t] += w[t, s] * (h @ W2[e])
    den = np.sqrt(np.mean(ref ** 2))
    e_ex = float(np.sqrt(np.mean((ex - ref) ** 2)) / den); e_fa = float(np.sqrt(np.mean((fa - ref) ** 2)) / den)
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/gemm_fast_err.txt", "a") as f:
            f.write(f"vs float64 on de-quantized weights: exact form {e_ex:.3e}, tolerance form {e_fa:.3e}\n")
    assert e_ex < 5e-3, e_ex            # the layout reading of _deq_t is right (the exact form is the oracle's arithmetic)
    assert e_fa <= max(3.0 * e_ex, 5e-4), (e_fa, e_ex)


@pytest.mark.parametrize("kinds,fp8", [(["la", "gqa", "la", "gqa"], True), (["gqa", "gqa"], False)])
def test_gemm_fast_prompt_pass_logits_and_perplexity(kinds, fp8):
    """whole-model prompt pass with KR_ATTN_FAST | KR_GEMM_FAST against the exact pass on the synthetic hybrid model: every projection, the shared
    expert, the routed experts and the scoring pass's lm_head run in the tolerance form.  STATED TOLERANCE: last-position logits within 1e-2 of the
    largest logit, |PPL_fast / PPL_exact - 1| <= 3e-3, |mean NLL difference| <= 3e-3 nats over 700 tokens in windows of 256 (the router sees inputs
    that differ in the last bits, so a near-tie can pick another expert for a token: the budget covers that, not only the rounding)."""
    from krasis_amd.perplexity import evaluate_perplexity
    from tests.test_attn_fast_gpu import build
    res = {}
    rng = np.random.default_rng(99)
    toks = None
    for mode in (False, True):
        st, eng, orc, keep, d = build(seed=17, kv_max=320, kinds=kinds, hd=128, nh=8)
        if fp8:
            st.set_kv_dtype(True)
        st.set_attention_mode(mode, gemm_fast=mode)
        if toks is None:
            toks = [int(x) for x in rng.integers(0, d["V"], 700)]
        ppl = evaluate_perplexity(st, toks, 256, 128)
        st.reset_decode_state(d["kv_max"])
        lg = np.empty(d["V"], F)
        st.set_prefill_chunk(64)
        st.prefill(toks[:200], 0, lg.ctypes.data)
        res[mode] = (ppl, lg.copy())
    a, b = res[False][0], res[True][0]
    rel = abs(b["perplexity"] / a["perplexity"] - 1.0); dn = abs(b["mean_loss"] - a["mean_loss"])
    lrel = float(np.abs(res[False][1] - res[True][1]).max() / np.abs(res[False][1]).max())
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/gemm_fast_err.txt", "a") as f:
            f.write(f"prompt pass kinds={'+'.join(kinds)} fp8={fp8}: ppl exact {a['perplexity']:.6f} fast {b['perplexity']:.6f} rel {rel:.3e} "
                    f"mean-nll diff {dn:.3e} logits rel {lrel:.3e}\n")
    assert np.isfinite(res[True][1]).all()
    assert rel <= 3e-3 and dn <= 3e-3, (rel, dn)
    assert lrel <= 1e-2, lrel


@pytest.mark.parametrize("cfg", [dict(kv_max=700), dict(lora=True, seed=2, kv_max=640)])
def test_gemm_fast_mla_prompt_pass(cfg):
    """MLA layers (direct and LoRA query paths: kv_a / q_a / q_b / o projections, dense MLP or experts) with both FAST bits against the exact pass.
    STATED TOLERANCE: last-position logits within 1e-2 of the largest logit (FP16 latent cache), greedy token unchanged."""
    from tests.test_mla_gpu import build as build_mla
    res = {}
    for mode in (False, True):
        st, eng, orc, keep, d = build_mla(**cfg)
        st.set_attention_mode(mode, gemm_fast=mode)
        toks = [int(x) for x in np.random.default_rng(3).integers(0, d["V"], 150)]
        st.set_prefill_chunk(64)
        pl = np.empty(d["V"], F)
        ptok = st.prefill(toks, 0, pl.ctypes.data)
        res[mode] = (pl.copy(), ptok)
    rel = float(np.abs(res[False][0] - res[True][0]).max() / np.abs(res[False][0]).max())
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/gemm_fast_err.txt", "a") as f:
            f.write(f"mla prompt pass cfg={cfg}: logits rel {rel:.3e}\n")
    assert np.isfinite(res[True][0]).all() and rel <= 1e-2, rel
    assert res[False][1] == res[True][1]


def test_gemm_mode_validation():
    from krasis_amd._lib import check
    eng, experts, shared, rng, torch = _setup(256, 128, 8, 2)
    with pytest.raises(ValueError):
        check(eng._lib.kr_moe_set_gemm_mode(eng._h, 2))
