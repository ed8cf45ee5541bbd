"""The LDS-ring form of the tolerance GEMM (krasis_amd/csrc/kr_prefill_ring.hip: operands by LDS-DMA into rings, 8-wave workgroups, one barrier per 64-k
unit) against the register-staged kernel it replaces for big INT4 problems (kr_prefill_h.hip).  Both feed v_mfma_f32_32x32x16_f16 the same fragments in
the same order per accumulator, so the check is BIT IDENTITY of the f32 / f16 outputs, not a tolerance; the tolerance of either form against the exact
(oracle-identical) kernel is stated in tests/test_gemm_fast_gpu.py, whose QCN-shaped cases run the ring kernel by default.
kr_moe_set_gemm_mode: 1 = default (ring for problems that fill the chip), 3 = register-staged kernels only, 5 = ring for every shape it takes."""
import numpy as np
import pytest

from tests.test_gemm_fast_gpu import _setup
from tests.util import rand_bf16

pytestmark = pytest.mark.gpu
F = np.float32


def _prefill_mode(eng, torch, x, ids, w, mode, routed_only=False):
    from krasis_amd import _lib
    from krasis_amd._lib import check
    M, H = x.shape
    xt = torch.from_numpy(x.view(np.int16)).cuda(); it = torch.from_numpy(ids).cuda(); wt = torch.from_numpy(w).cuda()
    out = torch.empty((M, H), dtype=torch.float32, device="cuda")
    check(eng._lib.kr_moe_set_gemm_mode(eng._h, mode))
    check(eng._lib.kr_moe_prefill(eng._h, 0, xt.data_ptr(), it.data_ptr(), wt.data_ptr(), out.data_ptr(), M, ids.shape[1], _lib.KR_OUT_F32, int(routed_only), 1))
    torch.cuda.synchronize()
    check(eng._lib.kr_moe_set_gemm_mode(eng._h, 1)); check(eng._lib.kr_moe_set_gemm_mode(eng._h, 0))      # ring back to its default, exact mode
    return out.cpu().numpy()


@pytest.mark.parametrize("H,I,E,k,M,n_shared", [
    (2048, 512, 32, 10, 700, 1),      # QCN expert shape: fused gate | up activation epilogue (I % 256 == 0), full and ragged 64-row tiles, shared expert (dense 128 x 256 tiles in mode 5)
    (512, 384, 16, 4, 333, 1),        # odd group count in w2 (the last stage holds one group), I % 256 != 0: gate | up stored, activation pass separate
    (256, 768, 8, 2, 1500, 0),        # K = 256: one stage, two groups; many full tiles; rows of 768 intermediates
    (1024, 256, 8, 2, 96, 0),         # tiles of <= 32 rows and column blocks past N (w13: N = 512 = one block; w2: N = 1024)
    (384, 128, 8, 2, 200, 0),         # K = 384 (odd group count in w13), N = 256 < one 512-column block
])
def test_ring_gemm_bit_identical_to_register_staged(H, I, E, k, M, n_shared):
    eng, experts, shared, rng, torch = _setup(H, I, E, k, n_shared, 2.0 if n_shared else 1.0)
    x = rand_bf16(rng, (M, H)); ids = np.stack([rng.choice(E, k, replace=False) for _ in range(M)]).astype(np.int32)
    ids[3, 1] = -1; ids[7, :] = -1
    if E >= 16:
        ids[:, 0] = 5                                                # a hot expert: several row tiles of one expert share an XCD run
    x[11] = 0
    w = rng.random((M, k)).astype(F)
    staged = _prefill_mode(eng, torch, x, ids, w, 3)
    ring = _prefill_mode(eng, torch, x, ids, w, 5)
    assert np.isfinite(ring).all()
    assert np.array_equal(staged.view(np.uint32), ring.view(np.uint32)), float(np.abs(staged - ring).max())
    auto = _prefill_mode(eng, torch, x, ids, w, 1)                   # the default dispatch, whichever kernel it picks
    assert np.array_equal(staged.view(np.uint32), auto.view(np.uint32))
    r_staged = _prefill_mode(eng, torch, x[:70], ids[:70], w[:70], 3, routed_only=True)
    r_ring = _prefill_mode(eng, torch, x[:70], ids[:70], w[:70], 5, routed_only=True)
    assert np.array_equal(r_staged.view(np.uint32), r_ring.view(np.uint32))


def test_ring_gemm_repeatable():
    """a race between an LDS-DMA and a fragment read would show as run-to-run differences: 20 runs of one problem, every output bit equal"""
    eng, experts, shared, rng, torch = _setup(2048, 512, 32, 10, 1, 2.0)
    M = 2000
    x = rand_bf16(rng, (M, 2048)); ids = np.stack([rng.choice(32, 10, replace=False) for _ in range(M)]).astype(np.int32)
    w = rng.random((M, 10)).astype(F)
    first = _prefill_mode(eng, torch, x, ids, w, 5)
    for _ in range(19):
        assert np.array_equal(first.view(np.uint32), _prefill_mode(eng, torch, x, ids, w, 5).view(np.uint32))
    assert np.array_equal(first.view(np.uint32), _prefill_mode(eng, torch, x, ids, w, 3).view(np.uint32))


@pytest.mark.parametrize("kinds", [["la", "gqa"], ["gqa", "gqa"]])
def test_ring_gemm_prompt_pass_bit_identical(kinds):
    """whole-model tolerance prompt pass (every dense projection through the 128 x 256 ring tiles with option gemm_ring = 2) against the same pass on the
    register-staged kernels: last-position logits bit for bit"""
    from tests.test_attn_fast_gpu import build
    rng = np.random.default_rng(5)
    toks = None
    res = []
    for ring in (0, 2):
        st, eng, orc, keep, d = build(seed=23, kv_max=400, kinds=kinds, hd=128, nh=8)
        st.set_attention_mode(True, gemm_fast=True)
        st.set_option("gemm_ring", ring)
        if toks is None:
            toks = [int(t) for t in rng.integers(0, d["V"], 300)]
        lg = np.empty(d["V"], F)
        st.prefill(toks, 0, lg.ctypes.data)
        res.append(lg.copy())
        st.set_option("gemm_ring", 1)
    assert np.isfinite(res[1]).all()
    assert np.array_equal(res[0].view(np.uint32), res[1].view(np.uint32)), float(np.abs(res[0] - res[1]).max())


@pytest.mark.parametrize("kinds", [["la", "gqa"]])
def test_norm_launch_writes_the_f16_row_image_bit_identically(kinds):
    """round 6: in the tolerance prompt pass the norm launch writes the f16 row image of its own output (kr_pfm_norm_kernel, KrPfmNormArgs::xf) instead of a separate
    kr_pfh_rows_kernel<0> launch over the stored f32 rows: same values, same row maximum, same rounding -- last-position logits AND the scoring pass's NLL bit for bit"""
    from tests.test_attn_fast_gpu import build
    rng = np.random.default_rng(6)
    toks = None
    res = []
    for fused in (0, 1):
        st, eng, orc, keep, d = build(seed=29, kv_max=400, kinds=kinds, hd=128, nh=8)
        st.set_attention_mode(True, gemm_fast=True)
        st.set_option("norm_rows", fused)
        if toks is None:
            toks = [int(t) for t in rng.integers(0, d["V"], 300)]
        lg = np.empty(d["V"], F)
        st.prefill(toks, 0, lg.ctypes.data)
        st.reset_decode_state(d["kv_max"])
        nll = st.prefill_nll(toks, 0)
        res.append((lg.copy(), np.array(nll, copy=True)))
    assert np.isfinite(res[1][0]).all()
    assert np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32))
    assert np.array_equal(res[0][1].view(np.uint32), res[1][1].view(np.uint32))


@pytest.mark.parametrize("kinds,ntok", [(["la", "gqa", "la"], 300), (["la", "la"], 131)])
def test_conv_inside_the_delta_rule_prep_launch_bit_identical(kinds, ntok):
    """round 6: in the KR_ATTN_FAST prompt pass the causal conv + SiLU + L2 norms + gates of a linear-attention layer are formed by the delta rule's prep launch from
    the in-projection's output (kr_lac_prep_kernel, KrLacArgs::fused) and the gated norm reads z in place; kr_pfm_la_conv_kernel does not run.  Same arithmetic value for
    value: last-position logits, recurrent state and carried conv state bit for bit against the stand-alone conv launch (option la_conv_fused = 0); chunks of 64 tokens so
    that the carried conv slots cross chunk borders, a ragged last sub-chunk."""
    from tests.test_attn_fast_gpu import build
    rng = np.random.default_rng(8)
    toks = None
    res = []
    for fused in (0, 1):
        st, eng, orc, keep, d = build(seed=31, kv_max=400, kinds=kinds, hd=128, nh=8)
        st.set_attention_mode(True, gemm_fast=False)
        st.set_option("la_conv_fused", fused)
        st.set_prefill_chunk(128)
        if toks is None:
            toks = [int(t) for t in rng.integers(0, d["V"], ntok)]
        lg = np.empty(d["V"], F)
        st.prefill(toks, 0, lg.ctypes.data)
        states = []
        for li, kind in enumerate(d["kinds"]):
            if kind == "la":
                cs = np.empty(d["conv_dim"] * 4, F); rs = np.empty(d["nv"] * d["dk"] * d["dv"], F)
                st.get_decode_state(li, None, None, cs, rs)
                states += [cs, rs]
        res.append((lg.copy(), states))
    assert np.isfinite(res[1][0]).all()
    assert np.array_equal(res[0][0].view(np.uint32), res[1][0].view(np.uint32)), float(np.abs(res[0][0] - res[1][0]).max())
    for a_, b_ in zip(res[0][1], res[1][1]):
        assert np.array_equal(a_.view(np.uint32), b_.view(np.uint32))
