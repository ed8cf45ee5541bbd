"""Tolerance-mode attention for long caches (kr_decode_set_attention_mode, KR_ATTN_FAST): split-KV softmax + p.v with a log-sum-exp merge.
The exact kernels (reference order, bit-identical to decode.rs:4194-4281) are the yardstick: the same decode steps are run in both modes and
the logits compared at a STATED tolerance.  GQA runs split-KV flash-decode on the f16 MFMA (kr_attn_flash.hip: q and the probabilities rounded to
f16 = 2^-11, K / V exact in f16, f32 accumulation, exp2 on the transcendental unit); MLA keeps exact f32 scores with a split softmax + p.v:
    max |logits_fast - logits_exact| <= 1e-3 * max |logits_exact|          (measured on MI355X: 1.8e-4 .. 4.2e-4 -- the attention output differs in
                                                                            its low bits, its INT16 re-quantisation for the o-projection then
                                                                            moves single values by one step, and that step is what the logits see)
and the greedy token is unchanged.  Router top-k ids must be identical (north_star: bit-exact ids)."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.test_decode_gpu import build

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.mark.parametrize("hd,fp8,kv_max", [(256, False, 1300), (128, True, 1300), (64, False, 2100), (256, True, 4400)])
def test_fast_attention_matches_exact_within_tolerance(hd, fp8, kv_max):
    outs = {}
    for mode in (False, True):
        st, eng, orc, keep, d = build(seed=31, kv_max=kv_max, hd=hd)
        if fp8:
            st.set_kv_dtype(True)
            rng = np.random.default_rng(5)
            kv = {li: (O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["nkv"] * d["hd"])) * 0.5).astype(F)),
                       O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["nkv"] * d["hd"])) * 0.5).astype(F))) for li, kind in enumerate(d["kinds"]) if kind == "gqa"}
            n = len(d["kinds"]); ptr = lambda a: a.ctypes.data
            st.set_decode_state(5, d["kv_max"], [ptr(kv[i][0]) if i in kv else 0 for i in range(n)], [ptr(kv[i][1]) if i in kv else 0 for i in range(n)],
                                [ptr(x) if x is not None else 0 for x in d["state"]["conv"]], [ptr(x) if x is not None else 0 for x in d["state"]["recur"]])
        st.set_attention_mode(mode)
        res = []
        tok = 5
        for pos in [3, 255, 256, 700, 1023, 1024, kv_max - 1]:
            lg = np.empty(d["V"], F); st.decode_step(tok, pos, lg.ctypes.data); res.append(lg.copy())
            tok = int(np.argmax(res[-1])) if mode is False else outs[False][1][len(res) - 1]      # both runs are fed the exact run's tokens
        outs[mode] = (res, [int(np.argmax(r)) for r in res])
    worst = 0.0
    for a, b in zip(outs[False][0], outs[True][0]):
        worst = max(worst, float(np.abs(a - b).max() / np.abs(a).max()))
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/r02_attn_fast_err.txt", "a") as f:
            f.write(f"hd={hd} fp8={fp8} kv_max={kv_max} worst_rel={worst:.3e}\n")
    assert worst <= 1e-3, worst
    assert outs[False][1] == outs[True][1]


def test_fast_mode_leaves_short_caches_bit_exact():
    """caches up to 1024 positions never take the split path: fast mode must be bit-identical there"""
    st, eng, orc, keep, d = build(seed=3, kv_max=300)
    st.set_attention_mode(True)
    tok = 7
    for pos in [5, 128, 299]:
        lg = np.empty(d["V"], F); st.decode_step(tok, pos, lg.ctypes.data)
        ref = orc.step(tok, pos)
        assert np.array_equal(lg.view(np.uint32), ref.view(np.uint32)), pos
        tok = O.sample_greedy(ref)


@pytest.mark.parametrize("hd,nh,fp8,n_tok,start,chunk", [(256, 16, False, 170, 37, 0), (256, 16, True, 170, 37, 64), (128, 8, False, 200, 0, 0), (128, 4, True, 130, 5, 50),
                                                         (64, 16, False, 170, 37, 0), (64, 4, True, 90, 3, 0)])
def test_flash_prompt_pass_matches_exact_within_tolerance(hd, nh, fp8, n_tok, start, chunk):
    """prompt pass in FAST mode: causal flash attention on f16 MFMA (kr_attn_flash.hip: S^T = K Q^T, online softmax in f32, O^T += V^T P^T; q and
    the probabilities rounded to f16) against the exact prompt pass (== token-by-token decode, bit for bit) of the same model and prompt.
    STATED TOLERANCE on the last-position logits: max |fast - exact| <= 3e-3 * max |exact| (f16 q / p: 2^-11 per product, eight times finer than the
    bf16 flash attention the reference's own GPU prefill uses); same greedy token.  Cache starts from random FP16 / E4M3 rows below `start`."""
    outs = {}
    for mode in (False, True):
        st, eng, orc, keep, d = build(seed=11, kv_max=260, hd=hd, nh=nh)
        if fp8:
            st.set_kv_dtype(True)
            rng = np.random.default_rng(5)
            kv = {li: (O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["nkv"] * d["hd"])) * 0.5).astype(F)), O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["nkv"] * d["hd"])) * 0.5).astype(F)))
                  for li, kind in enumerate(d["kinds"]) if kind == "gqa"}
            n = len(d["kinds"]); ptr = lambda a: a.ctypes.data
            st.set_decode_state(5, d["kv_max"], [ptr(kv[i][0]) if i in kv else 0 for i in range(n)], [ptr(kv[i][1]) if i in kv else 0 for i in range(n)],
                                [ptr(x) if x is not None else 0 for x in d["state"]["conv"]], [ptr(x) if x is not None else 0 for x in d["state"]["recur"]])
        st.set_attention_mode(mode)
        if chunk:
            st.set_prefill_chunk(chunk)
        rng = np.random.default_rng(hd + nh)
        toks = [int(x) for x in rng.integers(0, d["V"], n_tok)]
        lg = np.empty(d["V"], F)
        tok = st.prefill(toks, start, lg.ctypes.data)
        nxt = np.empty(d["V"], F); st.decode_step(tok, start + n_tok, nxt.ctypes.data)      # decoding continues on the state the prompt pass left
        outs[mode] = (lg.copy(), tok, nxt.copy())
    a, b = outs[False][0], outs[True][0]
    rel = float(np.abs(a - b).max() / np.abs(a).max())
    rel2 = float(np.abs(outs[False][2] - outs[True][2]).max() / np.abs(outs[False][2]).max())
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/r02_attn_fast_err.txt", "a") as f:
            f.write(f"flash prefill hd={hd} nh={nh} fp8={fp8} n={n_tok} chunk={chunk} rel={rel:.3e} next-step rel={rel2:.3e}\n")
    assert np.isfinite(b).all() and rel <= 3e-3, rel
    assert outs[False][1] == outs[True][1]
    assert rel2 <= 3e-3, rel2


@pytest.mark.parametrize("la_heads,n_tok,chunk", [((2, 4), 300, 0), ((2, 4), 333, 128), ((4, 16), 200, 0), ((1, 8), 64, 0)])
def test_chunked_delta_rule_matches_exact_within_tolerance(la_heads, n_tok, chunk):
    """linear-attention layers only, so the ONLY difference between the two modes is kr_la_chunk.hip: the gated delta rule over sub-chunks of 64
    tokens in closed form (triangular solve + f32 MFMA products) against the per-token recurrence (bit-identical to decode.rs:1293).  Same f32
    products, another summation order.  STATED TOLERANCE: max |fast - exact| <= 1e-3 * max |exact| on the last-position logits AND on the logits of
    the next decode step (which runs on the recurrent state the prompt pass left); same greedy token.  (Measured on MI355X: 0.6e-4 .. 3.7e-4 -- as
    with the attention fast mode, last-bit differences of the layer output move single INT16 digits of the next projection's input by one step.)  Covers a partial last sub-chunk, chunks
    that fall back to the exact kernel (< 64 tokens), and the head counts of both workgroup mappings (nv % 8 == 0 or not)."""
    outs = {}
    for mode in (False, True):
        st, eng, orc, keep, d = build(seed=13, kv_max=400, kinds=["la", "la", "la"], la_heads=la_heads)
        st.set_attention_mode(mode)
        if chunk:
            st.set_prefill_chunk(chunk)
        rng = np.random.default_rng(n_tok)
        toks = [int(x) for x in rng.integers(0, d["V"], n_tok)]
        lg = np.empty(d["V"], F)
        tok = st.prefill(toks, 0, lg.ctypes.data)
        nxt = np.empty(d["V"], F); st.decode_step(tok, n_tok, nxt.ctypes.data)
        outs[mode] = (lg.copy(), tok, nxt.copy())
    rel = float(np.abs(outs[False][0] - outs[True][0]).max() / np.abs(outs[False][0]).max())
    rel2 = float(np.abs(outs[False][2] - outs[True][2]).max() / np.abs(outs[False][2]).max())
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/r02_attn_fast_err.txt", "a") as f:
            f.write(f"chunked delta rule heads={la_heads} n={n_tok} chunk={chunk} rel={rel:.3e} next-step rel={rel2:.3e}\n")
    assert np.isfinite(outs[True][0]).all() and rel <= 1e-3, rel
    assert rel2 <= 1e-3, rel2
    assert outs[False][1] == outs[True][1]


@pytest.mark.parametrize("cfg", [dict(kv_max=700), dict(klr=256, nh=3, seed=4, kv_max=900), dict(lora=True, seed=2, kv_max=640)])
@pytest.mark.parametrize("fp8", [False, True])
def test_mla_fast_decode_and_flash_prompt_pass(cfg, fp8):
    """MLA in FAST mode: (a) decode over a long latent cache: split-KV flash-decode over [ckv | kpe] rows on the f16 MFMA, all heads share a
    chunk (kr_mla_flash.hip, SPLIT form) + log-sum-exp merge, (b) prompt pass: the same kernel over 64 (token, head) rows.  q and the
    probabilities are rounded to f16 in both.  Stated tolerances on the logits, FP16 caches: decode 3e-3, prompt pass 3e-3 relative (measured
    5.7e-4 .. 1.0e-3 and 1.0e-3 .. 2.2e-3).  E4M3 caches: the latent rows the steps APPEND are
    re-quantised to 3 mantissa bits, so a last-bit difference upstream can flip a stored code by 6 % of that element and the two runs' caches
    diverge -- the comparison then bounds that sensitivity, not the kernels: decode 5e-3, prompt pass 2e-2 (measured up to 1.9e-3 / 9.3e-3)."""
    from tests.test_mla_gpu import build as build_mla
    res = {}
    for mode in (False, True):
        st, eng, orc, keep, d = build_mla(**cfg)
        if fp8:
            st.set_kv_dtype(True)
            rng = np.random.default_rng(9)
            n = d["nL"]
            ck = [O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["klr"])) * 0.5).astype(F)) for _ in range(n)]
            kp = [O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["rd"])) * 0.5).astype(F)) for _ in range(n)]
            st.set_decode_state(5, d["kv_max"], [0] * n, [0] * n, [0] * n, [0] * n, [x.ctypes.data for x in ck], [x.ctypes.data for x in kp])
        st.set_attention_mode(mode)
        outs = []
        tok = 9
        for pos in [5, 300, 511, 512, d["kv_max"] - 200]:
            lg = np.empty(d["V"], F); st.decode_step(tok, pos, lg.ctypes.data); outs.append(lg.copy())
            tok = int(np.argmax(lg)) if mode is False else res[False][1][len(outs) - 1]
        toks = [int(x) for x in np.random.default_rng(3).integers(0, d["V"], 150)]
        st.set_prefill_chunk(64)
        pl = np.empty(d["V"], F)
        ptok = st.prefill(toks, d["kv_max"] - 180, pl.ctypes.data)
        res[mode] = (outs, [int(np.argmax(o)) for o in outs], pl.copy(), ptok)
    worst = max(float(np.abs(a - b).max() / np.abs(a).max()) for a, b in zip(res[False][0], res[True][0]))
    relp = float(np.abs(res[False][2] - res[True][2]).max() / np.abs(res[False][2]).max())
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/r02_attn_fast_err.txt", "a") as f:
            f.write(f"mla cfg={cfg} fp8={fp8} decode worst_rel={worst:.3e} prompt rel={relp:.3e}\n")
    assert worst <= (5e-3 if fp8 else 3e-3), worst
    assert np.isfinite(res[True][2]).all() and relp <= (2e-2 if fp8 else 3e-3), relp
    if not fp8:
        assert res[False][1] == res[True][1] and res[False][3] == res[True][3]


@pytest.mark.parametrize("kinds,fp8", [(["la", "gqa", "la", "gqa"], False), (["la", "gqa", "la", "gqa"], True), (["gqa", "gqa"], True)])
def test_fast_mode_perplexity_delta_on_the_synthetic_model(kinds, fp8):
    """the perplexity harness (krasis_amd.evaluate_perplexity == perplexity/measure_ppl.py:154-297) over a synthetic hybrid model, windows of 256
    tokens (4 sub-chunks of the closed-form delta rule, flash attention over the FP16 / E4M3 cache) in both modes.  STATED TOLERANCE on the
    result the reference reports: |PPL_fast / PPL_exact - 1| <= 1e-3 and |mean NLL difference| <= 1e-3 nats (measured on MI355X: see
    profiles/r02_fast_mode_errors.txt).  The exact mode is the reference CPU-decode arithmetic bit for bit (tests/test_perplexity.py)."""
    from krasis_amd.perplexity import evaluate_perplexity
    res = {}
    rng = np.random.default_rng(99)
    toks = None
    for mode in (False, True):
        st, eng, orc, keep, d = build(seed=17, kv_max=320, kinds=kinds, hd=128, nh=8)
        if fp8:
            st.set_kv_dtype(True)
        st.set_attention_mode(mode)
        if toks is None:
            toks = [int(x) for x in rng.integers(0, d["V"], 700)]
        res[mode] = evaluate_perplexity(st, toks, 256, 128)
    a, b = res[False], res[True]
    assert a["num_tokens_scored"] == b["num_tokens_scored"] == len(toks) - 1
    rel = abs(b["perplexity"] / a["perplexity"] - 1.0); dn = abs(b["mean_loss"] - a["mean_loss"])
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/r02_attn_fast_err.txt", "a") as f:
            f.write(f"ppl delta kinds={'+'.join(kinds)} fp8={fp8}: exact {a['perplexity']:.6f} fast {b['perplexity']:.6f} rel {rel:.3e} mean-nll diff {dn:.3e}\n")
    assert rel <= 1e-3 and dn <= 1e-3, (rel, dn)
