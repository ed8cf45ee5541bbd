"""What the three tolerance bits cost on a model whose predictions are PEAKED (the synthetic models of the other FAST tests sit at
perplexity ~ vocabulary size, where a relative perplexity bound says little).  tests/test_decode_gpu.build(peaked=True): embeddings tied to lm_head
so that the current token's logit leads by ~8; token stream = a Markov chain that repeats the previous token with probability 0.8 -- the exact
mode's perplexity is O(10), every layer (linear attention, gated GQA, router, experts, shared expert) contributes at the noise-floor level, and a
flipped expert or a last-bit difference moves the score of many positions.  Reported per mode (appended to gpurun_out/tolerance_peaked.txt) and
asserted:
    KR_ATTN_FAST, KR_ATTN_FAST | KR_GEMM_FAST   prompt pass (evaluate_perplexity = perplexity/measure_ppl.py:154-297): |PPL_fast / PPL_exact - 1| <= 2e-3
                                                (measured on MI355X: 0.8e-5 .. 3.0e-4); per-position NLL: at most 1 % of the positions move by more than 2e-2
                                                and none by more than 0.25 -- the large single-position moves (measured up to 0.098 with the tolerance GEMMs
                                                and an E4M3 cache) are tokens whose k-th / (k+1)-th router scores are a near-tie that flips when the router's
                                                INPUT differs in its last bits; ids are bit-exact for identical inputs, not across modes
    KR_DECODE_FAST                              token-by-token decode: the same bounds, top-1 agreement >= 99.5 %, router ids of the last MoE layer
                                                identical on >= 99 % of the tokens"""
import os

import numpy as np
import pytest

from tests.test_decode_gpu import build

pytestmark = pytest.mark.gpu
F = np.float32


def _stream(V, n, seed=0):
    rng = np.random.default_rng(seed)
    t = [int(rng.integers(0, V))]
    for _ in range(n - 1):
        t.append(t[-1] if rng.random() < 0.8 else int(rng.integers(0, V)))
    return t


def _log(msg):
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/tolerance_peaked.txt", "a") as f:
            f.write(msg + "\n")


def _nll(logits, label):
    x = logits.astype(np.float64); m = x.max()
    return float(m + np.log(np.exp(x - m).sum()) - x[label])


@pytest.mark.parametrize("fp8", [False, True])
def test_prompt_pass_modes_on_the_peaked_model(fp8):
    from krasis_amd.perplexity import evaluate_perplexity
    res, per_pos = {}, {}
    toks = None
    for name, mode in (("exact", (False, False)), ("attn_fast", (True, False)), ("attn+gemm_fast", (True, True))):
        st, eng, orc, keep, d = build(seed=9, kv_max=640, kinds=["la", "gqa", "la", "gqa"], peaked=True)
        if fp8:
            st.set_kv_dtype(True)
        st.set_attention_mode(mode[0], gemm_fast=mode[1])
        toks = toks or _stream(d["V"], 600, 1)
        res[name] = evaluate_perplexity(st, toks, 256, 128)
        st.reset_decode_state(640)
        per_pos[name] = np.asarray(st.prefill_nll(toks, 0), np.float64)
    a = res["exact"]
    assert 2.0 < a["perplexity"] < 40.0, a["perplexity"]           # peaked: far below the vocabulary size (512)
    for name in ("attn_fast", "attn+gemm_fast"):
        b = res[name]
        dd = np.abs(per_pos[name] - per_pos["exact"])
        rel = abs(b["perplexity"] / a["perplexity"] - 1.0); dmax = float(dd.max()); frac = float(np.mean(dd > 2e-2))
        _log(f"prompt pass fp8={fp8} {name}: PPL exact {a['perplexity']:.5f} fast {b['perplexity']:.5f} rel {rel:.3e}  per-position NLL diff: max {dmax:.3e}, "
             f"{100 * frac:.2f} % of {dd.size} positions above 2e-2")
        assert rel <= 2e-3, (name, rel)
        assert dmax <= 0.25 and frac <= 0.01, (name, dmax, frac)


def test_decode_fast_on_the_peaked_model():
    out = {}
    n = 400
    toks = None
    for fast in (False, True):
        st, eng, orc, keep, d = build(seed=9, kv_max=640, kinds=["la", "gqa", "la", "gqa"], peaked=True)
        st.set_attention_mode(False, decode_fast=fast)
        toks = toks or _stream(d["V"], n + 1, 2)
        nll, top1, ids = [], [], []
        lg = np.empty(d["V"], F)
        for i in range(n):
            st.decode_step(toks[i], i, lg.ctypes.data)
            nll.append(_nll(lg, toks[i + 1])); top1.append(int(np.argmax(lg)))
            ids.append(tuple(int(x) for x in st.read_router(16, 4)[1]))
        out[fast] = (np.asarray(nll), top1, ids)
    pe, pf = float(np.exp(out[False][0].mean())), float(np.exp(out[True][0].mean()))
    dd = np.abs(out[True][0] - out[False][0])
    rel = abs(pf / pe - 1.0); dmax = float(dd.max()); frac = float(np.mean(dd > 2e-2))
    agree = float(np.mean([a == b for a, b in zip(out[False][1], out[True][1])]))
    rid = float(np.mean([a == b for a, b in zip(out[False][2], out[True][2])]))
    _log(f"decode, {n} tokens: PPL exact {pe:.5f} KR_DECODE_FAST {pf:.5f} rel {rel:.3e}  per-position NLL diff: max {dmax:.3e}, {100 * frac:.2f} % above 2e-2  top-1 agreement {agree:.4f}  "
         f"router ids (last MoE layer) identical on {rid:.4f} of the tokens")
    assert 2.0 < pe < 40.0, pe
    assert rel <= 2e-3, rel
    assert dmax <= 0.25 and frac <= 0.01, (dmax, frac)
    assert agree >= 0.995, agree
    assert rid >= 0.99, rid
