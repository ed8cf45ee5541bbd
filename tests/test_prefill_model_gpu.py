"""Whole-model prompt pass (kr_decode_prefill) == token-by-token decode, BIT FOR BIT: last-position logits, greedy sample, FP16 KV caches,
conv and recurrent state.  The decode path is itself bit-exact against the oracle (tests/test_decode_gpu.py), so prefill == decode == oracle."""
import numpy as np
import pytest

from tests.test_decode_gpu import build

pytestmark = pytest.mark.gpu
F = np.float32


def _snapshot(st, d):
    out = []
    for li, kind in enumerate(d["kinds"]):
        if kind == "la":
            cs = np.empty(d["conv_dim"] * 4, F); rs = np.empty(d["nv"] * d["dk"] * d["dv"], F)
            st.get_decode_state(li, None, None, cs, rs); out.append((cs, rs))
        else:
            kc = np.empty((d["kv_max"], d["nkv"] * d["hd"]), np.uint16); vc = np.empty_like(kc)
            st.get_decode_state(li, kc, vc, None, None); out.append((kc, vc))
    return out


@pytest.mark.parametrize("cfg", [dict(), dict(norm_bias_one=False, scoring=0, rsf=2.5), dict(with_dense=True), dict(wbits=8, with_dense=True)])
@pytest.mark.parametrize("n_tok,start,chunk,depth", [(1, 5, 0, 0), (3, 5, 0, 0), (9, 0, 0, 0), (20, 7, 0, 0), (20, 7, 6, 0), (23, 3, 1, 0), (17, 0, 8, 0),
                                                     (23, 3, 2, 3), (23, 3, 1, 4), (20, 7, 6, 1)])
def test_prefill_equals_sequential_decode(cfg, n_tok, start, chunk, depth):
    """chunk > 0 forces several chunks: `depth` of them are in flight, one per stream / arena, in a (chunk, layer) wavefront."""
    st, eng, orc, keep, d = build(**cfg)
    st.set_prefill_chunk(chunk); st.set_prefill_depth(depth)
    rng = np.random.default_rng(n_tok * 31 + start)
    toks = [int(x) for x in rng.integers(0, d["V"], n_tok)]
    # reference: token-by-token decode from the same initial state
    st.set_use_graph(True)
    ref_logits = np.empty(d["V"], F)
    for i, t in enumerate(toks):
        st.decode_step(t, start + i, ref_logits.ctypes.data)
    ref_tok = st.last_token()
    ref_state = _snapshot(st, d)
    # prompt pass from the same initial state
    d["reset"]()
    logits = np.empty(d["V"], F)
    tok = st.prefill(toks, start, logits.ctypes.data)
    assert np.array_equal(logits.view(np.uint32), ref_logits.view(np.uint32)), float(np.max(np.abs(logits - ref_logits)))
    assert tok == ref_tok
    for li, (a, b) in enumerate(zip(_snapshot(st, d), ref_state)):
        assert np.array_equal(a[0].view(np.uint32) if a[0].dtype == F else a[0], b[0].view(np.uint32) if b[0].dtype == F else b[0]), ("state0", li)
        assert np.array_equal(a[1].view(np.uint32) if a[1].dtype == F else a[1], b[1].view(np.uint32) if b[1].dtype == F else b[1]), ("state1", li)
    # and decoding continues seamlessly after the prompt pass
    nxt = np.empty(d["V"], F)
    st.decode_step(tok, start + n_tok, nxt.ctypes.data)
    assert np.isfinite(nxt).all()


@pytest.mark.parametrize("dims,n_tok,chunk", [((256, 512, 16, 4, 128, 128), 9, 0), ((256, 512, 16, 4, 128, 128), 20, 6), ((512, 512, 8, 2, 256, 128), 12, 0)])
def test_prefill_with_native_gguf_experts_matches_the_oracle_driver(dims, n_tok, chunk):
    """A model whose ROUTED experts are native GGUF blocks (Q4_K gate / up, Q8_0 or Q4_K down) through kr_decode_prefill, against the oracle driver
    (decode.rs:2690-3520 control flow with moe_forward_gguf, moe.rs:990, as the routed-expert block).  The prompt pass runs the int8-MFMA block GEMM
    at every chunk size: exact integer sub-block sums, ONE f32 chain per output instead of the AVX2 kernel's 8 lane chains + hsum -- STATED TOLERANCE
    2e-5 of the largest logit (tests/test_gguf_gpu.py states the same for the operator; measured ~1e-6), same greedy token.  The decode STEP on such
    layers (round 4: the block kernels inside the captured step) is BIT-EXACT against the same driver: tests/test_decode_gpu.py."""
    st, eng, orc, keep, d = build(dims=dims, gguf=True, seed=11)
    st.set_prefill_chunk(chunk)
    rng = np.random.default_rng(n_tok)
    toks = [int(x) for x in rng.integers(0, d["V"], n_tok)]
    logits = np.empty(d["V"], F)
    st.prefill(toks, 5, logits.ctypes.data)
    ref = None
    for i, t in enumerate(toks):
        ref = orc.step(t, 5 + i)
    err = float(np.abs(logits - ref).max() / np.abs(ref).max())
    assert err <= 2e-5, err
    assert int(np.argmax(logits)) == int(np.argmax(ref))
    nxt = np.empty(d["V"], F)
    st.decode_step(1, 5 + n_tok, nxt.ctypes.data)                  # the step after the prompt pass runs on the same native blocks
    refn = orc.step(1, 5 + n_tok)
    assert float(np.abs(nxt - refn).max() / np.abs(refn).max()) <= 2e-5


def test_prefill_with_native_gguf_experts_mfma_chunks_and_tolerance_form():
    """80 tokens in one chunk through the int8-MFMA block GEMM (one f32 chain per output instead of 8 lane chains): the last-bit differences of every
    token's expert outputs reach the last token through the recurrent state and the KV cache -- STATED TOLERANCE 2e-4 of the largest logit against the
    oracle driver (measured 4.2e-5), same greedy token; with KR_GEMM_FAST the Q4_K layers run the f16 tolerance form
    (bound 5e-3 of the largest logit against the exact pass; Q8_0 down projections keep the exact form: I = 128 here -> the layer falls back whole)."""
    for dims, bound_fast in (((256, 512, 16, 4, 128, 128), 5e-3), ((512, 512, 8, 2, 256, 128), 5e-3)):
        st, eng, orc, keep, d = build(dims=dims, gguf=True, seed=12, kv_max=128)
        rng = np.random.default_rng(3)
        toks = [int(x) for x in rng.integers(0, d["V"], 80)]
        logits = np.empty(d["V"], F)
        st.prefill(toks, 0, logits.ctypes.data)
        ref = None
        for i, t in enumerate(toks):
            ref = orc.step(t, i)
        err = float(np.abs(logits - ref).max() / np.abs(ref).max())
        assert err <= 2e-4, err
        assert int(np.argmax(logits)) == int(np.argmax(ref))
        d["reset"]()
        st.set_attention_mode(False, gemm_fast=True)
        fast = np.empty(d["V"], F)
        st.prefill(toks, 0, fast.ctypes.data)
        errf = float(np.abs(fast - logits).max() / np.abs(logits).max())
        assert np.isfinite(fast).all() and errf <= bound_fast, errf


@pytest.mark.parametrize("hd,nh,fp8", [(128, 8, False), (256, 16, False), (256, 16, True), (128, 4, True), (64, 16, False)])
def test_prefill_wide_heads_long_prompt(hd, nh, fp8):
    """the head-dim / GQA-group / KV-dtype specialisations of the prompt-pass attention (QCN: head_dim 256, 8 query heads per KV head)
    on a prompt that spans several 64-position probability tiles, 32-position K tiles and chunks"""
    from oracle import oracle as O
    st, eng, orc, keep, d = build(seed=11, kv_max=260, hd=hd, nh=nh)
    if fp8:
        st.set_kv_dtype(True)
        rng = np.random.default_rng(5)
        kv = {li: (O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["nkv"] * d["hd"])) * 0.5).astype(F)), O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["nkv"] * d["hd"])) * 0.5).astype(F)))
              for li, kind in enumerate(d["kinds"]) if kind == "gqa"}
        n = len(d["kinds"]); ptr = lambda a: a.ctypes.data
        reset = lambda: st.set_decode_state(5, d["kv_max"], [ptr(kv[i][0]) if i in kv else 0 for i in range(n)], [ptr(kv[i][1]) if i in kv else 0 for i in range(n)],
                                            [ptr(x) if x is not None else 0 for x in d["state"]["conv"]], [ptr(x) if x is not None else 0 for x in d["state"]["recur"]])
    else:
        reset = d["reset"]
    reset()
    start, n_tok = 37, 170
    rng = np.random.default_rng(hd + nh)
    toks = [int(x) for x in rng.integers(0, d["V"], n_tok)]
    ref_logits = np.empty(d["V"], F)
    for i, t in enumerate(toks):
        st.decode_step(t, start + i, ref_logits.ctypes.data)
    ref_tok = st.last_token()
    esz = np.uint8 if fp8 else np.uint16
    def snap():
        out = []
        for li, kind in enumerate(d["kinds"]):
            if kind == "gqa":
                kc = np.empty((d["kv_max"], d["nkv"] * d["hd"]), esz); vc = np.empty_like(kc)
                st.get_decode_state(li, kc, vc, None, None); out.append((kc, vc))
        return out
    ref_kv = snap()
    reset()
    st.set_prefill_chunk(48); st.set_prefill_depth(3)
    logits = np.empty(d["V"], F)
    tok = st.prefill(toks, start, logits.ctypes.data)
    assert np.array_equal(logits.view(np.uint32), ref_logits.view(np.uint32)), float(np.max(np.abs(logits - ref_logits)))
    assert tok == ref_tok
    for a, b in zip(snap(), ref_kv):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_prefill_production_widths():
    """prompt pass == sequential decode at Qwen3-Coder-Next widths (hidden 2048, I 512, top-10 of 72, head_dim 256, 8 q heads per KV head)"""
    st, eng, orc, keep, d = build(seed=23, dims=(2048, 512, 72, 10, 512, 512), hd=256, nh=16, kv_max=160, kinds=["la", "gqa"])
    rng = np.random.default_rng(2)
    toks = [int(x) for x in rng.integers(0, d["V"], 90)]
    ref_logits = np.empty(d["V"], F)
    for i, t in enumerate(toks):
        st.decode_step(t, 11 + i, ref_logits.ctypes.data)
    ref_tok = st.last_token(); ref_state = _snapshot(st, d)
    d["reset"]()
    st.set_prefill_chunk(40); st.set_prefill_depth(3)
    logits = np.empty(d["V"], F)
    tok = st.prefill(toks, 11, logits.ctypes.data)
    assert np.array_equal(logits.view(np.uint32), ref_logits.view(np.uint32)), float(np.max(np.abs(logits - ref_logits)))
    assert tok == ref_tok
    for li, (a, b) in enumerate(zip(_snapshot(st, d), ref_state)):
        assert np.array_equal(a[0].view(np.uint32) if a[0].dtype == F else a[0], b[0].view(np.uint32) if b[0].dtype == F else b[0]), ("state0", li)
        assert np.array_equal(a[1].view(np.uint32) if a[1].dtype == F else a[1], b[1].view(np.uint32) if b[1].dtype == F else b[1]), ("state1", li)


def test_prefill_argument_errors():
    st, eng, orc, keep, d = build()
    with pytest.raises(ValueError):
        st.prefill([], 0)
    with pytest.raises(ValueError):
        st.prefill([1, 2, 3], d["kv_max"] - 1)            # does not fit kv_max_seq
    with pytest.raises(ValueError):
        st.prefill([d["V"] + 5], 0)                       # token id out of range
