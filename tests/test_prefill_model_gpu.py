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
