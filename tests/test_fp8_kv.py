"""FP8-E4M3 KV cache option (the reference's GPU cache dtype, python/krasis/kv_cache.py:38-135): codecs pinned against torch.float8_e4m3fn
(CPU), decode step and prompt pass with FP8 caches bit-exact against the oracle (GPU)."""
import numpy as np
import pytest

from oracle import oracle as O

F = np.float32


def test_e4m3_codecs_match_torch():
    import torch
    codes = np.arange(256, dtype=np.uint8)
    ref = torch.from_numpy(codes).view(torch.float8_e4m3fn).to(torch.float32).numpy()
    got = O.e4m3_to_f32(codes)
    nan = np.isnan(ref)
    assert np.array_equal(np.isnan(got), nan) and np.array_equal(got[~nan].view(np.uint32), ref[~nan].view(np.uint32))
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(4000).astype(F) * s for s in (1e-3, 0.02, 0.5, 8.0, 200.0)] +
                       [np.array([0.0, -0.0, 448.0, 464.0, 465.0, 479.9, 480.0, 1e9, -1e9, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 0.017578125], F)])
    want = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = O.f32_to_e4m3(x)
    assert np.array_equal(got, want), np.where(got != want)[0][:10]


@pytest.mark.gpu
@pytest.mark.parametrize("graph,kv_max,positions", [(True, 32, [5, 6, 7]), (False, 32, [5, 6, 7]), (True, 300, [127, 128, 129, 257, 299])])
def test_decode_step_fp8_kv_bit_exact(graph, kv_max, positions):
    from tests.test_decode_gpu import build
    st, eng, orc, keep, d = build(kv_max=kv_max)
    st.set_kv_dtype(True); O.set_kv_fp8(True)
    try:
        rng = np.random.default_rng(5)
        kv = {}
        for li, kind in enumerate(d["kinds"]):
            if kind == "gqa":
                kc = O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["nkv"] * d["hd"])) * 0.5).astype(F)); vc = O.f32_to_e4m3((rng.standard_normal((d["kv_max"], d["nkv"] * d["hd"])) * 0.5).astype(F))
                kv[li] = (kc, vc)
                orc.layers[li]["kv_k"] = kc.astype(np.uint16); orc.layers[li]["kv_v"] = vc.astype(np.uint16)   # oracle keeps one byte per u16 slot
        n = len(d["kinds"])
        ptr = lambda a: a.ctypes.data
        st.set_decode_state(5, d["kv_max"], [ptr(kv[i][0]) if i in kv else 0 for i in range(n)], [ptr(kv[i][1]) if i in kv else 0 for i in range(n)],
                            [ptr(x) if x is not None else 0 for x in d["state"]["conv"]], [ptr(x) if x is not None else 0 for x in d["state"]["recur"]])
        st.set_use_graph(graph)
        tok = 7
        for step, pos in enumerate(positions):
            logits = np.empty(d["V"], F)
            st.decode_step(tok, pos, logits.ctypes.data)
            ref = orc.step(tok, pos)
            assert np.array_equal(logits.view(np.uint32), ref.view(np.uint32)), (step, float(np.max(np.abs(logits - ref))))
            tok = st.last_token()
        for li in kv:
            kc = np.empty((d["kv_max"], d["nkv"] * d["hd"]), np.uint8); vc = np.empty_like(kc)
            st.get_decode_state(li, kc, vc, None, None)
            assert np.array_equal(kc, orc.layers[li]["kv_k"].astype(np.uint8)) and np.array_equal(vc, orc.layers[li]["kv_v"].astype(np.uint8))
        # prompt pass with FP8 caches == decoding the prompt token by token
        toks = [3, 9, 27, 81, 5, 15]
        st.set_decode_state(5, d["kv_max"], [ptr(kv[i][0]) if i in kv else 0 for i in range(n)], [ptr(kv[i][1]) if i in kv else 0 for i in range(n)],
                            [ptr(x) if x is not None else 0 for x in d["state"]["conv"]], [ptr(x) if x is not None else 0 for x in d["state"]["recur"]])
        seq = np.empty(d["V"], F)
        for i, t in enumerate(toks):
            st.decode_step(t, 8 + i, seq.ctypes.data)
        st.set_decode_state(5, d["kv_max"], [ptr(kv[i][0]) if i in kv else 0 for i in range(n)], [ptr(kv[i][1]) if i in kv else 0 for i in range(n)],
                            [ptr(x) if x is not None else 0 for x in d["state"]["conv"]], [ptr(x) if x is not None else 0 for x in d["state"]["recur"]])
        st.set_prefill_chunk(4)
        pf = np.empty(d["V"], F); st.prefill(toks, 8, pf.ctypes.data)
        assert np.array_equal(pf.view(np.uint32), seq.view(np.uint32))
    finally:
        O.set_kv_fp8(False)
