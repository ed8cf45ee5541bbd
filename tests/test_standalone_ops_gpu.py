"""The reference's stand-alone CpuDecodeStore operators (src/decode.rs:328-1086), the cancellable generate_stream loop (decode.rs:3611), the
`krasis.krasis` names (src/lib.rs:14-24), bench_decode_synthetic (decode.rs:4618) and the call sequence of the reference's own binding test
(tests/test_pyo3.py) -- through libkrasis_hip.so.  Operators are compared BIT FOR BIT with the oracle's restatements: kro_op_* for the scalar-loop
methods (which differ from the decode graph's AVX2 arithmetic exactly as they do in the reference), the graph's functions for the shared ones.
Every operator is called once with host pointers (what the reference's Python callers pass) and once with device pointers."""
import os
import struct

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu
F = np.float32
ptr = lambda a: a.ctypes.data


def store(norm_bias_one=False):
    from krasis_amd import CpuDecodeStore
    return CpuDecodeStore(128, True, norm_bias_one)


def same(a, b):
    return np.array_equal(np.ascontiguousarray(a, F).view(np.uint32), np.ascontiguousarray(b, F).view(np.uint32))


@pytest.mark.parametrize("bits", [4, 8])
def test_matmul_and_matmul_batch_bit_exact(bits):
    import torch
    rng = np.random.default_rng(bits)
    st = store()
    K, N1, N2 = 512, 96, 200
    w1 = (rng.standard_normal((N1, K)) * 0.05).astype(F); w2 = (rng.standard_normal((N2, K)) * 0.05).astype(F)
    a = st.store_weight_f32(ptr(w1), N1, K, bits); b = st.store_weight_f32(ptr(w2), N2, K, bits)
    assert st.num_weights() == 2
    x = rng.standard_normal(K).astype(F)
    q = O.quantize_f32_to_transposed_int4 if bits == 4 else O.quantize_f32_to_transposed_int8
    mv = O.matvec_int4_t if bits == 4 else O.matvec_int8_t
    xa, xs = O.quant_act_int16_f32(x)
    refs = [mv(*q(w), xa, xs) for w in (w1, w2)]
    y1 = np.empty(N1, F); st.matmul(a, ptr(x), ptr(y1)); assert same(y1, refs[0])
    y1b, y2b = np.empty(N1, F), np.empty(N2, F)
    st.matmul_batch([a, b], ptr(x), [ptr(y1b), ptr(y2b)])
    assert same(y1b, refs[0]) and same(y2b, refs[1])
    xd = torch.from_numpy(x).cuda(); yd = torch.empty(N2, dtype=torch.float32, device="cuda")     # device pointers
    st.matmul(b, xd.data_ptr(), yd.data_ptr()); torch.cuda.synchronize()
    assert same(yd.cpu().numpy(), refs[1])
    # bytes of one stored weight: packed words * 4 + bf16 scales * 2 (decode.rs:1107)
    assert st.weight_bytes(a) == (K // 8 if bits == 4 else K // 4) * N1 * 4 + (K // 128) * N1 * 2
    with pytest.raises(ValueError, match="out of range"):
        st.matmul(7, ptr(x), ptr(y1))
    with pytest.raises(ValueError, match="same length"):
        st.matmul_batch([a, b], ptr(x), [ptr(y1b)])


@pytest.mark.parametrize("bias_one", [False, True])
def test_norm_operators_bit_exact(bias_one):
    rng = np.random.default_rng(3)
    st = store(bias_one)
    n = 1000
    h = rng.standard_normal(n).astype(F); r = rng.standard_normal(n).astype(F); w = (rng.random(n) * 0.4 + (0.0 if bias_one else 0.8)).astype(F)
    nid = st.store_norm_weight(ptr(w), n)
    for first in (True, False):
        hh, rr = h.copy(), r.copy()
        st.fused_add_rmsnorm(ptr(hh), ptr(rr), ptr(w), 1e-6, n, first)
        eh, er = O.fused_add_rmsnorm(h.copy(), r.copy(), w, 1e-6, first, bias_one)
        assert same(hh, eh) and same(rr, er), first
        h2, r2 = h.copy(), r.copy()
        st.fused_add_rmsnorm_id(ptr(h2), ptr(r2), nid, 1e-6, n, first)
        assert same(h2, eh) and same(r2, er)
    out = np.empty(n, F)
    st.rmsnorm(ptr(h), ptr(w), 1e-5, ptr(out), n)
    assert same(out, O.op_rmsnorm(h, w, 1e-5, bias_one))
    with pytest.raises(ValueError, match="norm_id 5 out of range"):
        st.fused_add_rmsnorm_id(ptr(h), ptr(r), 5, 1e-6, n, False)


def test_silu_mul_and_fused_shared_expert_bit_exact():
    rng = np.random.default_rng(5)
    st = store()
    n = 777
    g = (rng.standard_normal(n) * 3).astype(F); u = rng.standard_normal(n).astype(F); out = np.empty(n, F)
    st.silu_mul(ptr(g), ptr(u), ptr(out), n)
    assert same(out, O.op_silu_mul(g, u))
    H, I = 256, 128
    wgu = (rng.standard_normal((2 * I, H)) * 0.08).astype(F); wd = (rng.standard_normal((H, I)) * 0.08).astype(F)
    a = st.store_weight_f32(ptr(wgu), 2 * I, H, 4); b = st.store_weight_f32(ptr(wd), H, I, 4)
    x = rng.standard_normal(H).astype(F); y = np.empty(H, F)
    st.fused_shared_expert(a, b, ptr(x), ptr(y))
    xa, xs = O.quant_act_int16_f32(x)
    gu = O.matvec_int4_t(*O.quantize_f32_to_transposed_int4(wgu), xa, xs)
    hid = O.op_silu_mul(gu[:I], gu[I:])                               # decode.rs:569-574: exact sigmoid, not the graph's polynomial
    ha, hs = O.quant_act_int16_f32(hid)
    assert same(y, O.matvec_int4_t(*O.quantize_f32_to_transposed_int4(wd), ha, hs))


@pytest.mark.parametrize("nk,nv,dk,dv,kd", [(2, 4, 128, 128, 4), (2, 2, 64, 128, 4), (1, 2, 128, 128, 3)])
def test_linear_attention_operators_bit_exact(nk, nv, dk, dv, kd):
    import torch
    rng = np.random.default_rng(nk * 10 + nv + kd)
    st = store()
    hr = nv // nk; conv_dim = 2 * nk * dk + nv * dv; group_dim = 2 * dk + 2 * dv * hr
    qkvz = rng.standard_normal(nk * group_dim).astype(F); ba = rng.standard_normal(nk * 2 * hr).astype(F)
    cs = (rng.standard_normal(conv_dim * kd) * 0.5).astype(F); cw = (rng.standard_normal(conv_dim * kd) * 0.5).astype(F)
    a_log = (rng.random(nv) * 2 - 1).astype(F); dtb = (rng.random(nv) - 0.5).astype(F); scale = float(1.0 / np.sqrt(dk))
    q, k = np.empty(nv * dk, F), np.empty(nv * dk, F); v, z = np.empty(nv * dv, F), np.empty(nv * dv, F); g, beta = np.empty(nv, F), np.empty(nv, F)
    cs_dev = cs.copy()
    st.linear_attention_conv(ptr(qkvz), ptr(ba), ptr(cs_dev), ptr(cw), ptr(a_log), ptr(dtb), scale, ptr(q), ptr(k), ptr(v), ptr(z), ptr(g), ptr(beta),
                             nk, nv, dk, dv, hr, kd)
    eq, ek, ev, ez, eg, eb, ecs = O.op_la_conv(qkvz, ba, cs, cw, a_log, dtb, scale, nk, nv, dk, dv, hr, kd)
    for got, exp, name in ((q, eq, "q"), (k, ek, "k"), (v, ev, "v"), (z, ez, "z"), (g, eg, "g"), (beta, eb, "beta"), (cs_dev, ecs, "conv_state")):
        assert same(got, exp), name
    # recurrence (decode.rs:609 -> linear_attention_recurrent_avx2, the function decode_step shares) on the conv outputs
    state = (rng.standard_normal(nv * dk * dv) * 0.02).astype(F); out = np.empty(nv * dv, F)
    s_dev = state.copy()
    st.linear_attention_recurrent(ptr(s_dev), ptr(q), ptr(k), ptr(v), ptr(g), ptr(beta), ptr(out), nv, dk, dv)
    o_ref, s_ref = O.la_recurrent(state, q, k, v, g, beta, nv, dk, dv)
    assert same(out, o_ref) and same(s_dev, s_ref)
    # gated RMSNorm + SiLU gate (decode.rs:650): scalar loops, libm exp
    nw = (rng.random(nv * dv) + 0.5).astype(F); go = np.empty(nv * dv, F)
    st.gated_rmsnorm_silu(ptr(out), ptr(z), ptr(nw), ptr(go), 1e-6, nv, dv)
    assert same(go, O.op_gated_rmsnorm_silu(out, z, nw, 1e-6, nv, dv))
    # device pointers
    od = torch.from_numpy(out).cuda(); zd = torch.from_numpy(z).cuda(); wd = torch.from_numpy(nw).cuda(); gd = torch.empty(nv * dv, dtype=torch.float32, device="cuda")
    st.gated_rmsnorm_silu(od.data_ptr(), zd.data_ptr(), wd.data_ptr(), gd.data_ptr(), 1e-6, nv, dv); torch.cuda.synchronize()
    assert same(gd.cpu().numpy(), go)


@pytest.mark.parametrize("scoring,norm,use_bias,use_esc,bf16_gate", [(1, True, False, False, True), (0, True, True, True, False), (2, False, False, False, True),
                                                                     (1, False, True, False, False), (0, False, False, True, True)])
def test_moe_route_bit_exact(scoring, norm, use_bias, use_esc, bf16_gate):
    rng = np.random.default_rng(scoring * 7 + use_bias)
    st = store()
    E, H, k = 72, 256, 6
    gate = (rng.standard_normal((E, H)) * 0.05).astype(F)
    if bf16_gate:
        gate = O.bf16_to_f32(O.f32_to_bf16(gate)).astype(F)
    bias = (rng.standard_normal(E) * 0.1).astype(F) if use_bias else None
    esc = (rng.standard_normal(E) * 0.05).astype(F) if use_esc else None
    rid = st.store_route_weight(ptr(gate), E, H, ptr(bias) if use_bias else None, E if use_bias else 0, ptr(esc) if use_esc else None, E if use_esc else 0)
    assert rid == 0 and st.num_route_weights() == 1
    for t in range(3):
        x = rng.standard_normal(H).astype(F)
        ids = np.empty(k, np.int32); w = np.empty(k, F)
        st.moe_route(rid, ptr(x), ptr(ids), ptr(w), k, scoring, norm)
        eids, ew, _ = O.route_decode(gate, x, k, scoring, norm, bias, esc)
        assert np.array_equal(ids, np.asarray(eids, np.int32)), (t, ids, eids)
        assert same(w, ew), t
    with pytest.raises(ValueError, match="route_id 3 out of range"):
        st.moe_route(3, ptr(x), ptr(ids), ptr(w), k, scoring, norm)
    with pytest.raises(ValueError, match="Unknown scoring_func"):
        st.moe_route(rid, ptr(x), ptr(ids), ptr(w), k, 9, norm)


def test_generate_stream_cancel_and_elapsed():
    from tests.test_decode_gpu import build
    st, eng, orc, keep, d = build(seed=2, kv_max=64)
    # reference order of the greedy tokens
    toks = st.generate_batch(5, 4, 8, 0.0, 0, 1.0, [])
    assert st.last_decode_elapsed_s > 0.0
    d["reset"]()
    seen = []
    n = st.generate_stream(5, 4, 8, 0.0, 0, 1.0, [], None, 0.0, lambda t, text, why: seen.append((t, text, why)) or True)
    assert n == 8 and [s[0] for s in seen] == toks and [s[2] for s in seen] == [None] * 7 + ["length"] and all(s[1] == "" for s in seen)
    # a stop id ends the run and is reported with reason "stop" (decode.rs:3679)
    d["reset"](); seen.clear()
    n = st.generate_stream(5, 4, 8, 0.0, 0, 1.0, [toks[2]], None, 0.0, lambda t, text, why: seen.append((t, why)) or True)
    assert n == 3 and seen[-1] == (toks[2], "stop")
    # the callback returning False cancels after that token (decode.rs:3690)
    d["reset"](); seen.clear()
    n = st.generate_stream(5, 4, 8, 0.0, 0, 1.0, [], None, 0.0, lambda t, text, why: (seen.append(t) or len(seen) < 2))
    assert n == 2 and seen == toks[:2]
    # cancel(): the flag is polled before every step; the run reports "cancelled" with the last token and generates nothing (decode.rs:3650)
    d["reset"](); seen.clear()
    st.cancel()
    n = st.generate_stream(5, 4, 8, 0.0, 0, 1.0, [], None, 0.0, lambda t, text, why: seen.append((t, why)) or True)
    assert n == 0 and seen == [(5, "cancelled")]
    st.reset_cancel(); seen.clear()
    n = st.generate_stream(5, 4, 2, 0.0, 0, 1.0, [], None, 0.0, lambda t, text, why: seen.append((t, why)) or True)
    assert n == 2 and [s[0] for s in seen] == toks[:2]
    # a tokenizer's text reaches the callback; an exception in the callback surfaces in Python after the loop stops
    class Tok:
        def decode(self, ids, skip_special_tokens=True):
            return "<%d>" % ids[0]
    d["reset"](); seen.clear()
    st.generate_stream(5, 4, 2, 0.0, 0, 1.0, [], Tok(), 0.0, lambda t, text, why: seen.append(text) or True)
    assert seen == ["<%d>" % t for t in toks[:2]]
    with pytest.raises(RuntimeError, match="boom"):
        st.generate_stream(5, 4, 2, 0.0, 0, 1.0, [], None, 0.0, lambda t, text, why: (_ for _ in ()).throw(RuntimeError("boom")))


def test_krasis_names_and_pyo3_sequence(tmp_path):
    """tests/test_pyo3.py of the reference, step by step, on a tiny DeepSeek-V2-Lite-shaped checkpoint (the reference test needs the real model)"""
    from krasis import KrasisEngine                                   # Test 1: import
    from krasis.krasis import CpuDecodeStore, WeightStore, bench_decode_synthetic    # src/lib.rs:17-22
    from tests.tiny_checkpoints import make_v2lite_tiny
    assert WeightStore() is not None and callable(bench_decode_synthetic) and CpuDecodeStore is not None
    engine = KrasisEngine(parallel=True)                               # Test 2
    assert engine.is_parallel() in (True, False)
    with pytest.raises(RuntimeError):                                  # Test 3: error before loading
        engine.hidden_size()
    model_dir = str(tmp_path / "v2l"); make_v2lite_tiny(model_dir)
    engine.load(model_dir, group_size=128)                             # Test 4
    H = engine.hidden_size()
    assert H > 0 and engine.num_experts() > 0 and engine.top_k() > 0 and engine.num_moe_layers() > 0
    act = bytearray(H * 2)                                             # Test 5: the reference's synthetic bf16 activation
    for i in range(H):
        val = ((i * 7 + 13) / H - 0.5) * 0.1
        act[2 * i:2 * i + 2] = struct.pack("f", val)[2:4]
    k = engine.top_k()
    idx = list(range(k)); wts = [1.0 / k] * k
    out_bytes = engine.moe_forward(0, bytes(act), idx, wts)
    assert len(out_bytes) == H * 4
    out = np.frombuffer(out_bytes, F)
    assert float(np.sqrt((out ** 2).mean())) > 1e-5 and int((np.abs(out) > 1e-10).sum()) > H // 2
    with pytest.raises(ValueError):                                    # Test 6: validation
        engine.moe_forward(0, b"too_short", [0], [1.0])
    with pytest.raises(ValueError):
        engine.moe_forward(0, bytes(act), [0, 1], [1.0])
    for _ in range(5):                                                 # Test 7: repeated calls give the same bytes
        assert engine.moe_forward(0, bytes(act), idx, wts) == out_bytes


@pytest.mark.parametrize("maker", ["qcn", "v2l"])
def test_bench_decode_synthetic_runs_from_config_json(tmp_path, maker, capsys):
    from krasis_amd import bench_decode_synthetic
    from tests.tiny_checkpoints import make_qcn_tiny, make_v2lite_tiny
    d = str(tmp_path / maker); (make_qcn_tiny if maker == "qcn" else make_v2lite_tiny)(d)
    r = bench_decode_synthetic(os.path.join(d, "config.json"), num_steps=12, warmup=3, timing=True, num_bits=4, max_experts=4)
    err = capsys.readouterr().err
    assert "=== RESULTS (12 steps) ===" in err and "Per token:" in err and "Speed:" in err       # decode.rs:5515-5518
    assert r["tok_per_s"] > 0 and r["steps"] == 12 and sum(r["per_kind_ms"].values()) > 0
    with pytest.raises(IOError):
        bench_decode_synthetic(os.path.join(d, "nope.json"))
    bad = tmp_path / "bad.json"; bad.write_text("{not json")
    with pytest.raises(ValueError, match="Invalid JSON"):
        bench_decode_synthetic(str(bad))
