"""Pins the CPU oracle against the reference's OWN unit tests (same closed-form inputs,
same assertions/thresholds).  Each test cites the Rust #[test] it restates.

The Rust crate cannot be compiled here (no rustc), so these known-answer / invariance
tests -- not a reference binary -- are what pins the oracle.
"""
import numpy as np
import pytest

from oracle import oracle as O


def _w(n, k, mul=1, add=0, amp=0.2):
    i = np.arange(n * k, dtype=np.int64)
    v = ((i * mul + add).astype(np.float32) / np.float32(n * k) - np.float32(0.5)) * np.float32(amp)
    return O.f32_to_bf16(v).reshape(n, k)


def _a(k, mul=1, add=0, amp=2.0):
    i = np.arange(k, dtype=np.int64)
    v = ((i * mul + add).astype(np.float32) / np.float32(k) - np.float32(0.5)) * np.float32(amp)
    return O.f32_to_bf16(v)


def _bf16_matvec(w_bf16, a_bf16):
    w = O.bf16_to_f32(w_bf16); a = O.bf16_to_f32(a_bf16)
    out = np.zeros(w.shape[0], np.float32)
    for r in range(w.shape[0]):
        acc = np.float32(0)
        for c in range(w.shape[1]):
            acc = np.float32(acc + w[r, c] * a[c])
        out[r] = acc
    return out


# ---- src/gguf.rs:891-933
def test_ggml_type_sizes():
    assert O.block_size(O.Q4_K) == 256 and O.block_bytes(O.Q4_K) == 144
    assert O.block_bytes(O.Q5_K) == 176 and O.block_bytes(O.Q6_K) == 210
    assert O.block_size(O.F32) == 1 and O.block_bytes(O.F32) == 4
    assert O.block_bytes(O.Q8_0) == 34 and O.block_bytes(O.Q4_0) == 18 and O.block_bytes(O.Q5_0) == 22


def test_get_scale_min_k4():
    scales = np.array([0x3F, 0x20, 0x10, 0x08, 0x04, 0x02, 0x01, 0x80, 0, 0, 0, 0], np.uint8)
    assert O.get_scale_min_k4(0, scales) == (0x3F, 0x04)
    # j>=4 branch, by the formula at gguf.rs:670-671
    s = np.array([0xC1, 0x82, 0x43, 0x04, 0xF5, 0xA6, 0x77, 0x38, 0x9A, 0xBC, 0xDE, 0xF0], np.uint8)
    for j in range(4, 8):
        sc = (s[j + 4] & 0xF) | ((s[j - 4] >> 6) << 4)
        mn = (s[j + 4] >> 4) | ((s[j] >> 6) << 4)
        assert O.get_scale_min_k4(j, s) == (int(sc), int(mn))


def test_dequant_f32_bf16_roundtrip():
    v = np.array([1.0, -2.5, 3.14, 0.0], np.float32)
    assert np.array_equal(O.dequantize(O.F32, v.view(np.uint8), 4), v)
    v2 = np.array([1.0, -2.0, 0.0, 0.5], np.float32)
    bf = (v2.view(np.uint32) >> 16).astype(np.uint16)
    assert np.array_equal(O.dequantize(O.BF16, bf.view(np.uint8), 4), v2)


# ---- src/gguf_kernels.rs:769-828
def test_q4_0_scalar_roundtrip():
    blk = np.zeros(18, np.uint8)
    blk[0:2] = np.array([1.0], np.float16).view(np.uint8)
    blk[2:] = 0x88
    out = O.gguf_matvec_f32(O.Q4_0, blk, np.ones(32, np.float32), 1, 32)
    assert abs(out[0]) < 1e-6


def test_q8_0_scalar_simple():
    blk = np.zeros(34, np.uint8)
    blk[0:2] = np.array([0.1], np.float16).view(np.uint8)
    blk[2:] = 10
    out = O.gguf_matvec_f32(O.Q8_0, blk, np.ones(32, np.float32), 1, 32)
    assert abs(out[0] - 32.0) < 0.5


def test_quantize_bf16_roundtrip():
    k = 64
    val = (np.arange(k, dtype=np.float32) - 32.0) * np.float32(0.1)
    bf = (val.view(np.uint32) >> 16).astype(np.uint16)  # truncation, as the Rust test does
    q, s, sm = O.gguf_quant_bf16(bf)
    orig = O.bf16_to_f32(bf)
    for g in range(k // 32):
        rec = q[g * 32:(g + 1) * 32].astype(np.float32) * s[g]
        assert np.max(np.abs(orig[g * 32:(g + 1) * 32] - rec)) < 0.01
        assert sm[g] == int(q[g * 32:(g + 1) * 32].astype(np.int64).sum())


# ---- src/weights/marlin.rs tests (quantize round trip)
@pytest.mark.parametrize("bits,tol", [(4, 0.02), (8, 0.002)])
def test_quantize_roundtrip(bits, tol):
    w = _w(8, 256, amp=0.2)
    if bits == 4:
        p, s = O.quantize_int4(w, 128); deq = O.dequantize_int4(p, s, 128)
    else:
        d, s = O.quantize_int8(w, 128); deq = O.dequantize_int8(d, s, 128)
    assert np.max(np.abs(deq - O.bf16_to_f32(w))) < tol


def test_quantize_int4_known_values():
    # one group whose amax = 0.875 -> scale = bf16(0.125) exactly -> q = round(v/0.125)
    v = np.zeros(128, np.float32); v[0] = 0.875; v[1] = -0.875; v[2] = 0.0625; v[3] = -0.0625; v[4] = 0.1875; v[5] = -1.0
    v[5] = -0.875
    p, s = O.quantize_int4(O.f32_to_bf16(v).reshape(1, 128), 128)
    assert O.bf16_to_f32(s)[0, 0] == 0.125
    nib = [(int(p[0, 0]) >> (4 * j)) & 0xF for j in range(8)]
    # 0.875/0.125=7 ; -7 ; 0.5 -> round-half-away = 1 ; -0.5 -> -1 ; 1.5 -> 2 ; -7
    assert nib[:6] == [15, 1, 9, 7, 10, 1]


# ---- src/kernel/avx2.rs tests
def test_scalar_int4_vs_avx2_synthetic_and_quant_error():      # avx2.rs:2382
    n, k = 16, 128
    w = _w(n, k, amp=0.2); a = _a(k, amp=2.0)
    ref = _bf16_matvec(w, a)
    p, s = O.quantize_int4(w, 128)
    q, qs = O.quant_act_int16_bf16(a, 128)
    out = O.matvec_int4_rowmajor(p, s, q, qs, 128)
    assert np.max(np.abs(ref - out)) < 1.0                      # :2430 quantization error bound


def test_activation_int16_quantization():                       # avx2.rs:2635
    k = 256
    a = _a(k, mul=7, add=3, amp=2.0)
    q, s = O.quant_act_int16_bf16(a, 128)
    orig = O.bf16_to_f32(a)
    rec = q.astype(np.float32) * np.repeat(s, 128)
    assert np.max(np.abs(orig - rec)) < 0.001


def test_integer_vs_fma_synthetic():                            # avx2.rs:2709 (rel RMSE < 1%)
    n, k = 32, 256
    w = _w(n, k, mul=3, add=7, amp=0.4); a = _a(k, mul=11, add=3, amp=0.5)
    p, s = O.quantize_int4(w, 128)
    deq = O.dequantize_int4(p, s, 128)
    fma = (deq.astype(np.float64) @ O.bf16_to_f32(a).astype(np.float64))
    q, qs = O.quant_act_int16_bf16(a, 128)
    out = O.matvec_int4_rowmajor(p, s, q, qs, 128)
    rel = np.sqrt(np.mean((fma - out) ** 2)) / np.sqrt(np.mean(fma ** 2))
    assert rel < 0.01


def test_transposed_matches_original_and_tiling_invariance():   # avx2.rs:2880,3054 (bit-identical)
    n, k = 1408, 2048
    rng = np.random.default_rng(0)
    w = O.f32_to_bf16(rng.standard_normal((n, k)).astype(np.float32) * 0.05)
    a = O.f32_to_bf16(rng.standard_normal(k).astype(np.float32))
    p, s = O.quantize_int4(w, 128)
    pt, st = O.transpose_int4(p, s, 128)
    q, qs = O.quant_act_int16_bf16(a, 128)
    out_t = O.matvec_int4_t(pt, st, q, qs, 128)
    # row-major scalar integer uses non-fused mul+add (avx2.rs:343); transposed AVX2 uses fma (:1175):
    out_r = O.matvec_int4_rowmajor(p, s, q, qs, 128)
    assert np.max(np.abs(out_t - out_r)) < 1e-3                 # the reference's scalar-vs-AVX2 tolerance
    # tiled layout + AVX2/OpenMP twin == scalar restatement, bit for bit (parallel == serial, :3110)
    tiled_p = O.repack_tiled_u32(pt); tiled_s = O.repack_tiled_u16(st)
    out_avx = O.matvec_int4_tiled_avx2(tiled_p, tiled_s, q, qs, k, n, 128, parallel=True)
    out_ser = O.matvec_int4_tiled_avx2(tiled_p, tiled_s, q, qs, k, n, 128, parallel=False)
    assert np.array_equal(out_avx, out_ser)
    assert np.array_equal(out_avx, out_t)


def test_transposed_int8_matches_dequant():                     # avx2.rs:3114
    n, k = 64, 256
    w = _w(n, k, mul=5, add=1, amp=0.3); a = _a(k, mul=3, add=1, amp=1.0)
    d, s = O.quantize_int8(w, 128)
    dt, st = O.transpose_int8(d, s, 128)
    q, qs = O.quant_act_int16_bf16(a, 128)
    out = O.matvec_int8_t(dt, st, q, qs, 128)
    deq = O.dequantize_int8(d, s, 128).astype(np.float64)
    ref = deq @ (q.astype(np.float64) * np.repeat(qs, 128))
    assert np.max(np.abs(out - ref)) < 0.001


# ---- src/moe.rs:3306-3482 synthetic expert forward sanity
def test_expert_forward_sanity_and_int_vs_f64():
    rng = np.random.default_rng(1)
    H, I = 256, 128
    g = O.f32_to_bf16(rng.standard_normal((I, H)).astype(np.float32) * 0.05)
    u = O.f32_to_bf16(rng.standard_normal((I, H)).astype(np.float32) * 0.05)
    d = O.f32_to_bf16(rng.standard_normal((H, I)).astype(np.float32) * 0.05)
    e = O.unified_from_bf16(g, u, d)
    x = O.f32_to_bf16(rng.standard_normal(H).astype(np.float32))
    q, qs = O.quant_act_int16_bf16(x)
    out = O.expert_forward_unified(e, q, qs)
    # f64 evaluation of the same quantized weights
    W13 = np.concatenate([O.dequantize_int4(*O.quantize_int4(g)), O.dequantize_int4(*O.quantize_int4(u))]).astype(np.float64)
    W2 = O.dequantize_int4(*O.quantize_int4(d)).astype(np.float64)
    xf = O.bf16_to_f32(x).astype(np.float64)
    gu = W13 @ xf
    hid = gu[:I] / (1 + np.exp(-gu[:I])) * gu[I:]
    ref = W2 @ hid
    rel = np.sqrt(np.mean((ref - out) ** 2)) / np.sqrt(np.mean(ref ** 2))
    assert rel < 0.01 and np.max(np.abs(ref - out)) < 0.01


# ---- sigmoid flavours: the rcp+Newton production form vs the exact-divide form the HIP kernels implement
def test_sigmoid_modes_agree_to_2ulp():
    xs = np.linspace(-25, 25, 20001).astype(np.float32)
    a = np.array([O.sigmoid(float(x), O.SIG_POLY5_DIV) for x in xs], np.float32)
    b = np.array([O.sigmoid(float(x), O.SIG_POLY5_RCPNR) for x in xs], np.float32)
    ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    assert ulp.max() <= 2
    c = np.array([O.sigmoid(float(x), O.SIG_LIBM) for x in xs], np.float32)
    inside = np.abs(xs) <= 20                                   # poly versions clamp to +-20 (avx2.rs:2278)
    assert np.max(np.abs(a - c)[inside]) < 5e-5                 # degree-5 minimax error of the reference's own poly


# ---- routers (decode.rs:1495 heap semantics, moe.rs:3223 iterative argmax)
def test_topk_tie_rules():
    v = np.array([0.5, 0.9, 0.9, 0.1, 0.9, 0.5, 0.2, 0.9], np.float32)
    # decode graph rule: heap seeded with 0..k, strict '>' replacement, stable desc sort
    ids = O.topk_indices(v, 3)
    assert sorted(v[ids].tolist(), reverse=True) == [np.float32(0.9)] * 3
    # heap [(.5,0),(.9,1),(.9,2)] -> idx4 replaces the min at the root; equal values keep HEAP order after the
    # stable descending sort (decode.rs:1531), i.e. [4,1,2], not index order; idx7 never enters (strict >)
    assert ids.tolist() == [4, 1, 2]
    # engine rule: iterative argmax, lowest index wins
    gate = np.zeros((8, 4), np.float32); gate[:, 0] = v
    act = np.array([1, 0, 0, 0], np.float32)
    ids2, w2 = O.route_engine(O.f32_to_bf16(gate), O.f32_to_bf16(act), 3, sigmoid=True, norm_topk=False)
    assert ids2.tolist() == [1, 2, 4]


def test_route_matmul_matches_f64():
    rng = np.random.default_rng(2)
    g = (rng.random((512, 2048), dtype=np.float32) - 0.5) * 0.04
    h = (rng.random(2048, dtype=np.float32) - 0.5)
    lg = O.route_matmul(g, h)
    ref = g.astype(np.float64) @ h.astype(np.float64)
    assert np.max(np.abs(lg - ref)) < 1e-5
    ids, w, _ = O.route_decode(g, h, 10, scoring=1, norm_topk=True)
    assert len(set(ids.tolist())) == 10 and abs(float(w.sum()) - 1.0) < 1e-5
    top_ref = np.argsort(-ref)[:10]
    assert set(ids.tolist()) == set(top_ref.tolist())


def test_reduce_sum_bf16():
    rng = np.random.default_rng(3)
    ins = [O.f32_to_bf16(rng.standard_normal(100).astype(np.float32)) for _ in range(3)]
    out = O.reduce_sum_bf16(ins)
    acc = np.zeros(100, np.float32)
    for a in ins:
        acc = (acc + O.bf16_to_f32(a)).astype(np.float32)
    assert np.array_equal(out, O.f32_to_bf16(acc))


def test_xorshift_matches_definition():
    r = O.Xorshift64()
    x = O.Xorshift64.SEED
    for _ in range(5):
        x ^= (x << 13) & 0xFFFFFFFFFFFFFFFF; x ^= x >> 7; x ^= (x << 17) & 0xFFFFFFFFFFFFFFFF
        assert r.next_u64() == x
