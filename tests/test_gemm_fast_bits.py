"""The de-quantization arithmetic of the tolerance GEMM (krasis_amd/csrc/kr_prefill_h.hip: pfh_dq4 / pfh_dq8 / the A permutation), restated in
numpy bit for bit (f16 fma = exact product + sum in float64, one rounding to f16).  No GPU: this pins the CLAIMS the kernel's header makes --
an INT4 weight becomes (nibble - 8) * s * 16 exactly for every nibble and every bf16 scale in the normal f16 range, an INT8 weight becomes
b exactly before its single rounding multiply, and the (0,4,1,5,2,6,3,7) order of the unpacked values is the order the A row image is stored in (the INT8
kernel's commit pass restores natural order)."""
import numpy as np

M0, M1, MH, KC = 0x000F000F, 0x00F000F0, 0x03C003C0, 0x64006400


def _h(x16):
    return x16.astype(np.uint16).view(np.float16).astype(np.float64)


def _dq4(w, s_bf16_bits):
    """pfh_dq4: 8 f16 values of one packed word, in the kernel's output order"""
    s = (s_bf16_bits.astype(np.uint32) << 16).view(np.float32)
    sq = (s * np.float32(0.25)).astype(np.float16)                       # (_Float16)(sc * 0.25f)
    cq = (np.float32(-1536.0) * sq.astype(np.float32)).astype(np.float16)
    t = [((w & M0) << 6) | KC, ((w & M1) << 2) | KC, ((w >> 2) & MH) | KC, ((w >> 6) & MH) | KC]
    out = np.zeros((len(w), 8), np.float16)
    for p in range(4):
        for half in range(2):
            v = _h((t[p] >> (16 * half)) & 0xFFFF)
            out[:, 2 * p + half] = (v * sq.astype(np.float64) + cq.astype(np.float64)).astype(np.float16)     # v_pk_fma_f16: one rounding
    return out, s, sq, cq


def test_int4_weights_are_dequantized_exactly():
    rng = np.random.default_rng(0)
    w = rng.integers(0, 2 ** 32, size=20000, dtype=np.uint64).astype(np.uint32)
    # every bf16 scale from 2^-12 to 64 (normal in f16 after the factor 1/4, no overflow of 1984 * s / 4): sign 0, exponent 115..133, all 7 mantissa bits
    exps = rng.integers(115, 134, size=20000); man = rng.integers(0, 128, size=20000)
    sb = ((exps << 7) | man).astype(np.uint16)
    out, s, sq, cq = _dq4(w, sb)
    assert np.array_equal(sq.astype(np.float32), s * np.float32(0.25))                    # s / 4 is exact in f16
    assert np.array_equal(cq.astype(np.float64), -1536.0 * sq.astype(np.float64))         # and so is -1536 * s / 4 (2 + 8 significant bits)
    order = [0, 4, 1, 5, 2, 6, 3, 7]
    for pos, k in enumerate(order):
        n = ((w >> (4 * k)) & 15).astype(np.float64)
        assert np.array_equal(out[:, pos].astype(np.float64), (n - 8.0) * 16.0 * s.astype(np.float64)), (pos, k)
    # all 16 nibble values occur at every position in a sample of this size
    assert all(len(np.unique((w >> (4 * k)) & 15)) == 16 for k in range(8))


def test_nibble_at_mantissa_bits_0_3_would_not_be_exact():
    """the design note in the kernel header: with the nibble at bits 0..3 the fma constant 1032 * s needs 16 significant bits and is rounded"""
    sb = np.array([0x3C2B, 0x3B55, 0x3D7F], np.uint16)
    s = (sb.astype(np.uint32) << 16).view(np.float32)
    c = (np.float32(-1032.0) * s.astype(np.float16).astype(np.float32)).astype(np.float16)
    assert np.any(c.astype(np.float64) != -1032.0 * s.astype(np.float16).astype(np.float64))


def _perm(s0, s1, sel):
    """v_perm_b32: selector bytes 0-3 pick from s1, 4-7 from s0"""
    by = np.concatenate([s1.view(np.uint8).reshape(-1, 4), s0.view(np.uint8).reshape(-1, 4)], axis=1)
    r = np.zeros(len(s0), np.uint32)
    for i in range(4):
        r |= by[:, (sel >> (8 * i)) & 0xFF].astype(np.uint32) << np.uint32(8 * i)
    return r


def test_int8_weights_and_the_a_permutation():
    rng = np.random.default_rng(1)
    b = rng.integers(-128, 128, size=(5000, 8)).astype(np.int8)
    w = np.ascontiguousarray(b).view(np.uint32)                  # [5000, 2] words, natural k order
    kc = np.full(5000, 0x64646464, np.uint32)
    vals = []
    for word in (w[:, 0] ^ np.uint32(0x80808080), w[:, 1] ^ np.uint32(0x80808080)):
        word = np.ascontiguousarray(word)
        for sel in (0x04010400, 0x04030402):
            t = _perm(kc, word, sel)
            vals += [_h(t & 0xFFFF) - 1152.0, _h(t >> 16) - 1152.0]
    got = np.stack(vals, axis=1)
    assert np.array_equal(got, b.astype(np.float64))             # (1024 + (b ^ 0x80)) - 1152 = b, natural order, exact in f16
    # A rows (round 6): the row kernels store the 8 values of a group in the order pfh_dq4 emits, (a0,a4 | a1,a5 | a2,a6 | a3,a7) -- the INT4 kernels (register-staged
    # and LDS-ring: LDS-DMA cannot permute) copy rows verbatim; the INT8 kernel, whose pfh_dq8 emits natural k order, un-permutes in its commit pass with four v_perm
    a = np.arange(8, dtype=np.uint16)[None, :].repeat(3, 0) + np.array([[0], [100], [200]], np.uint16)
    image = np.ascontiguousarray(a[:, [0, 4, 1, 5, 2, 6, 3, 7]])
    v = image.view(np.uint32)
    o = np.stack([_perm(np.ascontiguousarray(v[:, 1]), np.ascontiguousarray(v[:, 0]), 0x05040100), _perm(np.ascontiguousarray(v[:, 3]), np.ascontiguousarray(v[:, 2]), 0x05040100),
                  _perm(np.ascontiguousarray(v[:, 1]), np.ascontiguousarray(v[:, 0]), 0x07060302), _perm(np.ascontiguousarray(v[:, 3]), np.ascontiguousarray(v[:, 2]), 0x07060302)], axis=1)
    assert np.array_equal(np.ascontiguousarray(o).view(np.uint16), a)


def test_row_multiplier_is_a_power_of_two():
    """pfh_row_scale: largest |value| of the row lands in [1, 2); multiplier and inverse are exact powers of two"""
    mx = np.array([3e-5, 0.7, 1.0, 1.999, 2.0, 3000.0, 6.5e4, 1e10], np.float32)
    E = mx.view(np.uint32) >> 23
    scl = ((254 - E) << 23).astype(np.uint32).view(np.float32); inv = (E << 23).astype(np.uint32).view(np.float32)
    assert np.all(scl * inv == 1.0)
    assert np.all((mx * scl >= 1.0) & (mx * scl < 2.0))
