"""The AVX2 + OpenMP forms of the oracle's Q4_K / Q8_0 integer rows (oracle/krasis_oracle.c q4k_row_avx2 / q8_0_row_avx2, the intrinsics of
matvec_q4_k_avx2 / matvec_q8_0_avx2, gguf_kernels.rs:271-432) against the scalar lane-by-lane restatement the parity tests read: bit-identical
outputs on random raw blocks (every scale / min / quant byte pattern), on ragged shapes, and through the whole expert (gguf_kernels.rs:690).
bench.py's config-1 CPU leg runs on the AVX2 forms; this file is what makes that leg the same arithmetic as the oracle."""
import numpy as np
import pytest

from oracle import oracle as O

F = np.float32


def _blocks(rng, t, rows, K):
    be, bb = (256, 144) if t == O.Q4_K else (32, 34)
    nb = K // be
    raw = rng.integers(0, 256, size=(rows, nb, bb), dtype=np.uint8)
    d = ((0.005 + rng.random((rows, nb)).astype(F) * 0.045) / (63.0 if t == O.Q4_K else 127.0)).astype(np.float16).view(np.uint16)
    raw[:, :, 0] = d & 0xFF; raw[:, :, 1] = d >> 8
    if t == O.Q4_K:
        dm = (d.view(np.float16).astype(F) * 8.0).astype(np.float16).view(np.uint16)
        raw[:, :, 2] = dm & 0xFF; raw[:, :, 3] = dm >> 8
    return np.ascontiguousarray(raw.reshape(rows, nb * bb))


@pytest.fixture(autouse=True)
def _scalar_after():
    yield
    O.gguf_set_avx2(False)


@pytest.mark.parametrize("t,n,k", [(O.Q4_K, 300, 2048), (O.Q4_K, 7, 256), (O.Q4_K, 1408, 2048), (O.Q8_0, 2048, 1408), (O.Q8_0, 5, 32), (O.Q8_0, 257, 2816)])
def test_avx2_rows_bit_identical_to_the_scalar_lane_form(t, n, k):
    rng = np.random.default_rng(n * 7 + k)
    w = _blocks(rng, t, n, k)
    x = ((rng.random(k) - 0.5) * 4).astype(F); x[::37] *= 50.0
    q, s, sm = O.gguf_quant_f32(x)
    O.gguf_set_avx2(False); ref = O.gguf_matvec_int(t, w, q, s, sm, n, k)
    for threads in (1, 3, O.num_threads()):
        O.set_num_threads(threads)
        O.gguf_set_avx2(True); got = O.gguf_matvec_int(t, w, q, s, sm, n, k)
        assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), (t, n, k, threads)


def test_avx2_expert_and_moe_bit_identical():
    rng = np.random.default_rng(5)
    H, I = 512, 352          # 352 = 11 * 32: Q8_0 down like V2-Lite's 1408
    ex = [O.GgufExpert(_blocks(rng, O.Q4_K, I, H), _blocks(rng, O.Q4_K, I, H), _blocks(rng, O.Q8_0, H, I), O.Q4_K, O.Q8_0, H, I) for _ in range(4)]
    act = O.f32_to_bf16((rng.random(H) - 0.5).astype(F)); w = rng.random(3).astype(F)
    O.gguf_set_avx2(False); ref = O.moe_forward_gguf(ex[:3], w, act, shared=ex[3], rsf=2.5)
    O.gguf_set_avx2(True); got = O.moe_forward_gguf(ex[:3], w, act, shared=ex[3], rsf=2.5)
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32))
