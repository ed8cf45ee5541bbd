"""GPU parity of the native GGUF block experts (Q4_K / Q8_0 / Q4_0 integer path, Q5_0 / Q6_K scalar path) against the oracle's
restatement of src/gguf_kernels.rs + moe_forward_gguf (src/moe.rs:990).  Bit-exact.  (The reference has no test for these kernels:
the oracle side is 'parity unpinned', see oracle/README.md.)"""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import rand_bf16

pytestmark = pytest.mark.gpu
BB = {O.Q4_K: (256, 144), O.Q8_0: (32, 34), O.Q4_0: (32, 18), O.Q5_0: (32, 22), O.Q6_K: (256, 210)}


def rand_blocks(rng, t, rows, K):
    be, bb = BB[t]; nb = K // be
    raw = rng.integers(0, 256, size=(rows, nb, bb), dtype=np.uint8)
    d = (rng.random((rows, nb)) * 4e-3 + 5e-4).astype(np.float16).view(np.uint16)
    if t == O.Q6_K:
        raw[:, :, 208] = d & 0xFF; raw[:, :, 209] = d >> 8
    else:
        raw[:, :, 0] = d & 0xFF; raw[:, :, 1] = d >> 8
        if t == O.Q4_K:
            dm = (rng.random((rows, nb)) * 2e-3 + 1e-4).astype(np.float16).view(np.uint16)
            raw[:, :, 2] = dm & 0xFF; raw[:, :, 3] = dm >> 8
    return np.ascontiguousarray(raw.reshape(rows, nb * bb))


def make(rng, H, I, gu_t, dn_t):
    return O.GgufExpert(rand_blocks(rng, gu_t, I, H), rand_blocks(rng, gu_t, I, H), rand_blocks(rng, dn_t, H, I), gu_t, dn_t, H, I)


@pytest.mark.parametrize("H,I,gu_t,dn_t", [
    (512, 256, O.Q4_K, O.Q4_K),
    (2048, 512, O.Q4_K, O.Q4_K),       # QCN expert shape, Q4_K
    (512, 96 * 2, O.Q4_K, O.Q8_0),     # intermediate not a multiple of 256 -> 32-block type for down (V2-Lite situation)
    (512, 224, O.Q4_0, O.Q4_0),
    (256, 160, O.Q8_0, O.Q5_0),        # scalar f32 fallback for the down projection
    (512, 256, O.Q4_K, O.Q6_K),
    (2048, 512, O.Q8_0, O.Q8_0),       # QCN Q8 config
])
def test_moe_forward_gguf_bit_exact(H, I, gu_t, dn_t):
    from krasis_amd import KrasisEngine, ModelConfig
    rng = np.random.default_rng(H + I + gu_t * 7 + dn_t)
    E, k = 6, 3
    experts = [make(rng, H, I, gu_t, dn_t) for _ in range(E)]
    shared = make(rng, H, I, gu_t, dn_t)
    eng = KrasisEngine(); eng.configure(ModelConfig(H, I, E, k, 1, 1, 1.5))
    for e, ex in enumerate(experts):
        eng.load_gguf_expert(0, e, ex.gate, ex.up, ex.down, gu_t, dn_t, I)
    eng.load_gguf_expert(0, -1, shared.gate, shared.up, shared.down, gu_t, dn_t, I)
    assert eng.has_gguf()
    B = 4
    act = rand_bf16(rng, (B, H)); ids = np.stack([rng.choice(E, k, replace=False) for _ in range(B)]).astype(np.int32)
    ids[2, 1] = -1
    w = rng.random((B, k)).astype(np.float32)
    got = np.frombuffer(eng.moe_forward(0, act[0].tobytes(), ids[0].tolist(), w[0].tolist()), np.float32)
    ref = O.moe_forward_gguf([experts[i] for i in ids[0]], w[0], act[0], shared, 1.5)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), float(np.max(np.abs(got - ref)))
    out = np.empty((B, H), np.uint16)
    eng.forward_moe_direct(0, act.ctypes.data, ids.ctypes.data, w.ctypes.data, out.ctypes.data, B, k)
    for b in range(B):
        sel = [(experts[i], wi) for i, wi in zip(ids[b], w[b]) if i >= 0]
        ref = O.moe_forward_gguf([s[0] for s in sel], [s[1] for s in sel], act[b], shared, 1.5)
        assert np.array_equal(out[b], O.f32_to_bf16(ref)), b
