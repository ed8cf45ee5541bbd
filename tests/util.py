"""Shared helpers for the GPU parity tests (the oracle is the checker, never the thing measured)."""
import numpy as np

from oracle import oracle as O


def rand_bf16(rng, shape, scale=1.0):
    return O.f32_to_bf16((rng.standard_normal(shape) * scale).astype(np.float32))


def make_experts(rng, E, H, I, bits=4, w2_bits=None, scale=0.05):
    out = []
    for _ in range(E):
        g = rand_bf16(rng, (I, H), scale); u = rand_bf16(rng, (I, H), scale); d = rand_bf16(rng, (H, I), scale)
        out.append(O.unified_from_bf16(g, u, d, 128, bits, w2_bits))
    return out


def upload(engine, layer, experts, shared=None):
    for e, ex in enumerate(experts):
        engine.load_unified_expert(layer, e, ex.w13, ex.w13_scales, ex.w2, ex.w2_scales, ex.num_bits, ex.w2_bits)
    if shared is not None:
        engine.load_unified_expert(layer, -1, shared.w13, shared.w13_scales, shared.w2, shared.w2_scales, shared.num_bits, shared.w2_bits)
