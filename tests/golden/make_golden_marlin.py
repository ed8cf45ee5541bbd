#!/usr/bin/env python3
"""Bit-level golden vectors from the reference's importable Python: Marlin un-permute and the INT4 dequant / GEMV specification.

    python tests/golden/make_golden_marlin.py     # writes tests/golden/marlin_int4.npz (runs only where /root/reference exists)

Imported from where they lie (nothing is copied): python/krasis/triton_moe.py
  * _generate_weight_perm_int4 / _generate_scale_perms (:21-66)   -- the permutation tables themselves
  * inverse_marlin_repack (:71-128), inverse_scale_permute (:131-170) -- Marlin [K/16, 2N] words / [K/gs, N] scales -> standard [N, K/8] / [N, K/gs]
  * the arithmetic of _int4_gemv_kernel (:176-235: w = (u4 - 8) * scale, out[n] = sum_k x[k] * w[n, k]) is evaluated in torch float64 on
    inputs chosen so that every product and partial sum is an integer multiple of a power of two below 2^24: the f32 result is exact in ANY
    summation order, so the vector pins the dequantization semantics and the integer matvec of the CPU engine at bit level.

ANY u32 array is a valid Marlin tensor (the layout is a permutation of nibbles), so the fixtures start from random Marlin words and store
what the reference's inverse returns; the product's kr_marlin_unpack must reproduce it and kr_marlin_repack must map it back.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/python/krasis"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_triton_moe():
    pkg = types.ModuleType("krasis")
    pkg.__path__ = [REF]
    sys.modules["krasis"] = pkg
    return importlib.import_module("krasis.triton_moe")


def bf16_bits(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def main():
    tm = import_triton_moe()
    g = torch.Generator().manual_seed(20260925)
    out = {}
    out["weight_perm_int4"] = np.asarray(tm._generate_weight_perm_int4(), np.int32)
    sp, sps = tm._generate_scale_perms()
    out["scale_perm"] = np.asarray(sp, np.int32); out["scale_perm_single"] = np.asarray(sps, np.int32)
    for name, (N, K, gs) in {"a": (128, 256, 128), "b": (192, 512, 128), "c": (64, 128, 128)}.items():     # "c": channelwise (group_size == K)
        wm = torch.randint(-2**31, 2**31 - 1, (K // 16, 2 * N), generator=g, dtype=torch.int64).to(torch.int32)
        # bf16 scales with distinct bit patterns (finite): random 16-bit patterns with the exponent kept in a normal range
        sm_bits = torch.randint(0, 2**15, (K // gs, N), generator=g, dtype=torch.int64)
        sm_bits = (sm_bits & 0x807F) | (((sm_bits >> 7) % 64 + 96) << 7)
        sm = sm_bits.to(torch.int16).view(torch.bfloat16)
        w_std = tm.inverse_marlin_repack(wm, K, N, 4)                 # [N, K/8] int32
        s_std = tm.inverse_scale_permute(sm, K, N, gs)                 # [N, K/gs] bf16
        out[f"{name}.dims"] = np.array([N, K, gs], np.int32)
        out[f"{name}.w_marlin"] = wm.numpy().view(np.uint32).copy(); out[f"{name}.s_marlin"] = bf16_bits(sm)
        out[f"{name}.w_std"] = w_std.contiguous().numpy().view(np.uint32).copy(); out[f"{name}.s_std"] = bf16_bits(s_std)

    # ---- INT4 dequant + GEMV specification (triton_moe.py:183-235) with an exact result
    N, K, gs = 64, 512, 128
    w_std = torch.randint(-2**31, 2**31 - 1, (N, K // 8), generator=g, dtype=torch.int64).to(torch.int32)
    e = torch.randint(-9, -3, (N, 1), generator=g)                      # one power-of-two scale per ROW: 2^-9 .. 2^-4 (exact in bf16)
    scale = (2.0 ** e.double()).expand(N, K // gs).contiguous().to(torch.bfloat16)
    x = torch.randint(-200, 201, (K,), generator=g).double()
    x[::gs] = 32767.0                                                   # max |x| per group = 32767 -> the INT16 activation quantization is the identity
    x[gs // 2::gs] = -32767.0
    shifts = torch.arange(8, dtype=torch.int32) * 4
    u4 = ((w_std.unsqueeze(-1) >> shifts) & 0xF).reshape(N, K)         # nibble j of word c = k 8c + j
    wf = (u4.double() - 8.0) * scale.double().repeat_interleave(gs, dim=1)      # w_float * scales
    y = (wf * x[None, :]).sum(dim=1)                                    # float64: exact (all terms are multiples of 2^-9 below 2^40)
    assert torch.equal(y, y.float().double()), "GEMV fixture is not exactly representable in f32"
    out["gemv.dims"] = np.array([N, K, gs], np.int32)
    out["gemv.w_std"] = w_std.numpy().view(np.uint32).copy(); out["gemv.scales"] = bf16_bits(scale)
    out["gemv.x"] = x.float().numpy(); out["gemv.dequant"] = wf.float().numpy(); out["gemv.y"] = y.float().numpy()
    # ---- attention="int8" quantizer of the reference's loader (weight_loader.py:25-43): per-channel symmetric INT8 + bf16 scale
    try:
        wl = importlib.import_module("krasis.weight_loader")
        wb = (torch.randn((48, 320), generator=g) * 0.07).to(torch.bfloat16)
        wb[5] = 0.0                                                     # an all-zero row: amax clamps at 1e-10
        q8, sc8 = wl.quantize_to_int8(wb)
        out["q8.w_bf16"] = bf16_bits(wb); out["q8.q"] = q8.numpy().copy(); out["q8.scale_bf16"] = bf16_bits(sc8)
    except Exception as ex:                                             # heavy optional imports of weight_loader missing in this image
        print("quantize_to_int8 golden skipped:", repr(ex))
    np.savez_compressed(os.path.join(OUT, "marlin_int4.npz"), **out)
    print("marlin_int4.npz", os.path.getsize(os.path.join(OUT, "marlin_int4.npz")))


if __name__ == "__main__":
    torch.manual_seed(0)
    main()
