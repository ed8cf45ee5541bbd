#!/usr/bin/env python3
"""Golden vectors of the reference's LEGACY GPU quantizer `_quantize_and_pack_gpu` (python/krasis/gpu_prefill.py:240-291; SURVEY 8c: it imports on CPU tensors).
Run in the build container (needs /root/reference): PYTHONPATH=/root/reference/python python tests/golden/make_golden_legacy_quant.py
Writes tests/golden/legacy_quant.npz: bf16 inputs [K, N] and the reference's (packed int32 [K / vals, N], scale bf16 [K / 128, N]) for INT4 and INT8."""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, "/root/reference/python")
for name in ("flashinfer", "sgl_kernel", "sglang"):          # GPU-only dependencies of the module's import chain: not touched by the quantizer
    sys.modules.setdefault(name, types.ModuleType(name))
try:
    from krasis.gpu_prefill import _quantize_and_pack_gpu
except Exception as ex:   # pragma: no cover - reported to whoever regenerates the fixture
    raise SystemExit("cannot import the reference quantizer: %r" % (ex,))

rng = np.random.default_rng(20260926)
out = {}
for tag, (K, N) in {"a": (256, 24), "b": (128, 64)}.items():
    w = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
    w[5, 3] = 0.0; w[:128, 7] = 0.0                                   # an all-zero group: the scale clamps at 1e-10
    w[K - 100, 2] = -np.abs(w[K - 128:, 2]).max() * 1.5              # a group whose extreme is NEGATIVE: |min| / 8 decides the scale
    wt = torch.from_numpy(w).to(torch.bfloat16)
    out[f"w_{tag}"] = wt.view(torch.int16).numpy().view(np.uint16)
    for bits in (4, 8):
        packed, scale = _quantize_and_pack_gpu(wt, 128, bits)
        out[f"packed{bits}_{tag}"] = packed.numpy().astype(np.int32)
        out[f"scale{bits}_{tag}"] = scale.view(torch.int16).numpy().view(np.uint16)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "legacy_quant.npz"), **out)
print("wrote legacy_quant.npz:", {k: v.shape for k, v in out.items()})
