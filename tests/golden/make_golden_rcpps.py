#!/usr/bin/env python3
"""Committed vector for the ONE stated deviation of the default numerics: the reference's fused SiLU kernels take the sigmoid's
reciprocal with _mm256_rcp_ps + one Newton step (src/kernel/avx2.rs:2277-2291) -- a hardware table, different between CPU vendors -- where the
oracle's default and the HIP kernels use an IEEE divide.  This script runs the oracle's rcpps form (KRO_SIG_POLY5_RCPNR: the AVX2 instruction
sequence of the reference, compiled here) on THIS host over 8192 inputs and stores the bit patterns together with the CPU model, so the
"<= 2 ulp" claim is pinned to a committed vector rather than to whatever CPU runs the test.
    python tests/golden/make_golden_rcpps.py      # writes tests/golden/rcpps_sigmoid.npz"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

rng = np.random.default_rng(7)
x = np.concatenate([np.linspace(-22.0, 22.0, 4096), rng.standard_normal(3072) * 3.0, rng.standard_normal(1024) * 0.01]).astype(np.float32)
y = np.array([O.sigmoid(float(v), O.SIG_POLY5_RCPNR) for v in x], np.float32)
cpu = "unknown"
for line in open("/proc/cpuinfo"):
    if line.startswith("model name"):
        cpu = line.split(":", 1)[1].strip(); break
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rcpps_sigmoid.npz"), x=x, y_bits=y.view(np.uint32), cpu=np.array(cpu))
print(cpu, x.size, "values")
