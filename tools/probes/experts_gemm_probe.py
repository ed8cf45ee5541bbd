#!/usr/bin/env python3
"""Experts-only prompt-pass GEMMs on the QCN shape (bench.prefill_experts): exact int8-MFMA form, tolerance f16-MFMA form (KR_PFH_VARIANT selects the
64 x 256 kernel variant for an A/B), native Q4_K blocks.  argv: [layers=8] [tokens=8192] [modes=exact,fast,q4k]
Also the workload of the SQ-counter passes: rocprofv3 --pmc ... -- python tools/probes/experts_gemm_probe.py 4 8192 fast"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
modes = (sys.argv[3] if len(sys.argv) > 3 else "exact,fast,q4k").split(",")
from krasis_amd import KrasisEngine, ModelConfig  # noqa: E402

q = bench.QCN
if any(m in modes for m in ("exact", "fast", "staged", "ring")):
    eng = KrasisEngine(device=0); eng.configure(ModelConfig(q["hidden"], q["inter"], q["experts"], q["topk"], L, 0, 1.0)); eng.fill_synthetic(4, seed=5)
    for mode in [m for m in modes if m in ("exact", "fast", "staged", "ring")]:      # "staged" / "ring" = tolerance form on the register-staged kernels only / with the LDS-ring kernel forced (kr_moe_set_gemm_mode 3 / 5); list a mode twice to interleave
        r = bench.prefill_experts(eng, q, L, M, torch, gemm_fast=(mode != "exact"), gemm_mode=(3 if mode == "staged" else (5 if mode == "ring" else None)))
        print("experts-only %-5s variant=%s: %d layers x %d tokens: %.2f ms/layer  %.0f tok/s (48-layer equivalent %.0f)  %.1f %s = %.3f of peak" % (
            mode, os.environ.get("KR_PFH_VARIANT", "default"), L, M, r["ms"] / L, r["tok_s_experts_only"], r["tok_s_experts_only"] * L / 48,
            r["roofline"]["achieved"], "TFLOP/s f16" if mode != "exact" else "TOP/s int8", r["roofline"]["frac"]), flush=True)
    del eng
if "q4kfast" in modes:
    r = bench.prefill_experts_gguf(0, torch, L=L, gemm_fast=True)
    print("experts-only native Q4_K, tolerance form: %d layers x %d tokens: %.2f ms/layer  %.1f TFLOP/s f16 useful = %.3f of peak" % (L, 8192, r["ms"] / L, r["roofline"]["achieved"], r["roofline"]["frac"]), flush=True)
if "q4k" in modes:
    r = bench.prefill_experts_gguf(0, torch, L=L)
    print("experts-only native Q4_K: %d layers x %d tokens: %.2f ms/layer  %.1f TOP/s int8 useful = %.3f of peak" % (L, 8192, r["ms"] / L, r["roofline"]["achieved"], r["roofline"]["frac"]), flush=True)
