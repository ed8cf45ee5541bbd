#!/usr/bin/env python3
"""decode tok/s of the native-GGUF side configurations (bench.side_config): qcn-q4k-gguf and v2lite-q4k-gguf.  argv: [steps=30]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
args = argparse.Namespace(steps=steps, warmup=5, no_graph=False, prefill_reps=2)
for name in ("qcn-q4k-gguf", "v2lite-q4k-gguf"):
    orig = bench.prefill_model
    bench.prefill_model = lambda *a, **k: {"value": 0.0, "roofline": {"achieved": 0.0}}      # decode only
    try:
        r = bench.side_config(name, 0, 0, args, torch)
    finally:
        bench.prefill_model = orig
    print("%s: decode %.1f tok/s (%.3f ms/step, %.3f of the HBM peak for %.2f GB/token), KR_DECODE_FAST (routed slots on the GGUF blocks inside the mode's three MoE launches) %.1f tok/s" % (
        name, r["decode_tok_s"], r["ms_per_step"], r["step_frac_of_hbm_peak"], r["step_algorithmic_bytes"] / 1e9, r["decode_fast_tok_s"]), flush=True)
