#!/usr/bin/env python3
"""decode-only run of one bench side configuration (for rocprofv3 --kernel-trace --stats).  argv: name [steps=20]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
args = argparse.Namespace(steps=steps, warmup=5, no_graph=False, prefill_reps=2)
bench.prefill_model = lambda *a, **k: {"value": 0.0, "roofline": {"achieved": 0.0}}      # decode only
r = bench.side_config(name, 0, 0, args, torch)
print("%s: decode %.1f tok/s exact, %.1f tok/s KR_DECODE_FAST" % (name, r["decode_tok_s"], r.get("decode_fast_tok_s", 0.0)), flush=True)
