"""probe: the expert-parallel row path of bench.py's N > 1 leg at N = 1 (no collectives), for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng, st, keep = bench.build_qcn(0, 0, L, 0)
r = bench.prefill_ep(eng, L, 8192, 1, 0, torch, None)
print(r["ms"], "ms for", L, "layers ->", r["ms"] / L, "ms/layer")
r = bench.prefill_experts(eng, L, 8192, torch)
print("direct:", r["ms"] / L, "ms/layer")
