#!/usr/bin/env python3
"""Batch router of the tolerance prompt pass (kr_route_topk, rule DECODE, kr_moe_set_gemm_mode 1) at the QCN shape: E = 512, H = 2048, k = 10, one 2752-token chunk --
kr_route_logits_fast_kernel + kr_route_select_kernel.  Run under rocprofv3 --kernel-trace --stats (tools/gpu.sh routefast).  argv: [tokens=2752] [reps=20]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from krasis_amd import KrasisEngine, ModelConfig  # noqa: E402
from krasis_amd._lib import check  # noqa: E402
from oracle import oracle as O  # noqa: E402  (bf16 rounding of the synthetic gate only)

m = int(sys.argv[1]) if len(sys.argv) > 1 else 2752
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
E, H, k = 512, 2048, 10
rng = np.random.default_rng(1)
gate = O.bf16_to_f32(O.f32_to_bf16(((rng.random((E, H), dtype=np.float32) - 0.5) * 0.04).astype(np.float32))).reshape(E, H)
eng = KrasisEngine(); eng.configure(ModelConfig(H, 128, E, k, 1)); eng.set_routing_config("softmax", True, k, E, H)
eng.set_route_weight_f32(0, gate, None, None)
x = ((rng.random((m, H), dtype=np.float32) - 0.5) * 2).astype(np.float32)
check(eng._lib.kr_moe_set_gemm_mode(eng._h, 1))
for _ in range(3):
    eng.route(0, x, m, 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    eng.route(0, x, m, 1)
e1.record(); torch.cuda.synchronize()
print("kr_route_topk (tolerance logits + select), %d tokens: %.1f us per call incl. the host copies of x / ids / weights" % (m, e0.elapsed_time(e1) * 1e3 / reps))
