"""Probe: the prefill expert path alone (kr_moe_prefill, QCN dims, M tokens) for rocprofv3 stats / PMC passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from krasis_amd import KrasisEngine, ModelConfig, GpuPrefillManager

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
H, I, E, k, L = 2048, 512, 512, 10, 2
eng = KrasisEngine(device=0); eng.configure(ModelConfig(H, I, E, k, L, 0, 1.0)); eng.fill_synthetic(4, seed=1)
g = torch.Generator(device="cuda").manual_seed(7)
x = ((torch.rand((M, H), device="cuda", generator=g) - 0.5)).to(torch.bfloat16)
ids = torch.rand((M, E), device="cuda", generator=g).topk(k, dim=1).indices.to(torch.int32)
w = torch.softmax(torch.randn((M, k), device="cuda", generator=g), dim=1)
mgr = GpuPrefillManager(eng, k)
mgr.forward(0, x, ids, w, routed_only=True); torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for r in range(reps):
    mgr.forward(r % L, x, ids, w, routed_only=True)
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / reps
print("M", M, "ms/layer", ms, "tok/s(48 layers)", M / (ms * 48e-3), "TOPS", 2 * 2 * M * k * 3 * H * I / (ms * 1e-3) / 1e12)
