"""Probe: MoE-only launch loop (w13, w2, combine cycling over 48 cold layers) -- per-kernel durations with a warm instruction cache,
to compare against the same kernels inside the full decode step (rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from krasis_amd import KrasisEngine, ModelConfig

H, I, E, k, L = 2048, 512, 512, 10, 48
eng = KrasisEngine(device=0)
eng.configure(ModelConfig(H, I, E, k, L, 0, 1.0))
eng.fill_synthetic(4, seed=1)
x = (torch.randn(1, H, device="cuda") * 0.5).to(torch.bfloat16)
out = torch.empty(1, H, device="cuda", dtype=torch.bfloat16)
g = torch.Generator(device="cpu").manual_seed(0)
ids = [torch.randperm(E, generator=g)[:k].to(torch.int32).cuda().view(1, k) for _ in range(L)]
w = torch.full((1, k), 0.1, device="cuda")
s = torch.cuda.current_stream().cuda_stream or 1
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    for l in range(L):
        eng.forward_moe_direct(l, x.data_ptr(), ids[l].data_ptr(), w.data_ptr(), out.data_ptr(), 1, k, s)
torch.cuda.synchronize()
print("done")
