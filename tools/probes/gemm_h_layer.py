"""probe for rocprofv3: the expert path (kr_moe_prefill) of a few QCN-shaped layers, M tokens per call.  argv: M fast(0|1) [layers=4] [reps=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from krasis_amd import GpuPrefillManager, KrasisEngine, ModelConfig
from krasis_amd._lib import check
M = int(sys.argv[1]); fast = int(sys.argv[2]); L = int(sys.argv[3]) if len(sys.argv) > 3 else 4; reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
q = bench.QCN; H, I, E, k = q["hidden"], q["inter"], q["experts"], q["topk"]
eng = KrasisEngine(device=0); eng.configure(ModelConfig(H, I, E, k, L, 0, 1.0)); eng.fill_synthetic(4, seed=1)
g = torch.Generator(device="cuda").manual_seed(7)
x = ((torch.rand((M, H), device="cuda", generator=g) - 0.5)).to(torch.bfloat16)
ids = torch.rand((M, E), device="cuda", generator=g).topk(k, dim=1).indices.to(torch.int32)
w = torch.softmax(torch.randn((M, k), device="cuda", generator=g), dim=1)
mgr = GpuPrefillManager(eng, k)
check(eng._lib.kr_moe_set_gemm_mode(eng._h, fast))
for _ in range(reps):
    for l in range(L): mgr.forward(l, x, ids, w, routed_only=True)
torch.cuda.synchronize()
print("done")
