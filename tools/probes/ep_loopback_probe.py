#!/usr/bin/env python3
"""Expert parallelism at world sizes 1, 2, 4, 8 ON ONE GPU through the loopback transport (csrc/kr_ep.cpp): W engines of this process, engine r
holds experts [r E / W, (r + 1) E / W) of every layer, each virtual rank brings tokens / W tokens of the same 8192-token batch, rows travel by
device-to-device copies where RCCL would send them over xGMI.  What this measures: the dataflow around the exchange (owner sort, split-size
hand-off, row gather / scatter, receive-side GEMMs on top-1 rows, return, combine) at growing W with the transfer cost near zero and the W ranks
SHARING one GPU -- an upper bound for the per-rank software overhead, not a scaling number.  Every rank's output is checked against the single
engine operator (bf16 return rows: 2^-7 of the row's largest value).
    python tools/probes/ep_loopback_probe.py [layers=4] [tokens=8192]  ->  gpurun_out/r06_ep_loopback.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from krasis_amd import KrasisEngine, ModelConfig, _lib  # noqa: E402
from krasis_amd._lib import check  # noqa: E402
from krasis_amd.ep import ExpertParallel, LoopbackGroup  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
q = bench.QCN; H, I, E, k = q["hidden"], q["inter"], q["experts"], q["topk"]
SEED = 0x12345678ABCDEF01
g = torch.Generator(device="cuda").manual_seed(7)
x = ((torch.rand((M, H), device="cuda", generator=g) - 0.5)).to(torch.bfloat16)
ids = torch.rand((M, E), device="cuda", generator=g).topk(k, dim=1).indices.to(torch.int32)
w = torch.softmax(torch.randn((M, k), device="cuda", generator=g), dim=1)
lines = []
full = KrasisEngine(); full.configure(ModelConfig(H, I, E, k, L, 0, 1.0)); full.fill_synthetic(4, seed=SEED)
ref = torch.empty((M, H), dtype=torch.float32, device="cuda")
check(full._lib.kr_moe_prefill(full._h, 0, x.data_ptr(), ids.data_ptr(), w.data_ptr(), ref.data_ptr(), M, k, _lib.KR_OUT_F32, 1, 1)); torch.cuda.synchronize()
t0 = time.perf_counter()
for l in range(L):
    check(full._lib.kr_moe_prefill(full._h, l, x.data_ptr(), ids.data_ptr(), w.data_ptr(), ref.data_ptr(), M, k, _lib.KR_OUT_F32, 1, 1))
torch.cuda.synchronize(); base = (time.perf_counter() - t0) / L
check(full._lib.kr_moe_prefill(full._h, 0, x.data_ptr(), ids.data_ptr(), w.data_ptr(), ref.data_ptr(), M, k, _lib.KR_OUT_F32, 1, 1)); torch.cuda.synchronize()
lines.append("single engine (kr_moe_prefill, exact form): %.2f ms per layer, %d tokens x %d layers" % (base * 1e3, M, L))
for W in (1, 2, 4, 8):
    per = E // W
    grp = LoopbackGroup(W); engs, eps, outs = [], [], []
    for r in range(W):
        e = KrasisEngine(); e.configure(ModelConfig(H, I, per, k, L, 0, 1.0))
        # the slice of the SAME synthetic experts: generated on the full engine, copied expert by expert
        for l in range(L):
            for j in range(per):
                a = full.download_expert(l, r * per + j, 4)
                e.load_unified_expert(l, j, *a, num_bits=4)
        engs.append(e); eps.append(ExpertParallel(e, E, rank=r, loopback=grp, return_bf16=True))
        outs.append(torch.zeros((M // W, H), dtype=torch.float32, device="cuda"))
    sh = M // W

    def call(r, l):
        a = slice(r * sh, (r + 1) * sh)
        check(engs[r]._lib.kr_moe_prefill_ep(engs[r]._h, l, x[a].data_ptr(), ids[a].data_ptr(), w[a].data_ptr(), outs[r].data_ptr(), sh, k, _lib.KR_OUT_F32, 1, None))
        engs[r].synchronize()
    grp.run([lambda r=r: call(r, 0) for r in range(W)])
    torch.cuda.synchronize()
    worst = max(float((outs[r] - ref[r * sh:(r + 1) * sh]).abs().max() / ref[r * sh:(r + 1) * sh].abs().max()) for r in range(W))
    t0 = time.perf_counter()
    for l in range(L):
        grp.run([lambda r=r, l=l: call(r, l) for r in range(W)])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / L
    lines.append("loopback W = %d: %.2f ms per layer for the whole batch (%.2fx the single engine), %d tokens per rank, rccl-equivalent ranks %d, max |diff| / max |ref| %.2e (bf16 rows)"
                 % (W, dt * 1e3, dt / base, sh, eps[0].comm_ranks(), worst))
    for ep in eps:
        ep.close()
    grp.close(); del engs, eps, outs
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/r06_ep_loopback.txt", "w").write("# " + __doc__.split("\n")[0] + "\n" + "\n".join(lines) + "\n")
print("\n".join(lines))
