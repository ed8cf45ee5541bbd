"""probe: one whole-model prompt pass of QCN (synthetic) for rocprofv3 --kernel-trace --stats.  argv: tokens [fast=1 (2: + GEMM tolerance form)] [layers] [chunk] [depth]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
fast = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = int(sys.argv[3]) if len(sys.argv) > 3 else 48
eng, st, keep = bench.build_qcn(0, 0, L, P + 64, 4, True)
st.set_attention_mode(bool(fast), gemm_fast=(fast == 2))
if len(sys.argv) > 4 and int(sys.argv[4]): st.set_prefill_chunk(int(sys.argv[4]))
if len(sys.argv) > 5 and int(sys.argv[5]): st.set_prefill_depth(int(sys.argv[5]))
for kv in os.environ.get("KR_OPTS", "").split(","):          # A/B hooks: KR_OPTS="norm_rows=0,gemm_ring=2"
    if "=" in kv: st.set_option(kv.split("=")[0], int(kv.split("=")[1]))
st.fill_state_synthetic(P + 64, 7)
toks = [int(x) for x in np.random.default_rng(5).integers(0, bench.QCN["vocab"], P)]
st.prefill(toks, 0); torch.cuda.synchronize()          # full-length warm-up: the scratch arenas are sized by the chunk length
if os.environ.get("KR_PFM_TIMING"): st.set_option("pfm_timing", 1)
t0 = time.perf_counter(); st.prefill(toks, 0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("prompt pass %d tokens fast=%d: %.1f ms (%.0f tok/s)" % (P, fast, dt * 1e3, P / dt), flush=True)
