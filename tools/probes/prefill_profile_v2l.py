"""probe: one whole-model prompt pass of the DeepSeek-V2-Lite shape (synthetic) for rocprofv3 --kernel-trace --stats.  argv: tokens [fast=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
fast = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng, st, keep = bench.build_v2lite(0, 0, 27, P + 64, 4, True)
st.set_attention_mode(bool(fast))
st.fill_state_synthetic(P + 64, 7)
toks = [int(x) for x in np.random.default_rng(5).integers(0, bench.V2L["vocab"], P)]
st.prefill(toks, 0); torch.cuda.synchronize()          # full-length warm-up: the scratch arenas are sized by the chunk length
t0 = time.perf_counter(); st.prefill(toks, 0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("V2-Lite prompt pass %d tokens fast=%d: %.1f ms (%.0f tok/s)" % (P, fast, dt * 1e3, P / dt), flush=True)
