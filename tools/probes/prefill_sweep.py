"""probe: QCN prompt pass (synthetic, FAST mode by default) swept over the chunk size and the number of chunks in flight.
argv: tokens [fast=1]   -> one line per (chunk, depth)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
fast = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng, st, keep = bench.build_qcn(0, 0, 48, P + 64, 4, True)
st.set_attention_mode(bool(fast))
toks = [int(x) for x in np.random.default_rng(5).integers(0, bench.QCN["vocab"], P)]
for chunk in (512, 1024, 2048, 4096):
    for depth in (2, 3, 4):
        if chunk * depth > 2 * P:
            continue
        st.set_prefill_chunk(chunk); st.set_prefill_depth(depth)
        st.fill_state_synthetic(P + 64, 7)
        st.prefill(toks, 0); torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            st.fill_state_synthetic(P + 64, 7); torch.cuda.synchronize()
            t0 = time.perf_counter(); st.prefill(toks, 0); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print("tokens %d fast %d chunk %d depth %d: %.1f ms  %.0f tok/s" % (P, fast, chunk, depth, best * 1e3, P / best), flush=True)
