"""probe: chunk size x chunks in flight of the whole-model QCN prompt pass (synthetic), one model build for the whole sweep.
argv: tokens mode(0 exact / 1 KR_ATTN_FAST / 2 + KR_GEMM_FAST) "chunk:depth,chunk:depth,..." """
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
fast = int(sys.argv[2]) if len(sys.argv) > 2 else 2
combos = [tuple(int(v) for v in c.split(":")) for c in (sys.argv[3] if len(sys.argv) > 3 else "1024:3,2048:2,4096:2,8192:1").split(",")]
eng, st, keep = bench.build_qcn(0, 0, 48, P + 64, 4, True)
st.set_attention_mode(bool(fast), gemm_fast=(fast == 2))
toks = [int(x) for x in np.random.default_rng(5).integers(0, bench.QCN["vocab"], P)]
for chunk, depth in combos:
    st.set_prefill_chunk(chunk); st.set_prefill_depth(depth)
    st.fill_state_synthetic(P + 64, 7)
    st.prefill(toks, 0); torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        st.fill_state_synthetic(P + 64, 7); torch.cuda.synchronize()
        t0 = time.perf_counter(); st.prefill(toks, 0); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("mode %d  %5d tokens  chunk %5d x depth %d: %7.1f ms  %7.0f tok/s" % (fast, P, chunk, depth, best * 1e3, P / best), flush=True)
