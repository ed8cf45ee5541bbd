"""probe: decode step time vs KV position with a long cache (split decode attention: scores launch + softmax / p.v launch)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
bench.QCN["kv_max_seq"] = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
eng, st, keep = bench.build_qcn(0, 0, 48, bench.QCN["kv_max_seq"])
st.set_use_graph(True)
for i in range(3): st.decode_step(0, 10 + i)
torch.cuda.synchronize()
for pos in (10, 500, 1000, 2000, 4000, bench.QCN["kv_max_seq"] - 2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): st.decode_step(0, pos)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("pos %5d  %.3f ms/step  (%.1f tok/s)" % (pos, dt / 20 * 1e3, 20 / dt), flush=True)
