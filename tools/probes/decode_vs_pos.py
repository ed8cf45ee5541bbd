"""probe: decode step time as a function of the KV position (same graph, fixed position per measurement)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
eng, st, keep = bench.build_qcn(0, 0, 48, 64)
st.set_use_graph(True)
for i in range(5): st.decode_step(0, 10 + i)
torch.cuda.synchronize()
for rep in range(2):
    for pos in (10, 40, 80, 120, 200, 250, 10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(30): st.decode_step(0, pos)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("pos %3d  %.3f ms/step" % (pos, dt / 30 * 1e3), flush=True)
