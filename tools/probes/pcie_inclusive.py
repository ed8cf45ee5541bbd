"""probe: the PCIe-inclusive decode rate -- kr_decode_step handing the logits back to a HOST buffer every token (607 744 B DtoH + a stream
synchronisation per step), which is how a drop-in caller that samples on the host would use it; bench.py's `value` keeps everything in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
eng, st, keep = bench.build_qcn(0, 0, 48, 0)
st.set_use_graph(True)
V = bench.QCN["vocab"]; kvm = bench.QCN["kv_max_seq"]
lg = np.empty(V, np.float32)
for mode in ("device-resident", "host logits every step", "host greedy id every step"):
    for i in range(5): st.decode_step(0, 10 + i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100):
        if mode == "device-resident": st.decode_step(0, (15 + i) % (kvm - 1))
        elif mode == "host logits every step": st.decode_step(0, (15 + i) % (kvm - 1), lg.ctypes.data)
        else: st.decode_step(0, (15 + i) % (kvm - 1)); st.last_token()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-28s %.3f ms/step  %.1f tok/s" % (mode, dt * 10, 100 / dt), flush=True)
