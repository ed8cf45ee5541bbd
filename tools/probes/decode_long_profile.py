"""probe for rocprofv3 --kernel-trace --stats: 30 QCN decode steps at position 8190 of an 8192-position FP16 cache (split attention launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
bench.QCN["kv_max_seq"] = 8192
eng, st, keep = bench.build_qcn(0, 0, 48, 8192)
st.set_use_graph(True)
for i in range(3): st.decode_step(0, 8185 + i)
torch.cuda.synchronize()
for i in range(30): st.decode_step(0, 8190)
torch.cuda.synchronize()
