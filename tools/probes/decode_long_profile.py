"""probe for rocprofv3 --kernel-trace --stats: 30 QCN decode steps late in a long cache.  argv: kv_len [fast=1] [fp8=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
KV = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
fast = int(sys.argv[2]) if len(sys.argv) > 2 else 1
fp8 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
bench.QCN["kv_max_seq"] = KV
eng, st, keep = bench.build_qcn(0, 0, 48, KV + 64, 4, bool(fp8))
st.set_attention_mode(bool(fast))
st.set_use_graph(True)
for i in range(3): st.decode_step(0, KV - 7 + i)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(30): st.decode_step(0, KV - 2)
st.last_token(); dt = time.perf_counter() - t0
print("decode at position %d of %d, fast=%d fp8=%d: %.3f ms/step (%.1f tok/s)" % (KV - 2, KV, fast, fp8, dt / 30 * 1e3, 30 / dt), flush=True)
