"""probe: scoring prompt pass (kr_decode_prefill_nll) on the full QCN shape (vocab 151 936): tok/s of a 4096-token window vs the plain
prompt pass, and a spot check of the per-position negative log-likelihood against decode_step logits + float64 cross-entropy."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
eng, st, keep = bench.build_qcn(0, 0, 48, P + 64)
V = bench.QCN["vocab"]
toks = [int(x) for x in np.random.default_rng(5).integers(0, V, P)]
st.reset_decode_state(P + 64)
st.prefill(toks, 0); torch.cuda.synchronize()
for name, fn in (("prefill", lambda: st.prefill(toks, 0)), ("prefill_nll", lambda: st.prefill_nll(toks, 0))):
    st.reset_decode_state(P + 64); fn(); torch.cuda.synchronize()
    st.reset_decode_state(P + 64); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-12s %d tokens: %.1f ms (%.0f tok/s)" % (name, P, dt * 1e3, P / dt), flush=True)
nll = r
print("mean nll %.6f  ppl %.1f  (uniform would be %.6f)" % (float(nll.mean()), float(np.exp(nll.mean())), float(np.log(V))))
# spot check: first 6 positions by decode_step
st.reset_decode_state(P + 64)
lg = np.empty(V, np.float32); worst = 0.0
for i in range(6):
    st.decode_step(toks[i], i, lg.ctypes.data)
    x = lg.astype(np.float64); mx = x.max(); ref = np.log(np.exp(x - mx).sum()) + mx - x[toks[i + 1]]
    worst = max(worst, abs(ref - float(nll[i])))
print("max |nll - f64 reference| over 6 positions: %.2e" % worst)
