import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.test_decode_gpu import build
from tests.test_prefill_model_gpu import _snapshot
F = np.float32
st, eng, orc, keep, d = build(with_dense=True)
toks = [int(x) for x in np.random.default_rng(1).integers(0, d["V"], 20)]
start = 7
ref = np.empty(d["V"], F)
for i, t in enumerate(toks):
    st.decode_step(t, start + i, ref.ctypes.data)
ref_state = _snapshot(st, d)
for chunk in (0, 6, 20, 10):
    d["reset"](); st.set_prefill_chunk(chunk)
    lg = np.empty(d["V"], F); st.prefill(toks, start, lg.ctypes.data)
    snap = _snapshot(st, d)
    print("chunk", chunk, "logits equal", np.array_equal(lg.view(np.uint32), ref.view(np.uint32)))
    for li, (a, b) in enumerate(zip(snap, ref_state)):
        e0 = np.array_equal(a[0].view(np.uint32) if a[0].dtype == F else a[0], b[0].view(np.uint32) if b[0].dtype == F else b[0])
        e1 = np.array_equal(a[1].view(np.uint32) if a[1].dtype == F else a[1], b[1].view(np.uint32) if b[1].dtype == F else b[1])
        if a[0].dtype != F:
            rows = np.where((a[0] != b[0]).any(axis=1))[0]
            print("  layer", li, d["kinds"][li], e0, e1, "diff rows", rows[:12])
        else:
            print("  layer", li, d["kinds"][li], e0, e1)
