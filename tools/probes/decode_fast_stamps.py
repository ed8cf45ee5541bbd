#!/usr/bin/env python3
"""In-kernel phase times of the KR_DECODE_FAST kernels on the QCN-shaped model: wall-clock stamps (s_memrealtime, 10 ns) written by wave 0 of the
middle workgroup of each launch (last layer of the step), from the probe build `make -C krasis_amd/csrc timing` (libkrasis_hip_timing.so).
    KRASIS_HIP_LIB=krasis_amd/libkrasis_hip_timing.so python tools/probes/decode_fast_stamps.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("KRASIS_HIP_LIB", os.path.join(ROOT, "krasis_amd", "libkrasis_hip_timing.so"))
import numpy as np  # noqa: E402

import bench  # noqa: E402

NAMES = {0: ("kr_fdm (norm folded, qkvz|ba / q|k|v)", ["fetch issued", "norm + image", "barrier", "dot", "exchange + store"]),
         1: ("kr_fla (delta rule per value head)", ["loads issued, q/k sums, gates", "barrier", "pass 1", "kv reduce + delta (2 barriers)", "pass 2", "barrier + o reduce", "gated norm + image"]),
         2: ("kr_fdm (image input, out / o projection)", ["fetch issued", "image copy", "barrier", "dot", "exchange + store"]),
         3: ("kr_frt (norm + gate GEMV)", ["gate fetch issued + norm", "images + x to LDS + barrier", "GEMV", "exchange + store"]),
         4: ("kr_fw13 (select + gate|up + silu*up)", ["select (wave 0)", "barrier", "weight fetch issued", "dots", "exchange + store"]),
         5: ("kr_fw2 (quant + down + combine)", ["fetch issued", "quant of h", "dot", "barrier", "combine + store"])}


def main():
    import torch  # noqa: F401
    q = bench.QCN
    eng, st, keep = bench.build_qcn(0, 0, int(os.environ.get("LAYERS", "48")), 0, 4, kv_fp8=True, gguf=os.environ.get("GGUF", "0") == "1")
    st.set_attention_mode(False, decode_fast=True)
    st.set_option("lm_fused", 0)      # the one-launch final norm + vocabulary projection is a kr_fdm launch too: it would overwrite the in-projection's stamps
    lib = st._lib
    buf = (C.c_ulonglong * (8 * 16))()
    acc = {}; raws = []; raw_last = None; acc6 = []
    reps = 20
    for i in range(reps + 3):
        st.decode_step(0, 10 + i)
        torch.cuda.synchronize()
        assert lib.kr_debug_fstamps(buf) == 0
        a = np.frombuffer(buf, np.uint64).reshape(8, 16).astype(np.int64)
        if i < 3:
            continue
        for k, (nm, ph) in NAMES.items():
            d = np.diff(a[k, : len(ph) + 1]) * 0.01
            acc.setdefault(k, []).append(d)
        raws.append(a.copy()); raw_last = a
        acc6.append(np.diff(a[6, :7]) * 0.01)
    lines = []
    # the select prologue of kr_fw13 in finer steps (row 6) and the data-path gap between consecutive launches: last stamp of launch i -> first stamp of launch i + 1
    # (s_memrealtime is one constant-rate counter for the whole device, 10 ns)
    sel = ["entry -> logits in registers", "keys + lane maxima", "row-maximum rounds (threshold)", "ballot compaction into LDS", "rank from the LDS list + write", "tie check, weights, ids to LDS"]
    if raw_last is not None and raw_last[6, 0] > 0:
        d6 = np.mean(acc6, axis=0)
        lines.append("%-45s in-select %.2f us : " % ("kr_fw13 select prologue, finer", d6.sum()) + "  ".join("%s %.2f" % (p, v) for p, v in zip(sel, d6)))
    chain = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5)]
    gl = []
    for (a_, b_) in chain:
        na, nb = len(NAMES[a_][1]), 0
        g = np.mean([(r[b_, nb] - r[a_, na]) * 0.01 for r in raws])
        gl.append("%s -> %s %.2f" % (NAMES[a_][0].split(" ")[0] + str(a_), NAMES[b_][0].split(" ")[0] + str(b_), g))
    lines.append("gap on the data path, end of the middle workgroup's wave 0 -> start of the next launch's (us; rows 0..5 = K1..K6 of the last layer of the step): " + "  ".join(gl))
    for k, (nm, ph) in NAMES.items():
        m = np.mean(acc[k], axis=0)
        lines.append("%-45s in-kernel %.2f us : " % (nm, m.sum()) + "  ".join("%s %.2f" % (p, v) for p, v in zip(ph, m)))
    out = "\n".join(lines)
    print(out)
    os.makedirs("gpurun_out", exist_ok=True)
    open(os.environ.get("STAMPS_OUT", "gpurun_out/r05_decode_fast_stamps.txt"), "w").write("# QCN Q4 decode step, KR_DECODE_FAST: phases inside the kernels (us, wave 0 of the middle workgroup, last layer, mean of %d steps)\n" % reps + out + "\n")


if __name__ == "__main__":
    main()
