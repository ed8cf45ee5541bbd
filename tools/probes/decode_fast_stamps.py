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
    eng, st, keep = bench.build_qcn(0, 0, int(os.environ.get("LAYERS", "48")), 0, 4, kv_fp8=True)
    st.set_attention_mode(False, decode_fast=True)
    lib = st._lib
    buf = (C.c_ulonglong * (8 * 16))()
    acc = {}
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
    lines = []
    for k, (nm, ph) in NAMES.items():
        m = np.mean(acc[k], axis=0)
        lines.append("%-45s in-kernel %.2f us : " % (nm, m.sum()) + "  ".join("%s %.2f" % (p, v) for p, v in zip(ph, m)))
    out = "\n".join(lines)
    print(out)
    os.makedirs("gpurun_out", exist_ok=True)
    open("gpurun_out/r03_decode_fast_stamps.txt", "w").write("# QCN Q4 decode step, KR_DECODE_FAST: phases inside the kernels (us, wave 0 of the middle workgroup, last layer, mean of %d steps)\n" % reps + out + "\n")


if __name__ == "__main__":
    main()
