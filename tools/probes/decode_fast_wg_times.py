#!/usr/bin/env python3
"""Where the tail of a KR_DECODE_FAST launch comes from: entry / exit wall-clock time (s_memrealtime, 10 ns) and hardware id (XCC, SE, CU) of EVERY workgroup of
the six launches of the last layer of a step (timing build: make -C krasis_amd/csrc timing).
    KRASIS_HIP_LIB=krasis_amd/libkrasis_hip_timing.so LAYERS=47 python tools/probes/decode_fast_wg_times.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("KRASIS_HIP_LIB", os.path.join(ROOT, "krasis_amd", "libkrasis_hip_timing.so"))
import numpy as np  # noqa: E402

import bench  # noqa: E402

NAMES = {0: "K1 kr_fdm in-projection", 1: "K2 kr_fla delta rule", 2: "K3 kr_fdm out-projection", 3: "K4 kr_frt router", 4: "K5 kr_fw13 gate|up", 5: "K6 kr_fw2 down+combine"}


def main():
    import torch
    eng, st, keep = bench.build_qcn(0, 0, int(os.environ.get("LAYERS", "47")), 0, 4, kv_fp8=True)
    st.set_attention_mode(False, decode_fast=True)
    st.set_option("lm_fused", 0)      # the one-launch final norm + vocabulary projection is a kr_fdm launch too: it would overwrite the in-projection's stamps
    buf = (C.c_ulonglong * (6 * 1024 * 3))()
    lines = ["# QCN decode step, KR_DECODE_FAST, launches of the last (linear-attention) layer: per workgroup entry / exit (us, relative to the launch's first entry)"]
    acc = {}
    for i in range(8):
        st.decode_step(0, 10 + i); torch.cuda.synchronize()
        assert st._lib.kr_debug_fwg(buf) == 0
        a = np.frombuffer(buf, np.uint64).reshape(6, 1024, 3).astype(np.int64)
        if i < 3:
            continue
        for k in NAMES:
            m = (a[k][:, 0] > 0) & (a[k][:, 1] >= a[k][:, 0]) & (a[k][:, 1] - a[k][:, 0] < 100000)      # workgroups that ran to the exit stamp in THIS step
            m &= a[k][:, 0] >= a[k][:, 0][a[k][:, 0] > 0].max() - 100000 if (a[k][:, 0] > 0).any() else m
            n = int(m.sum())
            if n == 0:
                continue
            ent = a[k][m, 0] * 0.01; ext = a[k][m, 1] * 0.01; hw = a[k][m, 2]
            t0 = ent.min()
            cu = (hw & 0xFFFFFFFF); xcc = (hw >> 32) & 7
            cuid = ((cu >> 8) & 0xFF) | (xcc << 8)      # HW_ID bits 15:8 = cu_id, sh_id, se_id (+ XCC id): one key per physical CU
            uniq, cnt = np.unique(cuid, return_counts=True)
            per_cu = dict(zip(uniq.tolist(), cnt.tolist()))
            wcount = np.array([per_cu[int(c)] for c in cuid])
            dur = ext - ent
            acc.setdefault(k, []).append(dict(n=n, span=float(ext.max() - t0), last_entry=float(ent.max() - t0), med_dur=float(np.median(dur)), max_dur=float(dur.max()),
                                             dur1=float(np.median(dur[wcount == 1])) if (wcount == 1).any() else 0.0, dur2=float(np.median(dur[wcount >= 2])) if (wcount >= 2).any() else 0.0,
                                             cus=len(uniq), cus2=int((cnt >= 2).sum()), end1=float((ext[wcount == 1] - t0).max()) if (wcount == 1).any() else 0.0,
                                             end2=float((ext[wcount >= 2] - t0).max()) if (wcount >= 2).any() else 0.0))
    for k, rows in acc.items():
        f = lambda key: float(np.mean([r[key] for r in rows]))
        lines.append("%-26s %4d workgroups on %3d CUs (%3d CUs hold two or more): first entry -> last exit %.2f us; last entry at %.2f; duration of a workgroup median %.2f / max %.2f; "
                     "alone on its CU: median %.2f, last exit %.2f; sharing a CU: median %.2f, last exit %.2f" % (
                         NAMES[k], rows[0]["n"], rows[0]["cus"], rows[0]["cus2"], f("span"), f("last_entry"), f("med_dur"), f("max_dur"), f("dur1"), f("end1"), f("dur2"), f("end2")))
    out = "\n".join(lines)
    print(out)
    os.makedirs("gpurun_out", exist_ok=True)
    open(os.environ.get("WG_OUT", "gpurun_out/r05_decode_fast_wg_times.txt"), "w").write(out + "\n")


if __name__ == "__main__":
    main()
