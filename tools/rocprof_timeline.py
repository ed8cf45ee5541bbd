#!/usr/bin/env python3
"""Concurrency view of a rocprofv3 --kernel-trace CSV: which kernels own the wall clock when several streams are in flight.

  rocprof_timeline.py <dir> <out.txt> [title] [window]      # window = "last": only the dispatches after the longest idle gap (the timed pass of a probe)

Per kernel name: launches, summed duration, EXCLUSIVE time (wall time during which no kernel of another name is running), and the
time-weighted mean number of other kernels running beside it.  Header: wall time of the window, busy time (>= 1 kernel running), the share of
the wall at concurrency 0 / 1 / 2 / 3+."""
import collections
import csv
import glob
import sys


def main():
    d, out = sys.argv[1:3]
    title = sys.argv[3] if len(sys.argv) > 3 else out
    window = sys.argv[4] if len(sys.argv) > 4 else "last"
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    ev = []
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
    ev.sort()
    if window == "last":        # cut at the longest gap between consecutive dispatches (host-side synchronisation between the warm-up and the timed pass)
        best, cut, end = 0, 0, ev[0][1]
        for i in range(1, len(ev)):
            gap = ev[i][0] - end
            if gap > best and i > len(ev) // 4: best, cut = gap, i
            end = max(end, ev[i][1])
        ev = ev[cut:]
    pts = []
    for s, e, n in ev:
        pts.append((s, 1, n)); pts.append((e, -1, n))
    pts.sort(key=lambda p: (p[0], p[1]))
    running = collections.Counter()
    nrun = 0
    last = pts[0][0]
    conc = collections.Counter()
    excl = collections.Counter(); beside = collections.Counter(); tot = collections.Counter(); cnt = collections.Counter()
    for t, dlt, n in pts:
        dt = t - last
        if dt > 0:
            conc[min(nrun, 3)] += dt
            names = [k for k, v in running.items() if v > 0]
            for k in names:
                beside[k] += dt * (nrun - running[k])
                if len(names) == 1: excl[k] += dt
        last = t
        running[n] += dlt; nrun += dlt
    for s, e, n in ev: tot[n] += e - s; cnt[n] += 1
    wall = pts[-1][0] - pts[0][0]
    lines = [f"# {title}", f"# source: rocprofv3 --kernel-trace ({f.split('gpurun_out/')[-1]}), window = {window}: {len(ev)} dispatches",
             f"# wall {wall/1e6:.2f} ms; kernels running: none {100*conc[0]/wall:.1f} %, one {100*conc[1]/wall:.1f} %, two {100*conc[2]/wall:.1f} %, three or more {100*conc[3]/wall:.1f} %",
             f"{'calls':>7} {'sum_ms':>9} {'excl_ms':>9} {'excl/wall':>9} {'others':>7}  kernel   (excl = wall time with only this kernel name running; others = mean number of other kernels beside it)"]
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:28]:
        lines.append(f"{cnt[k]:7d} {v/1e6:9.3f} {excl[k]/1e6:9.3f} {100*excl[k]/wall:8.1f}% {beside[k]/max(v,1):7.2f}  {k[:90]}")
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
