#!/usr/bin/env python3
"""Format rocprofv3 CSV outputs into the small text summaries committed under profiles/.

  rocprof_csv_summary.py stats <dir> <out.txt> [title]      # *_kernel_stats.csv   (--kernel-trace --stats)
  rocprof_csv_summary.py statsdb <dir> <out.txt> [title]    # *_results.db (rocpd SQLite, the default output format)
  rocprof_csv_summary.py pmc   <dir> <out.txt> [title]      # *_counter_collection.csv (--pmc FETCH_SIZE): per-kernel HBM fetch bytes per launch

FETCH_SIZE is reported in KiB of 64-B requests; on gfx950 wide streaming reads are tallied at half their size
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section), so the table prints both the raw and the x2-corrected figure."""
import collections
import csv
import glob
import sys


def stats(d, out, title):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    lines = [f"# {title}", f"# source: rocprofv3 --kernel-trace --stats ({f.split('gpurun_out/')[-1]})",
             f"{'calls':>8} {'total_ms':>11} {'avg_us':>10} {'pct':>6} {'min_us':>9} {'max_us':>9}  kernel"]
    for r in rows[:40]:
        lines.append(f"{int(r['Calls']):8d} {float(r['TotalDurationNs'])/1e6:11.3f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f} "
                     f"{float(r['MinNs'])/1e3:9.2f} {float(r['MaxNs'])/1e3:9.2f}  {r['Name'][:110]}")
    open(out, "w").write("\n".join(lines) + "\n")


def statsdb(d, out, title):
    """same table from the rocpd SQLite output (rocprofv3 default output format): durations from the `kernels` view"""
    import sqlite3
    f = glob.glob(d + "/**/*results.db", recursive=True)[0]
    c = sqlite3.connect(f)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = [f"# {title}", f"# source: rocprofv3 --kernel-trace --stats ({f.split('gpurun_out/')[-1]}, kernels view: end - start per dispatch)",
             f"{'calls':>8} {'total_ms':>11} {'avg_us':>10} {'pct':>6} {'min_us':>9} {'max_us':>9}  kernel"]
    for r in rows[:40]:
        lines.append(f"{r[1]:8d} {r[2]/1e6:11.3f} {r[3]/1e3:10.2f} {100*r[2]/tot:6.2f} {r[4]/1e3:9.2f} {r[5]/1e3:9.2f}  {r[0][:110]}")
    open(out, "w").write("\n".join(lines) + "\n")


def pmc(d, out, title):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    raw = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            raw[r["Kernel_Name"].split("(")[0]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r.get("Grid_Size") or 0)))
    # one template instantiation launched on very different grids is two different jobs (round 5: kr_fdm_kernel<4,1,8,1> is the in-projection of every layer AND, once per
    # step, the final norm + vocabulary projection on a 12 x larger grid): launches whose grid is >= 4 x (or <= 1/4 of) the name's most frequent grid get their own row
    agg = collections.defaultdict(list)
    for k, v in raw.items():
        common = collections.Counter(x[2] for x in v).most_common(1)[0][0]
        for x in v:
            far = common > 0 and x[2] > 0 and (x[2] >= 4 * common or 4 * x[2] <= common)
            agg[(k + "@grid%d" % x[2]) if far else k].append(x[:2])
    lines = [f"# {title}", f"# source: rocprofv3 --kernel-trace --pmc FETCH_SIZE ({f.split('gpurun_out/')[-1]}); per launch averages",
             f"{'launches':>8} {'fetch_KiB_raw':>14} {'fetch_MiB_x2':>13} {'avg_us':>9}  kernel"]
    for k, v in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        n = len(v); fs = sum(x[0] for x in v) / n; t = sum(x[1] for x in v) / n
        lines.append(f"{n:8d} {fs:14.1f} {2 * fs / 1024:13.3f} {t / 1e3:9.2f}  {k[:100]}")
    open(out, "w").write("\n".join(lines) + "\n")
    import json   # machine-readable twin: corrected HBM fetch bytes per launch, read by bench.py for roofline.traffic
    json.dump({"source": f.split("gpurun_out/")[-1], "unit": "bytes per launch (FETCH_SIZE KiB x 1024 x 2, gfx950 correction)",
               "kernels": {k.replace("void ", "").replace(" ", ""): 2.0 * 1024.0 * sum(x[0] for x in v) / len(v) for k, v in agg.items()}},
              open(out.rsplit(".", 1)[0] + ".json", "w"), indent=1)


if __name__ == "__main__":
    mode, d, out = sys.argv[1:4]
    title = sys.argv[4] if len(sys.argv) > 4 else out
    {"stats": stats, "statsdb": statsdb, "pmc": pmc}[mode](d, out, title)
