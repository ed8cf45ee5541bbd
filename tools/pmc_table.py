#!/usr/bin/env python3
"""rocprofv3 --pmc CSV -> per (kernel, grid) averages of every counter.  pmc_table.py <dir> <out.txt> [title] [name-filter]"""
import collections, csv, glob, sys
d, out = sys.argv[1:3]; title = sys.argv[3] if len(sys.argv) > 3 else out; flt = sys.argv[4] if len(sys.argv) > 4 else ""
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if flt and flt not in name: continue
    key = (name[:70], r.get("Grid_Size", "?"))
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines = [f"# {title}", f"# source: {f.split('gpurun_out/')[-1]}; averages per launch"]
for key in sorted(agg, key=lambda k: -sum(dur[k])):
    n = len(next(iter(agg[key].values())))
    lines.append(f"{key[0]}  grid={key[1]}  launches={n}  avg_us={sum(dur[key]) / len(dur[key]) / 1e3:.1f}")
    for c, v in sorted(agg[key].items()):
        lines.append(f"    {c:32s} {sum(v) / len(v):16.1f}")
open(out, "w").write("\n".join(lines) + "\n")
