#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace into a small text table (the artefact committed under profiles/)."""
import glob
import sqlite3
import sys


def main(path, out=None):
    dbs = glob.glob(path + "/**/*_results.db", recursive=True) if not path.endswith(".db") else [path]
    lines = []
    for dbp in dbs:
        cur = sqlite3.connect(dbp).cursor()
        lines.append(f"# rocprofv3 --kernel-trace --stats   source: {dbp.split('gpurun_out/')[-1]}")
        lines.append(f"{'calls':>8} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
        for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            lines.append(f"{calls:8d} {total:12.1f} {avg:10.3f} {pct:6.2f}  {name[:150]}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
