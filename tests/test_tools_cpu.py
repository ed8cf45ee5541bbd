"""The profile tooling the committed summaries under profiles/ come from: tools/rocprof_timeline.py (concurrency view of a rocprofv3 kernel trace) on a
hand-made trace whose exclusive / overlapped times are known."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_timeline_exclusive_time_and_concurrency(tmp_path):
    d = tmp_path / "prof" / "host"
    d.mkdir(parents=True)
    rows = [
        # a warm-up dispatch long before (the "last" window must cut it off)
        ("warm(int)", 0, 1000),
        ("warm(int)", 1000, 2000),
        ("warm(int)", 2000, 3000),
        ("warm(int)", 3000, 4000),
        # the timed window: A alone 100..200, A+B 200..300, B alone 300..400, gap 400..500, C alone 500..600
        ("void kern_a<2>(Args)", 1_000_100, 1_000_300),
        ("kern_b(Args)", 1_000_200, 1_000_400),
        ("kern_c()", 1_000_500, 1_000_600),
    ]
    with open(d / "1_kernel_trace.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for i, (n, s, e) in enumerate(rows):
            w.writerow(["KERNEL_DISPATCH", 1, 1 + i % 2, 1 + i % 2, n, s, e])
    out = tmp_path / "tl.txt"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_timeline.py"), str(tmp_path / "prof"), str(out), "synthetic", "last"], check=True)
    text = out.read_text().splitlines()
    head = [l for l in text if l.startswith("# wall")][0]
    assert "wall 0.00 ms" in head                      # 500 ns
    assert "none 20.0 %" in head and "one 60.0 %" in head and "two 20.0 %" in head
    table = {l.split()[-1] if "kern" in l else None: l.split() for l in text if "kern" in l}
    a = [v for k, v in table.items() if k and k.startswith("kern_a")][0]
    b = [v for k, v in table.items() if k and k.startswith("kern_b")][0]
    c = [v for k, v in table.items() if k and k.startswith("kern_c")][0]
    # columns: calls sum_ms excl_ms excl/wall others kernel
    assert a[0] == "1" and b[0] == "1" and c[0] == "1"
    assert float(a[3].rstrip("%")) == 20.0 and float(b[3].rstrip("%")) == 20.0 and float(c[3].rstrip("%")) == 20.0
    assert float(a[4]) == 0.5 and float(b[4]) == 0.5 and float(c[4]) == 0.0


def test_pmc_summary_gives_a_kernel_on_two_grid_classes_two_rows(tmp_path):
    """tools/rocprof_csv_summary.py pmc: one template instantiation launched on very different grids is two different jobs (round 5: kr_fdm_kernel<4,1,8,1> is
    every layer's in-projection AND, once per step, the final norm + vocabulary projection on a 12 x larger grid) -- the rare grid class gets its own row / json key,
    the x2 gfx950 correction is applied to both"""
    import json
    d = tmp_path / "pmc" / "host"
    d.mkdir(parents=True)
    with open(d / "7_counter_collection.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Grid_Size", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
        for i in range(6):
            w.writerow([98816, "void kr_fdm_kernel<4, 1, 8, 1>(void const*, int)", "FETCH_SIZE", 6000.0, 1000 * i, 1000 * i + 700])
        w.writerow([1215488, "void kr_fdm_kernel<4, 1, 8, 1>(void const*, int)", "FETCH_SIZE", 78000.0, 9000, 9000 + 31000])
        w.writerow([98816, "void kr_fdm_kernel<4, 1, 8, 1>(void const*, int)", "SOMETHING_ELSE", 1.0, 0, 1])
        w.writerow([256, "kr_small(int)", "FETCH_SIZE", 10.0, 0, 100])
    out = tmp_path / "pmc.txt"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocprof_csv_summary.py"), "pmc", str(tmp_path / "pmc"), str(out), "synthetic"], check=True)
    ks = json.load(open(tmp_path / "pmc.json"))["kernels"]
    assert ks["kr_fdm_kernel<4,1,8,1>"] == 2.0 * 1024.0 * 6000.0
    assert ks["kr_fdm_kernel<4,1,8,1>@grid1215488"] == 2.0 * 1024.0 * 78000.0
    assert ks["kr_small"] == 2.0 * 1024.0 * 10.0 and len(ks) == 3
    text = out.read_text()
    assert "@grid1215488" in text and text.count("kr_fdm_kernel") == 2
