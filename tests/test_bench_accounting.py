"""bench.py's accounting that needs no GPU: algorithmic bytes per decoded token against SURVEY 8(d)'s per-unit figures, GEMM MACs per prompt token,
the PMC-traffic lookup of a kind launched in several template forms, and the CPU legs' oracle entry points (tiny budgets)."""
import json
import os

import bench


def test_algorithmic_bytes_match_survey_figures():
    ab = bench.algorithmic_bytes(48, bench.B4)
    assert abs(ab["total"] - 2.0626e9) / 2.0626e9 < 5e-3            # 2.06 GB per QCN Q4 token (VERDICT r2 recomputed 2.0626 GB)
    assert abs(ab["moe_w13"] / 48 - 11.9e6) / 11.9e6 < 2e-2          # 10 routed + shared gate|up tiles of one layer
    assert abs(ab["moe_w2"] / 48 - 5.9e6) / 5.9e6 < 3e-2
    ab8 = bench.algorithmic_bytes(48, bench.B8)
    assert 1.8 < ab8["total"] / ab["total"] < 2.0                    # INT8-g128: twice the weight bytes; router gates, states and KV do not scale
    q = bench.algorithmic_bytes_q235(94, bench.B4)
    assert abs(q["moe_w13"] + q["moe_w2"] - 7.32e9) / 7.32e9 < 1e-2  # 7.32 GB of expert weights per Qwen3-235B token (8 of 128 experts x 94 layers)
    assert 11.0e9 < q["total"] < 11.4e9                              # + attention projections 3.46 GB, lm_head 0.32 GB, router gates 0.10 GB


def test_gemm_macs_per_token():
    m = bench.qcn_gemm_macs_per_token(48)
    assert abs(2 * m - 6.40e9) / 6.40e9 < 2e-2                       # 6.40 GFLOP per prompt token
    assert bench.q235_gemm_macs_per_token(94) > 3 * m


def test_pmc_traffic_lookup_handles_template_families(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"; prof.mkdir()
    (prof / "r99_pmc_x.json").write_text(json.dumps({"kernels": {"kr_fdm_kernel<4,1,8>": 12.0e6, "kr_fdm_kernel<4,4,4>": 4.0e6, "kr_fw13_kernel<4,4>": 12.2e6}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    v, src = bench.pmc_traffic("kr_fw13_kernel<4,4>")
    assert v == 12.2e6 and src == "r99_pmc_x.json"
    v, src = bench.pmc_traffic("kr_fdm_kernel<4,1,8>|<4,4,4>")
    assert v == 8.0e6 and "mean of" in src
    assert bench.pmc_traffic("kr_nothing") == (None, None)


def test_side_config_names_are_described():
    for name in ("v2lite-q4", "qcn-q8", "qcn-q4k-gguf", "qwen3-235b-q4", "qcn-q4"):
        assert name in bench.WORKLOAD and len(bench.WORKLOAD[name]) > 20


def test_compact_line_fits_the_driver(tmp_path):
    """the LAST stdout line is one compact JSON object (<= 6 KB target, 12 KB hard cap) whatever the full result dict holds;
    the r02-r04 full lines (13-25 KB; r04's was the one the driver could not parse) are the inputs."""
    import glob
    import json
    import os
    import bench
    files = sorted(glob.glob(os.path.join(bench.ROOT, "profiles", "r0[234]_bench_line.json")))
    assert files
    for f in files:
        full = json.load(open(f))
        out, text = bench.compact_line(full, "gpurun_out/bench_detail.json")
        assert len(text) <= bench.LINE_TARGET_BYTES, (f, len(text))
        assert "\n" not in text
        back = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in back, (f, k)
        assert abs(back["value"] - full["value"]) < 1e-2 * full["value"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in back["roofline"], k
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in back["cpu_baseline"], k
        assert "workload" in back["config"] and "model" not in back["config"]
        # numbers only below the contract keys: no string longer than a label anywhere in the side legs
        def walk(x, path):
            if isinstance(x, dict):
                for k, v in x.items():
                    walk(v, path + [k])
            elif isinstance(x, str) and path[0] not in ("metric", "config", "dtype", "cpu_baseline"):
                assert len(x) <= 130, (path, x)
        walk(back, [])


def test_compact_line_hard_cap_drops_optional_groups():
    import json
    import bench
    full = {"metric": "m", "value": 1.0, "unit": "tok/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": 1.0, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int4", "data": "synthetic", "config": {"workload": "w"}, "roofline": {"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": None},
            "cpu_baseline": {"value": 1.0, "unit": "tok/s", "cores": 1, "kind": "port"},
            "configs": {("cfg%d" % i): {"decode_tok_s": 1.0 * i, "decode_fast_tok_s": 2.0 * i, "step_frac_of_hbm_peak": 0.1} for i in range(400)}}
    out, text = bench.compact_line(full, None)
    assert len(text) <= bench.LINE_HARD_CAP_BYTES
    assert "configs" not in out and out["roofline"]["frac"] == 0.1 and json.loads(text)["value"] == 1.0


def test_compact_line_of_a_multi_gpu_result():
    """the N > 1 line (main_multi): contract keys survive, every expert-parallel leg is reduced to numbers"""
    import json
    import bench
    leg = {"tok_s": 1234.5, "ms_per_step": 0.81, "hip_graph": True, "form": "x" * 300}
    full = {"metric": "decode tok/s, Qwen3-Coder-Next Q4 expert-parallel @8 MI355X", "value": 1234.5, "unit": "tok/s", "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 0.81,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int4-g128 w x int16 act -> i32, f32 scales", "dtype_note": "y" * 400, "data": "synthetic",
            "config": {"workload": "w", "layers": 48, "kv": "FP8", "parallelism": "ep8: " + "z" * 400, "scope": "s" * 300, "value_is": "expert-parallel decode, KR_DECODE_FAST, hipGraph replay"},
            "roofline": {"bound": "hbm", "peak": 8000.0, "unit": "GB/s per GPU", "step_algorithmic_bytes_per_gpu": 5e8, "achieved": 617.0, "frac": 0.077, "kernel": "whole step", "traffic": None},
            "decode_ep_exact": dict(leg), "decode_ep_fast": dict(leg), "decode_ep_fast_graph": dict(leg), "rccl_ranks": 8, "experts_per_gpu": 64,
            "prefill_model_ep": {"value": 3e5, "unit": "tok/s", "tokens_per_gpu": 8192, "tokens_total": 65536, "ms": 218.0, "roofline": {"frac": 0.03}, "attention": "a" * 200},
            "prefill_experts_ep_alltoall": {"tok_s_experts_only": 7e5, "ms": 93.0, "tokens_total": 65536, "roofline": {"frac": 0.05}, "note": "n" * 500},
            "replicas": {"tok_s_aggregate": 5000.0, "tok_s_per_replica": 625.0, "note": "r" * 200},
            "qwen3_235b_ep": {"value": 900.0, "frac_of_hbm_peak_per_gpu": 0.1, "decode_ep_fast_graph": dict(leg), "workload": "q" * 200}}
    out, text = bench.compact_line(full, "gpurun_out/bench_detail.json")
    back = json.loads(text)
    assert len(text) <= bench.LINE_TARGET_BYTES
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in back, k
    assert back["n_gpus"] == 8 and back["scaling"] == "strong" and back["decode_ep_fast_graph"]["tok_s"] == 1234.5
    assert back["prefill_model_ep"]["tok_s"] == 300000.0 and back["qwen3_235b_ep"]["decode_ep_fast_graph"] == 1234.5
    assert back["roofline"]["traffic"] is None and back["roofline"]["step_bytes_per_gpu"] == 5e8


def test_the_line_is_the_last_stdout_line_even_with_native_output(tmp_path):
    """RCCL prints its version banner through libc's stdout, which on a pipe stays buffered until exit -- behind the JSON line (seen with --ep-selftest, round 5).
    emit_line writes native buffers out BEFORE the line and closes stdout after it: whatever a native library prints later never reaches the driver."""
    import subprocess
    import sys
    prog = r"""
import ctypes, json, os, sys, glob, types
sys.path.insert(0, %r)
import bench
libc = ctypes.CDLL(None)
libc.printf(b"native banner, buffered in libc\n")            # like RCCL's version text: not flushed yet
full = json.load(open(sorted(glob.glob(os.path.join(bench.ROOT, "profiles", "r04_bench_line.json")))[0]))
args = types.SimpleNamespace(detail_file=%r, cpu_seconds=5)
bench.emit_line(full, args)
libc.printf(b"native text at teardown\n")                     # like a message at communicator destroy / exit
print("python text after the line")
""" % (bench.ROOT, str(tmp_path / "detail.json"))
    r = subprocess.run([sys.executable, "-c", prog], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert lines[0].startswith("native banner"), lines[:2]
    back = json.loads(lines[-1])                                  # the LAST stdout line is the compact object
    assert "value" in back and "roofline" in back
    assert len(lines) == 2
