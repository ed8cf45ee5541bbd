#!/usr/bin/env python3
"""print the fields of a bench.py JSON line that a round's notes quote (tools/gpu.sh r4a / r4b / bench)"""
import functools
import json
import sys

d = json.load(open(sys.argv[1]))
g = lambda *k: functools.reduce(lambda a, b: a.get(b, {}) if isinstance(a, dict) else {}, k, d)
print("metric", d.get("metric"))
print("value", d.get("value"), "exact", d.get("value_exact"), "fast", d.get("value_fast"))
print("decode_generate", json.dumps(d.get("decode_generate"))[:900])
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "frac_raw_events", "us_per_launch", "us_per_launch_raw_events", "step_frac_of_hbm_peak")})
for k in ("prefill", "prefill_fast", "prefill_fast_gemm"):
    print(k, g(k, "by_prompt_length"), "allocs", g(k, "allocs_in_timed_region"), "ms_all", g(k, "ms_all"))
print("experts", g("prefill_experts_only", "tok_s_experts_only"), g("prefill_experts_only_fast_gemm", "tok_s_experts_only"), g("prefill_experts_only_fast_gemm", "roofline", "frac"))
for c, v in (d.get("configs") or {}).items():
    print(c, {k: (round(v[k], 1) if isinstance(v.get(k), float) else v.get(k)) for k in ("decode_tok_s", "decode_fast_tok_s", "step_frac_of_hbm_peak", "decode_fast_frac_of_hbm_peak", "error") if k in v},
          g("configs", c, "prefill", "value"), g("configs", c, "prefill_fast_gemm", "value"))
if "expert_parallel_selftest_one_rank_rccl" in d:
    print("ep selftest", json.dumps(d["expert_parallel_selftest_one_rank_rccl"])[:1200])
print("cpu_baseline", g("cpu_baseline", "value"), g("cpu_baseline", "cores"), g("cpu_baseline", "v2lite_q4k_cpu", "value"))
