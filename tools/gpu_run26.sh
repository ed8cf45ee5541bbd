#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_router_gpu.py tests/test_decode_gpu.py tests/test_mla_gpu.py tests/test_fp8_kv.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --gpus 1 --steps 100 --warmup 10 --prefill-tokens "" --side-configs "" --no-cpu-baseline --no-long-context > gpurun_out/r02_bench_decode_only.json 2> gpurun_out/r02_bench_decode_only.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_bench_decode_only.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline'].get('avg_launch_us'), {k:round(v,2) for k,v in (d.get('per_launch_us') or d['roofline'].get('per_launch_us') or {}).items()} if isinstance(d.get('per_launch_us') or d['roofline'].get('per_launch_us'), dict) else '')
P
