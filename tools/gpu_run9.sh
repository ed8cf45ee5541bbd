#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_prefill_gpu.py tests/test_prefill_model_gpu.py tests/test_ep_gpu.py tests/test_moe_gpu.py tests/test_checkpoint_gpu.py -q -x 2>&1 | tail -4
timeout 400 python tools/probes/prefill_sweep.py 8192 1 2>&1 | grep tokens | grep "chunk 1024 depth 3\|chunk 2048 depth 3\|chunk 4096 depth 2"
timeout 400 python tools/probes/prefill_sweep.py 20434 1 2>&1 | grep tokens | grep "chunk 1024 depth 3\|chunk 2048 depth 3\|chunk 4096 depth 2\|chunk 4096 depth 3"
timeout 600 python bench.py --steps 5 --warmup 2 --prefill-tokens 8192 --side-configs "" --no-cpu-baseline --no-long-context > gpurun_out/r02_bench_f.json 2> gpurun_out/r02_bench_f.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_bench_f.json'))
for k in ['value','prefill','prefill_fast','prefill_experts_only']:
    v=d.get(k);
    if isinstance(v,dict): v={kk:vv for kk,vv in v.items() if kk in('value','by_prompt_length','tok_s','ms','tok_s_experts_only')}
    print(k, v)
P
