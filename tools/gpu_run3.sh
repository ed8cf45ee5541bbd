#!/bin/bash
# round-2 GPU call 3: flash prefill + fixes; profile of the prompt pass in fast mode
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/r02_gguf_prefill_err.txt gpurun_out/r02_attn_fast_err.txt
timeout 600 python -m pytest tests/test_attn_fast_gpu.py tests/test_gguf_gpu.py tests/test_ep_gpu.py -q > gpurun_out/r02_pytest3_new.log 2>&1; echo "new tests rc=$?"
tail -30 gpurun_out/r02_pytest3_new.log
cat gpurun_out/r02_attn_fast_err.txt 2>/dev/null
timeout 1200 python bench.py --steps 50 --warmup 5 --ep-selftest --side-configs "" --no-cpu-baseline > gpurun_out/r02_bench_c.json 2> gpurun_out/r02_bench_c.err; echo "bench rc=$?"
tail -5 gpurun_out/r02_bench_c.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_bench_c.json'))
for k in ['value','prefill','prefill_fast','decode_long_context','decode_long_context_32k','decode_long_context_fast','decode_long_context_32k_fast','prefill_experts_ep_alltoall','prefill_experts_only']:
    v=d.get(k);
    if isinstance(v,dict): v={kk:vv for kk,vv in v.items() if kk in('value','by_prompt_length','tok_s','ms_per_step','tok_s_experts_only','ms','error')}
    print(k, v)
P
