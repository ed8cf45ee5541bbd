#!/bin/bash
# round-2 GPU call 4: MLA fast mode tests, V2-Lite side config, rocprof of the fast prompt pass
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/r02_attn_fast_err.txt
timeout 600 python -m pytest tests/test_attn_fast_gpu.py tests/test_mla_gpu.py -q > gpurun_out/r02_pytest4_new.log 2>&1; echo "new tests rc=$?"
tail -25 gpurun_out/r02_pytest4_new.log
grep mla gpurun_out/r02_attn_fast_err.txt 2>/dev/null
timeout 600 python bench.py --config v2lite-q4 --steps 50 --warmup 5 --prefill-tokens 8192,20434 --side-configs "" --no-cpu-baseline > gpurun_out/r02_bench_v2l.json 2> gpurun_out/r02_bench_v2l.err; echo "bench rc=$?"
tail -3 gpurun_out/r02_bench_v2l.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_bench_v2l.json'))
for k in ['value','prefill','prefill_fast','decode_long_context','decode_long_context_32k','decode_long_context_fast','decode_long_context_32k_fast']:
    v=d.get(k);
    if isinstance(v,dict): v={kk:vv for kk,vv in v.items() if kk in('value','by_prompt_length','tok_s','ms_per_step','error')}
    print(k, v)
P
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_pf_fast -- python /root/repo/tools/probes/prefill_profile.py 8192 1 > /root/repo/gpurun_out/prof_pf_fast.log 2>&1
cd /root/repo
python tools/rocprof_csv_summary.py statsdb gpurun_out/prof_pf_fast gpurun_out/r02_a_prefill_fast_8192_kernel_stats.txt "QCN prompt pass, 8192 tokens, FAST attention mode (flash GQA on f16 MFMA), 48 layers" 2>&1 | tail -2
tail -3 gpurun_out/prof_pf_fast.log; head -25 gpurun_out/r02_a_prefill_fast_8192_kernel_stats.txt
