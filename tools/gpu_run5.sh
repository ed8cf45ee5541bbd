#!/bin/bash
# round-2 GPU call 5: chunked gated delta rule (FAST prompt pass), full GPU suite after the destroy-order fix, QCN prefill in both modes, rocprof
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/r02_attn_fast_err.txt
timeout 900 python -m pytest tests/test_attn_fast_gpu.py -q -x > gpurun_out/r02_pytest5_new.log 2>&1; echo "new tests rc=$?"
tail -15 gpurun_out/r02_pytest5_new.log
grep -i "chunked" gpurun_out/r02_attn_fast_err.txt 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02_pytest5_all.log 2>&1; echo "full suite rc=$?"
tail -8 gpurun_out/r02_pytest5_all.log
timeout 900 python bench.py --steps 20 --warmup 5 --prefill-tokens 8192,20434 --side-configs "" --no-cpu-baseline > gpurun_out/r02_bench_d.json 2> gpurun_out/r02_bench_d.err; echo "bench rc=$?"
tail -3 gpurun_out/r02_bench_d.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_bench_d.json'))
for k in ['value','prefill','prefill_fast']:
    v=d.get(k);
    if isinstance(v,dict): v={kk:vv for kk,vv in v.items() if kk in('value','by_prompt_length','tok_s','ms_per_step','error')}
    print(k, v)
P
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_pf_fast2 -- python /root/repo/tools/probes/prefill_profile.py 8192 1 > /root/repo/gpurun_out/prof_pf_fast2.log 2>&1
cd /root/repo
python tools/rocprof_csv_summary.py statsdb gpurun_out/prof_pf_fast2 gpurun_out/r02_b_prefill_fast_8192_kernel_stats.txt "QCN prompt pass, 8192 tokens, FAST mode (flash GQA on f16 MFMA + chunked gated delta rule on f32 MFMA), 48 layers" 2>&1 | tail -2
tail -3 gpurun_out/prof_pf_fast2.log; head -25 gpurun_out/r02_b_prefill_fast_8192_kernel_stats.txt
