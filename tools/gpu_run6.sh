#!/bin/bash
# round-2 GPU call 6: full GPU suite (stand-alone ops, per-device LDS table, EP rows path), EP selftest, chunked-LA prefill bench + profile
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02_pytest6_all.log 2>&1; echo "full suite rc=$?"
tail -6 gpurun_out/r02_pytest6_all.log
timeout 900 python bench.py --steps 20 --warmup 5 --prefill-tokens 8192 --side-configs "" --no-cpu-baseline --ep-selftest > gpurun_out/r02_bench_e.json 2> gpurun_out/r02_bench_e.err; echo "bench rc=$?"
tail -3 gpurun_out/r02_bench_e.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_bench_e.json'))
for k in ['value','prefill','prefill_fast','prefill_experts_only','prefill_ep','prefill_ep_235b']:
    v=d.get(k);
    if isinstance(v,dict): v={kk:vv for kk,vv in v.items() if kk in('value','by_prompt_length','tok_s','ms_per_step','error','ms','ms_direct','ratio_vs_direct','ms_per_pass','direct_ms_per_pass')}
    print(k, v)
print({k:(v if not isinstance(v,dict) else list(v.keys())) for k,v in d.items() if 'ep' in k})
P
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_pf_fast3 -- python /root/repo/tools/probes/prefill_profile.py 8192 1 > /root/repo/gpurun_out/prof_pf_fast3.log 2>&1
cd /root/repo
python tools/rocprof_csv_summary.py statsdb gpurun_out/prof_pf_fast3 gpurun_out/r02_c_prefill_fast_8192_kernel_stats.txt "QCN prompt pass, 8192 tokens, FAST mode (flash GQA on f16 MFMA + chunked gated delta rule v2 on f32 MFMA), 48 layers" 2>&1 | tail -2
tail -2 gpurun_out/prof_pf_fast3.log; head -16 gpurun_out/r02_c_prefill_fast_8192_kernel_stats.txt
