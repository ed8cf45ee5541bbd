#!/bin/bash
# round-2 final measurements: new PPL-delta test, the driver's bench command (timed), rocprof stats + PMC fetch pass of the decode step
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_attn_fast_gpu.py -q -k "perplexity_delta" 2>&1 | tail -3
grep "ppl delta" gpurun_out/r02_attn_fast_err.txt
SECONDS=0
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err; echo "bench rc=$? wall=${SECONDS}s"
tail -2 gpurun_out/r02_bench_final.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_bench_final.json'))
print({k:d[k] for k in ['value','ms_per_step']}, d['roofline']['step_frac_of_hbm_peak'], d['roofline']['kernel'], d['roofline']['frac'])
for k in ['decode_other_kv','prefill','prefill_fast','prefill_experts_only','prefill_experts_only_q4k_gguf','decode_long_context','decode_long_context_32k','decode_long_context_fast','decode_long_context_32k_fast','cpu_baseline']:
    v=d.get(k)
    if isinstance(v,dict): v={kk:vv for kk,vv in v.items() if kk in('value','by_prompt_length','tok_s','ms','tok_s_experts_only','int8_TOPS_useful','unit','cores','error','frac_of_i8_peak_useful')}
    print(k, v)
for n,c in (d.get('configs') or {}).items():
    print(n, {kk:(vv if not isinstance(vv,dict) else {a:b for a,b in vv.items() if a in ('value','by_prompt_length','tok_s')}) for kk,vv in c.items() if kk in ('value','prefill','prefill_fast','decode_long_context_fast','decode_long_context_32k_fast','error')})
P
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_dec_r02 -- python /root/repo/bench.py --steps 20 --warmup 5 --prefill-tokens "" --side-configs "" --no-cpu-baseline --no-long-context > /root/repo/gpurun_out/prof_dec_r02.log 2>&1
cd /root/repo
python tools/rocprof_csv_summary.py statsdb gpurun_out/prof_dec_r02 gpurun_out/r02_decode_step_kernel_stats.txt "QCN Q4 decode step, FP8-E4M3 KV, positions 10.., bench.py --steps 20 --warmup 5 (decode leg only)" 2>&1 | tail -1
head -16 gpurun_out/r02_decode_step_kernel_stats.txt
cd /tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE -d /root/repo/gpurun_out/pmc_dec_r02 --output-format csv -- python /root/repo/bench.py --steps 8 --warmup 2 --prefill-tokens "" --side-configs "" --no-cpu-baseline --no-long-context > /root/repo/gpurun_out/pmc_dec_r02.log 2>&1
cd /root/repo
python tools/rocprof_csv_summary.py pmc gpurun_out/pmc_dec_r02 gpurun_out/r02_decode_step_pmc_fetch_size.txt "QCN Q4 decode step: HBM fetch bytes per launch (rocprofv3 --pmc FETCH_SIZE, separate pass)" 2>&1 | tail -1
head -12 gpurun_out/r02_decode_step_pmc_fetch_size.txt
