#!/bin/bash
# round-2 closing run: whole GPU suite, the driver's bench command, smoke()
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/r02_attn_fast_err.txt gpurun_out/r02_gguf_prefill_err.txt
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r02_pytest_final.log 2>&1; echo "full suite rc=$?"; tail -4 gpurun_out/r02_pytest_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
SECONDS=0
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_final.json 2> gpurun_out/r02_bench_final.err; echo "bench rc=$? wall=${SECONDS}s"
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_bench_final.json'))
print({k:d[k] for k in ['value','ms_per_step']}, d['roofline']['step_frac_of_hbm_peak'], d['roofline']['kernel'], round(d['roofline']['frac'],4), d['roofline']['traffic_source'])
for k in ['prefill','prefill_fast','prefill_fast_gemm','prefill_experts_only','prefill_experts_only_fast_gemm','decode_long_context','decode_long_context_32k','decode_long_context_fast','decode_long_context_32k_fast','cpu_baseline']:
    v=d.get(k)
    if isinstance(v,dict): v={kk:vv for kk,vv in v.items() if kk in('value','by_prompt_length','tok_s','ms','tok_s_experts_only','unit','cores','error')}
    print(k, v)
for n,c in (d.get('configs') or {}).items():
    print(n, {kk:(vv if not isinstance(vv,dict) else {a:b for a,b in vv.items() if a in ('value','by_prompt_length','tok_s')}) for kk,vv in c.items() if kk in ('value','decode_tok_s','prefill','prefill_fast','prefill_fast_gemm','decode_long_context_fast','decode_long_context_32k_fast','error')})
P
