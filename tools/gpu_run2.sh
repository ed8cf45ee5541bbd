#!/bin/bash
# round-2 GPU call 2: new kernels (GGUF MFMA prefill, library EP, fast attention), full suite, bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/r02_gguf_prefill_err.txt gpurun_out/r02_attn_fast_err.txt
timeout 600 python -m pytest tests/test_gguf_gpu.py tests/test_ep_gpu.py tests/test_attn_fast_gpu.py -q > gpurun_out/r02_pytest2_new.log 2>&1; echo "new tests rc=$?"
tail -40 gpurun_out/r02_pytest2_new.log
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r02_pytest2.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r02_pytest2.log
cat gpurun_out/r02_gguf_prefill_err.txt gpurun_out/r02_attn_fast_err.txt 2>/dev/null
timeout 900 python bench.py --steps 50 --warmup 5 --ep-selftest --prefill-tokens 8192 --side-configs "" > gpurun_out/r02_bench_b.json 2> gpurun_out/r02_bench_b.err; echo "bench rc=$?"
tail -5 gpurun_out/r02_bench_b.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r02_bench_b.json'))
for k in ['value','decode_long_context','decode_long_context_32k','decode_long_context_fast','decode_long_context_32k_fast','prefill_experts_only_q4k_gguf','prefill_experts_ep_alltoall','prefill_experts_ep_235b','prefill_experts_only']:
    v=d.get(k); 
    if isinstance(v,dict): v={kk:vv for kk,vv in v.items() if kk in('tok_s','ms_per_step','tok_s_experts_only','ms','roofline','error','attention')}
    print(k, v)
P
