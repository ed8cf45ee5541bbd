#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mla_gpu.py tests/test_checkpoint_gpu.py tests/test_attn_fast_gpu.py tests/test_gemm_fast_gpu.py tests/test_prefill_model_gpu.py -x -q 2>&1 | tail -4
for v in 1 0; do
if [ $v = 1 ]; then export KR_EXACT_ATTN_VALU=1; else unset KR_EXACT_ATTN_VALU; fi
echo "== V2-Lite exact prompt pass, per-token attention launches = $v"
timeout 600 python tools/probes/prefill_profile_v2l.py 8192 0 2>&1 | grep -i "prompt pass"
done
