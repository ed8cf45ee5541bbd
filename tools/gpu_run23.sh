#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
R=/root/repo/gpurun_out
timeout 900 python -m pytest tests/test_router_gpu.py tests/test_prefill_model_gpu.py tests/test_mla_gpu.py tests/test_attn_fast_gpu.py tests/test_gemm_fast_gpu.py -x -q 2>&1 | tail -3
timeout 600 python tools/probes/gemm_fast_probe.py 8192 > $R/r02_gemm_fast_probe.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids $R/r02_gemm_fast_probe.txt | tail -11
