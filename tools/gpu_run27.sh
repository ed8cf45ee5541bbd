#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gemm_fast_gpu.py -x -q 2>&1 | tail -4; cat gpurun_out/r02_gemm_fast_err.txt | tail -4
