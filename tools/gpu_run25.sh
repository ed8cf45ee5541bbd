#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_prefill_model_gpu.py tests/test_attn_fast_gpu.py tests/test_gemm_fast_gpu.py tests/test_perplexity.py tests/test_checkpoint_gpu.py -x -q 2>&1 | tail -3
bash tools/gpu_run24.sh
