#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mla_gpu.py tests/test_checkpoint_gpu.py tests/test_prefill_model_gpu.py tests/test_fp8_kv.py -x -q 2>&1 | tail -3
timeout 600 python tools/probes/prefill_profile_v2l.py 8192 0 2>&1 | grep -i "prompt pass"
timeout 600 python tools/probes/prefill_profile.py 8192 0 2>&1 | grep "prompt pass"
timeout 600 python tools/probes/prefill_profile.py 35139 0 2>&1 | grep "prompt pass"
