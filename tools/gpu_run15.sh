#!/bin/bash
# tolerance GEMM: parity tests, then the probe (experts-only exact vs fast; whole pass swept over chunk / depth)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/r02_gemm_fast_err.txt
timeout 600 python -m pytest tests/test_gemm_fast_gpu.py -x -q > gpurun_out/r02_gemm_fast_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r02_gemm_fast_pytest.log
cat gpurun_out/r02_gemm_fast_err.txt 2>/dev/null
timeout 600 python tools/probes/gemm_fast_probe.py 8192 > gpurun_out/r02_gemm_fast_probe.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids gpurun_out/r02_gemm_fast_probe.txt | tail -30
