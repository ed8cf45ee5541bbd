#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for M in 4096; do
echo "== form 1 (64 x 256, two waves per SIMD), M=$M"; KR_PFH_FORM=1 timeout 60 ./tools/probes/gemm_h_timing $M
echo "== form 2 (128 x 256), M=$M"; KR_PFH_FORM=2 timeout 60 ./tools/probes/gemm_h_timing $M
done 2>&1 | tee gpurun_out/r02_gemm_h_timing2.txt
timeout 300 python -m pytest tests/test_gemm_fast_gpu.py -x -q 2>&1 | tail -3
