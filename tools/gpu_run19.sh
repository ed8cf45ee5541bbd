#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for M in 1024 4096; do
echo "== dense tolerance GEMM, M=$M"; timeout 60 ./tools/probes/gemm_h_timing $M
done 2>&1 | tee gpurun_out/r02_gemm_h_timing2.txt
bash tools/gpu_run21.sh
