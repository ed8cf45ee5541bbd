#!/bin/bash
# exact + tolerance GEMM changes (whole-line A requests, unguarded store path, vectorised combine): whole GPU suite, then the probe
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
R=/root/repo/gpurun_out
rm -f $R/r02_gemm_fast_err.txt
timeout 1800 python -m pytest tests -m gpu -x -q > $R/r02_pytest_full.log 2>&1; echo "pytest rc=$?"; tail -5 $R/r02_pytest_full.log
timeout 600 python tools/probes/gemm_fast_probe.py 8192 > $R/r02_gemm_fast_probe.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids $R/r02_gemm_fast_probe.txt | tail -20
cd /tmp
for fast in 1 0; do
rm -rf $R/gh_stats_$fast
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gh_stats_$fast -- python /root/repo/tools/probes/gemm_h_layer.py 8192 $fast > $R/gh_stats_$fast.log 2>&1
( cd /root/repo; python tools/rocprof_csv_summary.py statsdb gpurun_out/gh_stats_$fast gpurun_out/r02_gemm_h_stats_fast$fast.txt "experts of 4 QCN layers x 3, M = 8192, gemm fast=$fast" | tail -1; head -9 gpurun_out/r02_gemm_h_stats_fast$fast.txt )
done
exit 0
