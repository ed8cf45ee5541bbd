#!/bin/bash
# tolerance GEMM iteration: parity tests, probe, kernel stats + SQ counters of the experts at M = 8192
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
R=/root/repo/gpurun_out
rm -f $R/r02_gemm_fast_err.txt
timeout 600 python -m pytest tests/test_gemm_fast_gpu.py -x -q > $R/r02_gemm_fast_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $R/r02_gemm_fast_pytest.log
timeout 600 python tools/probes/gemm_fast_probe.py ${1:-8192} > $R/r02_gemm_fast_probe.txt 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids $R/r02_gemm_fast_probe.txt | tail -20
cd /tmp
rm -rf $R/gh_stats_1 $R/gh_pmc_sq
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gh_stats_1 -- python /root/repo/tools/probes/gemm_h_layer.py 8192 1 > $R/gh_stats_1.log 2>&1
( cd /root/repo; python tools/rocprof_csv_summary.py statsdb gpurun_out/gh_stats_1 gpurun_out/r02_gemm_h_stats_fast1.txt "experts of 4 QCN layers x 3, M = 8192, gemm fast=1" | tail -1; head -8 gpurun_out/r02_gemm_h_stats_fast1.txt )
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $R/gh_pmc_sq --output-format csv -- python /root/repo/tools/probes/gemm_h_layer.py 8192 1 2 2 > $R/gh_pmc_sq.log 2>&1
cd /root/repo
python tools/pmc_table.py gpurun_out/gh_pmc_sq gpurun_out/r02_gemm_h_pmc_sq.txt "tolerance GEMM, experts M = 8192: sq counters" pfh_gemm 2>&1 | tail -2; cat gpurun_out/r02_gemm_h_pmc_sq.txt
exit 0
