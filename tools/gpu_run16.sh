#!/bin/bash
# where the tolerance GEMM's time goes: kernel trace + SQ counters + L2 counters on 4 layers of experts at M = 8192
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
R=/root/repo/gpurun_out
cd /tmp
for fast in 1 0; do
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gh_stats_$fast -- python /root/repo/tools/probes/gemm_h_layer.py 8192 $fast > $R/gh_stats_$fast.log 2>&1
( cd /root/repo; python tools/rocprof_csv_summary.py statsdb gpurun_out/gh_stats_$fast gpurun_out/r02_gemm_h_stats_fast$fast.txt "experts of 4 QCN layers x 3, M = 8192, gemm fast=$fast" | tail -1; head -10 gpurun_out/r02_gemm_h_stats_fast$fast.txt )
done
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $R/gh_pmc_sq --output-format csv -- python /root/repo/tools/probes/gemm_h_layer.py 8192 1 2 2 > $R/gh_pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_SALU -d $R/gh_pmc_sq2 --output-format csv -- python /root/repo/tools/probes/gemm_h_layer.py 8192 1 2 2 > $R/gh_pmc_sq2.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $R/gh_pmc_tcc --output-format csv -- python /root/repo/tools/probes/gemm_h_layer.py 8192 1 2 2 > $R/gh_pmc_tcc.log 2>&1
cd /root/repo
for p in sq sq2 tcc; do python tools/pmc_table.py gpurun_out/gh_pmc_$p gpurun_out/r02_gemm_h_pmc_$p.txt "tolerance GEMM, experts M = 8192: $p counters" pfh_gemm 2>&1 | tail -2; cat gpurun_out/r02_gemm_h_pmc_$p.txt; tail -3 gpurun_out/gh_pmc_$p.log | grep -i "error\|invalid" ; done
