#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
R=/root/repo/gpurun_out
cd /tmp
for mode in 2; do
rm -rf $R/prof_pf_m$mode
timeout 900 rocprofv3 --kernel-trace --stats -d $R/prof_pf_m$mode -- python /root/repo/tools/probes/prefill_profile.py 8192 $mode > $R/prof_pf_m$mode.log 2>&1
cd /root/repo
python tools/rocprof_csv_summary.py statsdb gpurun_out/prof_pf_m$mode gpurun_out/r02_h_prefill_8192_mode$mode.txt "QCN prompt pass, 8192 tokens + 2048 warm-up, mode $mode (1 = KR_ATTN_FAST, 2 = KR_ATTN_FAST | KR_GEMM_FAST), 48 layers, chunk 1024 x 3" 2>&1 | tail -1
head -14 gpurun_out/r02_h_prefill_8192_mode$mode.txt | cut -c1-150
done
