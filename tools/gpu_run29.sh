#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
R=/root/repo/gpurun_out
cd /tmp
rm -rf $R/prof_pf_x
timeout 900 rocprofv3 --kernel-trace --stats -d $R/prof_pf_x -- python /root/repo/tools/probes/prefill_profile.py ${1:-20434} 0 > $R/prof_pf_x.log 2>&1
cd /root/repo
python tools/rocprof_csv_summary.py statsdb gpurun_out/prof_pf_x gpurun_out/r02_i_prefill_exact_kernel_stats.txt "QCN prompt pass, ${1:-20434} tokens + 2048 warm-up, EXACT mode (attention passes A and C on the f32 MFMA), 48 layers, chunk 1024 x 3" 2>&1 | tail -1
grep "prompt pass" gpurun_out/prof_pf_x.log; head -16 gpurun_out/r02_i_prefill_exact_kernel_stats.txt | cut -c1-160
