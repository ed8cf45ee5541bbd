#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
R=/root/repo/gpurun_out
cd /tmp
rm -rf $R/prof_v2l_x
timeout 900 rocprofv3 --kernel-trace --stats -d $R/prof_v2l_x -- python /root/repo/tools/probes/prefill_profile_v2l.py 8192 0 > $R/prof_v2l_x.log 2>&1
cd /root/repo
python tools/rocprof_csv_summary.py statsdb gpurun_out/prof_v2l_x gpurun_out/r02_v2lite_prefill_exact_kernel_stats.txt "DeepSeek-V2-Lite shape, prompt pass 8192 tokens + 2048 warm-up, EXACT mode (MLA scores / weighted sum on the f32 MFMA), 27 layers" 2>&1 | tail -1
grep "prompt pass" gpurun_out/prof_v2l_x.log; head -14 gpurun_out/r02_v2lite_prefill_exact_kernel_stats.txt | cut -c1-170
