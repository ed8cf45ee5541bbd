#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_dl32 -- python /root/repo/tools/probes/decode_long_profile.py 32768 1 1 > /root/repo/gpurun_out/prof_dl32.log 2>&1
cd /root/repo
python tools/rocprof_csv_summary.py statsdb gpurun_out/prof_dl32 gpurun_out/r02_decode_32k_fast_fp8_kernel_stats.txt "QCN decode at position 32766 of a 32768 FP8-E4M3 cache, FAST attention mode, 33 steps" 2>&1 | tail -1
grep "decode at" gpurun_out/prof_dl32.log; head -14 gpurun_out/r02_decode_32k_fast_fp8_kernel_stats.txt
