#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
for cfg in "1024 3"; do set -- $cfg
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_pf_e$1 -- python /root/repo/tools/probes/prefill_profile.py 8192 1 48 $1 $2 > /root/repo/gpurun_out/prof_pf_e$1.log 2>&1
cd /root/repo
python tools/rocprof_csv_summary.py statsdb gpurun_out/prof_pf_e$1 gpurun_out/r02_e_prefill_fast_8192_chunk$1_depth$2_kernel_stats.txt "QCN prompt pass, 8192 tokens, FAST mode, chunk $1, depth $2, router logits on the f32 MFMA, 48 layers" 2>&1 | tail -1
grep "prompt pass" gpurun_out/prof_pf_e$1.log; head -12 gpurun_out/r02_e_prefill_fast_8192_chunk$1_depth$2_kernel_stats.txt
done
timeout 400 python tools/probes/prefill_sweep.py 8192 1 2>&1 | grep tokens | grep -v "chunk 512"
timeout 400 python tools/probes/prefill_sweep.py 8192 0 2>&1 | grep tokens | grep "chunk 1024\|chunk 2048"
