#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
for tk in 8192 49863; do
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_pf_g$tk -- python /root/repo/tools/probes/prefill_profile.py $tk 1 > /root/repo/gpurun_out/prof_pf_g$tk.log 2>&1
cd /root/repo
python tools/rocprof_csv_summary.py statsdb gpurun_out/prof_pf_g$tk gpurun_out/r02_g_prefill_fast_${tk}_kernel_stats.txt "QCN prompt pass, $tk tokens, FAST mode (8-wave flash attention, chunked delta rule, MFMA router logits), 48 layers, chunk 1024 x 3" 2>&1 | tail -1
grep "prompt pass" gpurun_out/prof_pf_g$tk.log; head -14 gpurun_out/r02_g_prefill_fast_${tk}_kernel_stats.txt
done
