#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
R=/root/repo/gpurun_out
timeout 300 python bench.py --ep-selftest > $R/r02_ep_selftest.json 2> $R/r02_ep_selftest.err; echo "ep-selftest rc=$?"; tail -c 1500 $R/r02_ep_selftest.json
cd /tmp
rm -rf $R/pmc_xm
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVES -d $R/pmc_xm --output-format csv -- python /root/repo/tools/probes/prefill_profile.py 8192 0 12 > $R/pmc_xm.log 2>&1
cd /root/repo
python tools/pmc_table.py gpurun_out/pmc_xm gpurun_out/r02_exact_attention_mfma_pmc_sq.txt "QCN exact prompt pass (12 layers, 8192 tokens): SQ counters of the matrix-core attention passes" mfma_kernel 2>&1 | tail -2; head -40 gpurun_out/r02_exact_attention_mfma_pmc_sq.txt
