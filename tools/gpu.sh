#!/bin/bash
# tools/gpu.sh <step> [args] -- the command lines sent to the MI355X box with `gpurun` (one parametrised script; round 3 on).
#   gpurun --timeout 900 -- 'bash tools/gpu.sh fast1'
# Every step writes under gpurun_out/ (merged back by gpurun); summaries worth keeping are copied to profiles/ by hand.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
R=/root/repo/gpurun_out
step="$1"; shift

kstats() {   # kstats <name> <title> -- <command...>: rocprofv3 --kernel-trace --stats of a command, summary table -> gpurun_out/<name>_kernel_stats.txt
    local name="$1" title="$2"; shift 3
    rm -rf $R/prof_$name
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/prof_$name --output-format csv -- "$@" > $R/prof_$name.log 2>&1)
    python tools/rocprof_csv_summary.py stats $R/prof_$name $R/${name}_kernel_stats.txt "$title" 2>&1 | tail -2
    head -30 $R/${name}_kernel_stats.txt
}

case "$step" in
fast1)      # KR_DECODE_FAST bring-up: parity tests, exact vs fast decode bench, kernel trace of the fast graph
    timeout 900 python -m pytest tests/test_decode_fast_gpu.py -x -q 2>&1 | tail -25
    cat $R/r03_decode_fast_err.txt 2>/dev/null
    timeout 600 python tools/probes/decode_fast_bench.py "$@" 2>&1 | tail -60
    kstats r03_decode_fast "QCN Q4 decode step, KR_DECODE_FAST, FP8-E4M3 KV, positions 10.. (tools/probes/decode_fast_bench.py --only fast --steps 30)" -- \
        python /root/repo/tools/probes/decode_fast_bench.py --only fast --steps 30 --route-tokens 0 --out /root/repo/gpurun_out/r03_decode_fast_prof
    ;;
fastbench)  # only the bench probe (+ trace)
    timeout 600 python tools/probes/decode_fast_bench.py "$@" 2>&1 | tail -60
    kstats r03_decode_fast "QCN Q4 decode step, KR_DECODE_FAST, FP8-E4M3 KV, positions 10.. (tools/probes/decode_fast_bench.py --only fast --steps 30)" -- \
        python /root/repo/tools/probes/decode_fast_bench.py --only fast --steps 30 --route-tokens 0 --out /root/repo/gpurun_out/r03_decode_fast_prof
    ;;
fast2)      # remaining FAST tests, in-kernel phase stamps (probe build), then the whole GPU suite
    timeout 900 python -m pytest tests/test_decode_fast_gpu.py -q -k "router or generate" 2>&1 | tail -8
    cat $R/r03_decode_fast_err.txt 2>/dev/null | tail -4
    KRASIS_HIP_LIB=/root/repo/krasis_amd/libkrasis_hip_timing.so timeout 300 python tools/probes/decode_fast_stamps.py 2>&1 | tail -8
    timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6
    ;;
fast3)      # FAST tests + stamps + bench probe (no trace)
    timeout 900 python -m pytest tests/test_decode_fast_gpu.py -x -q 2>&1 | tail -5
    tail -16 $R/r03_decode_fast_err.txt
    KRASIS_HIP_LIB=/root/repo/krasis_amd/libkrasis_hip_timing.so timeout 300 python tools/probes/decode_fast_stamps.py 2>&1 | tail -7
    timeout 600 python tools/probes/decode_fast_bench.py --only fast --route-tokens 200 "$@" 2>&1 | tail -14
    ;;
stamps)
    KRASIS_HIP_LIB=/root/repo/krasis_amd/libkrasis_hip_timing.so timeout 300 python tools/probes/decode_fast_stamps.py 2>&1 | tail -8
    ;;
tests)      # the whole GPU suite, as the driver runs it
    timeout 2400 python -m pytest tests/ -x -q -m gpu "$@" 2>&1 | tail -15
    ;;
bench)      # the driver's bench line
    timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $R/bench_line.json 2> $R/bench_line.err; echo "bench rc=$?"; tail -c 3000 $R/bench_line.json
    ;;
*)
    echo "unknown step $step"; exit 2;;
esac
