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
    cat $R/decode_fast_err.txt 2>/dev/null
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
    cat $R/decode_fast_err.txt 2>/dev/null | tail -4
    KRASIS_HIP_LIB=/root/repo/krasis_amd/libkrasis_hip_timing.so timeout 300 python tools/probes/decode_fast_stamps.py 2>&1 | tail -8
    timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6
    ;;
fast3)      # FAST tests + stamps + bench probe (no trace)
    timeout 900 python -m pytest tests/test_decode_fast_gpu.py -x -q 2>&1 | tail -5
    tail -16 $R/decode_fast_err.txt
    KRASIS_HIP_LIB=/root/repo/krasis_amd/libkrasis_hip_timing.so timeout 300 python tools/probes/decode_fast_stamps.py 2>&1 | tail -7
    timeout 600 python tools/probes/decode_fast_bench.py --only fast --route-tokens 200 "$@" 2>&1 | tail -14
    ;;
ep)         # expert parallelism at world > 1 on one GPU (loopback transport) + bench --gpus refusal
    timeout 900 python -m pytest tests/test_ep_gpu.py -x -q 2>&1 | tail -12
    ;;
gemmab)     # A/B of the 64 x 256 tolerance-GEMM variants (experts only), then the GEMM parity tests on the shipped default
    for v in 0 1 2; do KR_PFH_VARIANT=$v timeout 300 python tools/probes/experts_gemm_probe.py 8 8192 fast 2>&1 | grep experts-only; done
    timeout 300 python tools/probes/experts_gemm_probe.py 8 8192 exact,q4k 2>&1 | grep experts-only
    timeout 900 python -m pytest tests/test_gemm_fast_gpu.py -x -q 2>&1 | tail -4
    ;;
gemmpmc)    # SQ counters of the shipped GEMM kernels (separate pass, counters only)
    rm -rf $R/pmc_gemm
    (cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_WAVES -d $R/pmc_gemm --output-format csv -- \
        python /root/repo/tools/probes/experts_gemm_probe.py 2 8192 exact,fast,q4k,q4kfast > $R/pmc_gemm.log 2>&1)
    python tools/pmc_table.py $R/pmc_gemm $R/r03_gemm_pmc_sq.txt "Experts-only prompt-pass GEMMs, QCN shape, 8192 tokens, 2 layers: SQ counters per launch (rocprofv3 --pmc, counters-only pass)" gemm 2>&1 | tail -2
    head -60 $R/r03_gemm_pmc_sq.txt
    ;;
gemmab2)    # shipped library vs the A/B build (make -C krasis_amd/csrc ab AB_SRC=... AB_DEFS=...), experts only, all 48 layers, interleaved
    for rep in 1 2; do
        timeout 300 python tools/probes/experts_gemm_probe.py 48 8192 fast 2>&1 | grep experts-only | sed 's/^/shipped: /'
        KRASIS_HIP_LIB=/root/repo/krasis_amd/libkrasis_hip_ab.so timeout 300 python tools/probes/experts_gemm_probe.py 48 8192 fast 2>&1 | grep experts-only | sed 's/^/ab:      /'
    done
    ;;
gemmab3)    # A/B build: parity tests of the tolerance forms on the A/B library, then interleaved timing
    KRASIS_HIP_LIB=/root/repo/krasis_amd/libkrasis_hip_ab.so timeout 900 python -m pytest tests/test_gemm_fast_gpu.py tests/test_gguf_gpu.py -x -q 2>&1 | tail -4
    for rep in 1 2; do
        timeout 300 python tools/probes/experts_gemm_probe.py 48 8192 fast,q4kfast 2>&1 | grep experts-only | sed 's/^/shipped: /'
        KRASIS_HIP_LIB=/root/repo/krasis_amd/libkrasis_hip_ab.so timeout 300 python tools/probes/experts_gemm_probe.py 48 8192 fast,q4kfast 2>&1 | grep experts-only | sed 's/^/ab:      /'
    done
    ;;
fastpmc)    # HBM fetch bytes per launch of the KR_DECODE_FAST kernels: counters-only pass (separate from the trace), then the kernel trace of the same command
    rm -rf $R/pmc_fast
    (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/pmc_fast --output-format csv -- python /root/repo/tools/probes/decode_fast_bench.py --only fast --steps 12 --route-tokens 0 --out /root/repo/gpurun_out/r03_decode_fast_pmcrun > $R/pmc_fast.log 2>&1)
    python tools/rocprof_csv_summary.py pmc $R/pmc_fast $R/r03_decode_fast_pmc_fetch_size.txt "QCN Q4 decode step, KR_DECODE_FAST: HBM fetch per launch (rocprofv3 --pmc FETCH_SIZE, counters-only pass; x2 = gfx950 correction)" 2>&1 | tail -2
    head -16 $R/r03_decode_fast_pmc_fetch_size.txt
    kstats r03_decode_fast "QCN Q4 decode step, KR_DECODE_FAST, FP8-E4M3 KV, positions 10.. (tools/probes/decode_fast_bench.py --only fast --steps 30)" -- \
        python /root/repo/tools/probes/decode_fast_bench.py --only fast --steps 30 --route-tokens 0 --out /root/repo/gpurun_out/r03_decode_fast_prof
    ;;
newtests)   # tests added since the last full-suite run
    timeout 900 python -m pytest tests/test_tolerance_peaked_gpu.py tests/test_moe_gpu.py tests/test_decode_gpu.py -x -q 2>&1 | tail -8
    cat $R/tolerance_peaked.txt 2>/dev/null
    ;;
q4k)        # Q4_K tolerance form: parity test, experts-only timing (exact int8 form vs f16 form), peaked-model + export tests that have not run yet
    timeout 900 python -m pytest tests/test_gguf_gpu.py -x -q -k "tolerance_form" 2>&1 | tail -6
    cat $R/r03_q4k_fast_err.txt 2>/dev/null
    timeout 300 python tools/probes/experts_gemm_probe.py 8 8192 q4kfast,q4k,fast 2>&1 | grep experts-only
    timeout 900 python -m pytest tests/test_tolerance_peaked_gpu.py tests/test_moe_gpu.py -x -q 2>&1 | tail -5
    cat $R/tolerance_peaked.txt 2>/dev/null
    ;;
gemmtrace)  # kernel trace of the experts-only tolerance pass (which launches carry the 1.2 ms per layer)
    kstats r03_experts_8192_gemm_fast "QCN experts only, 8192 tokens x 16 layers, tolerance GEMM (tools/probes/experts_gemm_probe.py 16 8192 fast)" -- python /root/repo/tools/probes/experts_gemm_probe.py 16 8192 fast
    ;;
gemmchk)    # tolerance GEMM after a change: parity tests of the tolerance forms, experts-only timing, kernel trace
    timeout 1200 python -m pytest tests/test_gemm_fast_gpu.py tests/test_gguf_gpu.py tests/test_tolerance_peaked_gpu.py -x -q 2>&1 | tail -6
    timeout 300 python tools/probes/experts_gemm_probe.py 48 8192 fast,q4kfast 2>&1 | grep experts-only
    kstats r03_experts_8192_gemm_fast "QCN experts only, 8192 tokens x 16 layers, tolerance GEMM (tools/probes/experts_gemm_probe.py 16 8192 fast)" -- python /root/repo/tools/probes/experts_gemm_probe.py 16 8192 fast
    ;;
pfexact)    # exact prompt pass at 8192 tokens: tok/s twice, then the kernel trace
    timeout 600 python tools/probes/prefill_profile.py 8192 0 2>&1 | tail -3
    timeout 600 python tools/probes/prefill_profile.py 8192 0 2>&1 | tail -3
    kstats r03_prefill_exact_8192 "QCN exact prompt pass, 8192 tokens (tools/probes/prefill_profile.py 8192 0)" -- python /root/repo/tools/probes/prefill_profile.py 8192 0
    ;;
pffast)     # tolerance prompt pass (KR_ATTN_FAST | KR_GEMM_FAST) at 8192 tokens: tok/s, then the kernel trace
    timeout 600 python tools/probes/prefill_profile.py 8192 2 2>&1 | tail -2
    kstats r03_prefill_8192_attn_fast_gemm_fast "QCN prompt pass, KR_ATTN_FAST | KR_GEMM_FAST, 8192 tokens (tools/probes/prefill_profile.py 8192 2)" -- python /root/repo/tools/probes/prefill_profile.py 8192 2
    ;;
ktrace)     # generic: bash tools/gpu.sh ktrace <name> "<title>" -- <command...>  -> gpurun_out/<name>_kernel_stats.txt
    kstats "$@"
    ;;
stamps)
    KRASIS_HIP_LIB=/root/repo/krasis_amd/libkrasis_hip_timing.so timeout 300 python tools/probes/decode_fast_stamps.py 2>&1 | tail -8
    ;;
r4a)        # round 4, first contact: the whole GPU suite (all failures listed), then the driver's bench line
    timeout 1500 python -m pytest tests/ -q -m gpu -x --maxfail=8 2>&1 | tail -40
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/r04_bench_line.json 2> $R/r04_bench_line.err; echo "bench rc=$?"; tail -c 1500 $R/r04_bench_line.err
    python tools/bench_summary.py $R/r04_bench_line.json
    ;;
pf4)        # round 4: kernel trace of the tolerance prompt pass (8192 tokens) and of the decode step
    kstats r04_prefill_8192_attn_fast_gemm_fast "QCN prompt pass, KR_ATTN_FAST | KR_GEMM_FAST, 8192 tokens (tools/probes/prefill_profile.py 8192 2)" -- python /root/repo/tools/probes/prefill_profile.py 8192 2
    kstats r04_decode_fast "QCN Q4 decode step, KR_DECODE_FAST, FP8-E4M3 KV, positions 10.. (tools/probes/decode_fast_bench.py --only fast --steps 30)" -- \
        python /root/repo/tools/probes/decode_fast_bench.py --only fast --steps 30 --route-tokens 0 --out /root/repo/gpurun_out/r04_decode_fast_prof
    ;;
gguf)       # native-GGUF experts inside the decode step: parity tests, decode tok/s of the two GGUF side configurations
    timeout 900 python -m pytest tests/test_gguf_gpu.py tests/test_decode_gpu.py -q -x -k "gguf" 2>&1 | tail -4
    timeout 600 python tools/probes/gguf_decode_bench.py 30 2>&1 | grep decode
    ;;
ks)         # kernel table of the tolerance prompt pass at a given length
    P=${1:-49863}
    kstats r04_prefill_${P}_attn_fast_gemm_fast "QCN prompt pass, KR_ATTN_FAST | KR_GEMM_FAST, $P tokens (tools/probes/prefill_profile.py $P 2)" -- python /root/repo/tools/probes/prefill_profile.py $P 2
    rm -rf $R/prof_*
    ;;
tl)         # concurrency view of the tolerance prompt pass: which kernels own the wall clock with three chunks in flight
    P=${1:-8192}
    rm -rf $R/prof_tl
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/prof_tl --output-format csv -- python /root/repo/tools/probes/prefill_profile.py $P 2 > $R/prof_tl.log 2>&1)
    grep "prompt pass" $R/prof_tl.log
    python tools/rocprof_timeline.py $R/prof_tl $R/r04_prefill_${P}_timeline.txt "QCN prompt pass, KR_ATTN_FAST | KR_GEMM_FAST, $P tokens: concurrency of the kernel trace (tools/probes/prefill_profile.py $P 2)"
    cat $R/r04_prefill_${P}_timeline.txt | cut -c1-170
    [ -n "$2" ] && python tools/probes/tl_queues.py $R/prof_tl > $R/tl_queues.txt 2>&1
    rm -rf $R/prof_tl
    ;;
r4final)    # round 4 closing run: the whole GPU suite as the driver runs it, the driver's bench line, PMC fetch pass + kernel trace of the decode step
    # (raw rocprof directories are removed on the box once summarised: gpurun merges at most 64 MiB back)
    timeout 1800 python -m pytest tests/ -x -q -m gpu > $R/r04_gpu_tests_full.txt 2>&1; echo "pytest rc=$?"
    grep -E " passed| failed| error|skipped" $R/r04_gpu_tests_full.txt | tail -3 | tee $R/r04_gpu_tests_tail.txt
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/r04_bench_line.json 2> $R/r04_bench_line.err; echo "bench rc=$?"; tail -c 400 $R/r04_bench_line.err
    python tools/bench_summary.py $R/r04_bench_line.json
    rm -rf $R/pmc_fast
    (cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/pmc_fast --output-format csv -- python /root/repo/tools/probes/decode_fast_bench.py --only fast --steps 12 --route-tokens 0 --out /root/repo/gpurun_out/r04_decode_fast_pmcrun > $R/pmc_fast.log 2>&1)
    python tools/rocprof_csv_summary.py pmc $R/pmc_fast $R/r04_decode_fast_pmc_fetch_size.txt "QCN Q4 decode step, KR_DECODE_FAST: HBM fetch per launch (rocprofv3 --pmc FETCH_SIZE, counters-only pass; x2 = gfx950 correction)" 2>&1 | tail -2
    head -14 $R/r04_decode_fast_pmc_fetch_size.txt
    kstats r04_decode_fast "QCN Q4 decode step, KR_DECODE_FAST, FP8-E4M3 KV, positions 10.. (tools/probes/decode_fast_bench.py --only fast --steps 30)" -- \
        python /root/repo/tools/probes/decode_fast_bench.py --only fast --steps 30 --route-tokens 0 --out /root/repo/gpurun_out/r04_decode_fast_prof
    kstats r04_prefill_8192_attn_fast_gemm_fast "QCN prompt pass, KR_ATTN_FAST | KR_GEMM_FAST, 8192 tokens (tools/probes/prefill_profile.py 8192 2)" -- python /root/repo/tools/probes/prefill_profile.py 8192 2 > /dev/null
    kstats r04_prefill_49863_attn_fast_gemm_fast "QCN prompt pass, KR_ATTN_FAST | KR_GEMM_FAST, 49863 tokens (tools/probes/prefill_profile.py 49863 2)" -- python /root/repo/tools/probes/prefill_profile.py 49863 2 > /dev/null
    (echo "# tools/probes/flash_timing (1024 queries at the end of a 32768-position E4M3 cache, QCN heads), then the same at 8192 positions"; timeout 120 tools/probes/flash_timing; timeout 120 tools/probes/flash_timing 8192) > $R/r04_flash_timing.txt 2>&1
    (echo "# tools/probes/lac_timing 2752 / 8192 (chunked gated delta rule, 32 heads): correctness against the per-token recurrence in double, launch times, phases of a prep workgroup and a scan step"; timeout 120 tools/probes/lac_timing 2752; timeout 120 tools/probes/lac_timing 8192) > $R/r04_lac_timing.txt 2>&1
    (echo "# tools/probes/clock_probe: MFMA 32x32x16 f16 issue rate and shader clock, one workgroup / every CU"; timeout 60 tools/probes/clock_probe) > $R/r04_clock_probe.txt 2>&1
    tail -3 $R/r04_flash_timing.txt | cut -c1-200
    rm -rf $R/prof_* $R/pmc_* $R/*.log
    ;;
r4c)        # round 4: lean select, MLA tolerance wiring, tolerance router logits, GGUF gate fuse -- their tests, then decode / prompt-pass timings
    timeout 1200 python -m pytest tests/test_decode_fast_gpu.py tests/test_mla_gpu.py tests/test_router_gpu.py tests/test_decode_gpu.py tests/test_gemm_fast_gpu.py tests/test_tolerance_peaked_gpu.py -q -x 2>&1 | tail -6
    timeout 600 python tools/probes/decode_fast_bench.py --only fast --route-tokens 200 2>&1 | tail -8
    timeout 300 python tools/probes/prefill_profile.py 8192 2 2>&1 | tail -1
    timeout 300 python tools/probes/prefill_profile.py 20434 2 2>&1 | tail -1
    ;;
r4b)        # round 4: tests that failed / are new since r4a, the bench line, the kernel trace of the bench command
    timeout 900 python -m pytest tests/test_ep_gpu.py tests/test_sampler_gpu.py tests/test_decode_gpu.py tests/test_gguf_gpu.py tests/test_prefill_model_gpu.py -q -x 2>&1 | tail -6
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/r04_bench_line.json 2> $R/r04_bench_line.err; echo "bench rc=$?"; tail -c 600 $R/r04_bench_line.err
    python tools/bench_summary.py $R/r04_bench_line.json
    ;;
r5a)        # round 5, first contact: ADVICE r4 fixes (their tests), the compact bench line as the driver reads it, cache-residency upper bound of the decode step, kernel trace
    timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_ep_gpu.py tests/test_decode_fast_gpu.py tests/test_prefill_model_gpu.py -q -x 2>&1 | tail -5
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/r05_bench_stdout.txt 2> $R/r05_bench_stderr.txt; echo "bench rc=$?"
    tail -1 $R/r05_bench_stdout.txt > $R/r05_bench_line.json; echo "last stdout line: $(wc -c < $R/r05_bench_line.json) bytes, stdout lines: $(wc -l < $R/r05_bench_stdout.txt)"
    python -c "import json,sys; d=json.load(open('$R/r05_bench_line.json')); print({k: d[k] for k in ('value','value_exact','ms_per_step')}, d['roofline'])"
    cp $R/bench_detail.json $R/r05_bench_detail.json 2>/dev/null
    timeout 600 python tools/probes/decode_mall_probe.py 2>&1 | tail -24
    kstats r05_decode_fast_a "QCN Q4 decode step, KR_DECODE_FAST, FP8-E4M3 KV, positions 10.. (tools/probes/decode_fast_bench.py --only fast --steps 30)" -- \
        python /root/repo/tools/probes/decode_fast_bench.py --only fast --steps 30 --route-tokens 0 --out /root/repo/gpurun_out/r05_decode_fast_prof
    rm -rf $R/prof_* $R/*.log
    ;;
r5final)    # round 5 closing run: GPU suite as the driver runs it; PMC fetch pass FIRST (its json lands in profiles/ on the box), then the driver's bench line (so
            # roofline.traffic in the line is the number of the json committed beside it); kernel traces; stamps
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
    timeout 1800 python -m pytest tests/ -x -q -m gpu > $R/r05_gpu_tests_full.txt 2>&1; echo "pytest rc=$?"
    grep -E " passed| failed| error|skipped" $R/r05_gpu_tests_full.txt | tail -3 | tee $R/r05_gpu_tests_summary.txt
    rm -rf $R/pmc_fast
    (cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/pmc_fast --output-format csv -- python /root/repo/tools/probes/decode_fast_bench.py --only fast --steps 12 --route-tokens 0 --out /root/repo/gpurun_out/r05_decode_fast_pmcrun > $R/pmc_fast.log 2>&1)
    python tools/rocprof_csv_summary.py pmc $R/pmc_fast $R/r05_decode_fast_pmc_fetch_size.txt "QCN Q4 decode step, KR_DECODE_FAST: HBM fetch per launch (rocprofv3 --pmc FETCH_SIZE, counters-only pass; x2 = gfx950 correction)" 2>&1 | tail -2
    cp $R/r05_decode_fast_pmc_fetch_size.json $R/r05_decode_fast_pmc_fetch_size.txt profiles/ 2>/dev/null
    head -14 $R/r05_decode_fast_pmc_fetch_size.txt
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/r05_bench_stdout.txt 2> $R/r05_bench_stderr.txt; echo "bench rc=$?"
    tail -1 $R/r05_bench_stdout.txt > $R/r05_bench_line.json; echo "last stdout line: $(wc -c < $R/r05_bench_line.json) bytes, stdout lines: $(wc -l < $R/r05_bench_stdout.txt)"
    cp $R/bench_detail.json $R/r05_bench_detail.json 2>/dev/null
    python tools/bench_summary.py $R/r05_bench_detail.json
    kstats r05_decode_fast "QCN Q4 decode step, KR_DECODE_FAST, FP8-E4M3 KV, positions 10.. (tools/probes/decode_fast_bench.py --only fast --steps 30)" -- \
        python /root/repo/tools/probes/decode_fast_bench.py --only fast --steps 30 --route-tokens 0 --out /root/repo/gpurun_out/r05_decode_fast_prof
    kstats r05_decode_exact "QCN Q4 decode step, exact mode, FP8-E4M3 KV, positions 10.. (tools/probes/decode_fast_bench.py --only exact --steps 30)" -- \
        python /root/repo/tools/probes/decode_fast_bench.py --only exact --steps 30 --route-tokens 0 --out /root/repo/gpurun_out/r05_decode_exact_prof
    kstats r05_prefill_8192_attn_fast_gemm_fast "QCN prompt pass, KR_ATTN_FAST | KR_GEMM_FAST, 8192 tokens (tools/probes/prefill_profile.py 8192 2)" -- python /root/repo/tools/probes/prefill_profile.py 8192 2 > /dev/null
    [ -f krasis_amd/libkrasis_hip_timing.so ] || make -C krasis_amd/csrc timing > $R/make_timing.log 2>&1      # the stamp build is not shipped: built on the box
    KRASIS_HIP_LIB=/root/repo/krasis_amd/libkrasis_hip_timing.so LAYERS=47 STAMPS_OUT=$R/r05_decode_fast_stamps.txt timeout 250 python tools/probes/decode_fast_stamps.py 2>&1 | tail -8
    KRASIS_HIP_LIB=/root/repo/krasis_amd/libkrasis_hip_timing.so LAYERS=47 WG_OUT=$R/r05_decode_fast_wg_times.txt timeout 250 python tools/probes/decode_fast_wg_times.py 2>&1 | tail -7
    rm -rf $R/prof_* $R/pmc_* $R/*.log
    ;;
r6final)    # round 6 closing run: GPU suite as the driver runs it; PMC fetch pass FIRST (its json lands in profiles/ on the box), then the driver's bench line (so
            # roofline.traffic in the line is the number of the json committed beside it); kernel traces; stamps
    timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
    timeout 1800 python -m pytest tests/ -x -q -m gpu > $R/r06_gpu_tests_full.txt 2>&1; echo "pytest rc=$?"
    grep -E " passed| failed| error|skipped" $R/r06_gpu_tests_full.txt | tail -3 | tee $R/r06_gpu_tests_summary.txt
    rm -rf $R/pmc_fast
    (cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/pmc_fast --output-format csv -- python /root/repo/tools/probes/decode_fast_bench.py --only fast --steps 12 --route-tokens 0 --out /root/repo/gpurun_out/r06_decode_fast_pmcrun > $R/pmc_fast.log 2>&1)
    python tools/rocprof_csv_summary.py pmc $R/pmc_fast $R/r06_decode_fast_pmc_fetch_size.txt "QCN Q4 decode step, KR_DECODE_FAST: HBM fetch per launch (rocprofv3 --pmc FETCH_SIZE, counters-only pass; x2 = gfx950 correction)" 2>&1 | tail -2
    cp $R/r06_decode_fast_pmc_fetch_size.json $R/r06_decode_fast_pmc_fetch_size.txt profiles/ 2>/dev/null
    head -14 $R/r06_decode_fast_pmc_fetch_size.txt
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/r06_bench_stdout.txt 2> $R/r06_bench_stderr.txt; echo "bench rc=$?"
    tail -1 $R/r06_bench_stdout.txt > $R/r06_bench_line.json; echo "last stdout line: $(wc -c < $R/r06_bench_line.json) bytes, stdout lines: $(wc -l < $R/r06_bench_stdout.txt)"
    cp $R/bench_detail.json $R/r06_bench_detail.json 2>/dev/null
    python tools/bench_summary.py $R/r06_bench_detail.json
    kstats r06_decode_fast "QCN Q4 decode step, KR_DECODE_FAST, FP8-E4M3 KV, positions 10.. (tools/probes/decode_fast_bench.py --only fast --steps 30)" -- \
        python /root/repo/tools/probes/decode_fast_bench.py --only fast --steps 30 --route-tokens 0 --out /root/repo/gpurun_out/r06_decode_fast_prof
    kstats r06_decode_exact "QCN Q4 decode step, exact mode, FP8-E4M3 KV, positions 10.. (tools/probes/decode_fast_bench.py --only exact --steps 30)" -- \
        python /root/repo/tools/probes/decode_fast_bench.py --only exact --steps 30 --route-tokens 0 --out /root/repo/gpurun_out/r06_decode_exact_prof
    kstats r06_prefill_8192_attn_fast_gemm_fast "QCN prompt pass, KR_ATTN_FAST | KR_GEMM_FAST, 8192 tokens (tools/probes/prefill_profile.py 8192 2)" -- python /root/repo/tools/probes/prefill_profile.py 8192 2 > /dev/null
    [ -f krasis_amd/libkrasis_hip_timing.so ] || make -C krasis_amd/csrc timing > $R/make_timing.log 2>&1      # the stamp build is not shipped: built on the box
    KRASIS_HIP_LIB=/root/repo/krasis_amd/libkrasis_hip_timing.so LAYERS=47 STAMPS_OUT=$R/r06_decode_fast_stamps.txt timeout 250 python tools/probes/decode_fast_stamps.py 2>&1 | tail -8
    KRASIS_HIP_LIB=/root/repo/krasis_amd/libkrasis_hip_timing.so LAYERS=47 WG_OUT=$R/r06_decode_fast_wg_times.txt timeout 250 python tools/probes/decode_fast_wg_times.py 2>&1 | tail -7
    rm -rf $R/prof_* $R/pmc_* $R/*.log
    ;;
r5t)        # round 5: a list of test files + the decode probe in both modes
    timeout 1200 python -m pytest "$@" -q -x > $R/r05_tests.txt 2>&1; grep -E " passed| failed|rror" $R/r05_tests.txt | tail -4
    timeout 300 python tools/probes/decode_fast_bench.py --route-tokens 0 --out gpurun_out/r05_df 2>&1 | grep -E "^fast|^exact|logits"
    ;;
r5u)        # round 5: whole GPU suite, then the decode probe with the one-launch final norm + vocabulary projection on / off (interleaved)
    timeout 2400 python -m pytest tests/ -x -q -m gpu > $R/r05_gpu_tests_full.txt 2>&1; grep -E " passed| failed|rror" $R/r05_gpu_tests_full.txt | tail -4 | tee $R/r05_gpu_tests_summary.txt
    for rep in 1 2; do for v in 1 0; do
        echo "lm_fused=$v"; timeout 300 python tools/probes/decode_fast_bench.py --only fast --route-tokens 0 --opt lm_fused=$v --out gpurun_out/r05_lm$v 2>&1 | grep -E "^fast|lm_head|rmsnorm"
    done; done
    ;;
r5line)     # round 5: PMC fetch pass (json into profiles/ on the box) -> the driver's bench line -> kernel trace of the fast step
    rm -rf $R/pmc_fast
    (cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/pmc_fast --output-format csv -- python /root/repo/tools/probes/decode_fast_bench.py --only fast --steps 12 --route-tokens 0 --out /root/repo/gpurun_out/r05_decode_fast_pmcrun > $R/pmc_fast.log 2>&1)
    python tools/rocprof_csv_summary.py pmc $R/pmc_fast $R/r05_decode_fast_pmc_fetch_size.txt "QCN Q4 decode step, KR_DECODE_FAST: HBM fetch per launch (rocprofv3 --pmc FETCH_SIZE, counters-only pass; x2 = gfx950 correction)" 2>&1 | tail -2
    cp $R/r05_decode_fast_pmc_fetch_size.json $R/r05_decode_fast_pmc_fetch_size.txt profiles/ 2>/dev/null
    head -16 $R/r05_decode_fast_pmc_fetch_size.txt
    timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/r05_bench_stdout.txt 2> $R/r05_bench_stderr.txt; echo "bench rc=$?"
    tail -1 $R/r05_bench_stdout.txt > $R/r05_bench_line.json; echo "last stdout line: $(wc -c < $R/r05_bench_line.json) bytes, stdout lines: $(wc -l < $R/r05_bench_stdout.txt)"
    cp $R/bench_detail.json $R/r05_bench_detail.json 2>/dev/null
    python tools/bench_summary.py $R/r05_bench_detail.json
    rm -rf $R/prof_* $R/pmc_* $R/*.log
    ;;
r5s)        # round 5: only the stamp / per-workgroup probes of the KR_DECODE_FAST launches (timing build made on the box)
    [ -f krasis_amd/libkrasis_hip_timing.so ] || make -C krasis_amd/csrc timing > $R/make_timing.log 2>&1
    KRASIS_HIP_LIB=/root/repo/krasis_amd/libkrasis_hip_timing.so LAYERS=47 STAMPS_OUT=$R/r05_decode_fast_stamps.txt timeout 250 python tools/probes/decode_fast_stamps.py 2>&1 | tail -8
    KRASIS_HIP_LIB=/root/repo/krasis_amd/libkrasis_hip_timing.so LAYERS=47 WG_OUT=$R/r05_decode_fast_wg_times.txt timeout 250 python tools/probes/decode_fast_wg_times.py 2>&1 | tail -7
    ;;
tests)      # the whole GPU suite, as the driver runs it
    timeout 2400 python -m pytest tests/ -x -q -m gpu "$@" 2>&1 | tail -15
    ;;
bench)      # the driver's bench line
    timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 "$@" > $R/bench_line.json 2> $R/bench_line.err; echo "bench rc=$?"; tail -c 3000 $R/bench_line.json
    ;;
exact1)     # round 6: exact decode step after the router / ADVICE changes: bit-exact tests, then the exact-vs-fast bench probe and the kernel trace of the exact graph
    timeout 1500 python -m pytest tests/test_decode_gpu.py tests/test_router_gpu.py tests/test_prefill_model_gpu.py -x -q 2>&1 | tail -5
    timeout 600 python tools/probes/decode_fast_bench.py --only exact --route-tokens 0 "$@" 2>&1 | tail -12
    kstats r06_decode_exact "QCN Q4 decode step, exact mode, FP8-E4M3 KV, positions 10.. (tools/probes/decode_fast_bench.py --only exact --steps 30)" -- \
        python /root/repo/tools/probes/decode_fast_bench.py --only exact --steps 30 --route-tokens 0 --out /root/repo/gpurun_out/r06_decode_exact_prof
    ;;
routefast)  # round 6: the tolerance router of the prompt pass: tests, then the kernel trace of one chunk's logits + select
    timeout 600 python -m pytest tests/test_router_gpu.py -x -q 2>&1 | tail -3
    kstats r06_route_fast "Batch router of the tolerance prompt pass, QCN shape, 2752 tokens x 23 calls (tools/probes/route_fast_probe.py)" -- python /root/repo/tools/probes/route_fast_probe.py 2752 20
    ;;
ringx)      # round 6: kernel trace of the experts-only pass at the prompt pass's chunk length, ring kernel forced vs register-staged
    kstats r06_experts_2752_ring "Experts only, QCN shape, 2752 tokens x 8 layers, LDS-ring GEMM forced (tools/probes/experts_gemm_probe.py 8 2752 ring)" -- python /root/repo/tools/probes/experts_gemm_probe.py 8 2752 ring
    kstats r06_experts_2752_staged "Experts only, QCN shape, 2752 tokens x 8 layers, register-staged GEMM (tools/probes/experts_gemm_probe.py 8 2752 staged)" -- python /root/repo/tools/probes/experts_gemm_probe.py 8 2752 staged
    ;;
pfmode)     # round 6: kernel trace of the prompt pass at 8192 tokens in mode $1 (0 exact, 1 KR_ATTN_FAST, 2 both bits), three chunks in flight
    kstats r06_prefill_8192_mode$1 "QCN prompt pass, mode $1 (0 exact, 1 KR_ATTN_FAST, 2 + KR_GEMM_FAST), 8192 tokens (tools/probes/prefill_profile.py 8192 $1)" -- python /root/repo/tools/probes/prefill_profile.py 8192 $1 > /dev/null
    head -34 $R/r06_prefill_8192_mode$1_kernel_stats.txt
    ;;
pf1)        # round 6: the tolerance prompt pass as ONE chunk on one stream (no overlap): stand-alone duration of every kernel
    kstats r06_prefill_8192_one_chunk "QCN prompt pass, KR_ATTN_FAST | KR_GEMM_FAST, 8192 tokens as ONE chunk, depth 1 (tools/probes/prefill_profile.py 8192 2 48 8192 1): stand-alone kernel durations" -- \
        python /root/repo/tools/probes/prefill_profile.py 8192 2 48 8192 1
    ;;
ringp)      # round 6: the stand-alone probe of the ring GEMM (dense problem): bit comparison, timing of both forms, stamps
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function -o /tmp/grp tools/probes/gemm_ring_probe.hip 2>&1 | grep -E "error" -A3
    /tmp/grp 4096 2048 12288 | grep -v "^  mismatch"; /tmp/grp 8192 2048 4096 | grep -v "^  mismatch"; /tmp/grp 2752 2048 12288 | grep -v "^  mismatch"
    ;;
ringabl)    # round 6: ablations of the ring GEMM's loop (results wrong by construction): what bounds a unit
    for abl in 0 1 2 4 8 3 7 15; do
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -Wno-unused-function -DPFR_ABL=$abl -o /tmp/grp$abl tools/probes/gemm_ring_probe.hip 2>&1 | grep -E "error" -A3
        echo "== PFR_ABL=$abl (1 no de-quantization, 2 no requests in the loop, 4 no fragment reads in the loop, 8 no barriers in the loop)"
        /tmp/grp$abl 4096 2048 12288 | grep "ring   rep [23]\|ring stamps"
    done
    ;;
ring1)      # round 6: LDS-ring tolerance GEMM -- bit identity against the register-staged kernels, tolerance tests, interleaved experts-only A/B
    timeout 900 python -m pytest tests/test_gemm_ring_gpu.py -x -q 2>&1 | tail -15
    timeout 900 python -m pytest tests/test_gemm_fast_gpu.py -x -q 2>&1 | tail -5
    timeout 600 python tools/probes/experts_gemm_probe.py 16 8192 staged,fast,staged,fast 2>&1 | grep experts-only
    timeout 600 python tools/probes/experts_gemm_probe.py 16 2752 staged,fast,staged,fast 2>&1 | grep experts-only
    ;;
*)
    echo "unknown step $step"; exit 2;;
esac
