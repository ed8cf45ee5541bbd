#!/bin/bash
# round-2 GPU call 1: full GPU test suite, timing probes, default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02_pytest1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest1.log
tail -15 gpurun_out/r02_pytest1.log
timeout 60 tools/probes/route_timing > gpurun_out/r02_route_timing.txt 2>&1
timeout 60 tools/probes/norm_timing > gpurun_out/r02_norm_timing.txt 2>&1
timeout 900 python bench.py --steps 100 --warmup 5 > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/r02_bench_a.json; tail -5 gpurun_out/r02_bench_a.err
