for dp in 2 3 4; do python bench.py --steps 2 --warmup 1 --no-cpu-baseline --prefill-tokens 20480 --prefill-reps 1 --prefill-depth $dp 2>&1 | tail -1 > gpurun_out/sw_$dp.json; done
