# prompt-pass sweep: prompt length x chunk size x chunks in flight (bench.py builds the model once per run)
for tk in ${TOKENS:-8192}; do for cfg in ${CFGS:-"1024 3" "1024 4" "1024 6" "1024 8" "512 8"}; do set -- $cfg
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --prefill-tokens $tk --prefill-reps 2 --prefill-chunk $1 --prefill-depth $2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tokens $tk chunk $1 depth $2', round(d['prefill']['value']))"
done; done
