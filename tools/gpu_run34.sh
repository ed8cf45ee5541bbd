#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gguf_gpu.py tests/test_loader.py -x -q 2>&1 | tail -3
timeout 300 python - <<'P'
import bench, torch
r = bench.prefill_experts_gguf(0, torch)
print("q4k gguf experts-only:", round(r["ms"], 2), "ms", round(r["tok_s_experts_only"]), "tok/s")
P
