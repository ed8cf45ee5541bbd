#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for d in 3 4 6; do
timeout 600 python tools/probes/prefill_profile.py 20434 0 48 1024 $d 2>&1 | grep "prompt pass" | sed "s/$/ depth $d/"
done
