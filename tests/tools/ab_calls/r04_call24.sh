#!/bin/bash
# Round 4, lease 24: replayed vs eager Adam moments over 16 steps (new regression test for the replay path).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04w
timeout 300 python -m pytest tests/test_gpu_model.py -q -m gpu -k "adam_moments" > gpurun_out/r04w/moments.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r04w/moments.log | cut -c1-300
grep -n "AssertionError\|^E  " gpurun_out/r04w/moments.log | head -8 | cut -c1-300
