#!/bin/bash
# Round 4, lease 23: the whole GPU suite on the final tree (after the c5 gate followed its two measurements).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/r04/r04_full_gputest_rerun.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r04/r04_full_gputest_rerun.log
grep -n "worst per-variable\|rel L2" gpurun_out/r04/r04_full_gputest_rerun.log | head -5
