#!/bin/bash
# Round 4, lease 15: diagnostic (eager generate between train replays).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04o
mkdir -p $OUT
timeout 200 python tests/tools/ab_calls/r04_call15.py small > $OUT/diag_small.log 2>&1; tail -14 $OUT/diag_small.log | cut -c1-330
timeout 200 python tests/tools/ab_calls/r04_call15.py > $OUT/diag_default.log 2>&1; tail -12 $OUT/diag_default.log | cut -c1-330
