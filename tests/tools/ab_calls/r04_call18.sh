#!/bin/bash
# Round 4, lease 18: which boundary of "train replay / eager generate, no host synchronize" is unordered.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04r
mkdir -p $OUT
run() { echo "== $1 ($2)"; env $2 timeout 200 python tests/tools/ab_calls/r04_call15.py small 3 nosync $1 > $OUT/diag_$1_$3.log 2>&1; tail -10 $OUT/diag_$1_$3.log | cut -c1-120 | grep -c nan; tail -2 $OUT/diag_$1_$3.log | cut -c1-120; }
run sync_before_generate X=1 a
run sync_after_generate X=1 a
run sync_in_stage X=1 a
run plain AMD_SERIALIZE_KERNEL=3 ser
run plain HIP_LAUNCH_BLOCKING=1 blk
run plain SAVP_INFER_GRAPH=1 x
