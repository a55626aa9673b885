#!/bin/bash
# Round 4, lease 16c: where the NaNs of "eager generate right behind a train replay, no host synchronize" come from.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04p
mkdir -p $OUT
for m in plain sidestream event; do
  timeout 200 python tests/tools/ab_calls/r04_call15.py small 3 nosync $m > $OUT/diag_$m.log 2>&1; echo "== $m"; tail -10 $OUT/diag_$m.log | cut -c1-150
done
echo "== eager train steps (SAVP_GRAPH=0)"
SAVP_GRAPH=0 timeout 200 python tests/tools/ab_calls/r04_call15.py small 3 nosync plain > $OUT/diag_eager.log 2>&1; tail -10 $OUT/diag_eager.log | cut -c1-150
