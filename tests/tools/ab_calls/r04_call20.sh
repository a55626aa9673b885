#!/bin/bash
# Round 4, lease 20b: bisect -- without generate; generate without staging copies; default-size arena.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04t
mkdir -p $OUT
echo "== no generate at all"; DIAG_NOGEN=1 timeout 200 python tests/tools/ab_calls/r04_call15.py small 3 nosync trace > $OUT/t_nogen.log 2>&1; grep -c "nonfinite: g" $OUT/t_nogen.log; grep "^i=" $OUT/t_nogen.log | tail -2 | cut -c1-110
echo "== unroll without staging copies"; DIAG_NOGEN=2 timeout 200 python tests/tools/ab_calls/r04_call15.py small 3 nosync trace > $OUT/t_nostage.log 2>&1; grep -c "nonfinite: g" $OUT/t_nostage.log; grep "^i=" $OUT/t_nostage.log | tail -2 | cut -c1-110
echo "== default arena"; timeout 200 python tests/tools/ab_calls/r04_call15.py big 3 nosync trace > $OUT/t_big.log 2>&1; grep -c "nonfinite: g" $OUT/t_big.log; grep "^i=" $OUT/t_big.log | tail -2 | cut -c1-110
echo "== 20 train replays, then generate"; DIAG_NOGEN=1 timeout 200 python tests/tools/ab_calls/r04_call15.py small 20 nosync trace > $OUT/t_pre20.log 2>&1; grep -c "nonfinite: g" $OUT/t_pre20.log; grep "^i=" $OUT/t_pre20.log | tail -2 | cut -c1-110
