#!/bin/bash
# Round 4, lease 22: hipMemsetAsync replaced by a fill kernel everywhere (csrc/zero_fill.h): the failing sequences, then the evidence script.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04v
mkdir -p $OUT
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $OUT/smoke.json 2> $OUT/smoke.err || { echo "SMOKE FAILED"; tail -25 $OUT/smoke.err; exit 1; }
DIAG_NOGEN=1 timeout 200 python tests/tools/ab_calls/r04_call15.py small 3 nosync trace > $OUT/t_nogen.log 2>&1; echo "train replays only: non-finite lines $(grep -c 'nonfinite: g' $OUT/t_nogen.log)"; grep "^i=" $OUT/t_nogen.log | tail -1 | cut -c1-120
timeout 200 python tests/tools/ab_calls/r04_call15.py small 3 nosync trace > $OUT/t_gen.log 2>&1; echo "with eager generate: non-finite lines $(grep -c 'nonfinite: g' $OUT/t_gen.log)"; grep "^i=" $OUT/t_gen.log | tail -1 | cut -c1-120
timeout 300 python -m pytest tests/test_gpu_model.py -q -m gpu -k "generate_replays" > $OUT/test_new.log 2>&1; rc=$?; tail -2 $OUT/test_new.log | cut -c1-200
[ $rc -ne 0 ] && { grep -n "Error\|assert" $OUT/test_new.log | head -10; exit 1; }
[ "$(grep -c 'nonfinite: g' $OUT/t_nogen.log)" != "0" ] && { echo "STILL NON-FINITE"; exit 1; }
bash tests/tools/r04_final.sh
