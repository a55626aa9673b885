#!/bin/bash
# Round 4, last lease: the statistics-shift build -- targeted parity first, then the whole evidence script again (source id changed).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04k
mkdir -p $OUT
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $OUT/smoke.json 2> $OUT/smoke.err || { echo "SMOKE FAILED"; tail -25 $OUT/smoke.err; exit 1; }
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x > $OUT/tests_ops.log 2>&1
rc=$?; echo "rc=$rc" >> $OUT/tests_ops.log; tail -4 $OUT/tests_ops.log
[ $rc -ne 0 ] && exit 1
bash tests/tools/r04_final.sh
