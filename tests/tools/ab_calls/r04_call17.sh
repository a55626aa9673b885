#!/bin/bash
# Round 4, lease 17: replays moved off the legacy default stream (_StepProgram.run): the failing sequence, the graph tests, step time A/B.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04q
mkdir -p $OUT
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $OUT/smoke.json 2> $OUT/smoke.err || { echo "SMOKE FAILED"; tail -25 $OUT/smoke.err; exit 1; }
timeout 200 python tests/tools/ab_calls/r04_call15.py small 3 nosync plain > $OUT/diag_plain.log 2>&1; echo "== diag (default stream, no host synchronize)"; tail -10 $OUT/diag_plain.log | cut -c1-150
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_dp.py -q -m gpu -k "generate_replays or hipgraph or best_of_n or rccl or training_reduces" > $OUT/tests.log 2>&1
echo "rc=$?" >> $OUT/tests.log; tail -6 $OUT/tests.log | cut -c1-300
B="--steps 40 --warmup 6 --no-f32 --no-cpu-baseline --inst-steps 0"
for rep in 1 2; do
  python bench.py $B > $OUT/bench_$rep.json 2> $OUT/bench_$rep.err; tail -1 $OUT/bench_$rep.json | cut -c1-260
done
