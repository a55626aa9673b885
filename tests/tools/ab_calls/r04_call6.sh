#!/bin/bash
# Round 4, sixth lease: new GPU checks (norm-backward sums from the DGRAD epilogue), then the tuning table for the round's new problem
# keys: isolated tuning of everything the three workloads launch + in-step tuning of the most expensive problems (tests/tools/insitu_tune.py).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04f
mkdir -p $OUT
# guard: two steps of the default workload must run before anything expensive is started
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $OUT/smoke.json 2> $OUT/smoke.err || { echo "SMOKE FAILED"; tail -25 $OUT/smoke.err; exit 1; }
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu > $OUT/tests_ops.log 2>&1
echo "rc=$?" >> $OUT/tests_ops.log; tail -4 $OUT/tests_ops.log
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "recipe or b16 or smoke or small" > $OUT/tests_model.log 2>&1
echo "rc=$?" >> $OUT/tests_model.log; tail -4 $OUT/tests_model.log
B="--steps 30 --warmup 6 --no-f32 --no-cpu-baseline --inst-steps 4"
for rep in 1 2; do
  SAVP_NORM_BWD_STATS=0 python bench.py $B > $OUT/bench_nb0_$rep.json 2> $OUT/bench_nb0_$rep.err
  SAVP_FUSE_SELECT=0 python bench.py $B > $OUT/bench_nofuse_$rep.json 2> $OUT/bench_nofuse_$rep.err
  SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_early.so python bench.py $B > $OUT/bench_early_$rep.json 2> $OUT/bench_early_$rep.err
  SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_mid.so python bench.py $B > $OUT/bench_mid_$rep.json 2> $OUT/bench_mid_$rep.err
  python bench.py $B > $OUT/bench_default_$rep.json 2> $OUT/bench_default_$rep.err
done
T=video_prediction_amd/tuning_gfx950_bf16.json
cp $T $OUT/table_before.json
python tests/tools/insitu_tune.py $OUT/table_c2.json 36 5 > $OUT/insitu_c2.log 2>&1; tail -3 $OUT/insitu_c2.log
[ -s $OUT/table_c2.json ] && cp $OUT/table_c2.json $T
CONFIG=c4 python tests/tools/insitu_tune.py $OUT/table_c4.json 16 4 > $OUT/insitu_c4.log 2>&1; tail -3 $OUT/insitu_c4.log
[ -s $OUT/table_c4.json ] && cp $OUT/table_c4.json $T
CONFIG=c5 python tests/tools/insitu_tune.py $OUT/table_c5.json 16 4 > $OUT/insitu_c5.log 2>&1; tail -3 $OUT/insitu_c5.log
[ -s $OUT/table_c5.json ] && cp $OUT/table_c5.json $T
cp $T $OUT/tuning_gfx950_bf16.json
for rep in 1; do
  python bench.py $B > $OUT/bench_tuned_$rep.json 2> $OUT/bench_tuned_$rep.err
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04f/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], 'ms %.2f' % d['ms_per_step'], 'gateconv us %.2f' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], d['config'].get('submission'), d['roofline_cell'].get('kernel_only'))
    except Exception as ex:
        print(f, 'FAILED', ex, open(f.replace('.json', '.err')).read()[-1500:])
P
bash tests/tools/prof_step.sh r04f/tuned
