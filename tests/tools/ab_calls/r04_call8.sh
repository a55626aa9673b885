#!/bin/bash
# Round 4, eighth lease: split-K for the gapped / norm-sum DGRADs, clean A/B of the first-slab overlap (two builds of the same source),
# in-step re-tune of the data-gradient problems (their split-K choice is new).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04h
mkdir -p $OUT
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $OUT/smoke.json 2> $OUT/smoke.err || { echo "SMOKE FAILED"; tail -25 $OUT/smoke.err; exit 1; }
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "tiled_z or norm_backward or cell or warmup" > $OUT/tests_ops.log 2>&1
echo "rc=$?" >> $OUT/tests_ops.log; tail -4 $OUT/tests_ops.log
T=video_prediction_amd/tuning_gfx950_bf16.json
MODES=1 python tests/tools/insitu_tune.py $OUT/table_c2.json 12 6 > $OUT/insitu_c2.log 2>&1; tail -3 $OUT/insitu_c2.log
[ -s $OUT/table_c2.json ] && cp $OUT/table_c2.json $T
MODES=1 CONFIG=c4 python tests/tools/insitu_tune.py $OUT/table_c4.json 8 6 > $OUT/insitu_c4.log 2>&1; tail -3 $OUT/insitu_c4.log
[ -s $OUT/table_c4.json ] && cp $OUT/table_c4.json $T
MODES=1 CONFIG=c5 python tests/tools/insitu_tune.py $OUT/table_c5.json 8 6 > $OUT/insitu_c5.log 2>&1; tail -3 $OUT/insitu_c5.log
[ -s $OUT/table_c5.json ] && cp $OUT/table_c5.json $T
cp $T $OUT/tuning_gfx950_bf16.json
B="--steps 30 --warmup 6 --no-f32 --no-cpu-baseline --inst-steps 4"
for rep in 1 2; do
  SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_nooverlap.so python bench.py $B > $OUT/bench_nooverlap_$rep.json 2> $OUT/bench_nooverlap_$rep.err
  python bench.py $B > $OUT/bench_default_$rep.json 2> $OUT/bench_default_$rep.err
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04h/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], 'ms %.2f' % d['ms_per_step'], 'gateconv us %.2f' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], 'cell kernel-only', d['roofline_cell']['kernel_only']['avg_cell_us'])
    except Exception as ex:
        print(f, 'FAILED', ex, open(f.replace('.json', '.err')).read()[-1500:])
P
SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_nooverlap.so bash tests/tools/prof_step.sh r04h/nooverlap
