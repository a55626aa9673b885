#!/bin/bash
# Round 4, seventh lease: early-DMA default + first-slab overlap (A/B against a build without it), chunk-parallel tiled-z, the new table.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04g
mkdir -p $OUT
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $OUT/smoke.json 2> $OUT/smoke.err || { echo "SMOKE FAILED"; tail -25 $OUT/smoke.err; exit 1; }
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu > $OUT/tests_ops.log 2>&1
echo "rc=$?" >> $OUT/tests_ops.log; tail -4 $OUT/tests_ops.log
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "recipe or b16" > $OUT/tests_model.log 2>&1
echo "rc=$?" >> $OUT/tests_model.log; tail -4 $OUT/tests_model.log
B="--steps 30 --warmup 6 --no-f32 --no-cpu-baseline --inst-steps 4"
for rep in 1 2; do
  SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_nooverlap.so python bench.py $B > $OUT/bench_nooverlap_$rep.json 2> $OUT/bench_nooverlap_$rep.err
  python bench.py $B > $OUT/bench_default_$rep.json 2> $OUT/bench_default_$rep.err
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04g/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], 'ms %.2f' % d['ms_per_step'], 'gateconv us %.2f' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], 'cell kernel-only', d['roofline_cell']['kernel_only']['avg_cell_us'])
    except Exception as ex:
        print(f, 'FAILED', ex, open(f.replace('.json', '.err')).read()[-1500:])
P
bash tests/tools/prof_step.sh r04g/default
