#!/bin/bash
# Round 4, fifth lease: whole GPU suite (no -x), DMA position A/B (early | mid | late builds), select fusion A/B.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04e
mkdir -p $OUT
timeout 1800 python -m pytest tests -q -m gpu > $OUT/tests_all.log 2>&1
echo "rc=$?" >> $OUT/tests_all.log; tail -8 $OUT/tests_all.log
B="--steps 30 --warmup 6 --no-f32 --no-cpu-baseline --inst-steps 4"
for rep in 1 2; do
  SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_early.so python bench.py $B > $OUT/bench_early_$rep.json 2> $OUT/bench_early_$rep.err
  SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_mid.so python bench.py $B > $OUT/bench_mid_$rep.json 2> $OUT/bench_mid_$rep.err
  python bench.py $B > $OUT/bench_late_$rep.json 2> $OUT/bench_late_$rep.err
  SAVP_FUSE_SELECT=0 python bench.py $B > $OUT/bench_nofuse_$rep.json 2> $OUT/bench_nofuse_$rep.err
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04e/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], 'ms %.2f' % d['ms_per_step'], 'gateconv us %.2f' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], d['config'].get('submission'))
    except Exception as ex:
        print(f, 'FAILED', ex, open(f.replace('.json', '.err')).read()[-1500:])
P
