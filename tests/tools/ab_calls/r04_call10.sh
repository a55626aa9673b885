#!/bin/bash
# Round 4, tenth lease: wide weight slabs (8 / 9 k-steps per ring entry, tile bit 0x1000): parity, then in-step tuning with the new candidates.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04j
mkdir -p $OUT
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $OUT/smoke.json 2> $OUT/smoke.err || { echo "SMOKE FAILED"; tail -25 $OUT/smoke.err; exit 1; }
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "tiled_z or norm_backward or cell" > $OUT/tests_ops.log 2>&1
echo "rc=$?" >> $OUT/tests_ops.log; tail -4 $OUT/tests_ops.log
T=video_prediction_amd/tuning_gfx950_bf16.json
B="--steps 30 --warmup 6 --no-f32 --no-cpu-baseline --inst-steps 4"
python bench.py $B > $OUT/bench_before_1.json 2> $OUT/bench_before_1.err
python tests/tools/insitu_tune.py $OUT/table_c2.json 14 8 > $OUT/insitu_c2.log 2>&1; tail -3 $OUT/insitu_c2.log
[ -s $OUT/table_c2.json ] && cp $OUT/table_c2.json $T
cp $T $OUT/tuning_gfx950_bf16.json
for rep in 1 2; do
  python bench.py $B > $OUT/bench_after_$rep.json 2> $OUT/bench_after_$rep.err
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04j/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], 'ms %.2f' % d['ms_per_step'], 'fps %.0f' % d['value'], 'gateconv us %.2f' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], 'cell kernel-only', d['roofline_cell']['kernel_only']['avg_cell_us'])
    except Exception as ex:
        print(f, 'FAILED', ex, open(f.replace('.json', '.err')).read()[-1500:])
P
grep "0x1[37]11" $OUT/insitu_c2.log | cut -c1-330
