#!/bin/bash
# Round 4, third lease: cycle attribution of conv_ring_kernel (developer build), the tiled-z kernels after the load unrolling, the
# c4 / c5 goldens again, and roles 0 / 1 / 2 in the step.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04c
mkdir -p $OUT
SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_ringdev.so python tests/tools/ring_attrib.py lstm_h0:fprop:712:cell16 lstm_h1:fprop:711:cell16 \
   lstm_h2:fprop:311:cell16 lstm_h0:dgrad:712:src16 lstm_h0:dgrad:711:src16:gap lstm_h1:dgrad:711:src16 lstm_h1:dgrad:711:src16:gap lstm_h0:fprop:721:cell16 > $OUT/ring_attrib.log 2>&1
cat $OUT/ring_attrib.log
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "c4_c5" > $OUT/tests_golden.log 2>&1
echo "rc=$?" >> $OUT/tests_golden.log; tail -3 $OUT/tests_golden.log
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "tiled_z or roles or cell or conv_fprop" > $OUT/tests_ops.log 2>&1
echo "rc=$?" >> $OUT/tests_ops.log; tail -3 $OUT/tests_ops.log
B="--steps 30 --warmup 6 --no-f32 --no-cpu-baseline --inst-steps 4"
for rep in 1 2; do
  for r in 0 1 2; do
    SAVP_RING_ROLES=$r python bench.py $B > $OUT/bench_roles${r}_$rep.json 2> $OUT/bench_roles${r}_$rep.err
  done
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04c/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], 'ms %.2f' % d['ms_per_step'], 'gateconv us %.2f' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], d['config'].get('submission'))
    except Exception as ex:
        print(f, 'FAILED', ex)
P
bash tests/tools/prof_step.sh r04c/on
python tests/tools/compare_stats.py gpurun_out/r04b/off_kernel_stats.csv $OUT/on_kernel_stats.csv 6 > $OUT/compare.txt 2>/dev/null
grep -i "tiled_z" $OUT/on_kernel_stats.csv
