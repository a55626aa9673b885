#!/bin/bash
# Round 4, second lease: re-run of the first lease's two failed gates (tolerances), the ring kernel's wave roles (bit identity + A/B),
# and a per-kernel A/B of the z-less DGRAD / roles against the same build with both off (rocprofv3 kernel stats, eager).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04b
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_dp.py -q -m gpu -k "rccl or bucket or torchrun" > $OUT/tests_rccl.log 2>&1
echo "rc=$?" >> $OUT/tests_rccl.log; tail -4 $OUT/tests_rccl.log
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "c4_c5" > $OUT/tests_golden.log 2>&1
echo "rc=$?" >> $OUT/tests_golden.log; tail -4 $OUT/tests_golden.log
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu > $OUT/tests_ops.log 2>&1
echo "rc=$?" >> $OUT/tests_ops.log; tail -4 $OUT/tests_ops.log
B="--steps 30 --warmup 6 --no-f32 --no-cpu-baseline --inst-steps 4"
for rep in 1 2; do
  SAVP_RING_ROLES=0 python bench.py $B > $OUT/bench_roles0_$rep.json 2> $OUT/bench_roles0_$rep.err
  SAVP_RING_ROLES=1 python bench.py $B > $OUT/bench_roles1_$rep.json 2> $OUT/bench_roles1_$rep.err
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04b/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], 'ms %.2f' % d['ms_per_step'], 'gateconv us %.2f' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], d['config'].get('submission'))
    except Exception as ex:
        print(f, 'FAILED', ex)
P
SAVP_RING_ROLES=0 SAVP_ZLESS_DGRAD=0 bash tests/tools/prof_step.sh r04b/off
bash tests/tools/prof_step.sh r04b/on
python tests/tools/compare_stats.py $OUT/off_kernel_stats.csv $OUT/on_kernel_stats.csv 6 > $OUT/compare_off_on.txt
cat $OUT/compare_off_on.txt
