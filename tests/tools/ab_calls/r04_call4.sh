#!/bin/bash
# Round 4, fourth lease: the late-DMA ring kernel (single variant, no spills) and bf16 conv I/O -- whole GPU suite, then the A/B in the step.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04d
mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/tests_all.log 2>&1
echo "rc=$?" >> $OUT/tests_all.log; tail -5 $OUT/tests_all.log
B="--steps 30 --warmup 6 --no-f32 --no-cpu-baseline --inst-steps 4"
for rep in 1 2; do
  SAVP_BF16_CONVIO=0 python bench.py $B > $OUT/bench_io0_$rep.json 2> $OUT/bench_io0_$rep.err
  SAVP_BF16_CONVIO=1 python bench.py $B > $OUT/bench_io1_$rep.json 2> $OUT/bench_io1_$rep.err
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04d/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], 'ms %.2f' % d['ms_per_step'], 'gateconv us %.2f' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], d['config'].get('submission'))
    except Exception as ex:
        print(f, 'FAILED', ex, open(f.replace('.json', '.err')).read()[-1500:])
P
SAVP_BF16_CONVIO=0 bash tests/tools/prof_step.sh r04d/io0
bash tests/tools/prof_step.sh r04d/io1
python tests/tools/compare_stats.py $OUT/io0_kernel_stats.csv $OUT/io1_kernel_stats.csv 6 > $OUT/compare_io.txt
cat $OUT/compare_io.txt
