#!/bin/bash
# in-step tuning of the C4 (KTH) and C5 (128x128) workloads; step A/B of the resulting tables
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03ab; mkdir -p $O
t0=$(date +%s)
R=$PWD
run() { name=$1; shift
  timeout 600 python bench.py --no-f32 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d['config']
    print(sys.argv[2], 'ms/step %.2f (%s)'%(d['ms_per_step'], c['submission']), 'frames/s %.0f'%d['value'], 'ring us %.2f frac %.4f'%(d['roofline']['avg_launch_us'], d['roofline']['frac']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
for c in c4 c5; do
  CONFIG=$c timeout 600 python tests/tools/insitu_tune.py $R/$O/insitu_$c.json 28 6 2>&1 | grep -v amdgpu.ids | tee $O/insitu_$c.log | grep "changed\|eager step\|table:\|conv launches" | cut -c1-300
  echo "tune $c $(( $(date +%s)-t0 ))s"
  run ${c}_shipped --config $c --steps 20 --warmup 4
  run ${c}_insitu --config $c --steps 20 --warmup 4 --tuning-table $R/$O/insitu_$c.json
  run ${c}_shipped2 --config $c --steps 20 --warmup 4
  run ${c}_insitu2 --config $c --steps 20 --warmup 4 --tuning-table $R/$O/insitu_$c.json
done
echo "total $(( $(date +%s)-t0 ))s"
