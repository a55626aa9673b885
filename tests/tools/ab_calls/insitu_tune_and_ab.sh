#!/bin/bash
# in-step tuning of the top conv problems, then step A/B of the resulting table against the shipped one
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03u; mkdir -p $O
t0=$(date +%s)
R=$PWD
timeout 600 python tests/tools/insitu_tune.py $R/$O/insitu.json 28 6 2>&1 | grep -v amdgpu.ids | tee $O/insitu.log | cut -c1-400
echo "tune $(( $(date +%s)-t0 ))s"
run() { name=$1; shift
  timeout 500 python bench.py --steps 40 --warmup 4 --no-f32 --no-cpu-baseline "$@" > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d['config']
    print(sys.argv[2], 'ms/step %.2f (%s)'%(d['ms_per_step'], c['submission']), 'eager %.2f'%c['eager_ms_per_step'], 'ring us %.2f frac %.4f'%(d['roofline']['avg_launch_us'], d['roofline']['frac']), 'cell frac %.4f'%d['roofline_cell']['mfma']['frac'], 'd_loss %.4f g_loss %.3f'%(d['losses']['d_loss'], d['losses']['g_loss']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run shipped
run insitu --tuning-table $R/$O/insitu.json
run shipped2
run insitu2 --tuning-table $R/$O/insitu.json
echo "total $(( $(date +%s)-t0 ))s"
