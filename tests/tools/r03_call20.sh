#!/bin/bash
# re-tune the bf16-activation conv problems on the LDS-DMA staging build: shipped table vs live-tuned entries, same call
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03s; mkdir -p $O
t0=$(date +%s)
R=$PWD
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
run livetune --tuning-table $R/tests/tools/tuning_bf16_without_bf16act.json --save-tuning $R/$O/retuned.json
run retuned --tuning-table $R/$O/retuned.json
run shipped2
run retuned2 --tuning-table $R/$O/retuned.json
BENCH_ARGS="--tuning-table $R/$O/retuned.json" bash tests/tools/prof_step.sh r03s/retuned 2>&1 | tail -1
bash tests/tools/prof_step.sh r03s/shipped 2>&1 | tail -1
echo "total $(( $(date +%s)-t0 ))s"
