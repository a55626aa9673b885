#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03af; mkdir -p $O
t0=$(date +%s)
run() { name=$1; shift
  env "$@" timeout 500 python bench.py --steps 40 --warmup 4 --no-f32 --no-cpu-baseline $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d['config']
    print(sys.argv[2], 'ms/step %.2f (%s)'%(d['ms_per_step'], c['submission']), 'eager %.2f'%c['eager_ms_per_step'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
EXTRA=""
run new A=1
run old768 SAVP_WGP_SPLIT=768
run new2 A=1
run old768b SAVP_WGP_SPLIT=768
EXTRA="--config c4" run c4_new A=1
EXTRA="--config c4" run c4_old SAVP_WGP_SPLIT=768
EXTRA="--config c5" run c5_new A=1
EXTRA="--config c5" run c5_old SAVP_WGP_SPLIT=768
timeout 500 python -m pytest tests/test_gpu_ops.py -q -x > $O/ops.log 2>&1; echo "ops rc=$? $(( $(date +%s)-t0 ))s"; tail -2 $O/ops.log | cut -c1-300
echo "total $(( $(date +%s)-t0 ))s"
