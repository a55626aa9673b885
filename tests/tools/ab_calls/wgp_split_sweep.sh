#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03ae; mkdir -p $O
t0=$(date +%s)
run() { name=$1; shift
  env "$@" timeout 500 python bench.py --steps 40 --warmup 4 --no-f32 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
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
run split128 SAVP_WGP_SPLIT=128
run split192 SAVP_WGP_SPLIT=192
run split256 SAVP_WGP_SPLIT=256
run split512 SAVP_WGP_SPLIT=512
SAVP_WGP_SPLIT=256 bash tests/tools/prof_step.sh r03ae/split256 2>&1 | tail -1
SAVP_WGP_SPLIT=128 bash tests/tools/prof_step.sh r03ae/split128 2>&1 | tail -1
echo "total $(( $(date +%s)-t0 ))s"
