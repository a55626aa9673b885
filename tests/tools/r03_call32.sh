#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03ad; mkdir -p $O
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
run default A=1
run split256 SAVP_WGP_SPLIT=256
run split384 SAVP_WGP_SPLIT=384
run split512 SAVP_WGP_SPLIT=512
run split640 SAVP_WGP_SPLIT=640
run default2 A=1
run split384b SAVP_WGP_SPLIT=384
run split512b SAVP_WGP_SPLIT=512
SAVP_WGP_SPLIT=512 bash tests/tools/prof_step.sh r03ad/split512 2>&1 | tail -1
bash tests/tools/prof_step.sh r03ad/default 2>&1 | tail -1
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "warmup" 2>&1 | tail -2
echo "total $(( $(date +%s)-t0 ))s"
