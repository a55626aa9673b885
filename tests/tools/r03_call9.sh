#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03i; mkdir -p $O
for v in "base" "inorm512 SAVP_INORM_MIN_HW=512" "inorm2048 SAVP_INORM_MIN_HW=2048" "inorm64 SAVP_INORM_MIN_HW=64" "base2"; do
  set -- $v; name=$1; shift
  env "$@" timeout 400 python bench.py --steps 40 --warmup 4 --no-f32 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], 'ms/step %.2f'%d['ms_per_step'], 'eager %.2f'%d['config']['eager_ms_per_step'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
