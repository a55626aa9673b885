#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03j; mkdir -p $O
for v in "base" "q1 SAVP_LSTM_Q=1" "q2 SAVP_LSTM_Q=2" "q4 SAVP_LSTM_Q=4" "base2"; do
  set -- $v; name=$1; shift
  env "$@" timeout 400 python bench.py --steps 40 --warmup 4 --no-f32 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], 'ms/step %.2f'%d['ms_per_step'], 'eager %.2f'%d['config']['eager_ms_per_step'], 'cell us %.1f'%d['roofline_cell']['avg_cell_us'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "instnorm or lstm or cell" 2>&1 | tail -2
