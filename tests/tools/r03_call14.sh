#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03m; mkdir -p $O
t0=$(date +%s)
SAVP_CDNA_SIDE=1 timeout 600 python -m pytest tests/test_gpu_model.py -q -x -k "generator_forward_vs_oracle or hipgraph or b16_t30 or test_train_step_vs_oracle or golden_vectors" > $O/model_side.log 2>&1; echo "model(side) rc=$? $(( $(date +%s)-t0 ))s"; tail -4 $O/model_side.log | cut -c1-400
for v in "side0" "side1 SAVP_CDNA_SIDE=1" "side0b" "side1b SAVP_CDNA_SIDE=1"; do
  set -- $v; name=$1; shift
  env "$@" timeout 400 python bench.py --steps 40 --warmup 4 --no-f32 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], 'ms/step %.2f (%s)'%(d['ms_per_step'], d['config']['submission']), 'eager %.2f'%d['config']['eager_ms_per_step'], 'd_loss %.4f g_loss %.3f'%(d['losses']['d_loss'], d['losses']['g_loss']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
echo "total $(( $(date +%s)-t0 ))s"
