#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03l; mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "not f32" --durations=5 > $O/ops.log 2>&1; echo "ops rc=$? $(( $(date +%s)-t0 ))s"; tail -9 $O/ops.log
for v in "dma1" "dma0 SAVP_RING_DMA=0" "dma1b" "dma0b SAVP_RING_DMA=0"; do
  set -- $v; name=$1; shift
  env "$@" timeout 400 python bench.py --steps 40 --warmup 4 --no-f32 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], 'ms/step %.2f'%d['ms_per_step'], 'eager %.2f'%d['config']['eager_ms_per_step'], 'gate conv %.1f us frac %.4f'%(d['roofline']['avg_launch_us'], d['roofline']['frac']), 'd_loss %.4f g_loss %.3f'%(d['losses']['d_loss'], d['losses']['g_loss']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
echo "bench done $(( $(date +%s)-t0 ))s"
timeout 900 python -m pytest tests/test_gpu_model.py -q -k "test_train_step_vs_oracle or b16_t30 or bf16_mode" --durations=3 > $O/model.log 2>&1; echo "model rc=$? $(( $(date +%s)-t0 ))s"; tail -8 $O/model.log
echo "total $(( $(date +%s)-t0 ))s"
