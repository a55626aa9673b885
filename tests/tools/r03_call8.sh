#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03h; mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_model.py -q -k "patch_kernels_match or best_of_n or checkpoint or scripts or plugin or golden_vectors or full_size or loss_curve or b16_t30" > $O/rest.log 2>&1; echo "rest rc=$? $(( $(date +%s)-t0 ))s"; tail -6 $O/rest.log | cut -c1-600
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "not tuning_table" > $O/ops.log 2>&1; echo "ops rc=$? $(( $(date +%s)-t0 ))s"; tail -3 $O/ops.log
for v in "base" "min128 SAVP_FUSED_MIN_HW=128" "min512 SAVP_FUSED_MIN_HW=512" "min2048 SAVP_FUSED_MIN_HW=2048" "base2"; do
  set -- $v; name=$1; shift
  env "$@" timeout 400 python bench.py --steps 40 --warmup 4 --no-f32 --no-cpu-baseline --save-tuning $O/tuning_$name.json > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], 'ms/step %.2f'%d['ms_per_step'], 'gate conv us %.1f (frac %.3f)'%(d['roofline']['avg_launch_us'], d['roofline']['frac']), 'cell us %.1f'%d['roofline_cell']['avg_cell_us'], 'd_loss %.4f g_loss %.3f'%(d['losses']['d_loss'], d['losses']['g_loss']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
echo "total $(( $(date +%s)-t0 ))s"
