#!/bin/bash
# round 3, third lease: slab-major gate path (parity, step A/B), profile
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03c; mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "lstm or cell or bf16_activation or weight_prep" > $O/ops.log 2>&1; echo "ops rc=$? $(( $(date +%s)-t0 ))s"; tail -15 $O/ops.log
timeout 600 python -m pytest tests/test_gpu_model.py -q -k "bf16_mode or hipgraph or reduces_l1 or full_size or patch_kernels_match or checkpoint" > $O/model_quick.log 2>&1; echo "model quick rc=$? $(( $(date +%s)-t0 ))s"; tail -6 $O/model_quick.log
for v in "base" "noslab SAVP_LSTM_SLAB=0" "slab_act SAVP_BF16_ACT=1 SAVP_BF16_DGATES=1" "base2"; do
  set -- $v; name=$1; shift
  env "$@" timeout 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --save-tuning $O/tuning_$name.json > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], 'ms/step %.2f'%d['ms_per_step'], 'ring us %.1f'%d['roofline']['avg_launch_us'], 'cell us %.1f'%d['roofline_cell']['avg_cell_us'], 'd_loss %.4f g_loss %.3f'%(d['losses']['d_loss'], d['losses']['g_loss']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
echo "bench done $(( $(date +%s)-t0 ))s"
bash tests/tools/prof_step.sh r03c/slab > $O/prof.log 2>&1; tail -2 $O/prof.log
echo "total $(( $(date +%s)-t0 ))s"
