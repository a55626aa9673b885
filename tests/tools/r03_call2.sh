#!/bin/bash
# round 3, second lease: one-launch ConvLSTM gate block (parity + microbench + step A/B), bf16 activation switches after the fix,
# option table / caller-owned workspaces (every op test), dp tests.
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03b; mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "not tuning_table" > $O/ops.log 2>&1; echo "ops rc=$? $(( $(date +%s)-t0 ))s"; tail -5 $O/ops.log
timeout 300 python tests/tools/bench_gate_block.py > $O/gate_block.log 2>&1; echo "gate bench rc=$? $(( $(date +%s)-t0 ))s"; cat $O/gate_block.log | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_model.py -q -k "bf16_mode or hipgraph or reduces_l1 or full_size or patch_kernels_match" > $O/model_quick.log 2>&1; echo "model quick rc=$? $(( $(date +%s)-t0 ))s"; tail -4 $O/model_quick.log
for v in "base" "nofuse SAVP_LSTM_FUSED=0" "act SAVP_BF16_ACT=1" "dg SAVP_BF16_DGATES=1" "actdg SAVP_BF16_ACT=1 SAVP_BF16_DGATES=1" "base2"; do
  set -- $v; name=$1; shift
  env "$@" timeout 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
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
timeout 600 python -m pytest tests/test_gpu_dp.py -q > $O/dp.log 2>&1; echo "dp rc=$? $(( $(date +%s)-t0 ))s"; tail -3 $O/dp.log
bash tests/tools/prof_step.sh r03b/fused > $O/prof.log 2>&1; tail -2 $O/prof.log
echo "total $(( $(date +%s)-t0 ))s"
