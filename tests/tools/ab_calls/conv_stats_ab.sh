#!/bin/bash
# instance-norm statistics from the conv epilogue (fp32 destination): op check, model parity, step A/B, tuned entries
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03y; mkdir -p $O
t0=$(date +%s)
R=$PWD
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "epilogue_statistics or instnorm or conv_views" > $O/ops.log 2>&1; echo "ops rc=$? $(( $(date +%s)-t0 ))s"; tail -5 $O/ops.log | cut -c1-400
run() { name=$1; shift
  env "$@" timeout 500 python bench.py --steps 40 --warmup 4 --no-f32 --no-cpu-baseline $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d['config']
    print(sys.argv[2], 'ms/step %.2f (%s)'%(d['ms_per_step'], c['submission']), 'eager %.2f'%c['eager_ms_per_step'], 'ring us %.2f frac %.4f'%(d['roofline']['avg_launch_us'], d['roofline']['frac']), 'd_loss %.4f g_loss %.3f'%(d['losses']['d_loss'], d['losses']['g_loss']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
EXTRA="--save-tuning $R/$O/tuned.json" run on SAVP_CONV_STATS=1
EXTRA="--tuning-table $R/$O/tuned.json" run on_tuned SAVP_CONV_STATS=1
EXTRA="" run off SAVP_CONV_STATS=0
EXTRA="--tuning-table $R/$O/tuned.json" run on_tuned2 SAVP_CONV_STATS=1
EXTRA="" run off2 SAVP_CONV_STATS=0
echo "bench done $(( $(date +%s)-t0 ))s"
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "generator_forward_vs_oracle or train_step_vs_oracle or golden or hipgraph or b16_t30" > $O/model.log 2>&1; echo "model rc=$? $(( $(date +%s)-t0 ))s"; tail -4 $O/model.log | cut -c1-400
echo "total $(( $(date +%s)-t0 ))s"
