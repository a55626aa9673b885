#!/bin/bash
# L2 warm-up of the ring kernel's weight block: ops parity, in-step per-launch times with the warm-up off / on, step A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03v; mkdir -p $O
t0=$(date +%s)
R=$PWD
timeout 500 python -m pytest tests/test_gpu_ops.py -q -x > $O/ops.log 2>&1; echo "ops rc=$? $(( $(date +%s)-t0 ))s"; tail -2 $O/ops.log | cut -c1-300
echo "--- in-step per-launch times, warm-up OFF"; SAVP_RING_WWARM=0 timeout 300 python tests/tools/insitu_tune.py $R/$O/t0.json 8 1 2>&1 | grep -v amdgpu.ids | tee $O/insitu_off.log | cut -c1-200
echo "--- in-step per-launch times, warm-up ON"; SAVP_RING_WWARM=1 timeout 300 python tests/tools/insitu_tune.py $R/$O/t1.json 8 1 2>&1 | grep -v amdgpu.ids | tee $O/insitu_on.log | cut -c1-200
for v in "off SAVP_RING_WWARM=0" "on SAVP_RING_WWARM=1" "off2 SAVP_RING_WWARM=0" "on2 SAVP_RING_WWARM=1"; do
  set -- $v; name=$1; shift
  env "$@" timeout 400 python bench.py --steps 40 --warmup 4 --no-f32 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d['config']
    print(sys.argv[2], 'ms/step %.2f (%s)'%(d['ms_per_step'], c['submission']), 'eager %.2f'%c['eager_ms_per_step'], 'ring us %.2f frac %.4f'%(d['roofline']['avg_launch_us'], d['roofline']['frac']), 'cell frac %.4f'%d['roofline_cell']['mfma']['frac'], 'd_loss %.4f g_loss %.3f'%(d['losses']['d_loss'], d['losses']['g_loss']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
echo "total $(( $(date +%s)-t0 ))s"
