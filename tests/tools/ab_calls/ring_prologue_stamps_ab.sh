#!/bin/bash
# kernarg warm-up + launch constants: cycle stamps (per wave) with the warm-up on / off, step-time A/B of three builds
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03o; mkdir -p $O
t0=$(date +%s)
SPECS="lstm_h0:fprop:712:cell16 lstm_h1:fprop:711:cell16 lstm_h2:fprop:311:cell16 lstm_h0:dgrad:712:src16 lstm_h0:dgrad:322:src16 lstm_h1:dgrad:312:src16 head3x3:fprop:712"
P=$PWD/video_prediction_amd
echo "--- stamps kwarm=0"; KWARM=0 SAVP_LIB=$P/libsavp_hip_ringdev.so timeout 200 python tests/tools/ring_times.py $SPECS 2>&1 | grep -v amdgpu.ids | tee $O/stamps_kw0.log
echo "--- stamps kwarm=1"; KWARM=1 SAVP_LIB=$P/libsavp_hip_ringdev.so timeout 200 python tests/tools/ring_times.py $SPECS 2>&1 | grep -v amdgpu.ids | tee $O/stamps_kw1.log
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "conv or tuning or cell or ring or patch" > $O/ops.log 2>&1; echo "ops rc=$? $(( $(date +%s)-t0 ))s"; tail -2 $O/ops.log | cut -c1-300
for v in "base SAVP_LIB=$P/libsavp_hip_base.so" "diet SAVP_LIB=$P/libsavp_hip_diet.so" "new" "base2 SAVP_LIB=$P/libsavp_hip_base.so" "diet2 SAVP_LIB=$P/libsavp_hip_diet.so" "new2"; do
  set -- $v; name=$1; shift
  env "$@" timeout 400 python bench.py --steps 40 --warmup 4 --no-f32 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d['config']
    print(sys.argv[2], 'ms/step %.2f (%s)'%(d['ms_per_step'], c['submission']), 'eager %.2f host-issue %.2f'%(c['eager_ms_per_step'], c.get('host_issue_ms_per_step') or -1), 'ring us %.2f frac %.4f'%(d['roofline']['avg_launch_us'], d['roofline']['frac']), 'd_loss %.4f g_loss %.3f'%(d['losses']['d_loss'], d['losses']['g_loss']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
echo "total $(( $(date +%s)-t0 ))s"
