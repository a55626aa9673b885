#!/bin/bash
# the 38 k-cycle stall of the 8-wave DGRAD: which workgroups, which shapes, which parts; + step A/B (ring-only diet + kernarg warm)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03q; mkdir -p $O
t0=$(date +%s)
P=$PWD/video_prediction_amd
export SAVP_LIB=$P/libsavp_hip_ringdev.so KWARM=1
D="lstm_h0:dgrad:712:src16"
echo "--- block 0 / 100 / 255"; for b in 0 100 255; do RING_BLOCK=$b timeout 100 python tests/tools/ring_times.py $D 2>&1 | grep -v amdgpu.ids; done | tee $O/blocks.log
echo "--- other shapes, tile 712"; timeout 100 python tests/tools/ring_times.py lstm_h1:dgrad:712:src16 lstm_h2:dgrad:712:src16 head3x3:dgrad:712 lstm_h0:dgrad:312:src16 lstm_h0:dgrad:711:src16 2>&1 | grep -v amdgpu.ids | tee $O/shapes.log
echo "--- N=8 / N=16"; for n in 8 16; do RING_N=$n timeout 100 python tests/tools/ring_times.py $D 2>&1 | grep -v amdgpu.ids; done | tee $O/n.log
echo "--- ablate 4 (no staging), 32 (no main loop), 1 (no weight DMA)"; for a in 4 32 1; do SAVP_ABLATE=$a timeout 100 python tests/tools/ring_times.py $D 2>&1 | grep -v amdgpu.ids; done | tee $O/abl.log
unset SAVP_LIB KWARM
for v in "base SAVP_LIB=$P/libsavp_hip_base.so" "new" "base2 SAVP_LIB=$P/libsavp_hip_base.so" "new2"; do
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
