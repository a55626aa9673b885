#!/bin/bash
# Round 5, ninth lease: eight-deep weight ring for the small-tile ring instantiations (option ring_deep) -- conv parity (every shipped table
# instantiation runs with it), stamps of the 16x16 gate convolutions with / without, in-call A/B of the step, per-kernel view.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05j; mkdir -p $O
t0=$(date +%s)
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $O/smoke.json 2> $O/smoke.err || { echo "SMOKE FAILED"; tail -25 $O/smoke.err; exit 1; }
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv or table or cell" > $O/ops.log 2>&1; echo "ops rc=$? $(( $(date +%s)-t0 ))s"; tail -3 $O/ops.log | cut -c1-300
for d in 0 1; do for spec in lstm_h1:fprop:711:cell16 lstm_h1:dgrad:711:src16 lstm_h0:dgrad:711:src16; do
  SAVP_RING_DEEP=$d SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_stamps.so python tests/tools/ring_times.py $spec 2>&1 | grep -v "amdgpu.ids\|per wave" | sed "s/^/deep$d /"
done; done | tee $O/ring_deep_stamps.log
for v in deep0 deep1; do
  SAVP_RING_DEEP=${v#deep} bash tests/tools/prof_step.sh r05j/$v 2>&1 | tail -1
done
python tests/tools/compare_stats.py $O/deep0_kernel_stats.csv $O/deep1_kernel_stats.csv 6 | tee $O/cmp_deep.txt
OUT=$O REPS=2 bash tests/tools/ab_run.sh deep0 "SAVP_RING_DEEP=0" deep1 "SAVP_RING_DEEP=1"
echo "total $(( $(date +%s)-t0 ))s"
