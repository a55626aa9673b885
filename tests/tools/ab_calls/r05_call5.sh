#!/bin/bash
# Round 5, fifth lease: row-wise patch staging, second version (incremental walk, no per-j array) -- conv parity, per-kernel view and step A/B
# against the slot-linear build.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
t0=$(date +%s)
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $O/smoke.json 2> $O/smoke.err || { echo "SMOKE FAILED"; tail -25 $O/smoke.err; exit 1; }
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv or table or cell" > $O/ops.log 2>&1; echo "ops rc=$? $(( $(date +%s)-t0 ))s"; tail -3 $O/ops.log | cut -c1-300
for v in det rowdma3; do
  lib=$PWD/video_prediction_amd/ab/libsavp_hip_$v.so; [ $v = rowdma3 ] && lib=$PWD/video_prediction_amd/libsavp_hip.so
  SAVP_LIB=$lib bash tests/tools/prof_step.sh r05f/$v 2>&1 | tail -1
done
python tests/tools/compare_stats.py $O/det_kernel_stats.csv $O/rowdma3_kernel_stats.csv 6 | tee $O/cmp_det_rowdma3.txt
OUT=$O REPS=2 bash tests/tools/ab_run.sh det "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_det.so" rowdma3 ""
echo "total $(( $(date +%s)-t0 ))s"
