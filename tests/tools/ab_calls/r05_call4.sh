#!/bin/bash
# Round 5, fourth lease: per-kernel view (rocprofv3 kernel stats of 6 eager steps) of the ring kernel's patch-staging variants -- slot-linear
# (before), row-wise (default now), row-wise + first group requested before the rest of the prologue -- and the repaired golden / replay tests.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05d; mkdir -p $O
t0=$(date +%s)
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $O/smoke.json 2> $O/smoke.err || { echo "SMOKE FAILED"; tail -25 $O/smoke.err; exit 1; }
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "replayed_bench or golden" > $O/model.log 2>&1; echo "model rc=$? $(( $(date +%s)-t0 ))s"; grep -E "^E  .*(Assert|assert)|passed|failed" $O/model.log | cut -c1-700 | head -20
for v in det rowdma earlypatch; do
  lib=$PWD/video_prediction_amd/ab/libsavp_hip_$v.so; [ $v = rowdma ] && lib=$PWD/video_prediction_amd/libsavp_hip.so
  SAVP_LIB=$lib bash tests/tools/prof_step.sh r05d/$v 2>&1 | tail -2
done
python tests/tools/compare_stats.py $O/det_kernel_stats.csv $O/rowdma_kernel_stats.csv 6 | tee $O/cmp_det_rowdma.txt
python tests/tools/compare_stats.py $O/rowdma_kernel_stats.csv $O/earlypatch_kernel_stats.csv 6 | tee $O/cmp_rowdma_early.txt
OUT=$O REPS=2 bash tests/tools/ab_run.sh det "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_det.so" rowdma "" early "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_earlypatch.so"
echo "total $(( $(date +%s)-t0 ))s"
