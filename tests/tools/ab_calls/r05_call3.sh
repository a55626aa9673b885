#!/bin/bash
# Round 5, third lease: row-wise LDS-DMA patch staging in the ring kernel -- parity (every shipped tuning-table instantiation, the bench-shape
# goldens now under ONE gradient gate, the replayed-vs-eager tests with their measured gates), in-call A/B against the build before it, stamps.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05c; mkdir -p $O
t0=$(date +%s)
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $O/smoke.json 2> $O/smoke.err || { echo "SMOKE FAILED"; tail -25 $O/smoke.err; exit 1; }
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu > $O/ops.log 2>&1; echo "ops rc=$? $(( $(date +%s)-t0 ))s"; tail -5 $O/ops.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "replay or golden or hipgraph or recipe_shapes" > $O/model.log 2>&1; echo "model rc=$? $(( $(date +%s)-t0 ))s"; grep -E "^E  .*(Assert|assert)|passed|failed" $O/model.log | cut -c1-600 | head -20
for c in c2 c4 c5; do cat gpurun_out/r05_replay_vs_eager_$c.json | tr -d '\n'; echo; done
OUT=$O REPS=2 bash tests/tools/ab_run.sh det "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_det.so" rowdma ""
OUT=$O/c5 REPS=1 CONFIG=c5 BENCH_ARGS="--steps 20 --warmup 4 --no-f32 --no-cpu-baseline --inst-steps 4" bash tests/tools/ab_run.sh det "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_det.so" rowdma ""
for spec in lstm_h0:fprop:712:cell16 lstm_h1:fprop:711:cell16 lstm_h2:fprop:311:cell16 lstm_h0:dgrad:711:src16; do
  SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_stamps.so python tests/tools/ring_times.py $spec 2>&1 | grep -v amdgpu.ids
done | tee $O/ring_stamps.log
echo "total $(( $(date +%s)-t0 ))s"
