#!/bin/bash
# Round 5, second lease: the deterministic-reduction build -- whole GPU suite (incl. the replayed-vs-eager test at the bench shapes), the
# round-4 library against it in one call, and cycle stamps of the shipped gate-convolution instantiations (stamps-only developer build).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05b; mkdir -p $O
t0=$(date +%s)
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $O/smoke.json 2> $O/smoke.err || { echo "SMOKE FAILED"; tail -25 $O/smoke.err; exit 1; }
timeout 1200 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_model.py::test_the_replayed_bench_step_is_the_eager_step_and_matches_the_golden > $O/gputest.log 2>&1; echo "gputest rc=$? $(( $(date +%s)-t0 ))s"; tail -12 $O/gputest.log | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "replayed_bench_step" > $O/replay_test.log 2>&1; echo "replay test rc=$? $(( $(date +%s)-t0 ))s"; grep -E "^E  |passed|failed" $O/replay_test.log | cut -c1-300 | head -20
for c in c2 c4 c5; do cat gpurun_out/r05_replay_vs_eager_$c.json | tr -d '\n'; echo; done
OUT=$O REPS=2 bash tests/tools/ab_run.sh r04 "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_r04.so" det ""
for spec in lstm_h0:fprop:712:cell16 lstm_h1:fprop:711:cell16 lstm_h2:fprop:311:cell16 lstm_h0:dgrad:711:src16 lstm_h2:dgrad:311:src16; do
  for blk in 0 100 200; do
    RING_BLOCK=$blk SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_stamps.so python tests/tools/ring_times.py $spec 2>&1 | grep -v amdgpu.ids | sed "s/^/blk$blk /"
  done
done | tee $O/ring_stamps.log
echo "total $(( $(date +%s)-t0 ))s"
