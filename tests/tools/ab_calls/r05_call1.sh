#!/bin/bash
# Round 5, first lease: (i) the new replayed-vs-eager test at the bench shapes (records the distances of the CURRENT kernels: float
# atomics), (ii) parity of the two patches staged by round 4 (ring lane table, generic-conv fragment pre-read) on every shipped tuning-table
# instantiation, (iii) their in-call A/B.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
t0=$(date +%s)
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $O/smoke.json 2> $O/smoke.err || { echo "SMOKE FAILED"; tail -25 $O/smoke.err; exit 1; }
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "replayed_bench_step" > $O/replay_test.log 2>&1; echo "replay test rc=$? $(( $(date +%s)-t0 ))s"; tail -15 $O/replay_test.log | cut -c1-600
SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_both.so timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "table or conv" > $O/both_ops.log 2>&1; echo "both ops rc=$? $(( $(date +%s)-t0 ))s"; tail -4 $O/both_ops.log | cut -c1-400
OUT=$O REPS=2 bash tests/tools/ab_run.sh base "" lanetab "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_lanetab.so" preread "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_preread.so" both "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_both.so"
echo "total $(( $(date +%s)-t0 ))s"
