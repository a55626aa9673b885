#!/bin/bash
# Round 4, lease 28: the measured distances behind the gates of the replay-vs-eager moments test.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 50 python -m pytest tests/test_gpu_model.py -q -m gpu -k "adam_moments" 2>&1 | tail -2
cat gpurun_out/r04_replay_vs_eager_moments_bf16.json | tr -d "\n" | cut -c1-700
