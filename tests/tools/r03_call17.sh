#!/bin/bash
# where do 38 k cycles go before the first barrier of the 8-wave h0 DGRAD? + per-kernel stats of the base and the new build
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03p; mkdir -p $O
t0=$(date +%s)
P=$PWD/video_prediction_amd
SPECS="lstm_h0:dgrad:712:src16 lstm_h0:dgrad:712 lstm_h0:dgrad:722:src16 lstm_h0:dgrad:312:src16 lstm_h0:dgrad:711:src16 lstm_h1:dgrad:712:src16 lstm_h0:fprop:712:cell16 lstm_h0:fprop:712"
KWARM=1 SAVP_LIB=$P/libsavp_hip_ringdev.so timeout 200 python tests/tools/ring_times.py $SPECS 2>&1 | grep -v amdgpu.ids | tee $O/stamps.log
echo "--- prof base"; SAVP_LIB=$P/libsavp_hip_base.so bash tests/tools/prof_step.sh r03p/base 2>&1 | tail -1
echo "--- prof new"; bash tests/tools/prof_step.sh r03p/new 2>&1 | tail -1
echo "total $(( $(date +%s)-t0 ))s"
