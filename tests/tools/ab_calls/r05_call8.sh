#!/bin/bash
# Round 5, eighth lease: wide weight slabs (8 / 9 k-steps per entry: the look-ahead of three entries then covers the LDS-DMA latency) against the
# shipped instantiations of the 16x16 / 8x8 gate convolutions, stamps-only build, isolated launches.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05h; mkdir -p $O
for spec in lstm_h1:fprop:711:cell16 lstm_h1:fprop:1711:cell16 lstm_h1:fprop:1311:cell16 lstm_h1:fprop:312:cell16 lstm_h2:fprop:311:cell16 lstm_h2:fprop:1311:cell16 lstm_h2:fprop:1711:cell16 lstm_h1:dgrad:711:src16 lstm_h1:dgrad:1711:src16 lstm_h2:dgrad:311:src16 lstm_h2:dgrad:1311:src16; do
  SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_stamps.so python tests/tools/ring_times.py $spec 2>&1 | grep -v "amdgpu.ids\|per wave"
done | tee $O/ring_wide.log
