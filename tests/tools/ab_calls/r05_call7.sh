#!/bin/bash
# Round 5, seventh lease: what the slab DMAs cost the ring kernel's main loop -- stamps-only builds with all / one / none of the LW slab DMA
# instructions per wave and entry (timing builds, results wrong by construction).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/r05g; mkdir -p $O
for v in stamps stampsd1 stampsd0; do
  for spec in lstm_h0:fprop:712:cell16 lstm_h1:fprop:711:cell16 lstm_h2:fprop:311:cell16 lstm_h0:dgrad:711:src16; do
    SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_$v.so python tests/tools/ring_times.py $spec 2>&1 | grep -v "amdgpu.ids\|per wave" | sed "s/^/$v /"
  done
done | tee $O/ring_dma_ablate.log
