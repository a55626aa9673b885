#!/bin/bash
# developer build: video_prediction_amd/ab/libsavp_hip_ringdev.so = the shipped objects with csrc/conv_ring.hip recompiled under
# -DSAVP_CONV_ABLATE (cycle stamps + ablation switches of conv_ring_kernel; tests/tools/ring_times.py, ring_attrib.py)
cd "$(dirname "$0")/../../video_prediction_amd/csrc" || exit 1
mkdir -p ../ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-value -DSAVP_CONV_ABLATE -c conv_ring.hip -o /tmp/conv_ring_dev.o || exit 1
objs=$(ls build/*.o | grep -v conv_ring.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ab/libsavp_hip_ringdev.so $objs /tmp/conv_ring_dev.o -ldl
