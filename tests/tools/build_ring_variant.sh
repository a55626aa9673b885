#!/bin/bash
# developer A/B build: video_prediction_amd/ab/libsavp_hip_<tag>.so = the shipped objects with csrc/conv_ring.hip recompiled under extra
# flags (e.g. build_ring_variant.sh mid -DSAVP_RING_DMA_MID; early -DSAVP_RING_EARLY_DMA); used through SAVP_LIB=... in one gpurun call
TAG=$1; shift
cd "$(dirname "$0")/../../video_prediction_amd/csrc" || exit 1
mkdir -p ../ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-value "$@" -c conv_ring.hip -o /tmp/conv_ring_$TAG.o || exit 1
objs=$(ls build/*.o | grep -v conv_ring.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ab/libsavp_hip_$TAG.so $objs /tmp/conv_ring_$TAG.o -ldl
