#!/bin/bash
# developer build: libsavp_hip_dev.so = the shipped objects with csrc/norm_lstm.hip recompiled under -DSAVP_LSTM_STAMPS
cd "$(dirname "$0")/../../video_prediction_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wno-unused-value -DSAVP_LSTM_STAMPS -c norm_lstm.hip -o /tmp/norm_lstm_dev.o || exit 1
objs=$(ls build/*.o | grep -v norm_lstm.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsavp_hip_dev.so $objs /tmp/norm_lstm_dev.o -ldl
