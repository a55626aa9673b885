AB=$PWD/video_prediction_amd/ab
for spec in lstm_h1:fprop:711:cell16 lstm_h0:fprop:712:cell16 lstm_h2:fprop:311:cell16 dec64:dgrad:721:src16; do
  SAVP_LIB=$AB/libsavp_hip_stamps.so python tests/tools/ring_times.py $spec 2>&1 | grep -v "amdgpu.ids\|per wave"
done
