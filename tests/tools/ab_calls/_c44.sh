O=gpurun_out/r05ai; mkdir -p $O
AB=$PWD/video_prediction_amd/ab
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x > $O/ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/ops.log | cut -c1-300
for spec in lstm_h1:fprop:711:cell16 lstm_h0:fprop:712:cell16 lstm_h2:fprop:311:cell16 dec64:dgrad:721:src16; do
  SAVP_LIB=$AB/libsavp_hip_stamps.so python tests/tools/ring_times.py $spec 2>&1 | grep -v "amdgpu.ids\|per wave"
done | tee $O/ring_stamps_hoist.log
OUT=$O REPS=2 bash tests/tools/ab_run.sh old "SAVP_LIB=$AB/libsavp_hip_ringold.so" hoist ""
