#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03f; mkdir -p $O
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -k "loss_curve" > $O/curve.log 2>&1; echo "curve rc=$? $(( $(date +%s)-t0 ))s"; tail -5 $O/curve.log | cut -c1-1500
timeout 900 python bench.py --steps 30 --warmup 4 --no-f32 --no-cpu-baseline --retune --save-tuning $O/tuning_retuned_bf16.json > $O/bench_retune.json 2> $O/bench_retune.err; echo "retune rc=$? $(( $(date +%s)-t0 ))s"
python -c "import json;d=json.loads(open('$O/bench_retune.json').read().strip().splitlines()[-1]);print('retuned-live ms/step %.2f'%d['ms_per_step'])"
cp video_prediction_amd/tuning_gfx950_bf16.json $O/tuning_shipped.json
for v in shipped retuned shipped2 retuned2; do
  if [ "${v:0:7}" = "retuned" ]; then cp $O/tuning_retuned_bf16.json video_prediction_amd/tuning_gfx950_bf16.json; else cp $O/tuning_shipped.json video_prediction_amd/tuning_gfx950_bf16.json; fi
  timeout 300 python bench.py --steps 40 --warmup 4 --no-f32 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  python -c "import json;d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]);print('$v ms/step %.2f ring %.1f us'%(d['ms_per_step'], d['roofline']['avg_launch_us']))"
done
cp $O/tuning_shipped.json video_prediction_amd/tuning_gfx950_bf16.json
echo "total $(( $(date +%s)-t0 ))s"
