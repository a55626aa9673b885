#!/bin/bash
# Round 4, first lease: (1) the product paths no earlier round executed (RCCL at world size 1 through the engine, bench.py under
# torchrun and savp_allreduce_bucket; the segmented replay; C4 / C5 golden steps at bench shapes; z-less gate DGRAD kernels),
# (2) in-call A/B of the step: round-3 library | new library without the z-less DGRAD | new library, each arm twice,
# (3) the rest of the GPU suite.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04a
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_dp.py tests/test_gpu_ops.py::test_tiled_z_gradient_and_gapped_gate_dgrad -x -q -m gpu > $OUT/tests_new.log 2>&1
echo "rc=$?" >> $OUT/tests_new.log
tail -5 $OUT/tests_new.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k golden > $OUT/tests_golden.log 2>&1
echo "rc=$?" >> $OUT/tests_golden.log
tail -5 $OUT/tests_golden.log
B="--steps 30 --warmup 6 --no-f32 --no-cpu-baseline --inst-steps 4"
for rep in 1 2; do
  SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_r3.so SAVP_ZLESS_DGRAD=0 python bench.py $B > $OUT/bench_r3lib_$rep.json 2> $OUT/bench_r3lib_$rep.err
  SAVP_ZLESS_DGRAD=0 python bench.py $B > $OUT/bench_new_nozless_$rep.json 2> $OUT/bench_new_nozless_$rep.err
  python bench.py $B > $OUT/bench_new_$rep.json 2> $OUT/bench_new_$rep.err
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04a/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], 'ms %.2f' % d['ms_per_step'], 'gateconv us %.2f' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], d['config'].get('submission'))
    except Exception as ex:
        print(f, 'FAILED', ex)
P
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_dp.py -k "not golden and not tiled_z" > $OUT/tests_rest.log 2>&1
echo "rc=$?" >> $OUT/tests_rest.log
tail -5 $OUT/tests_rest.log
