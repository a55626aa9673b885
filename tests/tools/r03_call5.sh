#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03e; mkdir -p $O
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -k "b16_t30 or loss_curve or hipgraph or reduces_l1" > $O/model_new.log 2>&1; echo "model new rc=$? $(( $(date +%s)-t0 ))s"; tail -12 $O/model_new.log
timeout 600 python bench.py --steps 40 --warmup 4 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? $(( $(date +%s)-t0 ))s"; cat $O/bench_default.json | cut -c1-3500; tail -3 $O/bench_default.err
timeout 300 python bench.py --steps 40 --warmup 4 --eager --no-f32 --no-cpu-baseline > $O/bench_eager.json 2> $O/bench_eager.err; python -c "import json;d=json.loads(open('$O/bench_eager.json').read().strip().splitlines()[-1]);print('eager ms/step %.2f'%d['ms_per_step'], d['config']['submission'])"
timeout 600 python -m pytest tests/test_gpu_dp.py -q -k "bench_two" > $O/dp.log 2>&1; echo "dp rc=$? $(( $(date +%s)-t0 ))s"; tail -3 $O/dp.log
timeout 600 python -m pytest tests/test_gpu_model.py -q -k "learned_prior" > $O/lp.log 2>&1; echo "lp rc=$? $(( $(date +%s)-t0 ))s"; tail -3 $O/lp.log
echo "total $(( $(date +%s)-t0 ))s"
