#!/bin/bash
# round 3 evidence: the ONE command (collect_profiles.sh) + the C4 / C5 lines with >= 10 s timed regions + the two-rank line over gloo
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
TAG=${1:-r03final}
O=gpurun_out/$TAG; mkdir -p $O
t0=$(date +%s)
bash tests/tools/collect_profiles.sh $TAG 2>&1 | tail -12
echo "collect done $(( $(date +%s)-t0 ))s"
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python bench.py --config c4 --steps 120 --warmup 5 --no-f32 --save-tuning $O/tuning_c4.json > $O/${TAG}_bench_c4_kth.json 2> $O/c4.err; echo "c4 rc=$? $(( $(date +%s)-t0 ))s"; cut -c1-260 $O/${TAG}_bench_c4_kth.json
timeout 900 python bench.py --config c5 --steps 80 --warmup 5 --no-f32 --save-tuning $O/tuning_c5.json > $O/${TAG}_bench_c5_128.json 2> $O/c5.err; echo "c5 rc=$? $(( $(date +%s)-t0 ))s"; cut -c1-260 $O/${TAG}_bench_c5_128.json
SAVP_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 > $O/${TAG}_bench_2ranks_one_gpu_gloo.json 2> $O/dp.err; echo "dp2 rc=$? $(( $(date +%s)-t0 ))s"; cut -c1-400 $O/${TAG}_bench_2ranks_one_gpu_gloo.json
echo "total $(( $(date +%s)-t0 ))s"
