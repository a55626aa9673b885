#!/bin/bash
# A round's evidence in ONE lease (gpurun -- 'bash tests/tools/final_evidence.sh r05'); everything lands in gpurun_out/<tag>/, what is judged is
# copied to profiles/<tag>_*:
#   collect_profiles.sh <tag> default bench line (f32 object, cpu_baseline), rocprofv3 kernel stats + last-step trace, gate-conv PMC (c2)
#   the whole GPU suite
#   bench lines of c4 / c5 / c1, the forced-RCCL world-1 line under torch.distributed.run, the two-rank line over gloo
#   gate-conv PMC of c4 / c5; inference throughput (bench_generate.py); the hipGraph memset-node probe
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
TAG=${1:-r05}
O=gpurun_out/$TAG; mkdir -p $O
t0=$(date +%s)
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $O/smoke.json 2> $O/smoke.err || { echo "SMOKE FAILED"; tail -25 $O/smoke.err; exit 1; }
bash tests/tools/collect_profiles.sh $TAG 2>&1 | tail -12
echo "collect done $(( $(date +%s)-t0 ))s"
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m video_prediction_amd.debug > $O/${TAG}_box_fingerprint.json 2>/dev/null
timeout 1500 python -m pytest tests -q -m gpu > $O/${TAG}_full_gputest.log 2>&1; echo "gputest rc=$? $(( $(date +%s)-t0 ))s"; tail -3 $O/${TAG}_full_gputest.log
# the same suite with every allocation, the caller-owned scratch and all of LDS NaN-poisoned before each launch, the zero arena checked (video_prediction_amd/debug.py)
SAVP_POISON=1 timeout 1500 python -m pytest tests -q -m gpu > $O/${TAG}_full_gputest_poisoned.log 2>&1; echo "poisoned gputest rc=$? $(( $(date +%s)-t0 ))s"; tail -3 $O/${TAG}_full_gputest_poisoned.log
cp gpurun_out/pytest_evidence/soak_*.json gpurun_out/r06_replay_vs_eager_*.json $O/ 2>/dev/null
# two default-shaped bench runs in ONE lease: identical `losses` (the step is bit-reproducible)
for i in 1 2; do timeout 300 python bench.py --steps 40 --no-f32 --no-cpu-baseline --no-workloads --inst-steps 0 > $O/${TAG}_bench_repeat_$i.json 2>/dev/null; python -c "import json; d=json.load(open('$O/${TAG}_bench_repeat_$i.json')); print('repeat $i', d['ms_per_step'], d['losses'])"; done
timeout 600 python bench.py --config c4 --steps 60 --warmup 5 --no-f32 --no-cpu-baseline > $O/${TAG}_bench_c4_kth.json 2> $O/c4.err; echo "c4 rc=$? $(( $(date +%s)-t0 ))s"; cut -c1-200 $O/${TAG}_bench_c4_kth.json
timeout 600 python bench.py --config c5 --steps 40 --warmup 5 --no-f32 --no-cpu-baseline > $O/${TAG}_bench_c5_128.json 2> $O/c5.err; echo "c5 rc=$? $(( $(date +%s)-t0 ))s"; cut -c1-200 $O/${TAG}_bench_c5_128.json
timeout 600 python bench.py --config c1 --steps 100 --warmup 5 --no-f32 --no-cpu-baseline > $O/${TAG}_bench_c1_det.json 2> $O/c1.err; echo "c1 rc=$? $(( $(date +%s)-t0 ))s"; cut -c1-200 $O/${TAG}_bench_c1_det.json
SAVP_FORCE_DIST=1 SAVP_BENCH_CHECK_REPLICAS=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 1 --steps 20 --warmup 5 --no-f32 --no-cpu-baseline > $O/${TAG}_bench_rccl_world1_forced.json 2> $O/rccl.err; echo "rccl rc=$? $(( $(date +%s)-t0 ))s"; cut -c1-300 $O/${TAG}_bench_rccl_world1_forced.json
SAVP_DIST_BACKEND=gloo SAVP_BENCH_CHECK_REPLICAS=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 --no-f32 --no-cpu-baseline > $O/${TAG}_bench_2ranks_one_gpu_gloo.json 2> $O/dp.err; echo "dp2 rc=$? $(( $(date +%s)-t0 ))s"; cut -c1-300 $O/${TAG}_bench_2ranks_one_gpu_gloo.json
for c in c2 c4 c5 c1; do
  timeout 300 python tests/tools/bench_generate.py --config $c > $O/${TAG}_generate_$c.json 2> $O/generate_$c.err || tail -5 $O/generate_$c.err
done
echo "generate $(( $(date +%s)-t0 ))s"; cut -c1-200 $O/${TAG}_generate_c2.json
timeout 300 python tests/tools/ab_calls/graph_memset_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_graph_memset_probe.log
cd /tmp
for set in c4 c5; do
  names="c4_h0 c4_h1 c4_h2"; [ $set = c5 ] && names="c5_h4 c5_h5"
  for name in $names; do
    for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
      d=/tmp/pc_${name}_$(echo $pass | cut -d' ' -f1); rm -rf $d
      SHAPE=$name:fprop CELL=1 SRC16=1 TABLE=1 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- python ${GRAFT_REPO_ROOT:-/root/repo}/tests/tools/pmc_one.py > /tmp/pc.log 2>&1
    done
  done
  PMC_SET=$set python ${GRAFT_REPO_ROOT:-/root/repo}/tests/tools/pmc_cell_report.py > ${GRAFT_REPO_ROOT:-/root/repo}/$O/${TAG}_convlstm_cell_pmc_bf16_$set.json
done
echo "total $(( $(date +%s)-t0 ))s"
