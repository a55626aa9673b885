#!/bin/bash
# final refresh of the round's evidence after the WGRAD grid change: full GPU suite, default bench line, kernel stats, C4 / C5 lines
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
TAG=r03final4
O=gpurun_out/$TAG; mkdir -p $O
t0=$(date +%s)
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$? $(( $(date +%s)-t0 ))s"; cut -c1-200 $O/${TAG}_bench.json
bash tests/tools/prof_step.sh $TAG/$TAG > $O/${TAG}_prof.log 2>&1; tail -1 $O/${TAG}_prof.log
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python bench.py --config c4 --steps 120 --warmup 5 --no-f32 > $O/${TAG}_bench_c4_kth.json 2> $O/c4.err; echo "c4 rc=$? $(( $(date +%s)-t0 ))s"
timeout 900 python bench.py --config c5 --steps 80 --warmup 5 --no-f32 > $O/${TAG}_bench_c5_128.json 2> $O/c5.err; echo "c5 rc=$? $(( $(date +%s)-t0 ))s"
timeout 1500 python -m pytest tests -x -q -m gpu > $O/full_gputest.log 2>&1; echo "gpu tests rc=$? $(( $(date +%s)-t0 ))s"; tail -2 $O/full_gputest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
echo "all $(( $(date +%s)-t0 ))s"
