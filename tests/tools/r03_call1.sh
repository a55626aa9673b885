#!/bin/bash
# round 3, first GPU lease: validate-or-delete the round-2 experimental switches, first runs of the new parity / multi-rank tests,
# bench A/B of every switch inside ONE call (same box).
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
t0=$(date +%s)
SAVP_TEST_EXPERIMENTAL=1 SAVP_S2FPROP=1 timeout 400 python -m pytest tests/test_gpu_ops.py -q -k "experimental" > $O/exp_tests.log 2>&1; echo "exp rc=$? $(( $(date +%s)-t0 ))s"; tail -3 $O/exp_tests.log
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "tuning_table" > $O/table.log 2>&1; echo "table rc=$? $(( $(date +%s)-t0 ))s"; tail -3 $O/table.log
timeout 900 python -m pytest tests/test_gpu_dp.py -q -k "bench_two or live_tuning" > $O/dp.log 2>&1; echo "dp rc=$? $(( $(date +%s)-t0 ))s"; tail -3 $O/dp.log
# numerics A/B of the switches on a small train step
python tests/tools/ab_switch_check.py dump /tmp/ab_base.pt > $O/ab.log 2>&1
for sw in SAVP_BF16_ACT SAVP_BF16_DGATES SAVP_EPI_BATCH SAVP_S2FPROP; do
  env $sw=1 python tests/tools/ab_switch_check.py dump /tmp/ab_$sw.pt >> $O/ab.log 2>&1
  echo "== $sw" >> $O/ab.log; python tests/tools/ab_switch_check.py cmp /tmp/ab_base.pt /tmp/ab_$sw.pt >> $O/ab.log 2>&1; echo "ab $sw rc=$?"
done
env SAVP_BF16_ACT=1 SAVP_BF16_DGATES=1 python tests/tools/ab_switch_check.py dump /tmp/ab_both.pt >> $O/ab.log 2>&1
echo "== ACT+DGATES" >> $O/ab.log; python tests/tools/ab_switch_check.py cmp /tmp/ab_base.pt /tmp/ab_both.pt >> $O/ab.log 2>&1; echo "ab both rc=$?"
echo "ab done $(( $(date +%s)-t0 ))s"
# bench A/B (same box): base twice (noise floor), each switch, graph
for v in "base" "base2" "act SAVP_BF16_ACT=1" "dg SAVP_BF16_DGATES=1" "actdg SAVP_BF16_ACT=1 SAVP_BF16_DGATES=1" "epi SAVP_EPI_BATCH=1" "s2f SAVP_S2FPROP=1" "all SAVP_BF16_ACT=1 SAVP_BF16_DGATES=1 SAVP_EPI_BATCH=1 SAVP_S2FPROP=1"; do
  set -- $v; name=$1; shift
  env "$@" timeout 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], 'ms/step %.2f'%d['ms_per_step'], 'ring us %.1f'%d['roofline']['avg_launch_us'], 'cell us %.1f'%d['roofline_cell']['avg_cell_us'], 'd_loss %.4f g_loss %.3f'%(d['losses']['d_loss'], d['losses']['g_loss']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
done
timeout 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --graph > $O/bench_graph.json 2> $O/bench_graph.err
python -c "import json;d=json.loads(open('$O/bench_graph.json').read().strip().splitlines()[-1]);print('graph ms/step %.2f'%d['ms_per_step'])"
echo "bench done $(( $(date +%s)-t0 ))s"
bash tests/tools/prof_step.sh r03a/base > $O/prof.log 2>&1; tail -2 $O/prof.log
echo "total $(( $(date +%s)-t0 ))s"
