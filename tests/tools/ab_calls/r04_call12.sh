#!/bin/bash
# Round 4, lease 12: fused-operator entry points (one host call per cell / conv+norm): parity, host-issue A/B, then the evidence script again.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04l
mkdir -p $OUT
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $OUT/smoke.json 2> $OUT/smoke.err || { echo "SMOKE FAILED"; tail -25 $OUT/smoke.err; exit 1; }
timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -k "recipe or c2 or b16 or small or config or hipgraph" > $OUT/tests_model.log 2>&1
rc=$?; echo "rc=$rc" >> $OUT/tests_model.log; tail -4 $OUT/tests_model.log
[ $rc -ne 0 ] && exit 1
B="--steps 30 --warmup 6 --no-f32 --no-cpu-baseline --inst-steps 4"
for rep in 1 2; do
  SAVP_FUSED_ENTRIES=0 python bench.py $B > $OUT/bench_apart_$rep.json 2> $OUT/bench_apart_$rep.err
  python bench.py $B > $OUT/bench_fused_$rep.json 2> $OUT/bench_fused_$rep.err
done
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04l/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], 'ms %.2f' % d['ms_per_step'], 'eager %.2f' % d['config']['eager_ms_per_step'], 'host issue %.2f' % d['config']['host_issue_ms_per_step'])
    except Exception as ex:
        print(f, 'FAILED', ex, open(f.replace('.json', '.err')).read()[-1500:])
P
bash tests/tools/r04_final.sh
