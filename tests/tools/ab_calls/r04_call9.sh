#!/bin/bash
# Round 4, ninth lease: in-step re-tune of the weight-gradient problems (bf16 operands are new this round), state of the three workloads.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04i
mkdir -p $OUT
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $OUT/smoke.json 2> $OUT/smoke.err || { echo "SMOKE FAILED"; tail -25 $OUT/smoke.err; exit 1; }
T=video_prediction_amd/tuning_gfx950_bf16.json
MODES=2 python tests/tools/insitu_tune.py $OUT/table_c2.json 16 4 > $OUT/insitu_c2.log 2>&1; tail -3 $OUT/insitu_c2.log
[ -s $OUT/table_c2.json ] && cp $OUT/table_c2.json $T
cp $T $OUT/tuning_gfx950_bf16.json
B="--steps 30 --warmup 6 --no-f32 --no-cpu-baseline --inst-steps 4"
for rep in 1 2; do
  python bench.py $B > $OUT/bench_default_$rep.json 2> $OUT/bench_default_$rep.err
done
python bench.py --config c4 $B > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python bench.py --config c5 $B > $OUT/bench_c5.json 2> $OUT/bench_c5.err
python - <<'P'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04i/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], 'ms %.2f' % d['ms_per_step'], 'fps %.0f' % d['value'], 'gateconv us %.2f' % d['roofline']['avg_launch_us'], 'frac %.3f' % d['roofline']['frac'], 'cell kernel-only', d['roofline_cell']['kernel_only']['avg_cell_us'])
    except Exception as ex:
        print(f, 'FAILED', ex, open(f.replace('.json', '.err')).read()[-1500:])
P
bash tests/tools/prof_step.sh r04i/default
