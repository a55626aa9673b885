#!/bin/bash
# In-call A/B runner (replaces the one-off tests/tools/ab_calls/r0X_callN.sh scripts): every arm is a set of environment assignments
# applied to the same bench.py command, arms are interleaved `REPS` times inside ONE gpurun call (boxes differ by several percent; only
# arms of one call are comparable), and one summary line per run is printed and written to $OUT/summary.txt.
#   usage: OUT=gpurun_out/r05a REPS=2 [CONFIG=c2] [BENCH_ARGS="--steps 30 --warmup 6"] ab_run.sh name1 "ENV1=a ENV2=b" name2 "" ...
#   an arm may carry extra bench.py arguments as ARGS=--flag,value (commas become spaces)
#   an arm's environment may name another build of the library: SAVP_LIB=video_prediction_amd/ab/libsavp_hip_<tag>.so (build_variant.sh)
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=${OUT:-gpurun_out/ab}; REPS=${REPS:-2}; mkdir -p $OUT
B=${BENCH_ARGS:---steps 30 --warmup 6 --no-f32 --no-cpu-baseline --inst-steps 4}
[ -n "$CONFIG" ] && B="--config $CONFIG $B"
names=(); envs=()
while [ $# -gt 0 ]; do names+=("$1"); envs+=("$2"); shift 2; done
for rep in $(seq 1 $REPS); do
  for i in "${!names[@]}"; do
    n=${names[$i]}; e=${envs[$i]}
    e=${e//SAVP_LIB=video_prediction_amd/SAVP_LIB=$PWD/video_prediction_amd}
    x=""; ee=""
    for tok in $e; do case "$tok" in ARGS=*) x="$x ${tok#ARGS=}";; *) ee="$ee $tok";; esac; done      # ARGS=--flag,value: extra bench.py arguments of this arm
    env $ee python bench.py $B ${x//,/ } > $OUT/bench_${n}_$rep.json 2> $OUT/bench_${n}_$rep.err
  done
done
python - "$OUT" <<'P' | tee $OUT/summary.txt
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        r = d.get('roofline') or {}
        print('%-28s ms %.2f  gateconv us %.2f frac %.3f  cell %s' % (os.path.basename(f)[6:-5], d['ms_per_step'], r.get('avg_launch_us', 0), r.get('frac', 0),
              (d.get('roofline_cell') or {}).get('kernel_only')))
    except Exception as ex:
        print(os.path.basename(f), 'FAILED', ex, open(f.replace('.json', '.err')).read()[-800:])
P
