#!/bin/bash
# weight gradients inside the step: total workgroups of the LDS-patch WGRAD (option wgp_split) and the tile choice of the WGRAD problems
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/r03ac; mkdir -p $O
t0=$(date +%s)
R=$PWD
run() { name=$1; shift
  env "$@" timeout 500 python bench.py --steps 40 --warmup 4 --no-f32 --no-cpu-baseline $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err
  python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d['config']
    print(sys.argv[2], 'ms/step %.2f (%s)'%(d['ms_per_step'], c['submission']), 'eager %.2f'%c['eager_ms_per_step'], 'ring us %.2f'%(d['roofline']['avg_launch_us']))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
EXTRA=""
run default A=1
run split512 SAVP_WGP_SPLIT=512
run split1024 SAVP_WGP_SPLIT=1024
run split1536 SAVP_WGP_SPLIT=1536
run default2 A=1
MODES=2 timeout 600 python tests/tools/insitu_tune.py $R/$O/insitu_wgrad.json 16 4 2>&1 | grep -v amdgpu.ids | tee $O/insitu_wgrad.log | cut -c1-260
EXTRA="--tuning-table $R/$O/insitu_wgrad.json" run wgrad_insitu A=1
EXTRA="" run default3 A=1
echo "total $(( $(date +%s)-t0 ))s"
