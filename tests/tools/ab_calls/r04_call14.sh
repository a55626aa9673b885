#!/bin/bash
# Round 4, lease 14: Python-side additions -- generate() as a replayed hipGraph + shared zero arena, use_tile_concat=False, the other shipped
# recipes, inference throughput.  No kernel source changed (source id of the committed evidence stays valid).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04n
mkdir -p $OUT
python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $OUT/smoke.json 2> $OUT/smoke.err || { echo "SMOKE FAILED"; tail -25 $OUT/smoke.err; exit 1; }
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "generate_replays or best_of_n or shipped" > $OUT/tests_new.log 2>&1
echo "rc=$?" >> $OUT/tests_new.log; tail -30 $OUT/tests_new.log | cut -c1-400
timeout 300 python - > $OUT/untiled.log 2>&1 <<'P'
import sys
sys.path.insert(0, '.')
from tests import gpu_model_checks as G
off = dict(video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.0, vae_gan_feature_cdist_weight=0.0)
res = G.check_generator_forward(nz=8, B=1, T=4, tag='gen_fwd_untiled_latent', use_tile_concat=False)
res += G.check_train_step(B=1, T=4, nz=8, steps=1, tag='train_untiled_latent', use_tile_concat=False, **off)
bad = 0
for n, e, t in res:
    bad += not (e <= t)
    print('%-4s %-90s %.3e (tol %.1e)' % ('ok' if e <= t else 'FAIL', n, e, t))
print('untiled: %d failures' % bad)
P
tail -25 $OUT/untiled.log | cut -c1-300
for c in c2 c4 c5 c1; do
  timeout 300 python tests/tools/bench_generate.py --config $c > $OUT/generate_$c.json 2> $OUT/generate_$c.err || tail -12 $OUT/generate_$c.err
  tail -1 $OUT/generate_$c.json | cut -c1-420
done
