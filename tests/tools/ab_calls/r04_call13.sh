#!/bin/bash
# Round 4, lease 13: why is the two-rank gloo record (two processes time-slicing ONE GPU) slower with the segmented replay than eager?
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04m
mkdir -p $OUT
for g in 0 1; do
  E=""; [ $g = 0 ] && E="--eager"; SAVP_DIST_BACKEND=gloo SAVP_BENCH_CHECK_REPLICAS=1 timeout 600 python bench.py --gpus 2 --steps 12 --warmup 3 --no-f32 --no-cpu-baseline --inst-steps 0 $E > $OUT/gloo2_graph$g.json 2> $OUT/gloo2_graph$g.err
  tail -1 $OUT/gloo2_graph$g.json | cut -c1-330
done
