#!/bin/bash
# Round 4, lease 25: the driver's default command on the final tree (the committed counter / family files now carry this build's source id).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 150 python bench.py > gpurun_out/r04/r04_bench_final.json 2> gpurun_out/r04/r04_bench_final.err; echo "rc=$?"
tail -1 gpurun_out/r04/r04_bench_final.json | cut -c1-400
