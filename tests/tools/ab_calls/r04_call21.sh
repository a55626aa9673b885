#!/bin/bash
# Round 4, lease 21: memset nodes inside a replayed graph (tests/tools/ab_calls/graph_memset_probe.py).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04u
mkdir -p $OUT
timeout 300 python tests/tools/ab_calls/graph_memset_probe.py > $OUT/memset_probe.log 2>&1; grep -v amdgpu.ids $OUT/memset_probe.log | tail -12 | cut -c1-200
