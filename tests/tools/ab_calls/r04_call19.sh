#!/bin/bash
# Round 4, lease 19: minimal reproducer of the replay / eager ordering (tests/tools/ab_calls/graph_order_probe.py).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
OUT=gpurun_out/r04s
mkdir -p $OUT
timeout 300 python tests/tools/ab_calls/graph_order_probe.py > $OUT/probe.log 2>&1; tail -12 $OUT/probe.log | cut -c1-200
