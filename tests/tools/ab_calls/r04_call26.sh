#!/bin/bash
# Round 4, lease 26: __graft_entry__.smoke() with the tiny train step added.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out/r04x
timeout 100 python __graft_entry__.py smoke > gpurun_out/r04x/smoke.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r04x/smoke.log | cut -c1-300
