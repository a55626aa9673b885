#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
mkdir -p gpurun_out/r03r
timeout 300 python tests/tools/bench_dgrad_noz.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03r/dgrad_noz.log
