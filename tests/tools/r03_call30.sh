#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
t0=$(date +%s)
mkdir -p gpurun_out/r03final3
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03final3/full_gputest.log 2>&1; echo "gpu tests rc=$? $(( $(date +%s)-t0 ))s"; tail -3 gpurun_out/r03final3/full_gputest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
bash tests/tools/r03_final.sh r03final3
echo "all $(( $(date +%s)-t0 ))s"
