#!/bin/bash
# HBM traffic of every kernel of the train step: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X guide: TCC has 4
# slots, FETCH_SIZE costs 3, WRITE_SIZE 2) with --kernel-trace only -> gpurun_out/<tag>_pmc_step.json (per-kernel mean bytes per
# dispatch; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md)
TAG=${1:-pmc}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > /tmp/pmc_$c.log 2>&1
done
mkdir -p $R/gpurun_out
python $R/tests/tools/pmc_step_report.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $R/gpurun_out/${TAG}_pmc_step.json
head -c 1500 $R/gpurun_out/${TAG}_pmc_step.json
