#!/bin/bash
# usage: tests/tools/prof_one.sh  (env SHAPE MODE TILE SK SAVP_DBG) -> prints mean kernel duration of the conv kernels
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof1
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof1 -- python /root/repo/tests/tools/micro_one.py > /tmp/prof1.log 2>&1
f=$(find /tmp/prof1 -name "*kernel_trace.csv" | head -1)
echo "dbg=$SAVP_DBG shape=$SHAPE mode=$MODE tile=$TILE sk=$SK: $(python /root/repo/tests/tools/trace_groups.py $f 20 | grep conv_ | sed 's/  */ /g' | cut -d' ' -f2-)"
