#!/bin/bash
# usage: SHAPE=lstm_h0:fprop TILE=0x712 tests/tools/pmc_one.sh   -> prints per-dispatch PMC means of the conv kernel (three counter passes)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmc1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc1/a -- python $R/tests/tools/pmc_one.py > /tmp/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/pmc1/b -- python $R/tests/tools/pmc_one.py >> /tmp/pmc1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1/c -- python $R/tests/tools/pmc_one.py >> /tmp/pmc1.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d /tmp/pmc1/d -- python $R/tests/tools/pmc_one.py >> /tmp/pmc1.log 2>&1
echo "== SHAPE=$SHAPE TILE=$TILE CELL=$CELL"
python $R/tests/tools/pmc_one.py report /tmp/pmc1
f=$(find /tmp/pmc1/a -name "*kernel_trace.csv" | head -1)
python - "$f" <<PY
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'conv_' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print('durations us', d, 'LDS', rows[0]['LDS_Block_Size'], 'VGPR', rows[0]['VGPR_Count'], 'grid', rows[0]['Grid_Size_X'], rows[0]['Workgroup_Size_X'])
PY
tail -3 /tmp/pmc1.log | cut -c1-200
