#!/bin/bash
# LDS / issue / MFMA counters of the LDS-patch weight gradient at the step's operands (both bf16, 928 images), separate passes per counter set
# usage: pmc_wgrad.sh <out dir under gpurun_out> [shape ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_wgrad}; mkdir -p $O; shift
SHAPES=${@:-lstm_h1}
cd /tmp; export TMPDIR=/tmp
for name in $SHAPES; do
  i=0
  for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM"; do
    i=$((i+1)); d=/tmp/pw_${name}_$i; rm -rf $d
    BF16=1 NOBIAS=1 timeout 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- python $R/tests/tools/bench_wgrad.py $name > /tmp/pw.log 2>&1
    echo "== $name pass $i"; KFILTER=wgrad_patch python $R/tests/tools/pmc_one.py report $d
  done
done 2>&1 | tee $O/wgrad_counters.log
