#!/bin/bash
# LDS / issue counters of the gate convolution (shipped instantiation, bf16 source, cell epilogue), three layer shapes, separate passes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_lds}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for name in lstm_h0 lstm_h1 lstm_h2; do
  i=0
  for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
    i=$((i+1)); d=/tmp/pl_${name}_$i; rm -rf $d
    SHAPE=$name:fprop CELL=1 SRC16=1 TABLE=1 timeout 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- python $R/tests/tools/pmc_one.py > /tmp/pl.log 2>&1
    echo "== $name pass $i"; python $R/tests/tools/pmc_one.py report $d | grep -v "^conv_" 
  done
done 2>&1 | tee $O/lds_counters.log
