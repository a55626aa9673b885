#!/bin/bash
# ONE command for a round's committed evidence (run on the GPU box: gpurun -- 'bash tests/tools/collect_profiles.sh r03'):
#   <tag>_bench.json                    the default `python bench.py` line (170 timed steps, f32 object, cpu_baseline)
#   <tag>_kernel_stats.csv, <tag>_last_step_trace.csv      rocprofv3 --kernel-trace --stats of 4 timed + 2 warm-up steps (tests/tools/prof_step.sh)
#   <tag>_convlstm_cell_pmc_bf16.json   ConvLSTM gate conv (cell epilogue, bf16 source, the shipped table's instantiation), three layer
#                                       shapes at N = 32: rocprofv3 --pmc in SEPARATE passes -- FETCH_SIZE, WRITE_SIZE (HBM traffic; FETCH x2
#                                       on gfx950 per MI355X_MICROARCH.md) and SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
# Everything lands in gpurun_out/<tag>/ ; copy what is to be judged into profiles/.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"; cut -c1-300 $O/${TAG}_bench.json
bash tests/tools/prof_step.sh $TAG/$TAG > $O/${TAG}_prof.log 2>&1; tail -1 $O/${TAG}_prof.log
python tests/tools/kernel_families.py $O/${TAG}_kernel_stats.csv 6 > $O/${TAG}_kernel_families.json
cd /tmp
for name in lstm_h0 lstm_h1 lstm_h2; do
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    d=/tmp/pc_${name}_$(echo $pass | cut -d' ' -f1); rm -rf $d
    SHAPE=$name:fprop CELL=1 SRC16=1 TABLE=1 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- python $R/tests/tools/pmc_one.py > /tmp/pc.log 2>&1
  done
done
python $R/tests/tools/pmc_cell_report.py > $O/${TAG}_convlstm_cell_pmc_bf16.json; head -c 1500 $O/${TAG}_convlstm_cell_pmc_bf16.json
