#!/bin/bash
# kernel-trace of a few bench steps -> gpurun_out/<tag>_kernel_stats.csv + last-step trace (one step, graph off)
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_s
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32 --eager --inst-steps 0 ${BENCH_ARGS} > /tmp/prof_s.log 2>&1
tail -1 /tmp/prof_s.log | cut -c1-400
mkdir -p $R/gpurun_out
f=$(find /tmp/prof_s -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/${TAG}_kernel_stats.csv
f=$(find /tmp/prof_s -name "*kernel_trace.csv" | head -1)
python - "$f" "$R/gpurun_out/${TAG}_last_step_trace.csv" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo=idx[-3]+1; hi=idx[-1]+1
keep=["Kernel_Name","Start_Timestamp","End_Timestamp","LDS_Block_Size","VGPR_Count","SGPR_Count","Workgroup_Size_X","Grid_Size_X","Grid_Size_Y","Grid_Size_Z"]
w=csv.DictWriter(open(sys.argv[2],"w"),keep); w.writeheader()
for r in rows[lo:hi]:
    r2={k:r[k] for k in keep}; r2["Kernel_Name"]=r2["Kernel_Name"][:90]; w.writerow(r2)
print("rows",len(rows),"step launches",hi-lo,"step ms",(int(rows[hi-1]["End_Timestamp"])-int(rows[lo]["Start_Timestamp"]))/1e6)
PY
