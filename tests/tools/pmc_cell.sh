#!/bin/bash
# HBM traffic of the fused ConvLSTM gate convolution (conv_ring_kernel, cell epilogue: bf16 gates + instance-norm statistics) on the
# three distinct layer shapes at N = 32: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in SEPARATE passes, --kernel-trace only.
# -> gpurun_out/r02_convlstm_cell_pmc_bf16.json  (copy to profiles/)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "lstm_h0 0x712" "lstm_h1 0x711" "lstm_h2 0x311"; do set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pc_$1_$c
    SHAPE=$1:fprop TILE=$2 CELL=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pc_$1_$c -- python $R/tests/tools/pmc_one.py > /tmp/pc.log 2>&1
  done
done
python - <<PY > $R/gpurun_out/r02_convlstm_cell_pmc_bf16.json
import collections, csv, json, os
def mean_counter(d, counter):
    per = collections.defaultdict(float); name = {}
    for r, _, fs in os.walk(d):
        for f in fs:
            if f.endswith('counter_collection.csv'):
                for row in csv.DictReader(open(os.path.join(r, f))):
                    if row['Counter_Name'] == counter and 'conv_ring' in row['Kernel_Name']:
                        per[row['Dispatch_Id']] += float(row['Counter_Value']); name[row['Dispatch_Id']] = row['Kernel_Name'].split('(')[0]
    v = list(per.values())
    return (sum(v) / len(v) if v else 0.0), (list(name.values())[0] if name else '')
def dur(d):
    t = []
    for r, _, fs in os.walk(d):
        for f in fs:
            if f.endswith('kernel_trace.csv'):
                for row in csv.DictReader(open(os.path.join(r, f))):
                    if 'conv_ring' in row['Kernel_Name']:
                        t.append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    return sum(t) / len(t) if t else 0.0
shapes = {'lstm_h0': (32, 32, 72, 128), 'lstm_h1': (16, 16, 136, 256), 'lstm_h2': (8, 8, 264, 512)}
out = {'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only), ConvLSTM gate conv FPROP with the fused cell '
               'epilogue (bf16 gates + statistics), N=32, bf16 datapath, tuned tiles; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950)',
       'layers': {}}
for name, (H, W, Cx, Cy) in shapes.items():
    f, kn = mean_counter('/tmp/pc_%s_FETCH_SIZE' % name, 'FETCH_SIZE')
    w, _ = mean_counter('/tmp/pc_%s_WRITE_SIZE' % name, 'WRITE_SIZE')
    alg = 32 * H * W * Cx * 4 + 25 * Cx * Cy * 2 + 32 * H * W * Cy * 2 + 32 * Cy * 2 * 4
    out['layers'][name] = {'kernel': kn, 'fetch_kb_raw': f, 'write_kb_raw': w, 'hbm_bytes_corrected': (2 * f + w) * 1024,
                           'algorithmic_bytes': alg, 'avg_us': dur('/tmp/pc_%s_FETCH_SIZE' % name)}
L = out['layers']
L['lstm_h3'] = dict(L['lstm_h1']); L['lstm_h4'] = dict(L['lstm_h0'])
out['avg_hbm_bytes_per_launch_five_layers'] = sum(v['hbm_bytes_corrected'] for v in L.values()) / 5
out['avg_algorithmic_bytes_five_layers'] = sum(v['algorithmic_bytes'] for v in L.values()) / 5
print(json.dumps(out, indent=1))
PY
cat $R/gpurun_out/r02_convlstm_cell_pmc_bf16.json | head -30
