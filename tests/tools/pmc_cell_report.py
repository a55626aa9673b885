"""Folds the three rocprofv3 --pmc passes of tests/tools/collect_profiles.sh into one JSON (stdout)."""
import collections
import csv
import json
import os


def per_dispatch(d, counter):
    per = collections.defaultdict(float)
    name = ''
    for r, _, fs in os.walk(d):
        for f in fs:
            if f.endswith('counter_collection.csv'):
                for row in csv.DictReader(open(os.path.join(r, f))):
                    if row['Counter_Name'] == counter and ('conv_ring' in row['Kernel_Name'] or 'conv_gate_kernel' in row['Kernel_Name']):
                        per[row['Dispatch_Id']] += float(row['Counter_Value'])
                        name = row['Kernel_Name'].split('(')[0]
    v = list(per.values())
    return (sum(v) / len(v) if v else 0.0), name


def dur_us(d):
    t = []
    for r, _, fs in os.walk(d):
        for f in fs:
            if f.endswith('kernel_trace.csv'):
                for row in csv.DictReader(open(os.path.join(r, f))):
                    if ('conv_ring' in row['Kernel_Name'] or 'conv_gate_kernel' in row['Kernel_Name']):
                        t.append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    return sum(t) / len(t) if t else 0.0


SETS = {'c2': {'lstm_h0': (32, 32, 32, 72, 128), 'lstm_h1': (32, 16, 16, 136, 256), 'lstm_h2': (32, 8, 8, 264, 512)},
        'c4': {'c4_h0': (32, 32, 32, 96, 128), 'c4_h1': (32, 16, 16, 160, 256), 'c4_h2': (32, 8, 8, 288, 512)},
        'c5': {'c5_h4': (16, 16, 16, 520, 1024), 'c5_h5': (16, 32, 32, 264, 512)}}
PMC_SET = os.environ.get('PMC_SET', 'c2')
shapes = SETS[PMC_SET]
out = {'note': 'rocprofv3 --pmc, separate passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE), --kernel-trace only; '
               'ConvLSTM gate conv FPROP with the fused cell epilogue (bf16 gates + statistics), N=32, bf16 cell input, the kernel the engine launches '
               "(conv_gate_kernel where it has an instantiation, else the shipped tuning table's conv_ring_kernel); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950); SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's "
               '1024 SIMDs (= 32 cycles x number of 32x32x16 MFMAs), GRBM_GUI_ACTIVE over its 8 XCDs: mfma_busy_frac = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)', 'layers': {}}
for name, (NB, H, W, Cx, Cy) in shapes.items():
    f, kn = per_dispatch('/tmp/pc_%s_FETCH_SIZE' % name, 'FETCH_SIZE')
    w, _ = per_dispatch('/tmp/pc_%s_WRITE_SIZE' % name, 'WRITE_SIZE')
    d3 = '/tmp/pc_%s_SQ_VALU_MFMA_BUSY_CYCLES' % name
    mb, _ = per_dispatch(d3, 'SQ_VALU_MFMA_BUSY_CYCLES')
    sb, _ = per_dispatch(d3, 'SQ_BUSY_CYCLES')
    ga, _ = per_dispatch(d3, 'GRBM_GUI_ACTIVE')
    alg = NB * H * W * Cx * 2 + 25 * Cx * Cy * 2 + NB * H * W * Cy * 2 + NB * Cy * 2 * 4       # bf16 in, bf16 weights, bf16 gates, stats
    flops = 2.0 * NB * H * W * Cy * 25 * Cx
    us = dur_us('/tmp/pc_%s_FETCH_SIZE' % name)
    out['layers'][name] = {'kernel': kn, 'fetch_kb_raw': f, 'write_kb_raw': w, 'hbm_bytes_corrected': (2 * f + w) * 1024,
                           'algorithmic_bytes': alg, 'traffic_over_algorithmic': ((2 * f + w) * 1024 / alg) if alg else None,
                           'avg_us': us, 'tflops': flops / us / 1e6 if us else None,
                           'SQ_VALU_MFMA_BUSY_CYCLES': mb, 'SQ_BUSY_CYCLES': sb, 'GRBM_GUI_ACTIVE': ga,
                           'mfma_busy_frac': (mb / (ga / 8.0 * 1024.0)) if ga else None,
                           'mfma_ideal_cycles_per_simd': flops / 2.0 / 512.0 / 1024.0}
L = out['layers']
out['workload'] = PMC_SET
if PMC_SET == 'c2':
    L['lstm_h3'] = dict(L['lstm_h1'])
    L['lstm_h4'] = dict(L['lstm_h0'])
import sys  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from video_prediction_amd import lib as _lib  # noqa: E402
out['source_id'] = _lib.source_id()            # bench.py quotes this file only for the same kernel sources + tuning tables
if PMC_SET == 'c2':
    out['avg_hbm_bytes_per_launch_five_layers'] = sum(v['hbm_bytes_corrected'] for v in L.values()) / 5
    out['avg_algorithmic_bytes_five_layers'] = sum(v['algorithmic_bytes'] for v in L.values()) / 5
print(json.dumps(out, indent=1))
