"""Kernel time by family from a rocprofv3 kernel_stats.csv (tests/tools/prof_step.sh: 6 profiled eager steps) -> JSON on stdout, stamped with the
checkout's source id (video_prediction_amd.lib.source_id) so that bench.py only quotes it for the same kernel sources + tuning tables.
usage: kernel_families.py <tag>_kernel_stats.csv [steps=6]"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_prediction_amd import lib

FAMILIES = (('gate conv FPROP (conv_gate_kernel: weights in B-fragment order from L2)', ('conv_gate_kernel',)), ('ring conv (FPROP / DGRAD, LDS patch + DMA weight ring)', ('conv_ring',)), ('RGB-side / thin / stride-2 DGRAD convs', ('thin_fprop', 'wthin_', 'thin8', 's2dgrad')),
            ('weight gradients', ('wgrad',)),
            ('ConvLSTM gate block', ('lstm_fused', 'lstm_fwd', 'lstm_bwd', 'lstm_gates')), ('instance norm', ('inorm',)),
            ('generic conv (implicit GEMM)', ('conv_fd',)), ('patch conv', ('conv_patch',)), ('CDNA + composite', ('cdna', 'composite')),
            ('tiled-z gradient', ('tiled_z',)), ('dense', ('dense',)),
            ('weight prep (spectral norm, packs, folds)', ('snb_', 'pack_', 'fold_', 'sn_')), ('fills / copies / selects', ('fill', 'copy', 'select', 'Fill', 'tile_channels')))


def family(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return 'everything else'


rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
agg = {}
for r in rows:
    f = family(re.sub(r'\(.*', '', r['Name']))
    c, t = agg.get(f, (0, 0.0))
    agg[f] = (c + int(r['Calls']), t + float(r['TotalDurationNs']))
out = {'what': 'kernel time per train step by family, rocprofv3 --kernel-trace --stats of %g eager steps of the default workload (profiler on: '
               'the sum is above the un-profiled step time)' % steps,
       'source_id': lib.source_id(), 'launches_per_step': sum(c for c, _ in agg.values()) / steps,
       'kernel_ms_per_step': sum(t for _, t in agg.values()) / steps / 1e6,
       'families': {f: {'ms_per_step': round(t / steps / 1e6, 3), 'launches_per_step': round(c / steps, 1)}
                    for f, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])}}
print(json.dumps(out, indent=1))
