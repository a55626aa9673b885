"""Within-process A/B of the LDS-patch conv kernels on the SAVP layer shapes (bf16 datapath, N = 2B = 32):
conv_patch.hip (0x2xx / 0x6xx) vs conv_ring.hip (0x3xx / 0x7xx) vs the ring kernel's fused cell epilogue (bf16 gates + statistics).
Every variant is a hipGraph of IT launches, variants are interleaved over ROUNDS rounds; prints median / min microseconds."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from video_prediction_amd import kernels as K, lib  # noqa: E402

DEV = 'cuda:0'
SHAPES = [  # name, mode, N, H, W, Cx, Cy, k
    ('lstm_h0', 'fprop', 32, 32, 32, 72, 128, 5),
    ('lstm_h1', 'fprop', 32, 16, 16, 136, 256, 5),
    ('lstm_h2', 'fprop', 32, 8, 8, 264, 512, 5),
    ('lstm_h0', 'dgrad', 32, 32, 32, 72, 128, 5),
    ('lstm_h1', 'dgrad', 32, 16, 16, 136, 256, 5),
    ('lstm_h2', 'dgrad', 32, 8, 8, 264, 512, 5),
    ('head3x3', 'fprop', 32, 64, 64, 32, 32, 3),
    ('head3x3', 'dgrad', 32, 64, 64, 32, 32, 3),
    ('masks_out', 'fprop', 32, 64, 64, 56, 8, 3),
    ('masks_out', 'dgrad', 32, 64, 64, 56, 8, 3),
    ('dec64', 'fprop', 32, 64, 64, 32, 64, 3), ('dec64', 'dgrad', 32, 64, 64, 32, 64, 3),        # the merged 3x3 heads on the last decoder layer
    # the ConvLSTM gate convolutions of bench.py's other workloads (tests/tools/pmc_cell_report.py PMC_SET=c4 / c5)
    ('c4_h0', 'fprop', 32, 32, 32, 96, 128, 5), ('c4_h1', 'fprop', 32, 16, 16, 160, 256, 5), ('c4_h2', 'fprop', 32, 8, 8, 288, 512, 5),
    ('c5_h4', 'fprop', 16, 16, 16, 520, 1024, 5), ('c5_h5', 'fprop', 16, 32, 32, 264, 512, 5),
]
IT, ROUNDS = 20, 7
only = set(a for a in sys.argv[1:] if not a.startswith('-'))


def graph_of(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(IT):
            fn()
    g.replay()
    torch.cuda.synchronize()
    return g


def main():
    K.set_conv_precision('bf16')
    res = []
    for name, mname, N, H, W, Cx, Cy, k in SHAPES:
        if only and name not in only and (name + ':' + mname) not in only:
            continue
        mode = lib.CONV_FPROP if mname == 'fprop' else lib.CONV_DGRAD
        x = torch.randn(N, H, W, Cx, device=DEV)
        y = torch.randn(N, H, W, Cy, device=DEV)
        w = torch.randn(k * k * Cx * Cy, device=DEV) * 0.05
        w16 = w.to(torch.bfloat16)
        geom = K.ConvGeom((k, k), (1, 1), (k // 2, k // 2))
        flops = 2.0 * N * H * W * Cx * Cy * k * k
        variants = {}
        for alg in (0x200, 0x600, 0x300, 0x700):
            for t in (0x11, 0x12, 0x21, 0x22):
                for sk in ((1, 2, 4) if H <= 16 else (1,)):
                    tile = alg | t
                    try:
                        variants['%03x/sk%d' % (tile, sk)] = graph_of(lambda: K.conv(mode, geom, x, y, w, tile=tile, w16=w16, splitk=sk))
                    except Exception:
                        pass
        if mname == 'fprop' and Cy % 64 == 0:
            y16 = torch.empty(N, H, W, Cy, device=DEV, dtype=torch.bfloat16)
            st = torch.zeros(N, Cy, 2, device=DEV, dtype=torch.float64)
            x16 = x.to(torch.bfloat16)
            for alg in (0x300, 0x700):
                for t in (0x11, 0x12, 0x21, 0x22):
                    tile = alg | t
                    try:
                        variants['%03x/cell' % tile] = graph_of(lambda: K.conv(mode, geom, x, y16, w, tile=tile, w16=w16, stats=st))
                        variants['%03x/cell+src16' % tile] = graph_of(lambda: K.conv(mode, geom, x16, y16, w, tile=tile, w16=w16, stats=st))
                    except Exception:
                        pass
        times = {v: [] for v in variants}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(ROUNDS):
            for v, g in variants.items():
                e0.record()
                g.replay()
                e1.record()
                e1.synchronize()
                times[v].append(e0.elapsed_time(e1) / IT * 1e3)
        rows = sorted(((statistics.median(t), min(t), v) for v, t in times.items()))
        best = {}
        for med, mn, v in rows:
            fam = ('patch' if v[0] in '26' else 'ring') + ('' if 'cell' not in v else v[v.index('/'):])
            if fam not in best:
                best[fam] = (med, mn, v)
        print('%-9s %-5s N=%d %dx%dx%d->%d k%d  (%.2f GFLOP)' % (name, mname, N, H, W, Cx, Cy, k, flops / 1e9))
        for fam, (med, mn, v) in best.items():
            print('    %-16s best %-16s median %7.1f us  min %7.1f us  %6.1f TF' % (fam, v, med, mn, flops / med / 1e6))
        for med, mn, v in rows[:10]:
            print('        %-16s %7.1f %7.1f' % (v, med, mn))
        res.append(dict(shape=name, mode=mname, best={f: dict(variant=v, median_us=med, min_us=mn) for f, (med, mn, v) in best.items()}))
        sys.stdout.flush()
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'ring_ab.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
