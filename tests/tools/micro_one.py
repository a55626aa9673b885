"""Time one conv configuration: SHAPE=lstm_h0 MODE=fprop TILE=0x222 python tests/tools/micro_one.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_prediction_amd import kernels as K, lib
from tests.tools.bench_conv_micro import SHAPES

def main():
    K.set_conv_precision('bf16')
    sh = dict((s[0], s) for s in SHAPES)[os.environ.get('SHAPE', 'lstm_h0')]
    name, N, H, W, Cx, Cy, k = sh
    mode = {'fprop': lib.CONV_FPROP, 'dgrad': lib.CONV_DGRAD, 'wgrad': lib.CONV_WGRAD}[os.environ.get('MODE', 'fprop')]
    N = int(os.environ.get('NIMG', N))
    tile = int(os.environ.get('TILE', '0x222'), 16)
    sk = int(os.environ.get('SK', '0'))
    x = torch.randn(N, H, W, Cx, device='cuda'); y = torch.randn(N, H, W, Cy, device='cuda')
    w = torch.randn(k * k * Cx * Cy, device='cuda') * 0.05
    w16 = w.to(torch.bfloat16) if mode != lib.CONV_WGRAD else None
    if mode == lib.CONV_WGRAD:
        w = w.view(k, k, Cx, Cy)
    geom = K.ConvGeom((k, k), (1, 1), (k // 2, k // 2))
    fn = lambda: K.conv(mode, geom, x, y, w, tile=tile, w16=w16, splitk=sk)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    # launch back-to-back inside a graph-free loop but measure GPU time with events around many launches
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = int(os.environ.get('ITERS', 50))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(it): fn()
    g.replay(); torch.cuda.synchronize()
    s.record()
    g.replay(); g.replay()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / (2 * it) * 1e-3
    flops = 2.0 * N * H * W * Cx * Cy * k * k
    print('%s %s tile=%s dbg=%s: %.1f us  %.1f TF' % (name, os.environ.get('MODE', 'fprop'), hex(tile), os.environ.get('SAVP_ABLATE', '0'), t * 1e6, flops / t / 1e12))

if __name__ == '__main__':
    main()
