"""Upper bound of leaving the tiled-z channels out of the ConvLSTM gate convolution's DGRAD: best (tile, split-K) and time of the
DGRAD with all f+nz+f input channels vs only the f+f channels whose gradient is needed per pixel (bf16 gate gradient, N = 32)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_prediction_amd import kernels as K, lib
K.set_conv_precision('bf16')
K.AUTOTUNE['enabled'] = True
IT = 20
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(IT): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / IT * 1e3)
    return best
for (H, f, nz) in ((32, 32, 8), (16, 64, 8), (8, 128, 8)):
    N, k = 32, 5
    Cfull, Cy = f + nz + f, 4 * f
    dg = torch.randn(N, H, H, Cy, device='cuda').to(torch.bfloat16)
    geom = K.ConvGeom((k, k), (1, 1), (2, 2))
    out = torch.zeros(N, H, H, Cfull, device='cuda')
    res = []
    for Cx in (Cfull, 2 * f):
        w = torch.randn(k * k * Cx * Cy, device='cuda') * 0.05
        w16 = w.to(torch.bfloat16)
        xv = out[..., :Cx]
        n0 = len(K.AUTOTUNE['log'])
        fn = lambda: K.conv(lib.CONV_DGRAD, geom, xv, dg, w, w16=w16)
        fn()
        cfg = K.AUTOTUNE['log'][-1][1] if len(K.AUTOTUNE['log']) > n0 else None
        res.append((Cx, cfg, timed(fn)))
    print('%dx%d f=%d: ' % (H, H, f) + '   '.join('Cx=%d tile=%s sk=%s %.1f us' % (c, hex(cfg[0]) if cfg else '?', cfg[1] if cfg else '?', t) for c, cfg, t in res), flush=True)
