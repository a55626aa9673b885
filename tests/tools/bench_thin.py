"""The discriminators' RGB first layer (10x64x64x3 -> 32, 3x3x3, N = 32 clips) FPROP (bias + LeakyReLU) and WGRAD (+ bias gradient):
median microseconds and the HBM bandwidth the algorithmic bytes imply.  Run with SAVP_THIN=0 for the general kernels (A/B)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from video_prediction_amd import kernels as K, lib  # noqa: E402

N, D, H, W, Cx, Cy = int(os.environ.get('NCLIP', 32)), 10, 64, 64, 3, 32
big = torch.zeros(128 << 20, dtype=torch.int32, device='cuda')


def t(fn, n=15):
    ts = []
    for _ in range(n):
        big.sum()                                   # operands arrive cold, as inside a train step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


K.set_conv_precision('bf16')
geom = K.ConvGeom((3, 3, 3), (1, 1, 1), (1, 1, 1))
x = torch.rand(N, D, H, W, Cx, device='cuda')
y = torch.empty(N, D, H, W, Cy, device='cuda')
dy = torch.randn(N, D, H, W, Cy, device='cuda')
w = torch.randn(27, Cx, Cy, device='cuda') * 0.1
wt, wd = torch.empty(Cy, 27 * Cx, device='cuda'), torch.empty(Cx, 27 * Cy, device='cuda')
K.pack_weights(w, wt, wd)
b = torch.randn(Cy, device='cuda')
dw, db = torch.zeros(3, 3, 3, Cx, Cy, device='cuda'), torch.zeros(Cy, device='cuda')
act_bytes = (x.numel() + y.numel()) * 4
us = t(lambda: K.conv(lib.CONV_FPROP, geom, x, y, wt, bias=b, act=lib.ACT_LRELU, alpha=0.2))
print('L0 fprop  %8.1f us  %6.1f MB -> %5.2f TB/s' % (us, act_bytes / 1e6, act_bytes / us / 1e6))
us = t(lambda: K.conv(lib.CONV_WGRAD, geom, x, dy, dw, bias=db))
print('L0 wgrad  %8.1f us  %6.1f MB -> %5.2f TB/s' % (us, act_bytes / 1e6, act_bytes / us / 1e6))
