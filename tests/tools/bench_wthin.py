"""The two wide -> thin 3x3 convolutions of the generator (scratch-image head 32 -> 4 + sigmoid into a 56-wide buffer, mask convolution 56 -> 8)
at the step's shapes, cold-ish (a 256 MB buffer is swept between launches): wthin_fprop_kernel (tile 0) against the general kernels
(option thin = 0, and the tiles the in-step tuner had picked).  SAVP_LIB=<other build> for an A/B."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from video_prediction_amd import kernels as K, lib  # noqa: E402

K.set_conv_precision('bf16')
N = 32
CASES = [('scratch_head', 64, 64, 32, 4, lib.ACT_SIGMOID, 56, 0x2221), ('masks', 64, 64, 56, 8, 0, 8, 0x1621)]
flush = torch.empty(64 * 1024 * 1024, device='cuda')
for name, H, W, Cx, Cy, act, ywidth, old_tile in CASES:
    x = torch.randn(N, H, W, Cx, device='cuda')
    big = torch.zeros(N, H, W, ywidth, device='cuda')
    y = big[..., ywidth - Cy:]
    w = torch.randn(3, 3, Cx, Cy, device='cuda') * 0.1
    wt = w.reshape(-1, Cy).t().contiguous()
    w16 = wt.to(torch.bfloat16)
    b = torch.randn(Cy, device='cuda')
    geom = K.ConvGeom((3, 3), (1, 1), (1, 1))
    for label, thin, tile in (('wthin', 1, 0), ('general', 0, old_tile)):
        lib.set_option('thin', thin)
        fn = lambda: K.conv(lib.CONV_FPROP, geom, x, y, wt, bias=b, act=act, alpha=0.2, precision=1, w16=w16, tile=tile, splitk=1 if tile else 0)
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(9):
            flush.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print('%-12s %-8s median %6.1f us  min %6.1f us' % (name, label, ts[4], ts[0]))
    lib.set_option('thin', 1)
# the mask convolution's data gradient (56 <- 8, accumulating)
dy = torch.randn(N, 64, 64, 8, device='cuda'); dx = torch.zeros(N, 64, 64, 56, device='cuda')
w = torch.randn(3, 3, 56, 8, device='cuda') * 0.1
wd = w.reshape(-1, 56, 8).permute(1, 0, 2).reshape(56, -1).contiguous(); wd16 = wd.to(torch.bfloat16)
geom = K.ConvGeom((3, 3), (1, 1), (1, 1))
for label, thin, tile in (('thin8', 1, 0), ('general', 0, 0x321)):
    lib.set_option('thin', thin)
    fn = lambda: K.conv(lib.CONV_DGRAD, geom, dx, dy, wd, beta=1, precision=1, w16=wd16, tile=tile, splitk=1 if tile else 0)
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        flush.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print('%-12s %-8s median %6.1f us  min %6.1f us' % ('masks_dgrad', label, ts[4], ts[0]))
lib.set_option('thin', 1)
