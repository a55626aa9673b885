"""Micro-benchmark of the HBM-bound per-timestep kernels at the BAIR bench shapes (N = 32 images of 64x64x3): CDNA apply forward /
backward, mask composite backward, the CDNA dense head, the feature-matching cosine distance.  Run once as is and once with
SAVP_CDNA_LEGACY=1 SAVP_DENSE_LEGACY=1 for the A/B (the switches are read once per process).  Prints median microseconds with the
algorithmic HBM bytes of the op and the bandwidth they imply."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from video_prediction_amd import kernels as K  # noqa: E402

DEV = 'cuda:0'
N, H, W, C, KK, M = 32, 64, 64, 3, 4, 7
big = torch.zeros(128 << 20, dtype=torch.int32, device=DEV)
FLUSH = os.environ.get('FLUSH', 'read')      # read: evict with clean lines (a 512 MB reduction); write: with dirty lines; none: hot


def t(fn, n=30, flush=True):
    ts = []
    for _ in range(n):
        if flush and FLUSH == 'read':
            big.sum()              # evict L2 / Infinity Cache: inside a train step these operands arrive cold
        elif flush and FLUSH == 'write':
            big.zero_()            # ... and leave the caches full of dirty lines (pessimistic: every miss first writes a line back)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def report(name, us, nbytes):
    print('%-28s %8.1f us   %6.1f MB  -> %6.2f TB/s' % (name, us, nbytes / 1e6, nbytes / us / 1e6))


in0 = torch.rand(N, H, W, 16, device=DEV)
maskin = torch.rand(N, H, W, 56, device=DEV)
dmaskin = torch.rand(N, H, W, 56, device=DEV)
raw = torch.randn(N, 25 * KK, device=DEV) * 0.1
kern = torch.empty(N, 25, KK, device=DEV)
K.cdna_kernels_fwd(raw, kern, 5, 5, KK)
img = in0[..., 0:C]
tslot = maskin[..., 32:32 + KK * C]
dslot = dmaskin[..., 32:32 + KK * C]
dimg = torch.empty(N, H, W, C, device=DEV)
dkern = torch.empty(N, 25, KK, device=DEV, dtype=torch.float64)
px = N * H * W
report('cdna_apply_fwd', t(lambda: K.cdna_apply_fwd(img, kern, tslot, 5, 5, KK)), px * (3 + 12) * 4)
report('cdna_apply_bwd (img+kern)', t(lambda: K.cdna_apply_bwd(img, kern, dslot, dimg, dkern, 5, 5, KK)), px * (3 + 12 + 3) * 4)
logits = torch.randn(N, H, W, 8, device=DEV)
dgen = torch.randn(N, H, W, C, device=DEV)
dlogits = torch.empty(N, H, W, 8, device=DEV)
report('composite_bwd', t(lambda: K.composite_bwd(logits, maskin[..., 32:32 + M * C], dgen, dlogits, dmaskin, 32, M=M)),
       px * (8 + 21 + 3 + 8 + 56) * 4)
x = torch.randn(N, 8192, device=DEV)
Wd = torch.randn(8192, 100, device=DEV)
b = torch.randn(100, device=DEV)
o = torch.empty(N, 100, device=DEV)
report('dense 32x8192x100', t(lambda: K.dense_fwd(x, Wd, b, o)), (8192 * 100 + N * 8192) * 4)
xd = torch.randn(16, 65536, device=DEV)
Wl = torch.randn(65536, 1, device=DEV)
o1 = torch.empty(16, 1, device=DEV)
report('dense 16x65536x1', t(lambda: K.dense_fwd(xd, Wl, None, o1)), (65536 + 16 * 65536) * 4)
for (P, Cc) in ((16 * 10 * 64 * 64, 32), (16 * 9 * 32 * 32, 64), (16 * 4 * 8 * 8, 256)):
    f0 = torch.randn(P, Cc, device=DEV)
    f1 = torch.randn(P, Cc, device=DEV)
    df = torch.empty(P, Cc, device=DEV)
    lo = torch.zeros(1, device=DEV)
    report('cosine P=%d C=%d' % (P, Cc), t(lambda: K.cosine_distance(f0, f1, 10.0, lo, df)), P * Cc * 3 * 4)
