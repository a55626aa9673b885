import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_prediction_amd import kernels as K
x = torch.randn(32, 8192, device='cuda'); W = torch.randn(8192, 100, device='cuda'); b = torch.randn(100, device='cuda')
o = torch.empty(32, 100, device='cuda')
big = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
def t(fn, n=20, flush=False):
    ts = []
    for _ in range(n):
        if flush: big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
fn = lambda: K.dense_fwd(x, W, b, o)
fn(); torch.cuda.synchronize()
print('hot  %.1f us' % t(fn)); print('cold %.1f us' % t(fn, flush=True))
