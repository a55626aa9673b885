"""Weight-gradient micro-benchmark at the step's real reduction length (N = (T-1) * 2B = 928 images): the generator's WGRAD
launches of one train step.  SAVP_LIB=<other build> runs the same script on another build for an A/B."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from video_prediction_amd import kernels as K, lib  # noqa: E402

SHAPES = [  # name, H, W, Cx, Cy, k
    ('lstm_h0', 32, 32, 72, 128, 5), ('lstm_h1', 16, 16, 136, 256, 5), ('lstm_h2', 8, 8, 264, 512, 5),
    ('head3x3', 64, 64, 32, 32, 3), ('masks_out', 64, 64, 56, 8, 3),
]
N = int(os.environ.get('NIMG', 928))
def main():
    K.set_conv_precision('bf16')
    only = set(sys.argv[1:])
    for name, H, W, Cx, Cy, k in SHAPES:
        if only and name not in only:
            continue
        x = torch.randn(N, H, W, Cx, device='cuda')
        y = torch.randn(N, H, W, Cy, device='cuda')
        if os.environ.get('BF16', '0') == '1':              # the step's operands: both tensors stored in bf16
            x = x.bfloat16(); y = y.bfloat16()
        w = torch.zeros(k, k, Cx, Cy, device='cuda')
        b = torch.zeros(Cy, device='cuda')
        geom = K.ConvGeom((k, k), (1, 1), (k // 2, k // 2))
        if os.environ.get('NOBIAS', '0') == '1':            # the ConvLSTM gate convolutions carry no bias (the instance norm follows)
            b = None
        fn = lambda: K.conv(lib.CONV_WGRAD, geom, x, y, w, bias=b)
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        flops = 2.0 * N * H * W * Cx * Cy * k * k
        print('%-10s N=%d %dx%dx%d->%d k%d: median %8.1f us  min %8.1f us  %6.1f TF' % (name, N, H, W, Cx, Cy, k, ts[3], ts[0], flops / ts[3] / 1e6))


if __name__ == '__main__':
    main()
