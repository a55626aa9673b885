"""Video-discriminator convolutions (networks.py:72-108 at the c2 shapes, N = 32 clips): every instantiation the tuner knows, timed back to back,
with fp32 and with bf16 source activations -- what would the bf16 storage path buy?   python tests/tools/bench_disc_convs.py [fprop|dgrad]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_prediction_amd import kernels as K, lib

# name, (D, H, W, Cx), (Do, Ho, Wo, Cy), k, stride
LAYERS = [('sn_conv0_1', (10, 64, 64, 32), (9, 32, 32, 64), 4, (1, 2, 2)),
          ('sn_conv1_0', (9, 32, 32, 64), (9, 32, 32, 64), 3, (1, 1, 1)),
          ('sn_conv1_1', (9, 32, 32, 64), (8, 16, 16, 128), 4, (1, 2, 2)),
          ('sn_conv2_0', (8, 16, 16, 128), (8, 16, 16, 128), 3, (1, 1, 1)),
          ('sn_conv2_1', (8, 16, 16, 128), (4, 8, 8, 256), 4, (2, 2, 2)),
          ('sn_conv3_0', (4, 8, 8, 256), (4, 8, 8, 256), 3, (1, 1, 1))]


def main():
    mode_name = sys.argv[1] if len(sys.argv) > 1 else 'fprop'
    mode = {'fprop': lib.CONV_FPROP, 'dgrad': lib.CONV_DGRAD}[mode_name]
    N = int(os.environ.get('NIMG', 32))
    K.set_conv_precision('bf16')
    for name, xs, ys, k, st in LAYERS:
        geom = K.ConvGeom((k, k, k), st, (1, 1, 1))
        taps = k ** 3
        Cx, Cy = xs[-1], ys[-1]
        flops = 2.0 * N * ys[0] * ys[1] * ys[2] * Cy * taps * Cx
        for src16 in (False, True):
            x = torch.randn((N,) + xs, device='cuda')
            y = torch.randn((N,) + ys, device='cuda')
            if mode == lib.CONV_FPROP:
                w = torch.randn(Cy, taps * Cx, device='cuda') * 0.02
                if src16:
                    x = x.to(torch.bfloat16)
                bias = torch.zeros(Cy, device='cuda')
                a = K._fill_conv_args(mode, geom, x, y, w, bias, 0, lib.ACT_LRELU, 0.1, None, 0, 0, None, w.to(torch.bfloat16))
                dst = y
            else:
                w = torch.randn(Cx, taps * Cy, device='cuda') * 0.02
                if src16:
                    y = y.to(torch.bfloat16)
                a = K._fill_conv_args(mode, geom, x, y, w, None, 0, 0, 0.0, None, 0, 0, None, w.to(torch.bfloat16))
                dst = x
            keep = (x, y, w)
            res = K._tune(a, mode, dst, w, return_all=True)
            top = ', '.join('%.1f us (%.0f TF) tile=0x%x sk=%d' % (ms * 1e3, flops / (ms * 1e-3) / 1e12, t, sk) for ms, t, sk in res[:4])
            print('%-11s %s N=%d src=%s: %s' % (name, mode_name, N, 'bf16' if src16 else 'f32 ', top), flush=True)


if __name__ == '__main__':
    main()
