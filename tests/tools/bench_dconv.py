"""The video discriminator's seven 3-D convolutions (networks.py:72-108, N = 2B = 32 clips of 10 frames) on every conv algorithm /
tile candidate: FPROP (bias + LeakyReLU epilogue) and DGRAD.  Prints the best of each algorithm family per problem."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from video_prediction_amd import kernels as K, lib  # noqa: E402

N = int(os.environ.get('NCLIP', 32))
LAYERS = [  # name, D, H, W, Cx, Cy, k, stride
    ('L0', 10, 64, 64, 3, 32, 3, (1, 1, 1)), ('L1', 10, 64, 64, 32, 64, 4, (1, 2, 2)), ('L2', 9, 32, 32, 64, 64, 3, (1, 1, 1)),
    ('L3', 9, 32, 32, 64, 128, 4, (1, 2, 2)), ('L4', 8, 16, 16, 128, 128, 3, (1, 1, 1)), ('L5', 8, 16, 16, 128, 256, 4, (2, 2, 2)),
    ('L6', 4, 8, 8, 256, 256, 3, (1, 1, 1)),
]
FAM = {0x100: 'generic', 0x200: 'patch4', 0x600: 'patch8', 0x300: 'ring4', 0x700: 'ring8'}


def main():
    K.set_conv_precision('bf16')
    only = set(sys.argv[1:])
    fn = lib.get().savp_conv
    st = lib.stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, D, H, W, Cx, Cy, k, s in LAYERS:
        if only and name not in only:
            continue
        geom = K.ConvGeom((k, k, k), s, (1, 1, 1))
        Do, Ho, Wo = [(i + 2 - k) // st_ + 1 for i, st_ in zip((D, H, W), s)]
        x = torch.randn(N, D, H, W, Cx, device='cuda')
        y = torch.randn(N, Do, Ho, Wo, Cy, device='cuda')
        w = torch.randn(k * k * k * Cx * Cy, device='cuda') * 0.05
        wt = torch.empty(Cy, k * k * k * Cx, device='cuda'); wd = torch.empty(Cx, k * k * k * Cy, device='cuda')
        wt16 = torch.empty_like(wt, dtype=torch.bfloat16); wd16 = torch.empty_like(wd, dtype=torch.bfloat16)
        K.pack_weights(w.view(k * k * k, Cx, Cy), wt, wd, wt16=wt16, wd16=wd16)
        bias = torch.randn(Cy, device='cuda')
        flops = 2.0 * N * Do * Ho * Wo * Cx * Cy * k ** 3
        for mode, mname in ((lib.CONV_FPROP, 'fprop'), (lib.CONV_DGRAD, 'dgrad')):
            if mode == lib.CONV_FPROP:
                a = K._fill_conv_args(mode, geom, x, y, wt, bias, 0, lib.ACT_LRELU, 0.1, None, 0, 0, None, wt16)
            else:
                a = K._fill_conv_args(mode, geom, x, y, wd, None, 0, 0, 0.0, None, 0, 0, None, wd16)
            best = {}
            for alg in FAM:
                for t in (0x22, 0x21, 0x12, 0x11):
                    for lim in ((0, 0x1000, 0x2000) if alg in (0x200, 0x600) else (0,)):
                        for sk in (1, 2, 4):
                            if mode == lib.CONV_FPROP and sk > 1:
                                continue
                            a.tile, a.splitk = alg | t | lim, sk
                            if fn(st, ctypes.byref(a)) != 0:
                                continue
                            tm = 1e30
                            for _ in range(2):
                                e0.record()
                                for _ in range(3):
                                    fn(st, ctypes.byref(a))
                                e1.record(); e1.synchronize()
                                tm = min(tm, e0.elapsed_time(e1) / 3 * 1e3)
                            f = FAM[alg]
                            if f not in best or tm < best[f][0]:
                                best[f] = (tm, '%03x/sk%d' % (a.tile, sk))
            a.tile, a.splitk = 0, 0
            fn(st, ctypes.byref(a)); torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                fn(st, ctypes.byref(a))
            e1.record(); e1.synchronize()
            auto = e0.elapsed_time(e1) / 3 * 1e3
            print('%s %-5s %dx%dx%dx%d->%d k%d s%s  %.1f GFLOP | auto(untuned) %.0f us | ' % (name, mname, D, H, W, Cx, Cy, k, s, flops / 1e9, auto) +
                  '  '.join('%s %.0f us (%s, %.0f TF)' % (f, v[0], v[1], flops / v[0] / 1e6) for f, v in sorted(best.items(), key=lambda kv: kv[1][0])), flush=True)


if __name__ == '__main__':
    main()
