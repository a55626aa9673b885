"""Micro-benchmark of the implicit-GEMM conv kernel on the ConvLSTM layer shapes (TFLOP/s per mode/tile)."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_prediction_amd import kernels as K, lib

DEV = 'cuda:0'
SHAPES = [  # name, N, H, W, Cx, Cy, k
    ('lstm_h0', 32, 32, 32, 72, 128, 5),
    ('lstm_h1', 32, 16, 16, 136, 256, 5),
    ('lstm_h2', 32, 8, 8, 264, 512, 5),
    ('head3x3', 32, 64, 64, 32, 32, 3),
]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    res = []
    K.set_conv_precision(os.environ.get('PREC', 'f32'))
    for name, N, H, W, Cx, Cy, k in SHAPES:
        x = torch.randn(N, H, W, Cx, device=DEV)
        y = torch.randn(N, H, W, Cy, device=DEV)
        w = torch.randn(k * k * Cx * Cy, device=DEV) * 0.05
        geom = K.ConvGeom((k, k), (1, 1), (k // 2, k // 2))
        flops = 2.0 * N * H * W * Cx * Cy * k * k
        for mode, mname in ((lib.CONV_FPROP, 'fprop'), (lib.CONV_DGRAD, 'dgrad'), (lib.CONV_WGRAD, 'wgrad')):
            bf = os.environ.get('PREC', 'f32') == 'bf16' and mode != lib.CONV_WGRAD
            w16 = w.to(torch.bfloat16) if bf else None
            tiles = (0, 0x122, 0x121, 0x112, 0x111, 0x222, 0x221, 0x212, 0x211) if bf else (0, 0x22, 0x21, 0x12, 0x11)
            for tile in tiles:
                try:
                    t = timeit(lambda: K.conv(mode, geom, x, y, w, tile=tile, w16=w16))
                except Exception as ex:
                    print(name, mname, hex(tile), 'ERR', ex)
                    continue
                r = dict(shape=name, mode=mname, tile=hex(tile), us=t * 1e6, tflops=flops / t / 1e12)
                res.append(r)
                print('%-8s %-6s tile=%-5s %8.1f us  %6.1f TFLOP/s' % (name, mname, hex(tile), t * 1e6, flops / t / 1e12), flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'conv_micro.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
