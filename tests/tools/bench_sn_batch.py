"""Batched spectral norm of one video discriminator's eight matrices (ops.py:1020-1049): forward (power iteration + sigma) and backward,
back to back in a captured graph.   python tests/tools/bench_sn_batch.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_prediction_amd import kernels as K

SHAPES = [(81, 32), (2048, 64), (1728, 64), (4096, 128), (3456, 128), (8192, 256), (6912, 256), (65536, 1)]      # [K, C] of sn_conv0_0 .. sn_fc4


def main():
    dev = 'cuda'
    fwd, bwd = [], []
    for k, c in SHAPES:
        W = torch.randn(k, c, device=dev) * 0.05
        u = torch.randn(c, device=dev)
        ws = torch.zeros(K.sn_ws_size(k, c), device=dev)
        G = torch.randn(k, c, device=dev)
        dW = torch.zeros(k, c, device=dev)
        fwd.append({'W': W, 'u': u, 'ws': ws, 'u_new': torch.empty_like(u)})
        bwd.append({'W': W, 'u': u, 'ws': ws, 'G': G, 'dW': dW, 'beta': 1})
    for name, fn in (('fwd', lambda: K.sn_fwd_batch(fwd)), ('bwd', lambda: K.sn_bwd_batch(bwd))):
        K.sn_fwd_batch(fwd)
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print('sn %s batch of %d matrices (%.1f MB): %.1f us per call' % (name, len(SHAPES), sum(k * c for k, c in SHAPES) * 4 / 1e6,
                                                                         e0.elapsed_time(e1) / 100 * 1e3))


if __name__ == '__main__':
    main()
