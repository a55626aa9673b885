"""Microbenchmark of the ConvLSTM gate block (everything after the gate convolution) at the bench shapes (N = 32): the one-launch
kernels (option lstm_fused = 1) against the three-pass kernels (0), production configuration = bf16 gates + IN(4F) statistics from
the conv epilogue, fp32 or bf16 h destinations / gate gradient.  HIP events around ITERS back-to-back calls, cold-ish caches (a
64 MB fill between repetitions is NOT done: the step itself runs these kernels right behind the conv that produced the gates)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from video_prediction_amd import kernels as K, lib  # noqa: E402

ITERS = int(os.environ.get('ITERS', 50))
N = int(os.environ.get('N', 32))


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(ITERS):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / ITERS)
    return best


def main():
    dev = 'cuda:0'
    g = torch.Generator(device=dev).manual_seed(0)
    for (H, W, F, nz) in ((32, 32, 32, 8), (16, 16, 64, 8), (8, 8, 128, 8)):
        for act16 in (False, True):
            adt = torch.bfloat16 if act16 else torch.float32
            gates = (torch.randn(N, H, W, 4 * F, generator=g, device=dev) * 1.5).to(torch.bfloat16)
            gf = gates.float()
            ws, s1 = K.lstm_stats_ws(torch.device(dev), N, F)
            ws.zero_()
            s1[..., 0] = gf.sum(dim=(1, 2))
            s1[..., 1] = (gf * gf).sum(dim=(1, 2))
            c = torch.randn(N, H, W, F, generator=g, device=dev)
            p = [torch.rand(4 * F, device=dev) + 0.5, torch.randn(4 * F, device=dev) * 0.1, torch.rand(F, device=dev) + 0.5, torch.randn(F, device=dev) * 0.1]
            c_new = torch.empty(N, H, W, F, device=dev)
            a_next = torch.zeros(N, H, W, 2 * F + nz, device=dev, dtype=adt)          # next step's [x | z | h]
            nxt = torch.zeros(N, H, W, F + nz, device=dev)                             # next layer's input
            stats = [torch.empty(N, 4 * F, device=dev), torch.empty(N, 4 * F, device=dev), torch.empty(N, F, device=dev), torch.empty(N, F, device=dev)]
            lws = torch.empty(K.lstm_ws_floats(N, H * W, F), device=dev)
            hs = [nxt[..., :F], a_next[..., F + nz:]]
            dh = [torch.randn(N, H, W, F + nz, generator=g, device=dev)[..., :F], torch.randn(N, H, W, 2 * F + nz, generator=g, device=dev)[..., F + nz:]]
            dcn = torch.randn(N, H, W, F, generator=g, device=dev)
            dgates = torch.empty(N, H, W, 4 * F, device=dev, dtype=adt)
            raw = torch.empty(N, H, W, 4 * F, device=dev) if act16 else None
            dcp = torch.empty(N, H, W, F, device=dev)
            dpar = [torch.zeros(4 * F, device=dev), torch.zeros(4 * F, device=dev), torch.zeros(F, device=dev), torch.zeros(F, device=dev)]

            def fwd():
                K.convlstm_gates_fwd(gates, c, p[0], p[1], p[2], p[3], c_new, hs, stats, ws=lws, stats1=ws)

            def bwd():
                K.convlstm_gates_bwd(gates, c, p[0], p[1], p[2], p[3], stats, dh, dcn, dgates, dcp, dpar, ws=lws, dgates_raw=raw)
            res = {}
            outs = {}
            for fused in (0, 1):
                lib.set_option('lstm_fused', fused)
                res[fused] = (timeit(fwd), timeit(bwd))
                ws[N * 4 * F * 2:].zero_()                   # the three-pass kernels accumulate their second reduction here
                fwd(); bwd()
                torch.cuda.synchronize()
                outs[fused] = (c_new.clone(), a_next.float().clone(), dgates.float().clone(), dcp.clone())
            lib.set_option('lstm_fused', 1)
            err = [float((a - b).abs().max() / max(float(b.abs().max()), 1e-30)) for a, b in zip(outs[1], outs[0])]
            print('%dx%dx%d %s : fwd 3-pass %.1f us fused %.1f us | bwd 3-pass %.1f us fused %.1f us | rel diff c %.1e h %.1e dgates %.1e dc %.1e'
                  % (H, W, F, 'bf16-act' if act16 else 'fp32-act', res[0][0], res[1][0], res[0][1], res[1][1], err[0], err[1], err[2], err[3]))


if __name__ == '__main__':
    main()
