"""Developer aid: in-kernel cycle stamps of the one-launch ConvLSTM gate kernels (csrc/norm_lstm.hip built with -DSAVP_LSTM_STAMPS
into libsavp_hip_dev.so: tests/tools/build_dev_lib.sh).  Prints, per shape and direction, the phase boundaries of three workgroups
(first / middle / last block) in cycles since the earliest stamp, plus the dispatch duration from HIP events.
  SAVP_LIB=video_prediction_amd/libsavp_hip_dev.so python tests/tools/lstm_stamps.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from video_prediction_amd import kernels as K, lib  # noqa: E402

N = int(os.environ.get('N', 32))
FWD = ('entry', 'loads issued', 'loads landed', 'c_pre done', 'sum 1 done', 'sum 2 done', 'stores issued', 'stores landed')
BWD = ('entry', 'loads issued', 'loads landed', 'phase 1 done', 'sum(8) done', 'phase 2 done', 'sum(32) done', 'stores landed')


def stamps():
    buf = (ctypes.c_ulonglong * 24)()
    raw = ctypes.CDLL(lib.LIB_PATH)
    assert raw.savp_debug_lstm_times(buf) == 0
    return [[buf[w * 8 + i] for i in range(8)] for w in range(3)]


def main():
    dev = 'cuda:0'
    g = torch.Generator(device=dev).manual_seed(0)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    for (H, W, F, nz) in ((32, 32, 32, 8), (16, 16, 64, 8), (8, 8, 128, 8)):
        gates = (torch.randn(N, H, W, 4 * F, generator=g, device=dev) * 1.5).to(torch.bfloat16)
        gf = gates.float()
        ws, s1 = K.lstm_stats_ws(torch.device(dev), N, F)
        ws.zero_()
        s1[..., 0] = gf.sum(dim=(1, 2))
        s1[..., 1] = (gf * gf).sum(dim=(1, 2))
        c = torch.randn(N, H, W, F, generator=g, device=dev)
        p = [torch.rand(4 * F, device=dev) + 0.5, torch.randn(4 * F, device=dev) * 0.1, torch.rand(F, device=dev) + 0.5, torch.randn(F, device=dev) * 0.1]
        c_new = torch.empty(N, H, W, F, device=dev)
        a_next = torch.zeros(N, H, W, 2 * F + nz, device=dev)
        nxt = torch.zeros(N, H, W, F + nz, device=dev)
        stats = [torch.empty(N, 4 * F, device=dev), torch.empty(N, 4 * F, device=dev), torch.empty(N, F, device=dev), torch.empty(N, F, device=dev)]
        hs = [nxt[..., :F], a_next[..., F + nz:]]
        dh = [torch.randn(N, H, W, F + nz, generator=g, device=dev)[..., :F], torch.randn(N, H, W, 2 * F + nz, generator=g, device=dev)[..., F + nz:]]
        dcn = torch.randn(N, H, W, F, generator=g, device=dev)
        dgates = torch.empty(N, H, W, 4 * F, device=dev)
        dcp = torch.empty(N, H, W, F, device=dev)
        dpar = [torch.zeros(4 * F, device=dev), torch.zeros(4 * F, device=dev), torch.zeros(F, device=dev), torch.zeros(F, device=dev)]

        def fwd():
            K.convlstm_gates_fwd(gates, c, p[0], p[1], p[2], p[3], c_new, hs, stats, stats1=ws)

        def bwd():
            K.convlstm_gates_bwd(gates, c, p[0], p[1], p[2], p[3], stats, dh, dcn, dgates, dcp, dpar)
        for name, fn, labels in (('fwd', fwd, FWD), ('bwd', bwd, BWD)):
            for cold in (True, False):
                fn()
                torch.cuda.synchronize()
                if cold:
                    flush.fill_(1)                    # push the operands out of L2 / MALL (512 MB > 256 MB infinity cache)
                    torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                t = stamps()
                base = min(w[0] for w in t)
                print('%dx%dx%d %s %s: dispatch+gaps %.1f us' % (H, W, F, name, 'cold' if cold else 'warm', e0.elapsed_time(e1) * 1e3))
                for wi, w in enumerate(t):
                    print('   wg %-6s ' % ('first', 'middle', 'last')[wi] + '  '.join('%s +%d' % (labels[i], w[i] - base) for i in range(8)))


if __name__ == '__main__':
    main()
