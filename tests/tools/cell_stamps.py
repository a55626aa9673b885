"""Developer tool: the ConvLSTM cell forward as one launch (option gate_cell = 1) against the two-launch path of the same entry, back to back at
N = 32, and -- with a -DSAVP_GATE_STAMPS library as SAVP_LIB -- the one-launch kernel's phase stamps (phase 'bf16 store' + 'statistics' = the cell
epilogue behind the K-slice reduce)."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from video_prediction_amd import kernels as K, lib  # noqa: E402

L = lib.get()
geom = K.ConvGeom((5, 5), (1, 1), (2, 2))
N = int(os.environ.get('GATE_N', '32'))
names = ['dma issued', 'table', 'patch landed', 'main loop', 'k-slice reduce', 'cell epilogue', '-']
for (S, Cx, F) in [(16, 136, 64), (8, 264, 128)]:
    x = torch.randn(N, S, S, Cx, device='cuda').to(torch.bfloat16)
    w = torch.randn(5, 5, Cx, 4 * F, device='cuda') * 0.05
    wt = w.reshape(25 * Cx, 4 * F).t().contiguous()
    w16 = wt.to(torch.bfloat16)
    n_el = K.gate_weights_elems(25, Cx, 4 * F)
    frag, frag_il = torch.empty(n_el, device='cuda', dtype=torch.bfloat16), torch.empty(n_el, device='cuda', dtype=torch.bfloat16)
    K.pack_gate_weights(w, frag)
    K.pack_gate_weights(w, frag_il, interleave=True)
    y = torch.empty(N, S, S, 4 * F, device='cuda', dtype=torch.bfloat16)
    c = torch.randn(N, S, S, F, device='cuda')
    c_new = torch.empty_like(c)
    p = [torch.ones(4 * F, device='cuda'), torch.zeros(4 * F, device='cuda'), torch.ones(F, device='cuda'), torch.zeros(F, device='cuda')]
    wide = torch.empty(N, S, S, 3 * F + 8, device='cuda', dtype=torch.bfloat16)
    hs = [wide[..., 8:8 + F], wide[..., 8 + F:8 + 2 * F]]
    stats = [torch.empty(N, 4 * F, device='cuda'), torch.empty(N, 4 * F, device='cuda'), torch.empty(N, F, device='cuda'), torch.empty(N, F, device='cuda')]
    lws = torch.empty(K.lstm_ws_floats(N, S * S, F), device='cuda')

    def run():
        ws, s1 = K.lstm_stats_ws(torch.device('cuda:0'), N, F)
        ca = K.conv(lib.CONV_FPROP, geom, x, y, wt, precision=1, w16=w16, stats=s1, w_frag=frag, w_frag_il=frag_il, defer=True)
        ga = K.convlstm_gates_fwd(y, c, p[0], p[1], p[2], p[3], c_new, hs, stats, ws=lws, stats1=ws, defer=True)
        K.convlstm_cell_fwd(ca, ga)
    for opt in (1, 0):
        lib.set_option('gate_cell', opt)
        for _ in range(3):
            run()
        K.zero_arena(torch.device('cuda:0')).reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        e1.synchronize()
        K.zero_arena(torch.device('cuda:0')).reset()
        print(json.dumps({'shape': [S, Cx, F], 'cell': 'one launch' if opt else 'two launches', 'us_per_cell_back_to_back': round(e0.elapsed_time(e1) / 20 * 1e3, 2)}), flush=True)
    lib.set_option('gate_cell', 1)
    if hasattr(L, 'savp_debug_gate_times'):
        L.savp_debug_gate_block(100)
        run()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 32)()
        L.savp_debug_gate_times(buf)
        for wv in (0, 3):
            t = [buf[wv * 8 + i] for i in range(8)]
            print(json.dumps({'shape': [S, Cx, F], 'wave': wv, 'total': t[7] - t[0], 'phases': {names[i]: t[i + 1] - t[i] for i in (0, 1, 2, 3, 4)}, 'cell: norms + gates': t[6] - t[5], 'cell: stores': t[7] - t[6]}), flush=True)
