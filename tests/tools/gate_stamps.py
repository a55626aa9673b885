"""Developer tool: cycle stamps of conv_gate_kernel's phases (a library built with SAVP_EXTRA_FLAGS=-DSAVP_GATE_STAMPS, handed over as SAVP_LIB).
Prints, per gate-convolution shape at N = 32 and per wave of one workgroup: prologue issue / table / wait for the patch / main loop / K-slice
reduce / store / statistics, in shader cycles, plus the launch's event time back to back (warm) -- the numbers behind DESIGN.md section 3."""
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
names = ['dma issued', 'table', 'patch landed', 'main loop', 'k-slice reduce', 'bf16 store', 'statistics']
for (S, Cx, F) in [(32, 72, 32), (16, 136, 64), (8, 264, 128)]:
    x = torch.randn(N, S, S, Cx, device='cuda').to(torch.bfloat16)
    w = torch.randn(5, 5, Cx, 4 * F, device='cuda') * 0.05
    wt = w.reshape(25 * Cx, 4 * F).t().contiguous()
    frag = torch.empty(K.gate_weights_elems(25, Cx, 4 * F), device='cuda', dtype=torch.bfloat16)
    K.pack_gate_weights(w, frag)
    y = torch.empty(N, S, S, 4 * F, device='cuda', dtype=torch.bfloat16)
    st = torch.zeros(N, 4 * F, 2, device='cuda', dtype=torch.float64)
    for opt in (1, 2, 0):
        lib.set_option('gate_kernel', 1 if opt else 0)
        lib.set_option('gate_alt', 1 if opt == 2 else 0)
        run = lambda: K.conv(lib.CONV_FPROP, geom, x, y, wt, precision=1, w16=wt.to(torch.bfloat16), stats=st, w_frag=frag)
        w16 = wt.to(torch.bfloat16)
        run = lambda: K.conv(lib.CONV_FPROP, geom, x, y, wt, precision=1, w16=w16, stats=st, w_frag=frag)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        e1.synchronize()
        flops = 2.0 * N * S * S * 4 * F * 25 * Cx
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(json.dumps({'shape': [S, Cx, F], 'kernel': {1: 'gate', 2: 'gate_alt', 0: 'ring'}[opt], 'us_back_to_back': round(us, 2), 'frac_of_2.5PF': round(flops / us / 1e6 / 2500, 3)}), flush=True)
    lib.set_option('gate_kernel', 1)
    for alt in ((0, 1) if hasattr(L, 'savp_debug_gate_times') else ()):
        lib.set_option('gate_alt', alt)
        for blk in (100,):
                L.savp_debug_gate_block(blk)
                run()
                torch.cuda.synchronize()
                buf = (ctypes.c_ulonglong * 32)()
                L.savp_debug_gate_times(buf)
                for wv in range(4):
                    t = [buf[wv * 8 + i] for i in range(8)]
                    print(json.dumps({'shape': [S, Cx, F], 'alt': alt, 'block': blk, 'wave': wv, 'total': t[7] - t[0],
                                      'phases': {names[i]: t[i + 1] - t[i] for i in range(7)}}), flush=True)
    lib.set_option('gate_alt', 0)
