"""Developer build only (tests/tools/build_ring_dev.sh; SAVP_LIB=video_prediction_amd/ab/libsavp_hip_ringdev.so): where the cycles of
conv_ring_kernel go.  For each problem (the ConvLSTM gate convolutions at N = 32): cycle stamps of workgroup 0 for waves 0 and 4,
and the main loop's length with parts of the kernel switched off
(no slab DMAs, no MFMAs, neither) -- what the loop costs when it only multiplies, only streams, or only synchronises."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_prediction_amd import kernels as K, lib
from tests.tools.bench_ring_ab import SHAPES
K.set_conv_precision('bf16')
L = lib.get()
for fn in ('savp_debug_ring_wave', 'savp_debug_ring_ablate', 'savp_debug_ring_block'):
    getattr(L, fn).argtypes = [ctypes.c_int]
L.savp_debug_ring_times.argtypes = [ctypes.c_void_p]
NAMES = ['start', 'pre-stage', 'staged', 'loop-start', 'loop-end', 'pre-epilogue', 'end']


def run(spec, roles, wave, ablate):
    parts = spec.split(':')
    name, mname, tile = parts[:3]
    flags = parts[3:]
    cell = 'cell16' in flags
    src16 = 'cell16' in flags or 'src16' in flags
    gap = [f for f in flags if f.startswith('gap')]
    sh = [s for s in SHAPES if s[0] == name and s[1] == mname][0]
    _, _, N, H, W, Cx, Cy, k = sh
    mode = lib.CONV_FPROP if mname == 'fprop' else lib.CONV_DGRAD
    x = torch.randn(N, H, W, Cx, device='cuda'); y = torch.randn(N, H, W, Cy, device='cuda')
    w = torch.randn(k * k * Cx * Cy, device='cuda') * 0.05
    if src16 and mode == lib.CONV_FPROP:
        x = x.to(torch.bfloat16)
    if src16 and mode == lib.CONV_DGRAD:
        y = y.to(torch.bfloat16)
    geom = K.ConvGeom((k, k), (1, 1), (k // 2, k // 2))
    st = None
    if cell:
        y = torch.empty(N, H, W, Cy, device='cuda', dtype=torch.bfloat16)
        st = torch.zeros(N, Cy, 2, device='cuda', dtype=torch.float64)
    dg = None
    if gap:
        f = (Cx - 8) // 2
        dg = (f, 8)
    w16 = w.to(torch.bfloat16)
    L.savp_debug_ring_wave(wave)
    L.savp_debug_ring_ablate(ablate)
    for _ in range(4):
        K.conv(mode, geom, x, y, w, tile=int(tile, 16), w16=w16, splitk=1, stats=st, dst_gap=dg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.conv(mode, geom, x, y, w, tile=int(tile, 16), w16=w16, splitk=1, stats=st, dst_gap=dg)
    e1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    L.savp_debug_ring_times(buf)
    t = list(buf)
    return e0.elapsed_time(e1) * 100.0, [t[i] - t[0] for i in range(1, 7)], t[11] - t[0]


for spec in sys.argv[1:]:
    print('==', spec, flush=True)
    for roles in (0,):
        for wave in (0, 4):
            try:
                us, st, gs = run(spec, roles, wave, 0)
            except RuntimeError as e:
                print('   refused:', e)
                break
            print('   roles %d wave %d: %6.1f us/launch | group-sync:+%d ' % (roles, wave, us, gs) + ' '.join('%s:+%d' % (n, v) for n, v in zip(NAMES[1:], st)) +
                  ' | loop %d' % (st[3] - st[2]), flush=True)
    for abl, what in ((1, 'no slab DMA'), (2, 'no MFMA'), (3, 'no DMA, no MFMA'), (4, 'no patch staging'), (8, 'no epilogue')):
        try:
            us, st, gs = run(spec, 1, 0, abl)
        except RuntimeError as e:
            print('   refused:', e)
            break
        print('   ablate %d (%s): %6.1f us/launch | loop %d | staging %d | epilogue %d' % (abl, what, us, st[3] - st[2], st[1] - st[0], st[5] - st[4]), flush=True)
L.savp_debug_ring_ablate(0)
