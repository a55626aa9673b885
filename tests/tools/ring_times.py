"""Developer build only (SAVP_EXTRA_FLAGS=-DSAVP_CONV_ABLATE): cycle stamps of workgroup 0 / wave 0 of conv_ring_kernel."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_prediction_amd import kernels as K, lib
from tests.tools.bench_ring_ab import SHAPES
K.set_conv_precision('bf16')
if os.environ.get('KWARM') is not None and hasattr(lib.get(), 'savp_debug_ring_kwarm'):
    lib.get().savp_debug_ring_kwarm.argtypes = [ctypes.c_int]
    lib.get().savp_debug_ring_kwarm(int(os.environ['KWARM']))
if os.environ.get('RING_BLOCK') is not None:
    lib.get().savp_debug_ring_block.argtypes = [ctypes.c_int]
    lib.get().savp_debug_ring_block(int(os.environ['RING_BLOCK']))
NB = int(os.environ.get('RING_N', '0'))
for spec in sys.argv[1:]:
    parts = spec.split(':')
    name, mname, tile = parts[:3]
    flags = parts[3:]
    cell = 'cell' in flags or 'cell16' in flags
    src16 = 'cell16' in flags or 'src16' in flags            # bf16 source: the LDS-DMA staged patch
    sh = [s for s in SHAPES if s[0] == name and s[1] == mname][0]
    _, _, N, H, W, Cx, Cy, k = sh
    N = NB or N
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
    w16 = w.to(torch.bfloat16)
    try:
        for _ in range(5):
            K.conv(mode, geom, x, y, w, tile=int(tile, 16), w16=w16, splitk=1, stats=st)
    except RuntimeError as e:
        print(spec, 'refused:', e)
        continue
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    fn = lib.get().savp_debug_ring_times
    fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
    fn(buf)
    t = list(buf)
    names = ['start', 'pre-stage', 'staged', 'loop-start', 'loop-end', 'pre-epilogue', 'end']
    extra = ['geom', 'goff', 'acc/arow', 'group-sync', 'loop-head', 'pre-barrier']
    print(spec, ' '.join('%s:+%d' % (extra[i - 7], t[i] - t[0]) for i in (7, 8, 9, 11, 12, 10)), '|', ' '.join('%s:+%d' % (names[i], t[i] - t[0]) for i in range(1, 7)))
    if hasattr(lib.get(), 'savp_debug_ring_wave_times'):
        wb = (ctypes.c_ulonglong * 16)()
        fw = lib.get().savp_debug_ring_wave_times
        fw.argtypes = [ctypes.c_void_p]; fw.restype = ctypes.c_int
        fw(wb)
        w = list(wb)
        nw = 8 if (int(tile, 16) & 0x400) else 4
        print('    per wave: entry', ' '.join('%+d' % (w[i] - t[0]) for i in range(nw)), '| at first barrier', ' '.join('%+d' % (w[8 + i] - t[0]) for i in range(nw)))
