"""Developer build only (SAVP_EXTRA_FLAGS=-DSAVP_CONV_ABLATE): cycle stamps of workgroup 0 / wave 0 of conv_ring_kernel."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from video_prediction_amd import kernels as K, lib
from tests.bench_ring_ab import SHAPES
K.set_conv_precision('bf16')
for spec in sys.argv[1:]:
    name, mname, tile = spec.split(':')
    sh = [s for s in SHAPES if s[0] == name and s[1] == mname][0]
    _, _, N, H, W, Cx, Cy, k = sh
    mode = lib.CONV_FPROP if mname == 'fprop' else lib.CONV_DGRAD
    x = torch.randn(N, H, W, Cx, device='cuda'); y = torch.randn(N, H, W, Cy, device='cuda')
    w = torch.randn(k * k * Cx * Cy, device='cuda') * 0.05
    geom = K.ConvGeom((k, k), (1, 1), (k // 2, k // 2))
    for _ in range(5):
        K.conv(mode, geom, x, y, w, tile=int(tile, 16), w16=w.to(torch.bfloat16), splitk=1)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    fn = lib.get().savp_debug_ring_times
    fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
    fn(buf)
    t = list(buf)
    names = ['start', 'pre-stage', 'staged', 'loop-start', 'loop-end', 'pre-epilogue', 'end']
    print(spec, ' '.join('%s:+%d' % (names[i], t[i] - t[0]) for i in range(1, 7)), '(s_memtime ticks = 100 MHz? see ratio)')
