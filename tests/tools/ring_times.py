"""Developer build only (SAVP_EXTRA_FLAGS=-DSAVP_CONV_ABLATE): cycle stamps of workgroup 0 / wave 0 of conv_ring_kernel."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_prediction_amd import kernels as K, lib
from tests.tools.bench_ring_ab import SHAPES
K.set_conv_precision('bf16')
for spec in sys.argv[1:]:
    parts = spec.split(':')
    name, mname, tile = parts[:3]
    cell = len(parts) > 3 and parts[3] == 'cell'
    sh = [s for s in SHAPES if s[0] == name and s[1] == mname][0]
    _, _, N, H, W, Cx, Cy, k = sh
    mode = lib.CONV_FPROP if mname == 'fprop' else lib.CONV_DGRAD
    x = torch.randn(N, H, W, Cx, device='cuda'); y = torch.randn(N, H, W, Cy, device='cuda')
    w = torch.randn(k * k * Cx * Cy, device='cuda') * 0.05
    geom = K.ConvGeom((k, k), (1, 1), (k // 2, k // 2))
    st = None
    if cell:
        y = torch.empty(N, H, W, Cy, device='cuda', dtype=torch.bfloat16)
        st = torch.zeros(N, Cy, 2, device='cuda')
    for _ in range(5):
        K.conv(mode, geom, x, y, w, tile=int(tile, 16), w16=w.to(torch.bfloat16), splitk=1, stats=st)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    fn = lib.get().savp_debug_ring_times
    fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
    fn(buf)
    t = list(buf)
    names = ['start', 'pre-stage', 'staged', 'loop-start', 'loop-end', 'pre-epilogue', 'end']
    extra = ['geom', 'goff', 'acc/arow', 'group-sync']
    print(spec, ' '.join('%s:+%d' % (extra[i - 7], t[i] - t[0]) for i in range(7, 11)), '|', ' '.join('%s:+%d' % (names[i], t[i] - t[0]) for i in range(1, 7)))
