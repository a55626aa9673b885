"""Developer build only (SAVP_EXTRA_FLAGS=-DSAVP_CONV_ABLATE): per-phase cycles of workgroup 0 / wave 0 of wgrad_patch_kernel."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_prediction_amd import kernels as K, lib
from tests.tools.bench_wgrad import SHAPES, N
K.set_conv_precision('bf16')
for name, H, W, Cx, Cy, k in SHAPES:
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    x = torch.randn(N, H, W, Cx, device='cuda'); y = torch.randn(N, H, W, Cy, device='cuda')
    if os.environ.get('BF16', '1') == '1':              # the step's operands: both tensors stored in bf16
        x = x.bfloat16(); y = y.bfloat16()
    w = torch.zeros(k, k, Cx, Cy, device='cuda')
    geom = K.ConvGeom((k, k), (1, 1), (k // 2, k // 2))
    for _ in range(3):
        K.conv(lib.CONV_WGRAD, geom, x, y, w)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    fn = lib.get().savp_debug_wgp_times
    fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
    fn(buf)
    t = list(buf)
    nt = max(1, t[6])
    names = ['first-fetch', 'stage(wait loads+cvt+ds_write)', 'barrier', 'fetch(issue)', 'mfma-loop']
    print(name, 'tiles', nt, ' '.join('%s:%d/tile' % (names[i], t[i] // nt if i else t[i]) for i in range(5)))
