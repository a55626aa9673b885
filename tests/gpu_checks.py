"""GPU parity checks: HIP kernels (through the C ABI) vs the CPU oracle on identical seeded inputs.

Used by tests/test_gpu_*.py (pytest -m gpu), by __graft_entry__.smoke() and by tests/tools/run_gpu_checks.py (which
dumps every result to gpurun_out/ instead of stopping at the first failure).  Each check returns a list of
(name, err, tol) tuples; err is max|hip - oracle| / max(|oracle|) with the oracle evaluated in float64.
"""
import ctypes

import numpy as np
import torch

from oracle import ops as O
from oracle import savp as OS
from oracle import tf_ops as TF
from video_prediction_amd import kernels as K
from video_prediction_amd import lib

DEV = 'cuda:0'
TOL_OP = 2e-5          # fp32 per-op tolerance: rel <= 2e-5 vs the fp64 oracle


def rel_err(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    denom = max(ref.abs().max().item(), 1e-30)
    return (got - ref).abs().max().item() / denom


def dev(t):
    return t.float().to(DEV).contiguous()


def rnd(rng, *shape):
    return torch.tensor(rng.standard_normal(shape), dtype=torch.float64)


# ---------------------------------------------------------------------------------------------------------------
# conv: all three modes against autograd of the fp64 torch conv
# ---------------------------------------------------------------------------------------------------------------
def _ref_conv(x, w, k, s, p, pa):
    """x [N,D,H,W,Cx] fp64, w [kd,kh,kw,Cx,Cy]; explicit pad-before p / pad-after pa; returns y [N,Do,Ho,Wo,Cy]."""
    xc = x.permute(0, 4, 1, 2, 3)
    xc = torch.nn.functional.pad(xc, (p[2], pa[2], p[1], pa[1], p[0], pa[0]))
    y = torch.nn.functional.conv3d(xc, w.permute(4, 3, 0, 1, 2), stride=s)
    return y.permute(0, 2, 3, 4, 1)


def pack_wt(w):
    """HWIO [taps..., Cx, Cy] -> WT [Cy, taps*Cx]."""
    cy = w.shape[-1]
    return w.reshape(-1, cy).t().contiguous()


def pack_wd(w):
    """HWIO -> WD [Cx, taps*Cy]."""
    cx, cy = w.shape[-2], w.shape[-1]
    return w.reshape(-1, cx, cy).permute(1, 0, 2).reshape(cx, -1).contiguous()


CONV_CASES = [
    # name, N, (D,H,W), Cx, Cy, k, s, p(before), pa(after)
    ('lstm5x5_32', 2, (1, 32, 32), 72, 128, (1, 5, 5), (1, 1, 1), (0, 2, 2), (0, 2, 2)),
    ('lstm5x5_8', 3, (1, 8, 8), 264, 512, (1, 5, 5), (1, 1, 1), (0, 2, 2), (0, 2, 2)),
    ('pool6x6s2', 2, (1, 64, 64), 16, 32, (1, 6, 6), (1, 2, 2), (0, 2, 2), (0, 2, 2)),
    ('pool4x4s2', 2, (1, 32, 32), 40, 64, (1, 4, 4), (1, 2, 2), (0, 1, 1), (0, 1, 1)),
    ('head3x3', 2, (1, 64, 64), 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), (0, 1, 1)),
    ('scalar_c14', 2, (1, 32, 32), 14, 32, (1, 6, 6), (1, 2, 2), (0, 2, 2), (0, 2, 2)),
    ('scalar_c53_n7', 2, (1, 32, 32), 53, 7, (1, 3, 3), (1, 1, 1), (0, 1, 1), (0, 1, 1)),
    ('scalar_c3', 1, (1, 32, 32), 32, 3, (1, 3, 3), (1, 1, 1), (0, 1, 1), (0, 1, 1)),
    ('d3_k3', 2, (6, 16, 16), 32, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    ('d3_k4_s122', 2, (6, 16, 16), 32, 64, (4, 4, 4), (1, 2, 2), (1, 1, 1), (1, 1, 1)),
    ('d3_k4_s222', 2, (6, 16, 16), 64, 32, (4, 4, 4), (2, 2, 2), (1, 1, 1), (1, 1, 1)),
    ('d3_first_c3', 2, (5, 16, 16), 3, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    ('dense_8192_100', 4, (1, 1, 1), 8192, 100, (1, 1, 1), (1, 1, 1), (0, 0, 0), (0, 0, 0)),
    ('dense_65536_1', 4, (1, 1, 1), 16384, 1, (1, 1, 1), (1, 1, 1), (0, 0, 0), (0, 0, 0)),
    ('enc4x4s2', 5, (1, 16, 16), 64, 128, (1, 4, 4), (1, 2, 2), (0, 1, 1), (0, 1, 1)),
    ('odd_sizes', 3, (1, 13, 10), 20, 36, (1, 3, 3), (1, 2, 2), (0, 1, 1), (0, 1, 1)),
    ('lstm5x5_16', 2, (1, 16, 16), 136, 256, (1, 5, 5), (1, 1, 1), (0, 2, 2), (0, 2, 2)),
    ('s1_odd', 3, (1, 13, 10), 24, 40, (1, 3, 3), (1, 1, 1), (0, 1, 1), (0, 1, 1)),
    ('s1_odd5', 2, (1, 19, 21), 40, 72, (1, 5, 5), (1, 1, 1), (0, 2, 2), (0, 2, 2)),
    ('s2_odd3', 3, (1, 13, 10), 24, 40, (1, 3, 3), (1, 2, 2), (0, 1, 1), (0, 1, 1)),
    ('s2_odd5', 2, (1, 21, 18), 40, 24, (1, 5, 5), (1, 2, 2), (0, 2, 2), (0, 2, 2)),
    ('up6x6s2', 2, (1, 32, 32), 32, 72, (1, 6, 6), (1, 2, 2), (0, 2, 2), (0, 2, 2)),
    # RGB / grey first layers of the discriminators (csrc/conv_thin.hip in bf16 mode): several column tiles, ragged edges, 2-D
    ('thin_rgb_w72', 2, (4, 24, 72), 3, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    ('thin_grey_2d', 3, (1, 20, 40), 1, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), (0, 1, 1)),
    ('thin_grey_3d', 2, (4, 12, 40), 1, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    ('thin_c4_odd', 1, (3, 9, 33), 4, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
    # 32 -> 4 / 1 channels: the DGRAD of these is the thin kernel with mirrored taps (the generator's scratch-image head)
    ('thin_dgrad_c4', 2, (1, 20, 70), 32, 4, (1, 3, 3), (1, 1, 1), (0, 1, 1), (0, 1, 1)),
    ('thin_dgrad_c1_3d', 1, (3, 10, 34), 32, 1, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1)),
]


def patch_lds_bytes(cred, kh, kw, wm, wn, nw=4, sh=1, sw=1, dgrad=False, hm=64):
    """LDS bytes of the patch conv kernel (mirror of conv_patch_try in csrc/conv_patch.hip); hm = rows of the output grid."""
    cp16 = (cred + 15) // 16 * 16
    best = None
    c0 = (cp16 + 95) // 96
    for c in range(c0, c0 + 3):
        kk = (cp16 // 16 + c - 1) // c
        if kk < 1 or kk > 6:
            continue
        cost = c * (kk + 1.5)
        if best is None or cost < best[0]:
            best = (cost, c, kk)
    if best is None:
        return 1 << 30
    _, nch, nks = best
    th = 2 * nw * wm
    tih = 4
    while tih < th and tih < hm:
        tih *= 2
    ni = th // tih
    ph = tih + (kh + sh - 1) // sh - 1 if dgrad else (tih - 1) * sh + kh
    pw = 8 + (kw + sw - 1) // sw - 1 if dgrad else 7 * sw + kw
    cp = nks * 16 + 8                     # one slab per patch group is the smallest footprint the launcher falls back to
    x = (8 - (pw * (cp // 8)) % 16 + 16) % 16
    pitch = pw * cp + 8 * x
    return ni * ph * pitch * 2 + 2 * 64 * wn * (nks * 16 + 8) * 2


def check_conv(cases=None, seed=0, tiles=(0,), precision=0, tol=None):
    tol = tol or TOL_OP
    out = []
    rng = np.random.default_rng(seed)
    for case in CONV_CASES:
        name, N, dhw, Cx, Cy, k, s, p, pa = case
        if cases and name not in cases:
            continue
        x = rnd(rng, N, *dhw, Cx)
        w = rnd(rng, *k, Cx, Cy) * 0.1
        b = rnd(rng, Cy)
        x.requires_grad_(True)
        w.requires_grad_(True)
        b.requires_grad_(True)
        y = _ref_conv(x, w, k, s, p, pa) + b
        dy = rnd(rng, *y.shape)
        (y * dy).sum().backward()
        geom = K.ConvGeom(k, s, p)
        for tile in tiles:
            tag = name + ('' if tile == 0 else '_t%x' % tile)
            xd, wd_, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
            if (tile & 0x300) == 0x300:          # LDS-DMA ring kernel forced (conv_ring.hip); a refused shape raises -> skipped
                if not (precision == 1 and s[0] == 1 and k[1] >= s[1] and k[2] >= s[2] and Cx % 8 == 0 and Cy % 8 == 0):
                    continue
                wtp, wdp = dev(pack_wt(w.detach())), dev(pack_wd(w.detach()))
                try:
                    yd2 = torch.empty(y.shape, device=DEV, dtype=torch.float32)
                    K.conv(lib.CONV_FPROP, geom, xd, yd2, wtp, bias=bd, tile=tile, precision=1, w16=wtp.to(torch.bfloat16))
                    out.append((tag + '/fprop_ring', rel_err(yd2, y), tol))
                except RuntimeError:
                    pass
                try:
                    dx2 = torch.full(x.shape, float('nan'), device=DEV, dtype=torch.float32)
                    K.conv(lib.CONV_DGRAD, geom, dx2, dyd, wdp, tile=tile, precision=1, w16=wdp.to(torch.bfloat16))
                    out.append((tag + '/dgrad_ring', rel_err(dx2, x.grad), tol))
                except RuntimeError:
                    pass
                continue
            if tile & 0x200:          # LDS patch kernel forced: 2-D stride-1, channels % 8, bf16 weight copy only
                if not (precision == 1 and s[0] == 1 and k[1] >= s[1] and k[2] >= s[2] and
                        Cx % 8 == 0 and Cy % 8 == 0 and y.shape[2] * y.shape[3] >= 16):
                    continue
                # the kernel parks a (8*WM+kh-1) x (8+kw-1) pixel patch of ALL reduction channels in LDS; shapes whose patch
                # does not fit 160 KB are (correctly) refused with EINVAL under a forced tile -> not part of this sweep
                wm_, wn_ = (tile >> 4) & 15, tile & 15
                fits = lambda cred, dg: patch_lds_bytes(cred, k[1], k[2], wm_, wn_, 8 if tile & 0x400 else 4, s[1], s[2], dg,
                                                        (dhw[1] + s[1] - 1) // s[1] if dg else y.shape[2]) <= 160 * 1024
                if not (fits(Cx, False) and fits(Cy, True)):
                    continue
                yd2 = torch.empty(y.shape, device=DEV, dtype=torch.float32)
                wtp = dev(pack_wt(w.detach()))
                K.conv(lib.CONV_FPROP, geom, xd, yd2, wtp, bias=bd, tile=tile, precision=1, w16=wtp.to(torch.bfloat16))
                out.append((tag + '/fprop_w16', rel_err(yd2, y), tol))
                dx2 = torch.full(x.shape, float('nan'), device=DEV, dtype=torch.float32)
                wdp = dev(pack_wd(w.detach()))
                K.conv(lib.CONV_DGRAD, geom, dx2, dyd, wdp, tile=tile, precision=1, w16=wdp.to(torch.bfloat16))
                out.append((tag + '/dgrad_w16', rel_err(dx2, x.grad), tol))
                continue
            # FPROP (+bias)
            yd = torch.empty(y.shape, device=DEV, dtype=torch.float32)
            K.conv(lib.CONV_FPROP, geom, xd, yd, dev(pack_wt(w.detach())), bias=bd, tile=tile, precision=precision)
            out.append((tag + '/fprop', rel_err(yd, y), tol))
            if precision == 1:      # bf16 pre-packed weight stream
                yd2 = torch.empty(y.shape, device=DEV, dtype=torch.float32)
                wtp = dev(pack_wt(w.detach()))
                K.conv(lib.CONV_FPROP, geom, xd, yd2, wtp, bias=bd, tile=tile, precision=1, w16=wtp.to(torch.bfloat16))
                out.append((tag + '/fprop_w16', rel_err(yd2, y), tol))
                dx2 = torch.full(x.shape, float('nan'), device=DEV, dtype=torch.float32)
                wdp = dev(pack_wd(w.detach()))
                K.conv(lib.CONV_DGRAD, geom, dx2, dyd, wdp, tile=tile, precision=1, w16=wdp.to(torch.bfloat16))
                out.append((tag + '/dgrad_w16', rel_err(dx2, x.grad), tol))
            # DGRAD
            dxd = torch.full(x.shape, float('nan'), device=DEV, dtype=torch.float32)
            K.conv(lib.CONV_DGRAD, geom, dxd, dyd, dev(pack_wd(w.detach())), tile=tile, precision=precision)
            out.append((tag + '/dgrad', rel_err(dxd, x.grad), tol))
            # WGRAD
            dwd = torch.zeros(w.shape, device=DEV, dtype=torch.float32)
            dbd = torch.zeros(Cy, device=DEV, dtype=torch.float32)
            K.conv(lib.CONV_WGRAD, geom, xd, dyd, dwd, bias=dbd, tile=tile, precision=precision)
            out.append((tag + '/wgrad', rel_err(dwd, w.grad), tol))
            out.append((tag + '/wgrad_bias', rel_err(dbd, b.grad), TOL_OP))
    torch.cuda.synchronize()
    return out


def check_conv_views_and_epilogues(seed=1):
    """Channel-slice views as source/destination, beta accumulation, fused activations."""
    out = []
    rng = np.random.default_rng(seed)
    N, H, W, Cx, Cy = 2, 16, 16, 24, 40
    xbig = rnd(rng, N, H, W, 64)
    x = xbig[..., 8:8 + Cx]
    w = rnd(rng, 3, 3, Cx, Cy) * 0.1
    b = rnd(rng, Cy)
    y = TF.conv2d(x, w, (1, 1), 'SAME') + b
    geom = K.ConvGeom((3, 3), (1, 1), (1, 1))
    xbd = dev(xbig)
    ybig = torch.zeros(N, H, W, 96, device=DEV)
    yv = ybig[..., 16:16 + Cy]
    K.conv(lib.CONV_FPROP, geom, xbd[..., 8:8 + Cx], yv, dev(pack_wt(w)), bias=dev(b), act=lib.ACT_LRELU, alpha=0.2)
    out.append(('views/fprop_lrelu', rel_err(yv, O.lrelu(y, 0.2)), TOL_OP))
    untouched = float(ybig[..., :16].abs().max() + ybig[..., 16 + Cy:].abs().max())
    out.append(('views/fprop_no_spill', untouched, 0.0))
    K.conv(lib.CONV_FPROP, geom, xbd[..., 8:8 + Cx], yv, dev(pack_wt(w)), bias=dev(b), act=lib.ACT_SIGMOID)
    out.append(('views/fprop_sigmoid', rel_err(yv, torch.sigmoid(y)), TOL_OP))
    # beta accumulate
    base = rnd(rng, N, H, W, Cy)
    yd = dev(base)
    K.conv(lib.CONV_FPROP, geom, xbd[..., 8:8 + Cx], yd, dev(pack_wt(w)), beta=1)
    out.append(('views/fprop_beta', rel_err(yd, base + y - b), TOL_OP))
    # dgrad with lrelu-derivative epilogue and beta: dx = (F^T dy + old) * lrelu'(aux)
    dy = rnd(rng, N, H, W, Cy)
    aux = rnd(rng, N, H, W, Cx)
    old = rnd(rng, N, H, W, Cx)
    xg = x.detach().clone().requires_grad_(True)
    (TF.conv2d(xg, w, (1, 1), 'SAME') * dy).sum().backward()
    ref = (xg.grad + old) * torch.where(aux > 0, torch.ones_like(aux), torch.full_like(aux, 0.1))
    dxd = dev(old)
    K.conv(lib.CONV_DGRAD, geom, dxd, dev(dy), dev(pack_wd(w)), beta=1, act=lib.ACT_DLRELU_FROM_OUT, alpha=0.1, aux=dev(aux))
    out.append(('views/dgrad_dlrelu_beta', rel_err(dxd, ref), TOL_OP))
    # upsample_conv2d == DGRAD mode with the bilinear-folded kernel (ops.py:643-719)
    xl = rnd(rng, 2, 8, 8, 40)
    kk = rnd(rng, 3, 3, 40, 24) * 0.1
    bb = rnd(rng, 24)
    ref = O.upsample_conv2d(xl, kk, bb, strides=(2, 2))
    kup = O.upsample_kernel(kk, (2, 2))                       # [6,6,F,Cin] = HWIO with Cx=F, Cy=Cin
    g6 = K.ConvGeom((6, 6), (2, 2), (2, 2))
    yo = torch.empty(2, 16, 16, 24, device=DEV)
    K.conv(lib.CONV_DGRAD, g6, yo, dev(xl), dev(pack_wd(kup)), bias=dev(bb))
    out.append(('upsample_conv2d_as_dgrad', rel_err(yo, ref), TOL_OP))
    # conv_pool2d == FPROP with the pool-folded kernel (ops.py:795-856)
    xh = rnd(rng, 2, 16, 16, 24)
    kk = rnd(rng, 5, 5, 24, 32) * 0.1
    ref = O.conv_pool2d(xh, kk, None, strides=(2, 2))
    kp = O.pool_kernel(kk, (2, 2))
    yo = torch.empty(2, 8, 8, 32, device=DEV)
    K.conv(lib.CONV_FPROP, g6, dev(xh), yo, dev(pack_wt(kp)))
    out.append(('conv_pool2d_as_fprop', rel_err(yo, ref), TOL_OP))
    torch.cuda.synchronize()
    return out


# ---------------------------------------------------------------------------------------------------------------
# instance norm + activation
# ---------------------------------------------------------------------------------------------------------------
def check_inorm(seed=2):
    out = []
    rng = np.random.default_rng(seed)
    for (N, H, W, C, act, alpha) in [(3, 64, 64, 32, 'relu', 0.0), (2, 8, 8, 128, 'lrelu', 0.2), (2, 5, 7, 8, 'none', 0.0)]:
        x = (rnd(rng, N, H, W, C) * 2 + 0.7).requires_grad_(True)
        g = (rnd(rng, C) * 0.5 + 1).requires_grad_(True)
        b = rnd(rng, C).requires_grad_(True)
        yn = O.fused_instance_norm(x, g, b)
        y = {'relu': torch.relu, 'lrelu': lambda t: O.lrelu(t, alpha), 'none': lambda t: t}[act](yn)
        dy1, dy2 = rnd(rng, *y.shape), rnd(rng, *y.shape)
        (y * (dy1 + dy2)).sum().backward()
        tag = 'inorm_%s_%dx%dx%d' % (act, H, W, C)
        xd, gd, bd = dev(x), dev(g), dev(b)
        big = torch.zeros(N, H, W, C + 8, device=DEV)
        o1, o2 = big[..., 8:], torch.empty(N, H, W, C, device=DEV)
        mean, rstd = torch.empty(N, C, device=DEV), torch.empty(N, C, device=DEV)
        K.instnorm_act_fwd(xd, gd, bd, [o1, o2], mean, rstd, act=act, alpha=alpha)
        out.append((tag + '/fwd', rel_err(o1, y), TOL_OP))
        out.append((tag + '/fwd2', rel_err(o2, y), TOL_OP))
        dx = torch.empty(N, H, W, C, device=DEV)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        K.instnorm_act_bwd(xd, gd, bd, o1, mean, rstd, [dev(dy1), dev(dy2)], dx, dg, db, act=act, alpha=alpha)
        out.append((tag + '/dx', rel_err(dx, x.grad), 5e-5))
        out.append((tag + '/dgamma', rel_err(dg, g.grad), 5e-5))
        out.append((tag + '/dbeta', rel_err(db, b.grad), 5e-5))
    torch.cuda.synchronize()
    return out


# ---------------------------------------------------------------------------------------------------------------
# fused ConvLSTM gate block
# ---------------------------------------------------------------------------------------------------------------
def _ref_lstm_gates(gates, c, g1, b1, g2, b2):
    """rnn_ops.py:148-165 after the conv (oracle.savp.conv_lstm_cell without the conv)."""
    concat = O.fused_instance_norm(gates, g1, b1)
    i, j, f, o = torch.chunk(concat, 4, dim=-1)
    new_c = c * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
    new_c = O.fused_instance_norm(new_c, g2, b2)
    new_h = torch.tanh(new_c) * torch.sigmoid(o)
    return new_c, new_h


def check_lstm(seed=3):
    """ConvLSTM gate block vs the fp64 oracle: the one-launch kernels (option lstm_fused = 1, every (Q, PPT) shape class: N = 32
    planes with 4 / 8 / 16-channel slabs, small N, ragged planes) and, with the option off, the three-pass / single-kernel paths."""
    out = []
    cases = [(2, 32, 32, 32, False, (0, 1)), (3, 16, 16, 64, False, (0, 1)), (2, 8, 8, 128, True, (0, 1)), (2, 4, 6, 8, False, (0, 1)),
             (2, 16, 24, 16, False, (1,)), (32, 32, 32, 32, False, (1,)), (32, 16, 16, 64, False, (1,)), (32, 8, 8, 128, True, (1,)),
             (32, 8, 8, 128, False, (1,))]
    try:
        for fused in (1, 0):
            lib.set_option('lstm_fused', fused)
            rng = np.random.default_rng(seed)
            for (N, H, W, F, zero_state, modes) in cases:
                if fused not in modes:
                    continue
                out += _check_lstm_case(rng, N, H, W, F, zero_state, '_fused' if fused else '')
    finally:
        lib.set_option('lstm_fused', 1)
    torch.cuda.synchronize()
    return out


def _check_lstm_case(rng, N, H, W, F, zero_state, sfx):
    out = []
    if True:
        gates = (rnd(rng, N, H, W, 4 * F) * 1.5 + 0.3).requires_grad_(True)
        c = (torch.zeros(N, H, W, F, dtype=torch.float64) if zero_state else rnd(rng, N, H, W, F)).requires_grad_(True)
        g1 = (rnd(rng, 4 * F) * 0.3 + 1).requires_grad_(True)
        b1 = (rnd(rng, 4 * F) * 0.3).requires_grad_(True)
        g2 = (rnd(rng, F) * 0.3 + 1).requires_grad_(True)
        b2 = (rnd(rng, F) * 0.3).requires_grad_(True)
        cn, hn = _ref_lstm_gates(gates, c, g1, b1, g2, b2)
        dh1, dh2, dcn = rnd(rng, *hn.shape), rnd(rng, *hn.shape), rnd(rng, *cn.shape)
        ((hn * (dh1 + dh2)).sum() + (cn * dcn).sum()).backward()
        tag = 'lstm%s_N%d_%dx%dx%d%s' % (sfx, N, H, W, F, '_zero' if zero_state else '')
        gd = dev(gates)
        cd = None if zero_state else dev(c)
        p = [dev(t) for t in (g1, b1, g2, b2)]
        c_new = torch.empty(N, H, W, F, device=DEV)
        hbig = torch.zeros(N, H, W, 2 * F + 8, device=DEV)
        h1, h2 = hbig[..., :F], hbig[..., F + 8:]
        stats = [torch.empty(N, 4 * F, device=DEV), torch.empty(N, 4 * F, device=DEV),
                 torch.empty(N, F, device=DEV), torch.empty(N, F, device=DEV)]
        K.convlstm_gates_fwd(gd, cd, p[0], p[1], p[2], p[3], c_new, [h1, h2], stats)
        out.append((tag + '/c', rel_err(c_new, cn), TOL_OP))
        out.append((tag + '/h', rel_err(h1, hn), TOL_OP))
        out.append((tag + '/h2', rel_err(h2, hn), TOL_OP))
        out.append((tag + '/pad_untouched', float(hbig[..., F:F + 8].abs().max()), 0.0))
        # coalesced three-pass forward (selected by the workspace when the one-launch kernels are off): same outputs and statistics
        three_pass = F >= 16 and not sfx
        if three_pass:
            ws = torch.empty(K.lstm_ws_floats(N, H * W, F), device=DEV)
            c_ws = torch.empty(N, H, W, F, device=DEV)
            hbig2 = torch.zeros(N, H, W, 2 * F + 8, device=DEV)
            stats2 = [torch.empty_like(t) for t in stats]
            K.convlstm_gates_fwd(gd, cd, p[0], p[1], p[2], p[3], c_ws, [hbig2[..., :F], hbig2[..., F + 8:]], stats2, ws=ws)
            out.append((tag + '/c_ws', rel_err(c_ws, cn), TOL_OP))
            out.append((tag + '/h_ws', rel_err(hbig2[..., :F], hn), TOL_OP))
            out.append((tag + '/h2_ws', rel_err(hbig2[..., F + 8:], hn), TOL_OP))
            out.append((tag + '/pad_untouched_ws', float(hbig2[..., F:F + 8].abs().max()), 0.0))
            for nm, a_, b_ in zip(('mean1', 'rstd1', 'mean2', 'rstd2'), stats2, stats):
                out.append((tag + '/%s_ws' % nm, rel_err(a_, b_.double().cpu()), 1e-5))
        dgates = torch.empty(N, H, W, 4 * F, device=DEV)
        dcp = torch.empty(N, H, W, F, device=DEV)
        dpar = [torch.zeros(4 * F, device=DEV), torch.zeros(4 * F, device=DEV), torch.zeros(F, device=DEV), torch.zeros(F, device=DEV)]
        K.convlstm_gates_bwd(gd, cd, p[0], p[1], p[2], p[3], stats, [dev(dh1), dev(dh2)], dev(dcn), dgates, dcp, dpar)
        out.append((tag + '/dgates', rel_err(dgates, gates.grad), 1e-4))
        out.append((tag + '/dc_prev', rel_err(dcp, c.grad), 1e-4))
        for nm, got, ref in zip(('dg1', 'db1', 'dg2', 'db2'), dpar, (g1.grad, b1.grad, g2.grad, b2.grad)):
            out.append((tag + '/' + nm, rel_err(got, ref), 1e-4))
        if sfx:        # one / three / four gradient sources of h', no d c' in or out: same result as the sum handed over as one source
            dsum = dev(dh1 + dh2)
            third = torch.zeros_like(dsum)
            for srcs in ([dsum], [dev(dh1), dev(dh2), third], [dev(dh1), third, dev(dh2), third]):
                dg3 = torch.empty_like(dgates)
                K.convlstm_gates_bwd(gd, cd, p[0], p[1], p[2], p[3], stats, srcs, dev(dcn), dg3, None, [torch.zeros_like(t) for t in dpar])
                out.append((tag + '/dgates_%dsrc' % len(srcs), rel_err(dg3, gates.grad), 1e-4))
        if three_pass:                       # coalesced three-pass backward
            dgates2 = torch.empty(N, H, W, 4 * F, device=DEV)
            dcp2 = torch.empty(N, H, W, F, device=DEV)
            dpar2 = [torch.zeros_like(t) for t in dpar]
            K.convlstm_gates_bwd(gd, cd, p[0], p[1], p[2], p[3], stats, [dev(dh1), dev(dh2)], dev(dcn), dgates2, dcp2, dpar2, ws=ws)
            out.append((tag + '/dgates_ws', rel_err(dgates2, gates.grad), 1e-4))
            out.append((tag + '/dc_prev_ws', rel_err(dcp2, c.grad), 1e-4))
            for nm, got, ref in zip(('dg1', 'db1', 'dg2', 'db2'), dpar2, (g1.grad, b1.grad, g2.grad, b2.grad)):
                out.append((tag + '/' + nm + '_ws', rel_err(got, ref), 1e-4))
            K.convlstm_gates_bwd(gd, cd, p[0], p[1], p[2], p[3], stats, [dev(dh1)], None, dgates2, None, dpar2, ws=ws)   # no dc in/out
            torch.cuda.synchronize()
    return out


# ---------------------------------------------------------------------------------------------------------------
# util ops
# ---------------------------------------------------------------------------------------------------------------
def check_dense(seed=14):
    """ops.dense for few rows (ops.py:5-16): the K-sliced kernel + reduction launch, incl. spectral-norm scale and row strides."""
    out = []
    rng = np.random.default_rng(seed)
    for (M, Kd, C, strided) in [(32, 8192, 100, False), (4, 65536, 1, False), (2, 4096, 1, True), (7, 1000, 36, False), (64, 520, 256, False)]:
        xb = rnd(rng, M, Kd + 8)
        x = xb[:, :Kd] if strided else xb[:, :Kd].contiguous()
        W, b = rnd(rng, Kd, C) * 0.05, rnd(rng, C)
        sc = 0.37
        ref = sc * (x @ W) + b
        xd = dev(xb)[:, :Kd] if strided else dev(x)
        o = torch.full((M, C), float('nan'), device=DEV)
        K.dense_fwd(xd, dev(W), dev(b), o, scale=torch.tensor([sc], device=DEV))
        out.append(('dense_%dx%dx%d%s' % (M, Kd, C, '_strided' if strided else ''), rel_err(o, ref), TOL_OP))
    torch.cuda.synchronize()
    return out


def check_util(seed=4):
    out = []
    rng = np.random.default_rng(seed)
    R, H, W, C = 6, 8, 8, 8
    z = rnd(rng, R, C)
    big = torch.zeros(R, H, W, 24, device=DEV)
    K.tile_channels(dev(z), big[..., 8:16], scale=0.5)
    ref = torch.zeros(R, H, W, 24, dtype=torch.float64)
    ref[..., 8:16] = 0.5 * z[:, None, None, :]
    out.append(('tile_channels', rel_err(big, ref), 1e-6))
    x = rnd(rng, R, H, W, 24)
    o1 = torch.zeros(C, device=DEV)
    K.colsum(dev(x)[..., 8:16], o1, scale=2.0)
    out.append(('colsum_all', rel_err(o1, 2.0 * x[..., 8:16].sum(dim=(0, 1, 2))), 1e-5))
    o2 = torch.zeros(R, C, device=DEV)
    K.colsum(dev(x)[..., 8:16], o2, per_row=True)
    out.append(('colsum_rows', rel_err(o2, x[..., 8:16].sum(dim=(1, 2))), 1e-5))
    # narrow-slice kernel: 16- / 8- / 4-byte aligned slices (float4 / float2 / scalar lanes), several pixel chunks per row
    xw = rnd(rng, 5, 33, 40, 30)
    for (lo, cc) in ((8, 8), (6, 8), (3, 4), (12, 16), (5, 1), (10, 2)):
        ow = torch.zeros(5, cc, device=DEV)
        K.colsum(dev(xw)[..., lo:lo + cc], ow, per_row=True, scale=0.5)
        out.append(('colsum_rows_narrow_c%d_off%d' % (cc, lo), rel_err(ow, 0.5 * xw[..., lo:lo + cc].sum(dim=(1, 2))), 1e-5))
    x200 = rnd(rng, 3, 5, 7, 200)
    o3 = torch.zeros(200, device=DEV)
    K.colsum(dev(x200), o3)
    out.append(('colsum_c200', rel_err(o3, x200.sum(dim=(0, 1, 2))), 1e-5))
    # wide power-of-two rows (the discriminators' bias gradients): float4 lanes, many pixel chunks per row
    for (cc, hw) in ((32, (9, 40)), (64, (33, 31)), (256, (8, 8)), (1024, (8, 8))):
        xc = rnd(rng, 3, hw[0], hw[1], cc)
        oc = torch.zeros(cc, device=DEV)
        K.colsum(dev(xc), oc, scale=0.25)
        out.append(('colsum_wide_c%d' % cc, rel_err(oc, 0.25 * xc.sum(dim=(0, 1, 2))), 1e-5))
        ocr = torch.zeros(3, cc, device=DEV)
        K.colsum(dev(xc), ocr, per_row=True)
        out.append(('colsum_wide_rows_c%d' % cc, rel_err(ocr, xc.sum(dim=(1, 2))), 1e-5))
    # all-pixel sums of large tensors take the partial-rows + reduce path (>= 32768 pixels, pixel-linear, C / 4 a power of two);
    # accumulation into a non-zero destination, ragged last chunk, a channel slice (pixel stride > C)
    for (cc, shape, lo, width) in ((32, (6, 128, 97), 0, 32), (64, (37, 33, 31), 0, 64), (32, (5, 100, 80), 8, 48), (1024, (40, 30, 30), 0, 1024)):
        xc = rnd(rng, *shape, width)
        o0 = rnd(rng, cc)
        oc = dev(o0)
        K.colsum(dev(xc)[..., lo:lo + cc], oc, scale=0.5)
        out.append(('colsum_2stage_c%d' % cc, rel_err(oc, o0 + 0.5 * xc[..., lo:lo + cc].sum(dim=(0, 1, 2))), 1e-5))
    # select fwd / bwd
    N, C3 = 4, 3
    a, b = rnd(rng, N, H, W, C3), rnd(rng, N, H, W, C3)
    mask = torch.tensor([1, 0, 0, 1], dtype=torch.int32)
    o_a = torch.zeros(N, H, W, 14, device=DEV)
    o_b = torch.zeros(N, H, W, C3, device=DEV)
    K.select(mask.to(DEV), dev(a), dev(b), [o_a[..., :3], o_b])
    ref = torch.where(mask.bool()[:, None, None, None], a, b)
    out.append(('select', rel_err(o_a[..., :3], ref) + rel_err(o_b, ref), 1e-7))
    # no second source (zeros), grey (1 channel) and 2-channel (generic kernel) variants
    for cc in (1, 2, 3):
        oz = torch.full((N, H, W, 6), 5.0, device=DEV)
        K.select(mask.to(DEV), dev(a[..., :cc]), None, [oz[..., 2:2 + cc]])
        refz = torch.where(mask.bool()[:, None, None, None], a[..., :cc], torch.zeros_like(a[..., :cc]))
        keep = bool((oz[..., :2] == 5.0).all() and (oz[..., 2 + cc:] == 5.0).all())
        out.append(('select_nob_c%d' % cc, rel_err(oz[..., 2:2 + cc], refz) + (0.0 if keep else 1.0), 1e-7))
    # the generator's first-frame fill: one frame broadcast over T steps (source sample stride 0), two destinations
    T_, Nn = 5, 3
    frame = rnd(rng, Nn, H, W, C3)
    fd = dev(frame)
    bufa, bufb = torch.zeros(T_, Nn, H, W, 16, device=DEV), torch.zeros(T_, Nn, H, W, 9, device=DEV)
    first = fd.reshape(1, Nn * H, W, C3).expand(T_, Nn * H, W, C3)
    K.select(torch.ones(T_, dtype=torch.int32, device=DEV), first, None,
             [bufa.reshape(T_, Nn * H, W, 16)[..., 3:6], bufb.reshape(T_, Nn * H, W, 9)[..., 6:9]])
    refb = frame[None].expand(T_, Nn, H, W, C3)
    out.append(('select_broadcast_first_frame', rel_err(bufa[..., 3:6], refb) + rel_err(bufb[..., 6:9], refb), 1e-7))
    d1, d2, db0 = rnd(rng, N, H, W, C3), rnd(rng, N, H, W, C3), rnd(rng, N, H, W, C3)
    dbd = dev(db0)
    K.select_bwd(mask.to(DEV), [dev(d1), dev(d2)], dbd)
    ref = db0 + torch.where(mask.bool()[:, None, None, None], torch.zeros_like(d1), d1 + d2)
    out.append(('select_bwd', rel_err(dbd, ref), 1e-6))
    # gather clips + adjoint
    L, B, clip = 9, 3, 4
    src = rnd(rng, L, B, 4, 4, 3)
    ts = torch.tensor([0, 5, 2], dtype=torch.int32)
    dst = torch.empty(B, clip, 4, 4, 3, device=DEV)
    K.gather_clips(dev(src), dst, ts.to(DEV))
    ref = torch.stack([src[ts[b]:ts[b] + clip, b] for b in range(B)], dim=0)
    out.append(('gather_clips', rel_err(dst, ref), 1e-7))
    g = rnd(rng, B, clip, 4, 4, 3)
    acc0 = rnd(rng, L, B, 4, 4, 3)
    accd = dev(acc0)
    K.gather_clips(accd, dev(g), ts.to(DEV), adjoint=True)
    ref = acc0.clone()
    for b in range(B):
        ref[ts[b]:ts[b] + clip, b] += g[b]
    out.append(('gather_clips_adjoint', rel_err(accd, ref), 1e-6))
    # axpby, fill
    xx, yy = rnd(rng, 1000), rnd(rng, 1000)
    od = torch.empty(1000, device=DEV)
    K.axpby(2.0, dev(xx), -0.5, dev(yy), od)
    out.append(('axpby', rel_err(od, 2 * xx - 0.5 * yy), 1e-6))
    fb = dev(rnd(rng, 2, 4, 4, 8))
    K.fill_view(fb[..., 2:6], 0.0)
    out.append(('fill_view', float(fb[..., 2:6].abs().max()), 0.0))
    # adam (two steps) vs oracle
    n = 1003
    p0, g1, g2 = rnd(rng, n), rnd(rng, n), rnd(rng, n)
    pd, md, vd = torch.zeros(1004, device=DEV), torch.zeros(1004, device=DEV), torch.zeros(1004, device=DEV)
    pd[:n] = dev(p0)
    pr, mr, vr = p0.clone(), torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    import math
    for t, g in ((1, g1), (2, g2)):
        gd = torch.zeros(1004, device=DEV)
        gd[:n] = dev(g) * 4.0
        lr_t = 2e-4 * math.sqrt(1 - 0.999 ** t) / (1 - 0.5 ** t)
        K.adam(pd[:n], gd[:n], md[:n], vd[:n], lr_t, 0.5, 0.999, gscale=0.25)
        pr, mr, vr = TF.adam_update(pr, g, mr, vr, 2e-4, 0.5, 0.999, t)
    out.append(('adam_p', rel_err(pd[:n], pr), 1e-6))
    # fp32 (1 - beta2) differs from the fp64 oracle's by 4.7e-5 relative -- same as TF's fp32 Adam kernel
    out.append(('adam_v', rel_err(vd[:n], vr), 1e-4))
    torch.cuda.synchronize()
    return out


# ---------------------------------------------------------------------------------------------------------------
# cdna + composite
# ---------------------------------------------------------------------------------------------------------------
def check_cdna_composite(seed=5):
    out = []
    rng = np.random.default_rng(seed)
    # 16x16 = one LDS tile; 8x12 / 40x24 / 19x35 = partial tiles and several tiles per image (tiled 5x5 kernels)
    for (N, H, W, C, Kk) in [(3, 16, 16, 3, 4), (2, 8, 12, 1, 4), (2, 40, 24, 3, 4), (2, 19, 35, 1, 4), (1, 64, 64, 3, 4)]:
        kh = kw = 5
        raw = (rnd(rng, N, kh, kw, Kk) * 0.3).requires_grad_(True)
        img = torch.tensor(rng.random((N, H, W, C)), dtype=torch.float64, requires_grad=True)
        ident = torch.as_tensor(OS.identity_kernel((kh, kw)))
        kern = raw + ident[None, :, :, None]
        kern = torch.relu(kern - OS.RELU_SHIFT) + OS.RELU_SHIFT
        kern = kern / kern.sum(dim=(1, 2), keepdim=True)
        timgs = OS.apply_cdna_kernels(img, kern)
        ref_out = torch.cat(timgs, dim=-1)                       # channel k*C + c
        dout = rnd(rng, *ref_out.shape)
        (ref_out * dout).sum().backward()
        tag = 'cdna_%dx%dx%d' % (H, W, C)
        rawd = dev(raw).reshape(N, kh * kw * Kk)
        kd = torch.empty(N, kh * kw, Kk, device=DEV)
        K.cdna_kernels_fwd(rawd, kd, kh, kw, Kk)
        out.append((tag + '/kernels', rel_err(kd.reshape(N, kh, kw, Kk), kern), TOL_OP))
        big = torch.zeros(N, H, W, 32 + Kk * C, device=DEV)
        ov = big[..., 32:]
        imgd = dev(img)
        K.cdna_apply_fwd(imgd, kd, ov, kh, kw, Kk)
        out.append((tag + '/apply', rel_err(ov, ref_out), TOL_OP))
        dimg = torch.empty(N, H, W, C, device=DEV)
        dk = torch.empty(N, kh * kw, Kk, device=DEV, dtype=torch.float64)
        K.cdna_apply_bwd(imgd, kd, dev(dout), dimg, dk, kh, kw, Kk)
        out.append((tag + '/dimg', rel_err(dimg, img.grad), 5e-5))
        # same through a 16-byte aligned channel slice of a wider buffer (the layout the generator uses: fast kernels)
        dwide = torch.zeros(N, H, W, 32 + 4 * ((Kk * C + 3) // 4) + 4, device=DEV)
        dwide[..., 32:32 + Kk * C] = dev(dout)
        dimg2 = torch.empty(N, H, W, C, device=DEV)
        dk2 = torch.full((N, kh * kw, Kk), float('nan'), device=DEV, dtype=torch.float64)
        K.cdna_apply_bwd(imgd, kd, dwide[..., 32:32 + Kk * C], dimg2, dk2, kh, kw, Kk)
        out.append((tag + '/dimg_fast', rel_err(dimg2, img.grad), 5e-5))
        draw2 = torch.empty(N, kh * kw * Kk, device=DEV)
        K.cdna_kernels_bwd(rawd, dk2, draw2, kh, kw, Kk)
        out.append((tag + '/draw_fast', rel_err(draw2.reshape(N, kh, kw, Kk), raw.grad), 5e-5))
        draw = torch.empty(N, kh * kw * Kk, device=DEV)
        K.cdna_kernels_bwd(rawd, dk, draw, kh, kw, Kk)
        out.append((tag + '/draw', rel_err(draw.reshape(N, kh, kw, Kk), raw.grad), 5e-5))
    # identity property: zero dense output => CDNA returns the input image exactly (savp_model.py:551,968-980)
    N, H, W, C, Kk = 2, 8, 8, 3, 4
    img = torch.tensor(rng.random((N, H, W, C)))
    kd = torch.empty(N, 25, Kk, device=DEV)
    K.cdna_kernels_fwd(torch.zeros(N, 25 * Kk, device=DEV), kd, 5, 5, Kk)
    ov = torch.empty(N, H, W, Kk * C, device=DEV)
    K.cdna_apply_fwd(dev(img), kd, ov, 5, 5, Kk)
    out.append(('cdna_identity', rel_err(ov, torch.cat([img] * Kk, dim=-1)), 1e-6))
    # composite
    for (N, H, W, C, M) in [(2, 16, 16, 3, 7), (2, 8, 8, 1, 7), (3, 20, 23, 3, 7)]:
        logits = (rnd(rng, N, H, W, M) * 2).requires_grad_(True)
        timgs = torch.tensor(rng.random((N, H, W, M * C)), dtype=torch.float64, requires_grad=True)
        masks = torch.softmax(logits, dim=-1)
        gen = sum(masks[..., k:k + 1] * timgs[..., k * C:(k + 1) * C] for k in range(M))
        dgen = rnd(rng, *gen.shape)
        (gen * dgen).sum().backward()
        tag = 'composite_c%d' % C
        big = dev(torch.cat([torch.zeros(N, H, W, 32, dtype=torch.float64), timgs.detach()], dim=-1))
        tv = big[..., 32:]
        ld = dev(logits)
        gd = torch.empty(N, H, W, C, device=DEV)
        md = torch.empty(N, H, W, M, device=DEV)
        K.composite_fwd(ld, tv, gd, md)
        out.append((tag + '/gen', rel_err(gd, gen), TOL_OP))
        out.append((tag + '/masks', rel_err(md, masks), TOL_OP))
        amism = int((md.argmax(dim=-1).cpu() != masks.argmax(dim=-1)).sum())
        out.append((tag + '/mask_argmax_mismatches', float(amism), 0.0))
        # padded logits row (stride 8) and whole-row gradient write
        l8 = torch.full((N, H, W, 8), 7.0, device=DEV)
        l8[..., :M] = ld
        g8 = torch.empty(N, H, W, C, device=DEV)
        K.composite_fwd(l8, tv, g8, None, M=M)
        out.append((tag + '/gen_padded_logits', rel_err(g8, gen), TOL_OP))
        dl = torch.full((N, H, W, 8), float('nan'), device=DEV)
        dbig = torch.full((N, H, W, 32 + M * C + 3), float('nan'), device=DEV)
        K.composite_bwd(l8, tv, dev(dgen), dl, dbig, 32, M=M)
        # the generator's layout: value rows and gradient rows of the same width (coalesced LDS-transposed kernel)
        rowc = (dbig.shape[-1] + 3) // 4 * 4
        vwide = torch.zeros(N, H, W, rowc, device=DEV)
        vwide[..., 32:32 + M * C] = tv
        dl2 = torch.full((N, H, W, 8), float('nan'), device=DEV)
        dbig2 = torch.full((N, H, W, rowc), float('nan'), device=DEV)
        K.composite_bwd(l8, vwide[..., 32:32 + M * C], dev(dgen), dl2, dbig2, 32, M=M)
        out.append((tag + '/tiled_dlogits', rel_err(dl2[..., :M], logits.grad), 5e-5))
        out.append((tag + '/tiled_dlogits_pad_zero', float(dl2[..., M:].abs().max()), 0.0))
        out.append((tag + '/tiled_dtimgs', rel_err(dbig2[..., 32:32 + M * C], timgs.grad), 5e-5))
        out.append((tag + '/tiled_drow_rest_zero', float(dbig2[..., :32].abs().max() + dbig2[..., 32 + M * C:].abs().max()), 0.0))
        out.append((tag + '/dlogits', rel_err(dl[..., :M], logits.grad), 5e-5))
        out.append((tag + '/dlogits_pad_zero', float(dl[..., M:].abs().max()), 0.0))
        out.append((tag + '/dtimgs', rel_err(dbig[..., 32:32 + M * C], timgs.grad), 5e-5))
        out.append((tag + '/drow_rest_zero', float(dbig[..., :32].abs().max() + dbig[..., 32 + M * C:].abs().max()), 0.0))
    torch.cuda.synchronize()
    return out


# ---------------------------------------------------------------------------------------------------------------
# small ops
# ---------------------------------------------------------------------------------------------------------------
def check_small(seed=6):
    out = []
    rng = np.random.default_rng(seed)
    for (T, B, nz) in [(7, 4, 8), (5, 3, 32)]:
        zs = rnd(rng, T, B, nz).requires_grad_(True)
        W = (rnd(rng, 2 * nz, 4 * nz) * 0.3).requires_grad_(True)
        b = (rnd(rng, 4 * nz) * 0.1).requires_grad_(True)
        c = torch.zeros(B, nz, dtype=torch.float64)
        h = torch.zeros(B, nz, dtype=torch.float64)
        hs = []
        for t in range(T):
            ho, (c, h) = TF.lstm_cell(zs[t], c, h, W, b)
            hs.append(ho)
        hs = torch.stack(hs)
        dh = rnd(rng, T, B, nz)
        (hs * dh).sum().backward()
        tag = 'lstm_z_nz%d' % nz
        zd, Wd, bd = dev(zs), dev(W), dev(b)
        ho = torch.empty(T, B, nz, device=DEV)
        gates = torch.empty(T, B, 4 * nz, device=DEV)
        cs = torch.empty(T, B, nz, device=DEV)
        K.lstm_z_fwd(zd, Wd, bd, ho, gates, cs)
        out.append((tag + '/h', rel_err(ho, hs), TOL_OP))
        dz = torch.empty(T, B, nz, device=DEV)
        dW = torch.zeros(2 * nz, 4 * nz, device=DEV)
        db = torch.zeros(4 * nz, device=DEV)
        K.lstm_z_bwd(zd, Wd, ho, gates, cs, dev(dh), dz, dW, db)
        out.append((tag + '/dz', rel_err(dz, zs.grad), 5e-5))
        out.append((tag + '/dW', rel_err(dW, W.grad), 5e-5))
        out.append((tag + '/db', rel_err(db, b.grad), 5e-5))
    # reparam + KL
    T, B, nz = 5, 4, 8
    mu = rnd(rng, T, B, nz).requires_grad_(True)
    lsr = (rnd(rng, T, B, nz) * 6).requires_grad_(True)       # some values beyond the [-10, 10] clip
    eps = rnd(rng, T, B, nz)
    ls = torch.clamp(lsr, -10, 10)
    z = mu + torch.sqrt(torch.exp(ls)) * eps
    from oracle import train as OT
    kl = OT.kl_loss(mu, ls)
    dz = rnd(rng, T, B, nz)
    ((z * dz).sum() + 0.7 * kl).backward()
    lsd, zd_, kld = torch.empty(T, B, nz, device=DEV), torch.empty(T, B, nz, device=DEV), torch.zeros(1, device=DEV)
    K.reparam_fwd(dev(mu), dev(lsr), dev(eps), lsd, zd_, kld)
    out.append(('reparam/z', rel_err(zd_, z), TOL_OP))
    out.append(('reparam/kl', rel_err(kld, kl.reshape(1)), 1e-5))
    dmu, dls = torch.empty(T, B, nz, device=DEV), torch.empty(T, B, nz, device=DEV)
    K.reparam_bwd(dev(mu), dev(lsr), dev(eps), dev(dz), 0.7, dmu, dls)
    out.append(('reparam/dmu', rel_err(dmu, mu.grad), 1e-5))
    out.append(('reparam/dls', rel_err(dls, lsr.grad), 1e-5))
    # l1 / l2
    pred = torch.tensor(rng.random((3, 2, 8, 8, 3)), dtype=torch.float64, requires_grad=True)
    targ = torch.tensor(rng.random((3, 2, 8, 8, 3)), dtype=torch.float64)
    for p2, fn, nm in ((False, OT.l1_loss, 'l1'), (True, OT.l2_loss, 'l2')):
        pred.grad = None
        l = fn(pred, targ)
        (100.0 * l).backward()
        lo = torch.zeros(1, device=DEV)
        dp = torch.zeros(pred.shape, device=DEV)
        K.lp_loss(dev(pred), dev(targ), 100.0, lo, dp, p2=p2)
        out.append((nm + '/loss', rel_err(lo, l.reshape(1)), 1e-5))
        out.append((nm + '/dpred', rel_err(dp, pred.grad), 1e-5))
    # lsgan
    lg = rnd(rng, 16, 1).requires_grad_(True)
    l = OT.gan_loss(lg, 1.0, 'LSGAN')
    (0.1 * l).backward()
    lo = torch.zeros(1, device=DEV)
    dl = torch.empty(16, 1, device=DEV)
    K.lsgan_loss(dev(lg), 1.0, 0.1, lo, dl)
    out.append(('lsgan/loss', rel_err(lo, l.reshape(1)), 1e-5))
    out.append(('lsgan/dlogits', rel_err(dl, lg.grad), 1e-5))
    # cosine distance
    for Cc in (32, 256, 64, 128, 48):           # 48: the one-wave-per-position fallback
        f0 = rnd(rng, 4, 2, 5, 5, Cc).requires_grad_(True)
        f1 = rnd(rng, 4, 2, 5, 5, Cc)
        l = OT.cosine_distance(f0, f1)
        (10.0 * l).backward()
        lo = torch.zeros(1, device=DEV)
        df = torch.empty(f0.shape, device=DEV)
        K.cosine_distance(dev(f0), dev(f1), 10.0, lo, df)
        out.append(('cosine_c%d/loss' % Cc, rel_err(lo, l.reshape(1)), 1e-5))
        out.append(('cosine_c%d/df0' % Cc, rel_err(df, f0.grad), 5e-5))
    torch.cuda.synchronize()
    return out


# ---------------------------------------------------------------------------------------------------------------
# weight prep: packing, folds (+adjoints), spectral norm fwd/bwd
# ---------------------------------------------------------------------------------------------------------------
def check_weight_prep(seed=7):
    out = []
    rng = np.random.default_rng(seed)
    w = rnd(rng, 3, 3, 20, 12)
    sc = torch.tensor([0.37], dtype=torch.float64)
    wt = torch.empty(12, 9 * 20, device=DEV)
    wd = torch.empty(20, 9 * 12, device=DEV)
    K.pack_weights(dev(w), wt, wd, scale=dev(sc))
    out.append(('pack/wt', rel_err(wt, pack_wt(w) * 0.37), 1e-6))
    out.append(('pack/wd', rel_err(wd, pack_wd(w) * 0.37), 1e-6))
    # batched pack of several layers == the single-layer packs, bit for bit (fp32 and bf16 copies, optional outputs / scale)
    w2 = rnd(rng, 5, 5, 7, 32)
    w3 = rnd(rng, 1, 64, 10)
    wt_b, wd_b = torch.empty_like(wt), torch.empty_like(wd)
    wt2, wt2_16 = torch.empty(32, 25 * 7, device=DEV), torch.empty(32, 25 * 7, device=DEV, dtype=torch.bfloat16)
    wd3, wd3_16 = torch.empty(64, 10, device=DEV), torch.empty(64, 10, device=DEV, dtype=torch.bfloat16)
    K.pack_weights_batch([{'src': dev(w), 'wt': wt_b, 'wd': wd_b, 'scale': dev(sc)},
                          {'src': dev(w2), 'wt': wt2, 'wt16': wt2_16},
                          {'src': dev(w3), 'wd': wd3, 'wd16': wd3_16, 'scale': dev(sc)}])
    r2, r2_16 = torch.empty_like(wt2), torch.empty_like(wt2_16)
    r3, r3_16 = torch.empty_like(wd3), torch.empty_like(wd3_16)
    K.pack_weights(dev(w2), r2, None, wt16=r2_16)
    K.pack_weights(dev(w3), None, r3, scale=dev(sc), wd16=r3_16)
    same = bool((wt_b == wt).all() and (wd_b == wd).all() and (wt2 == r2).all() and (wt2_16 == r2_16).all() and
                (wd3 == r3).all() and (wd3_16 == r3_16).all())
    out.append(('pack_batch/bit_exact', 0.0 if same else 1.0, 0.5))
    # fold_pool + adjoint
    for k in (5, 3):
        w = rnd(rng, k, k, 6, 8).requires_grad_(True)
        kp = O.pool_kernel(w, (2, 2))
        g = rnd(rng, *kp.shape)
        (kp * g).sum().backward()
        o = torch.empty(kp.shape, device=DEV)
        K.fold_pool(dev(w), o, k)
        out.append(('fold_pool_k%d' % k, rel_err(o, kp), 1e-6))
        dw = torch.zeros(w.shape, device=DEV)
        K.fold_pool(dev(g), dw, k, adjoint=True)
        out.append(('fold_pool_k%d_adj' % k, rel_err(dw, w.grad), 1e-6))
    # fold_bilinear + adjoint
    w = rnd(rng, 3, 3, 10, 6).requires_grad_(True)
    ku = O.upsample_kernel(w, (2, 2))
    g = rnd(rng, *ku.shape)
    (ku * g).sum().backward()
    o = torch.empty(ku.shape, device=DEV)
    K.fold_bilinear(dev(w), o, 3, 10, 6)
    out.append(('fold_bilinear', rel_err(o, ku), 1e-6))
    dw = torch.zeros(w.shape, device=DEV)
    K.fold_bilinear(dev(g), dw, 3, 10, 6, adjoint=True)
    out.append(('fold_bilinear_adj', rel_err(dw, w.grad), 1e-6))
    # spectral norm
    batch = []
    for shape in [(3, 3, 3, 16, 32), (4, 4, 4, 8, 16), (640, 1), (4, 4, 4, 16, 128), (3, 3, 3, 8, 24), (2, 2, 5, 256)]:
        W = (rnd(rng, *shape) * 0.05).requires_grad_(True)
        C = shape[-1]
        u = rnd(rng, 1, C)
        Wb, u_fin = O.spectral_normed_weight(W, u)
        G = rnd(rng, *shape)
        (Wb * G).sum().backward()
        Kdim = W.numel() // C
        ws = torch.zeros(K.sn_ws_size(Kdim, C), device=DEV)
        Wd, ud = dev(W), dev(u).reshape(C)
        un = torch.empty(C, device=DEV)
        K.sn_fwd(Wd, ud, ws, un)
        tag = 'sn_%s' % 'x'.join(map(str, shape))
        sigma_ref = (W.detach() / Wb.detach()).flatten()[0]
        out.append((tag + '/sigma', rel_err(ws[0:1], sigma_ref.reshape(1)), 1e-5))
        out.append((tag + '/u_final', rel_err(un, u_fin.detach().reshape(C)), 1e-5))
        dW = torch.empty(shape, device=DEV)
        K.sn_bwd(Wd, ud, ws, dev(G), dW)
        out.append((tag + '/dW', rel_err(dW, W.grad), 1e-4))
        batch.append(dict(W=Wd, u=ud, ws=torch.zeros_like(ws), u_new=torch.empty(C, device=DEV), G=dev(G), dW=torch.zeros(shape, device=DEV), beta=1,
                          ref=(sigma_ref.reshape(1), u_fin.detach().reshape(C), W.grad), tag=tag))
    # the batched entries (all six tensors in 4 launches each way) against the same oracle values; beta = 1 accumulates into dW
    K.sn_fwd_batch(batch)
    K.sn_bwd_batch(batch)
    K.sn_bwd_batch(batch)
    for e in batch:
        out.append((e['tag'] + '/batch_sigma', rel_err(e['ws'][0:1], e['ref'][0]), 1e-5))
        out.append((e['tag'] + '/batch_u_final', rel_err(e['u_new'], e['ref'][1]), 1e-5))
        out.append((e['tag'] + '/batch_dW_twice', rel_err(e['dW'], 2 * e['ref'][2]), 1e-4))
    torch.cuda.synchronize()
    return out



# ---------------------------------------------------------------------------------------------------------------
# flow warp + DNA
# ---------------------------------------------------------------------------------------------------------------
def check_warp_dna(seed=8):
    out = []
    rng = np.random.default_rng(seed)
    for (N, H, W, C, Kk) in [(2, 16, 12, 3, 4), (2, 8, 8, 1, 2)]:
        img = torch.tensor(rng.random((N, H, W, C)), dtype=torch.float64, requires_grad=True)
        flows = (rnd(rng, N, H, W, 2, Kk) * 2.5).requires_grad_(True)          # [N,H,W,2,K] as in savp_model.py:530
        outs = OS.apply_flows(img, flows)
        ref = torch.cat(outs, dim=-1)
        dout = rnd(rng, *ref.shape)
        (ref * dout).sum().backward()
        tag = 'warp_%dx%dx%d' % (H, W, C)
        fl = dev(flows).reshape(N, H, W, 2 * Kk)
        big = torch.zeros(N, H, W, 8 + Kk * C, device=DEV)
        K.image_warp_fwd(dev(img), fl, big[..., 8:], Kk)
        out.append((tag + '/fwd', rel_err(big[..., 8:], ref), TOL_OP))
        dfl = torch.empty(N, H, W, 2 * Kk, device=DEV)
        dimg = torch.empty(N, H, W, C, device=DEV)
        K.image_warp_bwd(dev(img), fl, dev(dout), dfl, dimg, Kk)
        out.append((tag + '/dflows', rel_err(dfl.reshape(N, H, W, 2, Kk), flows.grad), 5e-5))
        out.append((tag + '/dimg', rel_err(dimg, img.grad), 5e-5))
    for (N, H, W, C, Kk) in [(2, 12, 10, 3, 4), (1, 8, 8, 1, 4)]:
        kh = kw = 5
        img = torch.tensor(rng.random((N, H, W, C)), dtype=torch.float64, requires_grad=True)
        raw = (rnd(rng, N, H, W, kh, kw, Kk) * 0.3).requires_grad_(True)
        ident = torch.as_tensor(OS.identity_kernel((kh, kw)))
        kern = raw + ident[None, None, None, :, :, None]
        kern = torch.relu(kern - OS.RELU_SHIFT) + OS.RELU_SHIFT
        kern = kern / kern.sum(dim=(3, 4), keepdim=True)
        ref = torch.cat(OS.apply_dna_kernels(img, kern), dim=-1)
        dout = rnd(rng, *ref.shape)
        (ref * dout).sum().backward()
        tag = 'dna_%dx%dx%d' % (H, W, C)
        rawd = dev(raw).reshape(N, H, W, kh * kw * Kk)
        kd = torch.empty_like(rawd)
        od = torch.empty(N, H, W, Kk * C, device=DEV)
        K.dna_apply_fwd(dev(img), rawd, kd, od, kh, kw, Kk)
        out.append((tag + '/fwd', rel_err(od, ref), TOL_OP))
        out.append((tag + '/kern', rel_err(kd.reshape(N, H, W, kh, kw, Kk), kern), TOL_OP))
        draw = torch.empty_like(rawd)
        dimg = torch.empty(N, H, W, C, device=DEV)
        K.dna_apply_bwd(dev(img), rawd, kd, dev(dout), draw, dimg, kh, kw, Kk)
        out.append((tag + '/draw', rel_err(draw.reshape(N, H, W, kh, kw, Kk), raw.grad), 5e-5))
        out.append((tag + '/dimg', rel_err(dimg, img.grad), 5e-5))
    torch.cuda.synchronize()
    return out


def check_conv_bf16():
    """bf16-operand / fp32-accumulate mode of the implicit-GEMM kernel: per-op rel <= 1e-2."""
    res = check_conv(precision=1, tol=1e-2, tiles=(0, 0x22, 0x11, 0x212, 0x221, 0x122, 0x612, 0x621, 0x611,
                                                   0x311, 0x312, 0x321, 0x322, 0x711, 0x712, 0x721, 0x722))
    # 0x2xx: LDS patch kernel (0x6xx: 8 waves), 0x1xx: generic, 0x3xx / 0x7xx: LDS-DMA ring kernel (4 / 8 waves)
    return [('bf16/' + n, e, t) for (n, e, t) in res]


def check_conv_thin(seed=23):
    """conv_thin.hip beyond the plain cases of CONV_CASES: fused bias + LeakyReLU into a channel slice of a wider buffer, and
    WGRAD accumulating into non-zero dW / db (the `+=` contract of SAVP_CONV_WGRAD)."""
    out = []
    rng = np.random.default_rng(seed)
    N, dhw, Cx, Cy, k = 2, (3, 18, 70), 3, 32, (3, 3, 3)
    x, w, b = rnd(rng, N, *dhw, Cx), rnd(rng, *k, Cx, Cy) * 0.2, rnd(rng, Cy)
    y = torch.nn.functional.leaky_relu(_ref_conv(x, w, k, (1, 1, 1), (1, 1, 1), (1, 1, 1)) + b, 0.2)
    geom = K.ConvGeom(k, (1, 1, 1), (1, 1, 1))
    wide = torch.full((N,) + dhw + (48,), 7.0, device=DEV)
    K.conv(lib.CONV_FPROP, geom, dev(x), wide[..., 8:40], dev(pack_wt(w)), bias=dev(b), act=lib.ACT_LRELU, alpha=0.2, precision=1)
    out.append(('thin/fprop_lrelu_slice', rel_err(wide[..., 8:40], y), 1e-2))
    untouched = bool((wide[..., :8] == 7.0).all() and (wide[..., 40:] == 7.0).all())
    out.append(('thin/fprop_slice_untouched', 0.0 if untouched else 1.0, 0.5))
    dy = rnd(rng, N, *dhw, Cy)
    dw0, db0 = rnd(rng, *k, Cx, Cy), rnd(rng, Cy)
    dwd, dbd = dev(dw0), dev(db0)
    K.conv(lib.CONV_WGRAD, geom, dev(x), dev(dy), dwd, bias=dbd, precision=1)
    xp = torch.nn.functional.pad(x.permute(0, 4, 1, 2, 3), (1, 1, 1, 1, 1, 1))
    ref_dw = torch.zeros(*k, Cx, Cy, dtype=torch.float64)
    for a in range(3):
        for u in range(3):
            for v in range(3):
                xs = xp[:, :, a:a + dhw[0], u:u + dhw[1], v:v + dhw[2]]           # [N, Cx, D, H, W]
                ref_dw[a, u, v] = torch.einsum('ncdhw,ndhwo->co', xs, dy)
    out.append(('thin/wgrad_accumulates', rel_err(dwd, dw0 + ref_dw), 1e-2))
    out.append(('thin/wgrad_bias_accumulates', rel_err(dbd, db0 + dy.sum(dim=(0, 1, 2, 3))), TOL_OP))
    # conv_s2dgrad.hip: data gradient of the 4x4 stride-(1,2,2) layer into a 32-channel activation; ragged tiles, accumulation into
    # the destination (beta) and the fused LeakyReLU backward from the saved activation; 3-D (kd = 4) and 2-D (kd = 1, Cy = 32)
    for (tag, N, dhw, Cy, k, s_, pp, beta, act) in (('s2dgrad/3d_beta_dlrelu', 2, (5, 18, 40), 64, (4, 4, 4), (1, 2, 2), (1, 1, 1), 1, True),
                                                    ('s2dgrad/3d_plain', 1, (4, 32, 64), 64, (4, 4, 4), (1, 2, 2), (1, 1, 1), 0, False),
                                                    ('s2dgrad/2d_c32', 3, (1, 32, 34), 32, (1, 4, 4), (1, 2, 2), (0, 1, 1), 0, True)):
        x = rnd(rng, N, *dhw, 32).requires_grad_(True)
        w = rnd(rng, *k, 32, Cy) * 0.1
        y = _ref_conv(x, w, k, s_, pp, pp)
        dy = rnd(rng, *y.shape)
        (y * dy).sum().backward()
        old = rnd(rng, N, *dhw, 32)
        aux = rnd(rng, N, *dhw, 32)
        ref = x.grad + (old if beta else 0.0)
        if act:
            ref = ref * torch.where(aux > 0, torch.ones_like(aux), torch.full_like(aux, 0.1))
        dxd = dev(old) if beta else torch.full(x.shape, float('nan'), device=DEV)
        wdp = dev(pack_wd(w))
        K.conv(lib.CONV_DGRAD, K.ConvGeom(k, s_, pp), dxd, dev(dy), wdp, beta=beta, act=lib.ACT_DLRELU_FROM_OUT if act else 0, alpha=0.1,
               aux=dev(aux) if act else None, precision=1, w16=wdp.to(torch.bfloat16))
        out.append((tag, rel_err(dxd, ref), 1e-2))
    torch.cuda.synchronize()
    return out


def check_conv_cell(seed=21):
    """The fused ConvLSTM cell of the bf16 datapath (csrc/conv_ring.hip + the coalesced gate kernels): gate convolution with the
    bf16 / statistics epilogue, then IN(4F) + gates + IN(F) + h from the bf16 gate tensor and the epilogue's sums, against the
    fp64 oracle of the whole cell (rnn_ops.py:137-171).  Also a bf16 SOURCE tensor."""
    out = []
    rng = np.random.default_rng(seed)
    for (N, H, W, Cx, F) in [(2, 32, 32, 72, 32), (4, 16, 16, 136, 64), (4, 8, 8, 264, 128), (2, 16, 24, 40, 16)]:
        x = rnd(rng, N, H, W, Cx)
        w = rnd(rng, 5, 5, Cx, 4 * F) * 0.05
        c = rnd(rng, N, H, W, F)
        g1, b1 = rnd(rng, 4 * F) * 0.3 + 1, rnd(rng, 4 * F) * 0.3
        g2, b2 = rnd(rng, F) * 0.3 + 1, rnd(rng, F) * 0.3
        gates = TF.conv2d(x, w, (1, 1), 'SAME')
        cn, hn = _ref_lstm_gates(gates, c, g1, b1, g2, b2)
        tag = 'cell_%dx%dx%d' % (H, W, F)
        geom = K.ConvGeom((5, 5), (1, 1), (2, 2))
        wt = dev(pack_wt(w))
        for src16 in (False, True):
            t2 = tag + ('_src16' if src16 else '')
            xd = dev(x).to(torch.bfloat16) if src16 else dev(x)
            yd = torch.empty(N, H, W, 4 * F, device=DEV, dtype=torch.bfloat16)
            ws, s1 = K.lstm_stats_ws(torch.device(DEV), N, F)
            ws.zero_()
            K.conv(lib.CONV_FPROP, geom, xd, yd, wt, precision=1, w16=wt.to(torch.bfloat16), stats=s1)
            out.append((t2 + '/gates_bf16', rel_err(yd.float(), gates), 1e-2))
            if Cx > 96:        # wide weight slabs (tile bit 0x1000: 8 / 9 k-steps per entry): same sums in another slab order
                for wide in (0x1311, 0x1711):
                    yw = torch.empty_like(yd)
                    sw = torch.zeros_like(s1)
                    try:
                        K.conv(lib.CONV_FPROP, geom, xd, yw, wt, precision=1, w16=wt.to(torch.bfloat16), stats=sw, tile=wide)
                    except RuntimeError:
                        continue                       # this workgroup size cannot hold the wider ring (LDS)
                    out.append((t2 + '/wide%x_gates_bf16' % wide, rel_err(yw.float(), gates), 1e-2))
                    out.append((t2 + '/wide%x_stats_sum' % wide, rel_err(sw[..., 0], torch.stack([gates.sum(dim=(1, 2))], dim=-1)[..., 0]), 1e-2))
            ref_s = torch.stack([gates.sum(dim=(1, 2)), (gates ** 2).sum(dim=(1, 2))], dim=-1)         # [N, 4F, 2]
            out.append((t2 + '/stats_sum', rel_err(s1[..., 0], ref_s[..., 0]), 1e-2))
            out.append((t2 + '/stats_sumsq', rel_err(s1[..., 1], ref_s[..., 1]), 1e-2))
            p = [dev(t) for t in (g1, b1, g2, b2)]
            c_new = torch.empty(N, H, W, F, device=DEV)
            h1 = torch.empty(N, H, W, F, device=DEV)
            stats = [torch.empty(N, 4 * F, device=DEV), torch.empty(N, 4 * F, device=DEV), torch.empty(N, F, device=DEV),
                     torch.empty(N, F, device=DEV)]
            lws = torch.empty(K.lstm_ws_floats(N, H * W, F), device=DEV)
            K.convlstm_gates_fwd(yd, dev(c), p[0], p[1], p[2], p[3], c_new, [h1], stats, ws=lws, stats1=ws)
            out.append((t2 + '/c', rel_err(c_new, cn), 2e-2))
            out.append((t2 + '/h', rel_err(h1, hn), 2e-2))
        # the gate kernels alone, exact: bf16 gate tensor + exact unshifted sums vs the oracle run on the SAME rounded tensor
        gq = gates.float().to(torch.bfloat16)
        gq64 = gq.double().requires_grad_(True)
        c64 = c.clone().requires_grad_(True)
        cn2, hn2 = _ref_lstm_gates(gq64, c64, g1, b1, g2, b2)
        dh, dcn = rnd(rng, *hn2.shape), rnd(rng, *cn2.shape)
        ((hn2 * dh).sum() + (cn2 * dcn).sum()).backward()
        ws, s1 = K.lstm_stats_ws(torch.device(DEV), N, F)
        s1.copy_(torch.stack([gq64.detach().sum(dim=(1, 2)), (gq64.detach() ** 2).sum(dim=(1, 2))], dim=-1).float().to(DEV))
        K.convlstm_gates_fwd(gq.to(DEV), dev(c), p[0], p[1], p[2], p[3], c_new, [h1], stats, ws=lws, stats1=ws)
        out.append((tag + '/gate_kernels_bf16in/c', rel_err(c_new, cn2), 2e-4))
        out.append((tag + '/gate_kernels_bf16in/h', rel_err(h1, hn2), 2e-4))
        dgates = torch.empty(N, H, W, 4 * F, device=DEV)
        dcp = torch.empty(N, H, W, F, device=DEV)
        dpar = [torch.zeros(4 * F, device=DEV), torch.zeros(4 * F, device=DEV), torch.zeros(F, device=DEV), torch.zeros(F, device=DEV)]
        K.convlstm_gates_bwd(gq.to(DEV), dev(c), p[0], p[1], p[2], p[3], stats, [dev(dh)], dev(dcn), dgates, dcp, dpar, ws=lws)
        out.append((tag + '/gate_kernels_bf16in/dgates', rel_err(dgates, gq64.grad), 1e-3))
        out.append((tag + '/gate_kernels_bf16in/dc_prev', rel_err(dcp, c64.grad), 1e-3))
    torch.cuda.synchronize()
    return out


def check_gate_conv_kernel(seed=67):
    """The gate convolution's own kernel (csrc/conv_gate.hip; rnn_ops.py:115-126,143) at every instantiated (image side, input channels): bf16
    input [x | z | h], weights in B-fragment order (pack_gate_weights), bf16 gate pre-activations + the instance norm's float64 sums -- against
    the fp64 cross-correlation of the SAME bf16-rounded operands (so the gate is accumulation error only), against the ring kernel on the same
    call (option gate_kernel = 0), run twice (bit-identical), and at batch sizes that leave the last 8 x 8 tile pair / column tiles full.  Also:
    the fragment pack itself, element by element, and what savp_conv_special says."""
    out = []
    rng = np.random.default_rng(seed)
    geom = K.ConvGeom((5, 5), (1, 1), (2, 2))
    for (N, S, Cx, F) in [(4, 32, 72, 32), (6, 16, 136, 64), (8, 8, 264, 128), (3, 32, 96, 32), (5, 16, 160, 64), (32, 16, 136, 64),
                          (3, 32, 136, 64), (2, 32, 264, 128), (5, 16, 264, 128), (4, 32, 64, 32), (4, 16, 128, 64), (4, 8, 256, 128)]:          # ... and the 128 x 128 model's layers
        tag = 'gate_%dx%d_c%d_n%d' % (S, S, Cx, N)
        x = (rnd(rng, N, S, S, Cx)).float().to(torch.bfloat16)
        w = (rnd(rng, 5, 5, Cx, 4 * F) * 0.05).float()
        wq = w.to(torch.bfloat16)
        ref = TF.conv2d(x.double(), wq.double(), (1, 1), 'SAME')                      # fp64 on the rounded operands
        ref_s = torch.stack([ref.sum(dim=(1, 2)), (ref ** 2).sum(dim=(1, 2))], dim=-1)
        wd = dev(w)
        frag = torch.empty(K.gate_weights_elems(25, Cx, 4 * F), device=DEV, dtype=torch.bfloat16)
        K.pack_gate_weights(wd, frag)
        # the pack, element by element: [cb][ks][lane][j] = W[tap][ch8 * 8 + j][cb * 32 + (lane & 31)], chunk 2 ks + (lane >> 5)
        C8 = Cx // 8
        KS = (25 * C8 + 1) // 2
        npack = (4 * F // 32) * KS * 64 * 8
        out.append((tag + '/pack_pad_zero', float(frag[npack:].float().abs().max()), 0.0))      # 8 k-steps of readable zeros behind the pack
        fr = frag[:npack].float().cpu().reshape(4 * F // 32, KS, 64, 8)
        wf = wq.float().reshape(25, Cx, 4 * F)
        exp = torch.zeros_like(fr)
        for ks in range(KS):
            for half in (0, 1):
                c = 2 * ks + half
                if c < 25 * C8:
                    tap, ch = divmod(c, C8)
                    blk = wf[tap, ch * 8:ch * 8 + 8, :]                               # [8, Cy]
                    exp[:, ks, half * 32:(half + 1) * 32, :] = blk.t().reshape(4 * F // 32, 32, 8)
        out.append((tag + '/pack_exact', float((fr - exp).abs().max()), 0.0))
        wt = dev(pack_wt(w.double()))
        xd = x.to(DEV)
        res = []
        for rep in range(2):
            yd = torch.full((N, S, S, 4 * F), float('nan'), device=DEV, dtype=torch.bfloat16)
            s1 = torch.zeros(N, 4 * F, 2, device=DEV, dtype=torch.float64)
            a = K._fill_conv_args(lib.CONV_FPROP, geom, xd, yd, wt, None, 0, 0, 0.0, None, 0, 0, 1, wt.to(torch.bfloat16), s1, None, None, frag)
            assert lib.get().savp_conv_special(ctypes.byref(a)) == 1, tag
            K.conv(lib.CONV_FPROP, geom, xd, yd, wt, precision=1, w16=wt.to(torch.bfloat16), stats=s1, w_frag=frag)
            torch.cuda.synchronize()
            res.append((yd.clone(), s1.clone()))
        yd, s1 = res[0]
        out.append((tag + '/gates_vs_fp64', rel_err(yd.float(), ref), 6e-3))          # bf16 rounding of the result: 2^-9 relative to the element
        out.append((tag + '/stats_sum', rel_err(s1[..., 0], ref_s[..., 0]), 2e-5))      # the sums are taken of the fp32 accumulators
        out.append((tag + '/stats_sumsq', rel_err(s1[..., 1], ref_s[..., 1]), 2e-5))
        out.append((tag + '/repeat_bits', float((res[1][0].view(torch.int16) != yd.view(torch.int16)).sum() + (res[1][1] != s1).sum()), 0.0))
        # the ring kernel on the same call
        lib.set_option('gate_kernel', 0)
        try:
            yr = torch.empty_like(yd)
            sr = torch.zeros_like(s1)
            K.conv(lib.CONV_FPROP, geom, xd, yr, wt, precision=1, w16=wt.to(torch.bfloat16), stats=sr, w_frag=frag)
        finally:
            lib.set_option('gate_kernel', 1)
        out.append((tag + '/vs_ring_gates', rel_err(yd.float(), yr.float()), 8e-3))     # both round the same fp32 sums (other order) to bf16
        out.append((tag + '/vs_ring_stats', rel_err(s1, sr), 2e-5))
    torch.cuda.synchronize()
    return out


def check_one_launch_cell(seed=71):
    """savp_convlstm_cell_fwd as ONE kernel (csrc/conv_gate.hip, CELL instantiations; rnn_ops.py:137-171): the 16 x 16 and 8 x 8 ConvLSTM layers,
    weights in the interleaved fragment order, against the fp64 oracle of the whole cell on the bf16-rounded operands -- gate tensor (bf16, the
    backward pass reads it), c', h' into up to four destinations (fp32 and bf16, contiguous and channel slices of wider buffers), the four saved
    statistics vectors -- with and without a previous state, an odd batch at 8 x 8 (the last tile's second image is absent), and against the
    two-launch path of the same entry (option gate_cell = 0); the launch count is checked through the statistics workspace, which only the
    two-launch path fills."""
    out = []
    rng = np.random.default_rng(seed)
    geom = K.ConvGeom((5, 5), (1, 1), (2, 2))
    for (N, S, Cx, F, zero_state, nh) in [(3, 16, 136, 64, False, 3), (5, 8, 264, 128, False, 4), (2, 8, 264, 128, True, 1), (2, 16, 160, 64, False, 2),
                                          (4, 16, 128, 64, False, 2), (3, 8, 256, 128, False, 2)]:          # ... and the deterministic (nz = 0) model's layers
        tag = 'cell1_%dx%d_c%d_n%d' % (S, S, Cx, N)
        x = rnd(rng, N, S, S, Cx).float().to(torch.bfloat16)
        w = (rnd(rng, 5, 5, Cx, 4 * F) * 0.05).float()
        c = rnd(rng, N, S, S, F)
        g1, b1 = rnd(rng, 4 * F) * 0.3 + 1, rnd(rng, 4 * F) * 0.3
        g2, b2 = rnd(rng, F) * 0.3 + 1, rnd(rng, F) * 0.3
        gates = TF.conv2d(x.double(), w.to(torch.bfloat16).double(), (1, 1), 'SAME')
        # the cell reads the bf16-ROUNDED gate tensor with the statistics of the unrounded one (as the two-launch path): the oracle on the rounded
        # tensor differs from that by the rounding's effect on mean / variance, far below the gates
        gq = gates.float().to(torch.bfloat16).double()
        cz = torch.zeros_like(c) if zero_state else c
        cn, hn = _ref_lstm_gates(gq, cz, g1, b1, g2, b2)
        wd = dev(w)
        wt = dev(pack_wt(w.double()))
        w16 = wt.to(torch.bfloat16)
        n_el = K.gate_weights_elems(25, Cx, 4 * F)
        frag, frag_il = torch.empty(n_el, device=DEV, dtype=torch.bfloat16), torch.empty(n_el, device=DEV, dtype=torch.bfloat16)
        K.pack_gate_weights(wd, frag)
        K.pack_gate_weights(wd, frag_il, interleave=True)
        xd = x.to(DEV)
        p = [dev(t) for t in (g1, b1, g2, b2)]
        res = {}
        for mode in (1, 0):
            lib.set_option('gate_cell', mode)
            try:
                yd = torch.full((N, S, S, 4 * F), float('nan'), device=DEV, dtype=torch.bfloat16)
                ws, s1 = K.lstm_stats_ws(torch.device(DEV), N, F)
                ws.zero_()
                c_new = torch.full((N, S, S, F), float('nan'), device=DEV)
                wide = torch.full((N, S, S, 3 * F + 8), float('nan'), device=DEV, dtype=torch.bfloat16)      # a consumer's [.. | h | ..] buffer
                hs = [torch.full((N, S, S, F), float('nan'), device=DEV), wide[..., 8:8 + F],
                      torch.full((N, S, S, F), float('nan'), device=DEV, dtype=torch.bfloat16), wide[..., 8 + F:8 + 2 * F]][:nh]
                stats = [torch.full((N, 4 * F), float('nan'), device=DEV), torch.full((N, 4 * F), float('nan'), device=DEV),
                         torch.full((N, F), float('nan'), device=DEV), torch.full((N, F), float('nan'), device=DEV)]
                lws = torch.empty(K.lstm_ws_floats(N, S * S, F), device=DEV)
                ca = K.conv(lib.CONV_FPROP, geom, xd, yd, wt, precision=1, w16=w16, stats=s1, w_frag=frag, w_frag_il=frag_il, defer=True)
                ga = K.convlstm_gates_fwd(yd, None if zero_state else dev(c), p[0], p[1], p[2], p[3], c_new, hs, stats, ws=lws, stats1=ws, defer=True)
                K.convlstm_cell_fwd(ca, ga)
                torch.cuda.synchronize()
                res[mode] = dict(y=yd.float().cpu(), c=c_new.cpu(), h=[h.float().cpu() for h in hs], st=[t.cpu() for t in stats],
                                 one_launch=bool(float(s1.abs().sum()) == 0.0))
            finally:
                lib.set_option('gate_cell', 1)
        out.append((tag + '/is_one_launch', 0.0 if (res[1]['one_launch'] and not res[0]['one_launch']) else 1.0, 0.0))
        r = res[1]
        out.append((tag + '/gates_bf16', rel_err(r['y'], gates), 6e-3))
        out.append((tag + '/c', rel_err(r['c'], cn), 2e-2))
        for k, h in enumerate(r['h']):
            out.append((tag + '/h%d' % k, rel_err(h, hn), 2e-2))
        m1 = gates.mean(dim=(1, 2))
        v1 = gates.var(dim=(1, 2), unbiased=False)
        out.append((tag + '/mean1', rel_err(r['st'][0], m1), 1e-4))
        out.append((tag + '/rstd1', rel_err(r['st'][1], 1.0 / torch.sqrt(v1 + 1e-6)), 1e-3))
        # against the two-launch path (another tile split of the same sums: bf16 rounding flips, nothing more)
        t = res[0]
        out.append((tag + '/vs_two_launch_c', rel_err(r['c'], t['c']), 2e-2))
        out.append((tag + '/vs_two_launch_h', max(rel_err(a_, b_) for a_, b_ in zip(r['h'], t['h'])), 2e-2))
        out.append((tag + '/vs_two_launch_mean2', rel_err(r['st'][2], t['st'][2]), 5e-3))
        out.append((tag + '/vs_two_launch_rstd2', rel_err(r['st'][3], t['st'][3]), 5e-3))
    torch.cuda.synchronize()
    return out


ALL_CHECKS = [('conv', check_conv), ('conv_bf16', check_conv_bf16), ('conv_cell', check_conv_cell),
              ('conv_views', check_conv_views_and_epilogues), ('inorm', check_inorm),
              ('lstm', check_lstm), ('util', check_util), ('dense', check_dense), ('cdna_composite', check_cdna_composite),
              ('small', check_small), ('weight_prep', check_weight_prep), ('warp_dna', check_warp_dna)]


def failures(results):
    return [(n, e, t) for (n, e, t) in results if not (e <= t)]


def smoke():
    """One small hot-path invocation on cuda:0 checked against the oracle (used by __graft_entry__.smoke): the gate convolution and the
    gate block on their own (fp32 and the bf16 bench datapath), then ONE train step of a tiny SAVP model (B = 1, T = 4: forward unroll,
    BPTT, encoder, losses, Adam) against the oracle's step, and two more steps of the same engine so that the captured step is replayed."""
    res = check_conv(cases=('lstm5x5_32',)) + _check_lstm_case(np.random.default_rng(3), 2, 32, 32, 32, False, '_fused')
    # the bench datapath: bf16 LDS-patch FPROP / DGRAD / WGRAD kernels on the ConvLSTM gate-conv shape
    res += [('bf16/' + n, e, t) for (n, e, t) in check_conv(cases=('lstm5x5_32',), precision=1, tol=1e-2)]
    from tests import gpu_model_checks as G
    res += G.check_train_step(B=1, T=4, nz=8, steps=1, tag='smoke_train', video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.0,
                              vae_gan_feature_cdist_weight=0.0)
    hp = G.make_hparams(context_frames=2, sequence_length=4, nz=8, video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.0,
                        vae_gan_feature_cdist_weight=0.0)
    from video_prediction_amd.models.savp_model import SAVPEngine
    eng = SAVPEngine(hp, (64, 64, 3), 1, mode='train', seed=4, device=DEV)
    eng.set_images(torch.rand(4, 1, 64, 64, 3, generator=torch.Generator().manual_seed(2)).to(DEV), time_major=True)
    losses = [float(eng.train_step()['g_loss']) for _ in range(3)]
    res.append(('smoke_train/replayed_step_is_finite', 0.0 if (eng.graph is not None and all(np.isfinite(losses))) else 1.0, 0.0))
    bad = failures(res)
    if bad:
        raise AssertionError('smoke parity failures: %r' % bad)
    return res


# ---------------------------------------------------------------------------------------------------------------
# evaluation metrics (csrc/metrics.hip) vs oracle/metrics.py
# ---------------------------------------------------------------------------------------------------------------
def check_metrics(seed=11):
    from oracle import metrics as OM
    out = []
    rng = np.random.default_rng(seed)
    for (T, B, H, W, C) in [(3, 2, 64, 64, 3), (2, 3, 24, 20, 1), (4, 2, 32, 32, 3)]:
        tgt = torch.tensor(rng.random((T, B, H, W, C)))
        big = torch.tensor(rng.random((T, 2 * B, H, W, C)))                  # the generator's [T, 2B, ...] buffer
        big[:, B:] = (tgt + 0.1 * torch.tensor(rng.standard_normal((T, B, H, W, C)))).clamp(0, 1)
        pred = big[:, B:]
        td, bd = dev(tgt), dev(big)
        pd = bd[:, B:]                                                       # strided batch half, used in place
        tag = 'metrics_%dx%dx%d' % (H, W, C)
        mse = torch.empty(T, B, device=DEV); psnr = torch.empty(T, B, device=DEV); ssim = torch.empty(T, B, device=DEV)
        K.frame_mse_psnr(td, pd, mse=mse, psnr=psnr)
        K.frame_ssim(td, pd, ssim)
        out.append((tag + '/mse', rel_err(mse, OM.mse(tgt, pred)), 1e-5))
        out.append((tag + '/psnr', rel_err(psnr, OM.psnr(tgt, pred)), 1e-5))
        out.append((tag + '/ssim', rel_err(ssim, OM.ssim(tgt, pred)), 2e-5))
        K.frame_ssim(td, td, ssim)
        out.append((tag + '/ssim_identity', float((ssim - 1).abs().max()), 1e-5))
    # the sampling fold (base_model.py:170-198) on random metrics / images
    T, B = 4, 5
    shape = (T + 2, B, 8, 8, 3)
    vmin = torch.full((T, B), float('inf'), device=DEV); vmax = torch.full((T, B), float('-inf'), device=DEV)
    vsum = torch.zeros(T, B, device=DEV)
    gmin = torch.zeros(shape, device=DEV); gmax = torch.zeros(shape, device=DEV); gsum = torch.zeros(shape, device=DEV)
    cmin = torch.zeros(B, dtype=torch.int32, device=DEV); cmax = torch.zeros(B, dtype=torch.int32, device=DEV)
    r_min = torch.full((T, B), float('inf'), dtype=torch.float64); r_max = torch.full((T, B), float('-inf'), dtype=torch.float64)
    r_sum = torch.zeros(T, B, dtype=torch.float64)
    rg = {k: torch.zeros(shape, dtype=torch.float64) for k in ('min', 'max', 'sum')}
    for s in range(6):
        m = torch.tensor(rng.standard_normal((T, B)))
        g = torch.tensor(rng.random(shape))
        lo = m.mean(0) < r_min.mean(0); hi = m.mean(0) > r_max.mean(0)
        r_min = torch.where(lo[None], m, r_min); r_max = torch.where(hi[None], m, r_max); r_sum = r_sum + m
        sel = lambda c, x, y: torch.where(c.reshape(1, -1, 1, 1, 1), x, y)
        rg['min'] = sel(lo, g, rg['min']); rg['max'] = sel(hi, g, rg['max']); rg['sum'] = rg['sum'] + g
        gd = dev(g)
        K.eval_accumulate(dev(m), vmin, vsum, vmax, cmin, cmax)
        K.select_batch(cmin, gd, gmin); K.select_batch(cmax, gd, gmax); K.select_batch(None, gd, gsum, mode=1)
    out.append(('fold/min', rel_err(vmin, r_min), 1e-6)); out.append(('fold/max', rel_err(vmax, r_max), 1e-6))
    out.append(('fold/sum', rel_err(vsum, r_sum), 1e-6))
    out.append(('fold/gmin', rel_err(gmin, rg['min']), 1e-6)); out.append(('fold/gmax', rel_err(gmax, rg['max']), 1e-6))
    out.append(('fold/gsum', rel_err(gsum, rg['sum']), 1e-6))
    torch.cuda.synchronize()
    return out


ALL_CHECKS.append(('metrics', check_metrics))


def check_bf16_activation_io(seed=31):
    """bf16 destinations / operands of the kernels around the ConvLSTM gate convolution (SAVP_BF16_ACT=1 / SAVP_BF16_DGATES=1 in the engine):
    a bf16 view receives exactly the fp32 result rounded to nearest even, and the weight gradient from bf16 operand tensors equals
    the one from the same values held in fp32."""
    out = []
    rng = np.random.default_rng(seed)
    N, H, W, C = 2, 32, 32, 32
    # instance norm + ReLU into one fp32 and one bf16 slice (large-plane two-kernel path and the single-kernel path)
    for (hh, ww) in ((32, 32), (8, 8)):
        x = dev(rnd(rng, N, hh, ww, C))
        g, b = dev(rnd(rng, C) * 0.3 + 1), dev(rnd(rng, C) * 0.3)
        o32 = torch.zeros(N, hh, ww, C + 8, device=DEV)
        o16 = torch.zeros(N, hh, ww, C + 8, device=DEV, dtype=torch.bfloat16)
        mean, rstd = torch.empty(N, C, device=DEV), torch.empty(N, C, device=DEV)
        K.instnorm_act_fwd(x, g, b, [o32[..., :C], o16[..., 8:]], mean, rstd, act='relu')
        same = bool((o16[..., 8:] == o32[..., :C].to(torch.bfloat16)).all()) and float(o16[..., :8].float().abs().max()) == 0.0
        out.append(('bf16io/inorm_%dx%d' % (hh, ww), 0.0 if same else 1.0, 0.5))
    # ConvLSTM gate block: h' into an fp32 and a bf16 destination (coalesced and single-kernel paths)
    for (hh, ww, F, use_ws) in ((32, 32, 32, True), (8, 8, 16, False)):
        gates = dev(rnd(rng, N, hh, ww, 4 * F) * 1.5)
        c = dev(rnd(rng, N, hh, ww, F))
        p = [dev(rnd(rng, 4 * F) * 0.3 + 1), dev(rnd(rng, 4 * F) * 0.3), dev(rnd(rng, F) * 0.3 + 1), dev(rnd(rng, F) * 0.3)]
        c_new = torch.empty(N, hh, ww, F, device=DEV)
        h32 = torch.zeros(N, hh, ww, F, device=DEV)
        h16 = torch.zeros(N, hh, ww, 2 * F + 8, device=DEV, dtype=torch.bfloat16)
        stats = [torch.empty(N, 4 * F, device=DEV), torch.empty(N, 4 * F, device=DEV), torch.empty(N, F, device=DEV), torch.empty(N, F, device=DEV)]
        ws = torch.empty(K.lstm_ws_floats(N, hh * ww, F), device=DEV) if use_ws else None
        K.convlstm_gates_fwd(gates, c, p[0], p[1], p[2], p[3], c_new, [h32, h16[..., F + 8:]], stats, ws=ws)
        same = bool((h16[..., F + 8:] == h32.to(torch.bfloat16)).all()) and float(h16[..., :F + 8].float().abs().max()) == 0.0
        out.append(('bf16io/lstm_h_%dx%d' % (hh, ww), 0.0 if same else 1.0, 0.5))
        if use_ws:        # backward: bf16 gate gradient == the fp32 one rounded; everything else unchanged
            dh, dcn = dev(rnd(rng, N, hh, ww, F)), dev(rnd(rng, N, hh, ww, F))
            res = []
            for dt in (torch.float32, torch.bfloat16):
                dg = torch.empty(N, hh, ww, 4 * F, device=DEV, dtype=dt)
                dcp = torch.empty(N, hh, ww, F, device=DEV)
                dpar = [torch.zeros(4 * F, device=DEV), torch.zeros(4 * F, device=DEV), torch.zeros(F, device=DEV), torch.zeros(F, device=DEV)]
                raw = torch.empty(N, hh, ww, 4 * F, device=DEV) if dt == torch.bfloat16 else None
                K.convlstm_gates_bwd(gates, c, p[0], p[1], p[2], p[3], stats, [dh], dcn, dg, dcp, dpar, ws=ws, dgates_raw=raw)
                res.append((dg, dcp, dpar))
            same = bool((res[1][0] == res[0][0].to(torch.bfloat16)).all()) and bool((res[1][1] == res[0][1]).all())
            out.append(('bf16io/lstm_dgates_%dx%d' % (hh, ww), 0.0 if same else 1.0, 0.5))
            for nm, a_, b_ in zip(('dg1', 'db1', 'dg2', 'db2'), res[1][2], res[0][2]):
                out.append(('bf16io/lstm_%s' % nm, rel_err(a_, b_.double().cpu()), 1e-5))
    # tile_channels into a bf16 slice
    z = dev(rnd(rng, 6, 8))
    t16 = torch.zeros(6, 16, 16, 24, device=DEV, dtype=torch.bfloat16)
    t32 = torch.zeros(6, 16, 16, 8, device=DEV)
    K.tile_channels(z, t16[..., 8:16], scale=0.5)
    K.tile_channels(z, t32, scale=0.5)
    same = bool((t16[..., 8:16] == t32.to(torch.bfloat16)).all()) and float(t16[..., :8].float().abs().max()) == 0.0 and \
        float(t16[..., 16:].float().abs().max()) == 0.0
    out.append(('bf16io/tile_channels', 0.0 if same else 1.0, 0.5))
    # weight gradient of a 5x5 gate convolution (8-wave x 8-tile shape) from bf16 x and / or dy tensors
    Nn, hh, ww, Cx, Cy = 4, 16, 16, 72, 128
    geom = K.ConvGeom((1, 5, 5), (1, 1, 1), (0, 2, 2))
    xr = rnd(rng, Nn, hh, ww, Cx).to(torch.bfloat16)                   # values exactly representable in bf16
    dyr = rnd(rng, Nn, hh, ww, Cy).to(torch.bfloat16)
    x32, dy32 = xr.float().to(DEV).contiguous(), dyr.float().to(DEV).contiguous()
    x16, dy16 = xr.to(DEV).contiguous(), dyr.to(DEV).contiguous()
    ref = torch.zeros(5, 5, Cx, Cy, device=DEV)
    refb = torch.zeros(Cy, device=DEV)
    K.conv(lib.CONV_WGRAD, geom, x32, dy32, ref, bias=refb, precision=1)
    for tag, xa, ya in (('x16', x16, dy32), ('y16', x32, dy16), ('x16_y16', x16, dy16)):
        dw = torch.zeros(5, 5, Cx, Cy, device=DEV)
        db = torch.zeros(Cy, device=DEV)
        K.conv(lib.CONV_WGRAD, geom, xa, ya, dw, bias=db, precision=1)
        out.append(('bf16io/wgrad_' + tag, rel_err(dw, ref.double().cpu()), 1e-5))
        out.append(('bf16io/wgrad_bias_' + tag, rel_err(db, refb.double().cpu()), 1e-5))
    torch.cuda.synchronize()
    return out


# ---------------------------------------------------------------------------------------------------------------
# Every (problem, tile, split-K) instantiation the shipped tuning tables select -- i.e. exactly the savp_conv launches bench.py
# runs at B=16 / N=32 -- against an independent fp64 reference at the table's own shapes.
# The reference is a loop over the kernel taps of fp64 matmuls on strided slices (torch on the device: rocBLAS dgemm, no
# convolution code of this repository, and no CPU convolution of 1.2 M-pixel batches).
# ---------------------------------------------------------------------------------------------------------------
def _taps_ref(mode, x, w, y, k, s, p):
    """x [N,D,H,W,Cx], w [kd,kh,kw,Cx,Cy], y [N,Do,Ho,Wo,Cy] device tensors of one dtype (fp64; fp32 where the datapath under test
    carries 1e-2: rocBLAS sgemm is exact fp32 and 20x faster than dgemm on these shapes).  FPROP: returns F(x); DGRAD: F^T(y);
    WGRAD: dW = x (*) y."""
    N, D, H, W, Cx = x.shape
    Do, Ho, Wo, Cy = y.shape[1:]
    need = [(o - 1) * st + kk for o, st, kk in zip((Do, Ho, Wo), s, k)]
    ext = [max(i + pb, nd) for i, pb, nd in zip((D, H, W), p, need)]
    xp = torch.zeros(N, ext[0], ext[1], ext[2], Cx, dtype=x.dtype, device=x.device)
    if mode != lib.CONV_DGRAD:
        xp[:, p[0]:p[0] + D, p[1]:p[1] + H, p[2]:p[2] + W] = x
    out = None
    if mode == lib.CONV_FPROP:
        out = torch.zeros(y.shape, dtype=x.dtype, device=x.device)
    elif mode == lib.CONV_WGRAD:
        out = torch.zeros(w.shape, dtype=x.dtype, device=x.device)
        y2 = y.reshape(-1, Cy)
    for a in range(k[0]):
        for u in range(k[1]):
            for v in range(k[2]):
                sl = (slice(None), slice(a, a + (Do - 1) * s[0] + 1, s[0]), slice(u, u + (Ho - 1) * s[1] + 1, s[1]),
                      slice(v, v + (Wo - 1) * s[2] + 1, s[2]))
                if mode == lib.CONV_FPROP:
                    out += xp[sl] @ w[a, u, v]
                elif mode == lib.CONV_WGRAD:
                    out[a, u, v] = xp[sl].reshape(-1, Cx).t() @ y2
                else:
                    xp[sl] += y @ w[a, u, v].t()
    if mode == lib.CONV_DGRAD:
        return xp[:, p[0]:p[0] + D, p[1]:p[1] + H, p[2]:p[2] + W].contiguous()
    return out


def check_conv_stats_fp32(seed=43):
    """Statistics epilogue of the ring kernel on an fp32 destination (round 3: the generator's down / upsample convolutions leave
    the sums of the instance norm that follows, taken around the bias since round 4, behind; the norm then runs its apply pass alone): output, sums, and
    savp_instnorm_act_fwd(stats_ready) against fused_instance_norm of the fp32 tap-loop convolution; every tile code that accepts the
    problem; a problem with partial tiles must be refused by savp_conv_stats_ok."""
    out = []
    rng = torch.Generator(device=DEV).manual_seed(seed)

    def rn(*shape):
        return torch.randn(*shape, generator=rng, device=DEV, dtype=torch.float32)

    cases = [  # name, mode, N, (H, W, Cx) x-side, (Ho, Wo, Cy) y-side, k, s, p
        ('pool4x4s2', lib.CONV_FPROP, 4, (32, 32, 40), (16, 16, 64), (1, 4, 4), (1, 2, 2), (0, 1, 1)),
        ('pool6x6s2', lib.CONV_FPROP, 3, (64, 64, 16), (32, 32, 32), (1, 6, 6), (1, 2, 2), (0, 2, 2)),
        ('up6x6s2', lib.CONV_DGRAD, 4, (32, 32, 32), (16, 16, 136), (1, 6, 6), (1, 2, 2), (0, 2, 2)),
        ('up6x6s2_b', lib.CONV_DGRAD, 2, (16, 16, 64), (8, 8, 136), (1, 6, 6), (1, 2, 2), (0, 2, 2)),
        ('conv3x3', lib.CONV_FPROP, 2, (64, 64, 32), (64, 64, 64), (1, 3, 3), (1, 1, 1), (0, 1, 1)),
        # |mean| >> std: a bias of ~300 on a convolution output of std ~1.  The sums are taken around the bias (round 4), so the one-pass
        # variance sum(v^2)/HW - mean(v)^2 keeps its digits; around 0 it would cancel to ~1e-2 relative in fp32 and the norm would be off
        ('conv3x3_bigbias', lib.CONV_FPROP, 2, (64, 64, 32), (64, 64, 64), (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ]
    for name, mode, N, (H, W, Cx), (Ho, Wo, Cy), k, s, p in cases:
        x32, y32 = rn(N, 1, H, W, Cx), rn(N, 1, Ho, Wo, Cy)
        w32 = rn(*k, Cx, Cy) * 0.1
        fprop = mode == lib.CONV_FPROP
        cdst = Cy if fprop else Cx
        bias = rn(cdst) + (300.0 if 'bigbias' in name else 0.0)
        ref = _taps_ref(mode, x32, w32, y32, k, s, p) + bias                     # [N, 1, h, w, cdst]
        ref = ref[:, 0]
        geom = K.ConvGeom(k, s, p)
        wp = (pack_wt(w32) if fprop else pack_wd(w32)).contiguous()
        gam, bet = rn(cdst) * 0.3 + 1, rn(cdst) * 0.3
        yn = torch.relu(O.fused_instance_norm(ref.cpu().double(), gam.cpu().double(), bet.cpu().double()))
        ran = 0
        for tile in (0, 0x311, 0x312, 0x321, 0x322, 0x711, 0x712, 0x721):
            xv, yv = x32[:, 0].clone(), y32[:, 0].clone()
            dst = yv if fprop else xv
            dst.fill_(float('nan'))
            stats = torch.zeros(N, cdst, 2, device=DEV, dtype=torch.float64)
            try:
                K.conv(mode, geom, xv, yv, wp, bias=bias, tile=tile, precision=1, w16=wp.to(torch.bfloat16), stats=stats)
            except RuntimeError:
                continue                                   # this tile cannot honour the statistics for this problem
            ran += 1
            tag = 'convstats_%s_t%x' % (name, tile)
            out.append((tag + '/out', rel_err(dst, ref), 1e-2))
            r2 = (ref - bias).reshape(N, -1, cdst)                   # the sums are taken around the bias
            out.append((tag + '/sum', rel_err(stats[..., 0], r2.sum(1)), 1e-2))
            out.append((tag + '/sumsq', rel_err(stats[..., 1], (r2 * r2).sum(1)), 1e-2))
            o1 = torch.empty_like(dst)
            mean, rstd = torch.empty(N, cdst, device=DEV), torch.empty(N, cdst, device=DEV)
            K.instnorm_act_fwd(dst, gam, bet, [o1], mean, rstd, act='relu', stats=stats, stats_shift=bias)
            out.append((tag + '/inorm_from_stats', rel_err(o1, yn), 1e-2))
            o2 = torch.empty_like(dst)
            K.instnorm_act_fwd(dst, gam, bet, [o2], mean, rstd, act='relu')
            out.append((tag + '/inorm_same_as_own_stats', rel_err(o1, o2), 1e-3))
        out.append(('convstats_%s/tiles_that_ran>=2' % name, 0.0 if ran >= 2 else float('inf'), 1.0))
        a = K._fill_conv_args(mode, geom, x32[:, 0], y32[:, 0], wp, bias, 0, 0, 0.0, None, 0, 0, 1, wp.to(torch.bfloat16), None)
        out.append(('convstats_%s/stats_ok' % name, 0.0 if lib.get().savp_conv_stats_ok(ctypes.byref(a)) == 1 else float('inf'), 1.0))
    # partial tiles (12 columns: not a multiple of the 8-column tile) and the fp32 datapath: refused, and the probe says so
    x32, y32, w32 = rn(2, 12, 12, 32), rn(2, 12, 12, 64), rn(1, 3, 3, 32, 64) * 0.1
    wp = pack_wt(w32).contiguous()
    geom = K.ConvGeom((1, 3, 3), (1, 1, 1), (0, 1, 1))
    for prec in (1, 0):
        a = K._fill_conv_args(lib.CONV_FPROP, geom, x32, y32, wp, None, 0, 0, 0.0, None, 0, 0, prec, wp.to(torch.bfloat16), None)
        ok = lib.get().savp_conv_stats_ok(ctypes.byref(a))
        if prec == 0:
            x32, y32 = rn(2, 16, 16, 32), rn(2, 16, 16, 64)
            a = K._fill_conv_args(lib.CONV_FPROP, geom, x32, y32, wp, None, 0, 0, 0.0, None, 0, 0, prec, wp.to(torch.bfloat16), None)
            ok = lib.get().savp_conv_stats_ok(ctypes.byref(a))
        out.append(('convstats_refused_prec%d/probe' % prec, 0.0 if ok == 0 else float('inf'), 1.0))
        st = torch.zeros(2, 64, 2, device=DEV, dtype=torch.float64)
        try:
            K.conv(lib.CONV_FPROP, geom, x32, y32, wp, precision=prec, w16=wp.to(torch.bfloat16), stats=st)
            out.append(('convstats_refused_prec%d/call' % prec, float('inf'), 1.0))
        except RuntimeError:
            out.append(('convstats_refused_prec%d/call' % prec, 0.0, 1.0))
    torch.cuda.synchronize()
    return out


def check_ring_weight_warmup_invisible(seed=47, option='ring_wwarm'):
    """Option ring_wwarm only moves bytes into the L2 ahead of time (csrc/conv_ring.hip): outputs with it on and off are bit-identical,
    for FPROP / DGRAD, one and several column tiles, several slab groups, and a bf16 source (LDS-DMA staged patch)."""
    out = []
    rng = torch.Generator(device=DEV).manual_seed(seed)

    def rn(*shape):
        return torch.randn(*shape, generator=rng, device=DEV, dtype=torch.float32)

    cases = [('lstm16', 8, 16, 16, 136, 256, 5, 0x711), ('lstm8', 8, 8, 8, 264, 512, 5, 0x311), ('lstm32', 4, 32, 32, 72, 128, 5, 0x712),
             ('head3', 4, 64, 64, 32, 64, 3, 0x321)]
    old = lib.get_option(option)
    try:
        for name, N, H, W, Cx, Cy, k, tile in cases:
            x, y = rn(N, H, W, Cx), rn(N, H, W, Cy)
            w = rn(k, k, Cx, Cy) * 0.1
            geom = K.ConvGeom((k, k), (1, 1), (k // 2, k // 2))
            for mode, mname in ((lib.CONV_FPROP, 'fprop'), (lib.CONV_DGRAD, 'dgrad')):
                wp = (pack_wt(w) if mode == lib.CONV_FPROP else pack_wd(w)).contiguous()
                w16 = wp.to(torch.bfloat16)
                for src16 in (False, True):
                    res = []
                    for on in (1, 0):
                        lib.set_option(option, on)
                        xv, yv = x.clone(), y.clone()
                        if src16:
                            if mode == lib.CONV_FPROP:
                                xv = xv.to(torch.bfloat16)
                            else:
                                yv = yv.to(torch.bfloat16)
                        dst = yv if mode == lib.CONV_FPROP else xv
                        dst.fill_(float('nan'))
                        K.conv(mode, geom, xv, yv, wp, tile=tile, splitk=1, precision=1, w16=w16)
                        res.append(dst.float().clone())
                    same = torch.equal(res[0], res[1]) and bool(torch.isfinite(res[0]).all())
                    out.append(('%s_%s_%s%s/bit_identical' % (option[5:], name, mname, '_src16' if src16 else ''), 0.0 if same else float('inf'), 1.0))
    finally:
        lib.set_option(option, old)
    torch.cuda.synchronize()
    return out


def check_wgrad_dma_staging(seed=53):
    """LDS-patch weight gradient of two bf16 operands (csrc/conv_wgrad_patch.hip): the LDS-DMA staged kernel (option wgp_dma, default) against
    the register-staged one on the same bf16 tensors and against the exact-fp32 generic kernel on the same values -- every workgroup shape, ragged
    planes (border tiles, halo rows / columns from the zero slot), a 16-channel group cut by Cx, channel-slice views of wider buffers,
    stride 2 and a 3-D layer."""
    out = []
    rng = torch.Generator(device=DEV).manual_seed(seed)
    bf = torch.bfloat16

    def rn(*shape):
        return torch.randn(*shape, generator=rng, device=DEV, dtype=torch.float32).to(bf)

    def sliced(t, off, width):
        big = torch.zeros(*t.shape[:-1], width, device=DEV, dtype=t.dtype)
        v = big[..., off:off + t.shape[-1]]
        v.copy_(t)
        return v

    cases = [  # name, N, D, H, W, Cx, Cy, k, s, p, (x slice offset, width), (y slice offset, width)
        ('lstm16', 6, 1, 16, 16, 136, 256, (1, 5, 5), (1, 1, 1), (0, 2, 2), None, None),
        ('lstm32', 3, 1, 32, 32, 72, 128, (1, 5, 5), (1, 1, 1), (0, 2, 2), (8, 96), None),
        ('lstm8', 5, 1, 8, 8, 264, 512, (1, 5, 5), (1, 1, 1), (0, 2, 2), None, (64, 640)),
        ('ragged', 3, 1, 20, 12, 40, 24, (1, 3, 3), (1, 1, 1), (0, 1, 1), (16, 64), (8, 40)),
        ('head3', 2, 1, 64, 64, 32, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1), None, None),
        ('stride2', 3, 1, 32, 32, 64, 128, (1, 4, 4), (1, 2, 2), (0, 1, 1), None, None),
        ('video3d', 2, 5, 16, 16, 32, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), None, None),
    ]
    old = lib.get_option('wgp_dma')
    try:
        for name, N, D, H, W, Cx, Cy, k, st, pd, xs, ys in cases:
            Do = (D + 2 * pd[0] - k[0]) // st[0] + 1
            Ho = (H + 2 * pd[1] - k[1]) // st[1] + 1
            Wo = (W + 2 * pd[2] - k[2]) // st[2] + 1
            x, y = rn(N, D, H, W, Cx), rn(N, Do, Ho, Wo, Cy)
            xv = sliced(x, *xs) if xs else x
            yv = sliced(y, *ys) if ys else y
            if D == 1:
                xv, yv, x32, y32 = xv[:, 0], yv[:, 0], x[:, 0].float().contiguous(), y[:, 0].float().contiguous()
            else:
                x32, y32 = x.float().contiguous(), y.float().contiguous()
            geom = K.ConvGeom(k, st, pd)
            wshape = (k if D > 1 else k[1:]) + (Cx, Cy)
            ref = torch.zeros(*wshape, device=DEV)
            K.conv(lib.CONV_WGRAD, geom, x32, y32, ref, precision=0)        # exact fp32 on the generic kernel: an independent reference
            res = []
            for on in (1, 0):
                lib.set_option('wgp_dma', on)
                dw = torch.full(wshape, 0.25, device=DEV)               # `+=` contract: accumulates into what is there
                K.conv(lib.CONV_WGRAD, geom, xv, yv, dw, precision=1)
                res.append(dw - 0.25)
            out.append(('wgp_dma_%s/vs_register_staging' % name, rel_err(res[0], res[1].double().cpu()), 1e-5))
            out.append(('wgp_dma_%s/vs_exact_fp32_kernel' % name, rel_err(res[0], ref.double().cpu()), 5e-5))
    finally:
        lib.set_option('wgp_dma', old)
    torch.cuda.synchronize()
    return out


def check_wide_thin_fprop(seed=59):
    """3x3 SAME convolutions from a feature tensor to a few channels (csrc/conv_thin.hip: wthin_fprop_kernel, taken under tile 0 in bf16 precision):
    against an fp64 tap loop on the bf16-rounded operands (the datapath's contract) and against the general kernels (option thin = 0) --
    the scratch-image head (32 -> 4, sigmoid, into a channel slice of a wider buffer), the mask convolution (56 -> 8), ragged planes,
    16 / 64 input channels, 1 / 3 output channels, LeakyReLU, no bias."""
    out = []
    rng = torch.Generator(device=DEV).manual_seed(seed)
    bf = torch.bfloat16

    def rn(*shape):
        return torch.randn(*shape, generator=rng, device=DEV, dtype=torch.float32)

    cases = [  # name, N, H, W, Cx, Cy, act, has bias, (y slice offset, width), x width
        ('scratch_head', 4, 64, 64, 32, 4, lib.ACT_SIGMOID, True, (44, 56), 32),
        ('masks', 3, 64, 64, 56, 8, 0, True, None, 56),
        ('ragged', 2, 20, 37, 16, 3, lib.ACT_LRELU, True, (1, 8), 24),
        ('wide64', 2, 33, 64, 64, 8, 0, False, None, 64),
        ('one_out', 5, 8, 8, 40, 1, 0, True, None, 40),
    ]
    old = lib.get_option('thin')
    try:
        for name, N, H, W, Cx, Cy, act, hb, ys, xw in cases:
            x = rn(N, H, W, Cx)
            xv = torch.zeros(N, H, W, xw, device=DEV)[..., :Cx]
            xv.copy_(x)
            w = rn(3, 3, Cx, Cy) * 0.2
            bias = rn(Cy) if hb else None
            wt = pack_wt(w)
            ref = _taps_ref(lib.CONV_FPROP, x.to(bf).double()[:, None], w.to(bf).double()[None], torch.empty(N, 1, H, W, Cy, device=DEV),
                            (1, 3, 3), (1, 1, 1), (0, 1, 1))[:, 0]
            if hb:
                ref = ref + bias.double()
            if act == lib.ACT_SIGMOID:
                ref = torch.sigmoid(ref)
            elif act == lib.ACT_LRELU:
                ref = torch.where(ref > 0, ref, 0.2 * ref)
            geom = K.ConvGeom((3, 3), (1, 1), (1, 1))
            res = []
            for on in (1, 0):
                lib.set_option('thin', on)
                if ys:
                    big = torch.full((N, H, W, ys[1]), 7.0, device=DEV)
                    yv = big[..., ys[0]:ys[0] + Cy]
                else:
                    big = None
                    yv = torch.full((N, H, W, Cy), float('nan'), device=DEV)
                K.conv(lib.CONV_FPROP, geom, xv, yv, wt, bias=bias, act=act, alpha=0.2, precision=1, w16=wt.to(bf))
                res.append(yv.clone())
                if big is not None and on == 1:                      # the rest of the wider buffer is untouched
                    mask = torch.ones(ys[1], dtype=torch.bool, device=DEV)
                    mask[ys[0]:ys[0] + Cy] = False
                    out.append(('wthin_%s/neighbours_untouched' % name, float((big[..., mask] - 7.0).abs().max()), 0.5))
            out.append(('wthin_%s/vs_fp64_taps' % name, rel_err(res[0], ref), 2e-5))
            out.append(('wthin_%s/vs_general_kernels' % name, rel_err(res[0], res[1].double().cpu()), 2e-5))
    finally:
        lib.set_option('thin', old)
    torch.cuda.synchronize()
    return out


def check_thin8_wide_dgrad(seed=61):
    """Data gradient of a 3x3 SAME convolution with 8 output channels (csrc/conv_thin.hip: thin8_wide_kernel, taken under tile 0 in bf16 precision;
    the mask convolution's): against an fp64 tap loop on the bf16-rounded operands and the general kernels (option thin = 0); beta 0 / 1
    (accumulating into the gradient already there), 56 / 44 / 64 / 13 input channels, ragged planes, a channel-slice destination."""
    out = []
    rng = torch.Generator(device=DEV).manual_seed(seed)
    bf = torch.bfloat16

    def rn(*shape):
        return torch.randn(*shape, generator=rng, device=DEV, dtype=torch.float32)

    cases = [('masks', 3, 64, 64, 56, 1, 56), ('kth', 2, 64, 64, 44, 1, 44), ('full', 2, 16, 33, 64, 0, 64), ('odd', 4, 21, 9, 13, 1, 20), ('beta0', 2, 40, 40, 56, 0, 72)]
    old = lib.get_option('thin')
    try:
        for name, N, H, W, Cx, beta, xw in cases:
            dy = rn(N, H, W, 8)
            w = rn(3, 3, Cx, 8) * 0.2
            wd = pack_wd(w)
            x0 = rn(N, H, W, Cx)
            ref = _taps_ref(lib.CONV_DGRAD, torch.zeros(N, 1, H, W, Cx, dtype=torch.float64, device=DEV), w.to(bf).double()[None], dy.to(bf).double()[:, None],
                            (1, 3, 3), (1, 1, 1), (0, 1, 1))[:, 0]
            if beta:
                ref = ref + x0.double()
            geom = K.ConvGeom((3, 3), (1, 1), (1, 1))
            res = []
            for on in (1, 0):
                lib.set_option('thin', on)
                big = torch.full((N, H, W, xw), 7.0, device=DEV)
                xv = big[..., xw - Cx:]
                xv.copy_(x0)
                K.conv(lib.CONV_DGRAD, geom, xv, dy, wd, beta=beta, precision=1, w16=wd.to(bf))
                res.append(xv.clone())
                if on == 1 and xw > Cx:
                    out.append(('thin8_%s/neighbours_untouched' % name, float((big[..., :xw - Cx] - 7.0).abs().max()), 0.5))
            out.append(('thin8_%s/vs_fp64_taps' % name, rel_err(res[0], ref), 2e-5))
            out.append(('thin8_%s/vs_general_kernels' % name, rel_err(res[0], res[1].double().cpu()), 2e-5))
    finally:
        lib.set_option('thin', old)
    torch.cuda.synchronize()
    return out


def check_tuning_table(precision='bf16', max_entries=None, seed=41):
    """Runs every entry of video_prediction_amd/tuning_gfx950_<precision>.json as that exact savp_conv call (mode, shapes, view
    strides, bias / w16 / act / beta / bf16 source / bf16 destination / statistics epilogue, the table's tile code and split-K)."""
    import ast
    import json
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    table = json.load(open(os.path.join(here, 'video_prediction_amd', 'tuning_gfx950_%s.json' % precision)))
    prec = 1 if precision == 'bf16' else 0
    out = []
    rng = torch.Generator(device=DEV).manual_seed(seed)

    def rn(*shape):
        return torch.randn(*shape, generator=rng, device=DEV, dtype=torch.float32)

    def strided(shape5, sw, dtype, fill):
        """[N,D,H,W,C] view with pixel stride sw (>= C: a channel slice of a wider buffer); sw == 0: the 2-D dense form [N,C]."""
        N, D, H, W, C = shape5
        if sw == 0:
            assert D == H == W == 1
            t = torch.empty(N, C, device=DEV, dtype=dtype)
            t.copy_(fill.reshape(N, C))
            return t
        big = torch.zeros(N, D, H, W, max(sw, C), device=DEV, dtype=dtype)
        v = big[..., :C]
        v.copy_(fill)
        return v if D > 1 else v[:, 0]

    for n_done, (kstr, (tile, sk)) in enumerate(sorted(table.items())):
        if max_entries and n_done >= max_entries:
            break
        key = ast.literal_eval(kstr)
        (mode, pr, N, D, H, W, Cx, Do, Ho, Wo, Cy, k, s, p, act, beta, x_sw, y_sw, has_b, has_w16) = key[:20]
        src16, out16, has_stats = (key[20:23] if len(key) >= 23 else (0, 0, False))
        if pr != prec:
            continue
        # round 4 key extensions: ((gap_at, gap),) = data gradient without `gap` destination channels (the keyed Cx counts the
        # LOGICAL ones); ('nb', c0, nc) = the norm-backward sums epilogue (same tiling question; run here without the sums, which
        # check_norm_bwd_stats_epilogue covers)
        gap = None
        for extra in key[23:]:
            if isinstance(extra, tuple) and len(extra) == 2 and not isinstance(extra[0], str):
                gap = (int(extra[0]), int(extra[1]))
        if gap is not None:
            if mode != lib.CONV_DGRAD:
                continue
            Cx = Cx + gap[1]                         # physical channel count of the destination and of the packed weights
        tag = 'table_%s/m%d_N%d_%dx%dx%dx%d->%dx%dx%dx%d_k%s_s%s_t%x_sk%d%s' % (
            precision, mode, N, D, H, W, Cx, Do, Ho, Wo, Cy, 'x'.join(map(str, k)), 'x'.join(map(str, s)), tile, sk,
            ('_a%d' % act if act else '') + ('_beta' if beta else '') + ('_s16' if src16 else '') + ('_o16' if out16 else ''))
        x32 = rn(N, D, H, W, Cx)
        y32 = rn(N, Do, Ho, Wo, Cy)
        w32 = rn(*k, Cx, Cy) * 0.1
        bf = torch.bfloat16
        # operand dtypes exactly as the keyed call had them
        x_dt = bf if ((mode != lib.CONV_DGRAD and src16) or (mode == lib.CONV_DGRAD and out16)) else torch.float32
        y_dt = bf if ((mode == lib.CONV_DGRAD and src16) or (mode != lib.CONV_DGRAD and out16)) else torch.float32
        if x_dt == bf:
            x32 = x32.to(bf).float()
        if y_dt == bf:
            y32 = y32.to(bf).float()
        xv = strided((N, D, H, W, Cx), x_sw, x_dt, x32.to(x_dt))
        yv = strided((N, Do, Ho, Wo, Cy), y_sw, y_dt, y32.to(y_dt))
        geom = K.ConvGeom(k, s, p)
        rdt = torch.float64 if prec == 0 else torch.float32        # reference precision: >= 3 orders below the tolerance either way
        x64, y64, w64 = x32.to(rdt), y32.to(rdt), w32.to(rdt)
        alpha = 0.2
        try:
            if mode == lib.CONV_WGRAD:
                ref = _taps_ref(mode, x64, w64, y64, k, s, p)
                dw = torch.zeros(w32.shape, device=DEV)
                db = torch.zeros(Cy, device=DEV) if has_b else None
                K.conv(mode, geom, xv, yv, dw, bias=db, splitk=sk, tile=tile, precision=prec)
                out.append((tag + '/wgrad', rel_err(dw, ref), 1e-2 if prec else 2e-5 * max(1.0, (N * Do * Ho * Wo / 65536.0) ** 0.5)))
                if has_b:
                    out.append((tag + '/wgrad_bias', rel_err(db, y64.reshape(-1, Cy).sum(0)), 1e-4))
                continue
            fprop = mode == lib.CONV_FPROP
            dst_shape = (N, Do, Ho, Wo, Cy) if fprop else (N, D, H, W, Cx)
            cdst = dst_shape[-1]
            ref = _taps_ref(mode, x64, w64, y64, k, s, p)
            bias = rn(cdst) if has_b else None
            dst = yv if fprop else xv
            old = None
            if beta:
                old = (y32 if fprop else x32)
                ref = ref + old.to(rdt)
            else:
                dst.fill_(float('nan'))
            if bias is not None:
                ref = ref + bias.to(rdt)
            aux = None
            if act == lib.ACT_LRELU:
                ref = torch.where(ref > 0, ref, alpha * ref)
            elif act == lib.ACT_SIGMOID:
                ref = torch.sigmoid(ref)
            elif act == lib.ACT_DLRELU_FROM_OUT:
                a32 = rn(*dst_shape)
                aux = strided(dst_shape, (y_sw if fprop else x_sw), torch.float32, a32)
                ref = ref * torch.where(a32 > 0, torch.ones_like(ref), torch.full_like(ref, alpha))
            wp = (pack_wt(w32) if fprop else pack_wd(w32)).contiguous()
            stats = torch.zeros(N, cdst, 2, device=DEV, dtype=torch.float64) if has_stats else None
            K.conv(mode, geom, xv, yv, wp, bias=bias, beta=beta, act=act, alpha=alpha, aux=aux, splitk=sk, tile=tile, precision=prec,
                   w16=wp.to(bf) if has_w16 else None, stats=stats, dst_gap=gap)
            got = dst.float().reshape(dst_shape)
            refd = ref.reshape(dst_shape)
            if gap is not None:                      # the gap's channels are not computed: compare the logical ones
                keep = [c for c in range(dst_shape[-1]) if not (gap[0] <= c < gap[0] + gap[1])]
                got, refd = got[..., keep], refd[..., keep]
            out.append((tag + ('/fprop' if fprop else '/dgrad') + ('_gap' if gap else ''), rel_err(got, refd), 1e-2 if prec else 2e-5))
            if has_stats:
                r2 = (ref - bias.to(rdt) if bias is not None else ref).reshape(N, -1, cdst)        # sums are taken around the bias
                out.append((tag + '/stats_sum', rel_err(stats[..., 0], r2.sum(1)), 1e-2))
                out.append((tag + '/stats_sumsq', rel_err(stats[..., 1], (r2 * r2).sum(1)), 1e-2))
        except RuntimeError as ex:
            out.append((tag + '/REFUSED:%s' % str(ex)[:60], float('inf'), 0.0))
    torch.cuda.synchronize()
    return out


def check_tiled_z_and_gapped_dgrad(seed=53):
    """ConvLSTM gate convolution backward without the tiled-z channels (csrc/tiled_z.hip, SavpConvArgs.dst_gap):
    (1) dz from savp_tiled_z_weff + savp_tiled_z_grad == autograd of conv2d([x | tile(z) | h], W) w.r.t. z (fp64 CPU, both the bf16
        and the fp32 gate-gradient layouts, planes 8x8 / 16x16 / 32x32 / 4x8, several images);
    (2) the DGRAD with dst_gap = (f, nz) writes exactly the x and h channels the full DGRAD writes (every ring tile code that takes
        the problem) and leaves the z channels of the destination untouched."""
    import torch.nn.functional as F_
    out = []
    rng = np.random.default_rng(seed)
    geom = K.ConvGeom((1, 5, 5), (1, 1, 1), (0, 2, 2))
    for (IMG, H, W, f, nz, dt) in ((3, 8, 8, 16, 8, torch.bfloat16), (2, 16, 16, 16, 8, torch.float32), (2, 32, 32, 32, 8, torch.bfloat16),
                                   (5, 4, 8, 16, 3, torch.bfloat16)):
        Cin, C = f + nz + f, 4 * f
        w = rnd(rng, 5, 5, Cin, C) * 0.1
        dy = (rnd(rng, IMG, H, W, C)).to(dt)
        z = torch.zeros(IMG, nz, dtype=torch.float64, requires_grad=True)
        a = torch.cat([torch.zeros(IMG, H, W, f, dtype=torch.float64), z[:, None, None, :].expand(IMG, H, W, nz),
                       torch.zeros(IMG, H, W, f, dtype=torch.float64)], dim=-1)
        g = F_.conv2d(a.permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1), padding=2).permute(0, 2, 3, 1)
        (g * dy.double()).sum().backward()
        wd_, dyd = dev(w), dy.to(DEV).contiguous()
        weff = torch.empty(25, C, 8, device=DEV)
        dz = torch.full((IMG, nz), 0.5, device=DEV)
        K.tiled_z_weff(wd_, geom, f, nz, weff)
        K.tiled_z_grad(dyd, weff, dz, beta=1)
        out.append(('tiled_z/dz_%dx%d_%s' % (H, W, 'bf16' if dt == torch.bfloat16 else 'f32'), rel_err(dz - 0.5, z.grad), 2e-5))
        dz2 = torch.full((IMG, nz), 7.0, device=DEV)
        K.tiled_z_grad(dyd, weff, dz2, beta=0)
        out.append(('tiled_z/beta0_%dx%d' % (H, W), rel_err(dz2, z.grad), 2e-5))
        out.append(('tiled_z/ok_%dx%d' % (H, W), 0.0 if K.tiled_z_ok(H, W, C, nz, geom) else 1.0, 0.5))
    # (2) gapped DGRAD vs the full one, bf16 gate gradient, the recipe's three plane sizes (N kept small)
    for (N, H, f, nz) in ((4, 32, 32, 8), (4, 16, 64, 8), (8, 8, 128, 8)):
        Cin, C = f + nz + f, 4 * f
        w = rnd(rng, 5, 5, Cin, C) * 0.05
        wd32 = dev(pack_wd(w))
        wd16 = wd32.to(torch.bfloat16)
        dy = rnd(rng, N, H, H, C).to(torch.bfloat16).to(DEV).contiguous()
        full = torch.zeros(N, H, H, Cin, device=DEV)
        K.conv(lib.CONV_DGRAD, geom, full, dy, wd32, w16=wd16, precision=1)
        took = 0
        for tile in (0, 0x311, 0x312, 0x321, 0x322, 0x711, 0x712, 0x721, 0x722, 0x1311, 0x1711):
            got = torch.full((N, H, H, Cin), 123.0, device=DEV)
            try:
                K.conv(lib.CONV_DGRAD, geom, got, dy, wd32, w16=wd16, precision=1, tile=tile, dst_gap=(f, nz))
            except RuntimeError:
                continue                      # this tile code does not take the problem
            took += 1
            keep = torch.cat([got[..., :f], got[..., f + nz:]], dim=-1)
            ref = torch.cat([full[..., :f], full[..., f + nz:]], dim=-1)
            out.append(('zless_dgrad/%dx%d_tile%x' % (H, H, tile), rel_err(keep, ref.double().cpu()), 1e-5))
            # the gap is never computed: it keeps its old contents, or is cleared with the rest of the block by a split-K launch's memset
            gapv = got[..., f:f + nz]
            untouched = bool((gapv == 123.0).all()) or bool((gapv == 0.0).all())
            out.append(('zless_dgrad/%dx%d_tile%x_gap_untouched_or_cleared' % (H, H, tile), 0.0 if untouched else 1.0, 0.5))
        out.append(('zless_dgrad/%dx%d_tiles_taken' % (H, H), 0.0 if took >= 3 else 1.0, 0.5))
    # a gap is refused outside the ring kernel (fp32 precision) instead of being ignored
    try:
        K.conv(lib.CONV_DGRAD, geom, torch.zeros(2, 8, 8, 40, device=DEV), torch.zeros(2, 8, 8, 64, device=DEV),
               torch.zeros(40, 25 * 64, device=DEV), precision=0, dst_gap=(16, 8))
        out.append(('zless_dgrad/fp32_refused', 1.0, 0.5))
    except RuntimeError:
        out.append(('zless_dgrad/fp32_refused', 0.0, 0.5))
    torch.cuda.synchronize()
    return out


def check_norm_bwd_stats_epilogue(seed=61):
    """savp_conv's nb_* epilogue (round 4): a DGRAD whose destination channels [0, C) are the output gradient of an instance norm + ReLU
    leaves that norm's backward sums sum(dy'), sum(dy' * xhat) behind.  Checked: the data gradient itself is unchanged (bit for bit),
    the sums equal an fp64 evaluation on the kernel's own output, and savp_instnorm_act_bwd(stats_ready) gives the dx / dgamma / dbeta
    of the two-launch path.  Cases: the gate convolution's DGRAD with the z gap (x slice = the first f channels) and a 3x3 head DGRAD
    (all channels)."""
    out = []
    rng = np.random.default_rng(seed)
    K.set_conv_precision('bf16')              # conv_stats_ok answers for the datapath in use
    for (name, N, H, k, Cin, Cout, f, gap) in (('gate16', 4, 16, 5, 136, 256, 64, (64, 8)), ('gate32', 2, 32, 5, 72, 128, 32, (32, 8)),
                                              ('head64', 2, 64, 3, 32, 64, 32, None)):
        geom = K.ConvGeom((1, k, k), (1, 1, 1), (0, k // 2, k // 2))
        w = rnd(rng, k, k, Cin, Cout) * 0.05
        wd32 = dev(pack_wd(w))
        wd16 = wd32.to(torch.bfloat16)
        dy = rnd(rng, N, H, H, Cout).to(torch.bfloat16).to(DEV).contiguous()
        x = dev(rnd(rng, N, H, H, f))                                    # the norm's input
        gamma, beta = dev(rnd(rng, f) * 0.3 + 1), dev(rnd(rng, f) * 0.3)
        mean, rstd = torch.empty(N, f, device=DEV), torch.empty(N, f, device=DEV)
        y = torch.empty(N, H, H, f, device=DEV)
        K.instnorm_act_fwd(x, gamma, beta, [y], mean, rstd, act='relu')
        plain = torch.zeros(N, H, H, Cin, device=DEV)
        K.conv(lib.CONV_DGRAD, geom, plain, dy, wd32, w16=wd16, precision=1, dst_gap=gap, splitk=1)
        ws = torch.zeros(N, f, 2, device=DEV, dtype=torch.float64)
        nb = dict(x=x, mean=mean, rstd=rstd, gamma=gamma, beta=beta, ws=ws, c0=0, act='relu')
        ok = K.conv_stats_ok(lib.CONV_DGRAD, geom, plain, dy, wd32, w16=wd16, dst_gap=gap, norm_bwd=dict(nb, ws=None))
        out.append(('nbstats/%s_offered' % name, 0.0 if ok else 1.0, 0.5))
        if not ok:
            continue
        got = torch.zeros(N, H, H, Cin, device=DEV)
        K.conv(lib.CONV_DGRAD, geom, got, dy, wd32, w16=wd16, precision=1, dst_gap=gap, norm_bwd=nb, splitk=1)
        out.append(('nbstats/%s_dx_bit_identical' % name, 0.0 if torch.equal(got, plain) else 1.0, 0.5))
        g = got[..., :f].double().cpu()
        xh = (x.double().cpu() - mean.double().cpu()[:, None, None, :]) * rstd.double().cpu()[:, None, None, :]
        mask = ((x.cpu() - mean.cpu()[:, None, None, :]) * rstd.cpu()[:, None, None, :] * gamma.cpu() + beta.cpu() > 0).double()
        d = g * mask
        ref = torch.stack([d.sum(dim=(1, 2)), (d * xh).sum(dim=(1, 2))], dim=-1)
        out.append(('nbstats/%s_sums' % name, rel_err(ws, ref), 2e-5))
        # split-K: both sums are linear in the accumulators, every split adds its share (the 8x8 gate DGRAD needs the splits to fill the chip)
        ws2 = torch.zeros(N, f, 2, device=DEV, dtype=torch.float64)
        got2 = torch.full((N, H, H, Cin), 123.0, device=DEV)
        try:
            K.conv(lib.CONV_DGRAD, geom, got2, dy, wd32, w16=wd16, precision=1, tile=0x311, splitk=2, dst_gap=gap, norm_bwd=dict(nb, ws=ws2))
            keep = [got2[..., :f]] + ([got2[..., gap[0] + gap[1]:]] if gap else [got2[..., f:]])
            refk = [plain[..., :f]] + ([plain[..., gap[0] + gap[1]:]] if gap else [plain[..., f:]])
            out.append(('nbstats/%s_splitk2_dx' % name, rel_err(torch.cat(keep, -1), torch.cat(refk, -1).double().cpu()), 1e-5))
            out.append(('nbstats/%s_splitk2_sums' % name, rel_err(ws2, ref), 2e-5))
        except RuntimeError as ex:
            out.append(('nbstats/%s_splitk2_refused_%s' % (name, str(ex)[-40:]), 1.0, 0.5))
        res = []
        for st in (None, ws):
            dx = torch.empty(N, H, H, f, device=DEV)
            dg, db = torch.zeros(f, device=DEV), torch.zeros(f, device=DEV)
            K.instnorm_act_bwd(x, gamma, beta, y, mean, rstd, [got[..., :f]], dx, dg, db, act='relu', stats=st)
            res.append((dx, dg, db))
        for nm, a_, b_ in zip(('dx', 'dgamma', 'dbeta'), res[1], res[0]):
            out.append(('nbstats/%s_%s' % (name, nm), rel_err(a_, b_.double().cpu()), 2e-5))
    K.set_conv_precision('f32')
    torch.cuda.synchronize()
    return out
