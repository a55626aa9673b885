"""GPU parity checks: HIP kernels (through the C ABI) vs the CPU oracle on identical seeded inputs.

Used by tests/test_gpu_*.py (pytest -m gpu), by __graft_entry__.smoke() and by tests/run_gpu_checks.py (which
dumps every result to gpurun_out/ instead of stopping at the first failure).  Each check returns a list of
(name, err, tol) tuples; err is max|hip - oracle| / max(|oracle|) with the oracle evaluated in float64.
"""
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
]


def check_conv(cases=None, seed=0, tiles=(0,)):
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
        y = _ref_conv(x, w, k, s, p, pa) + b
        dy = rnd(rng, *y.shape)
        (y * dy).sum().backward()
        geom = K.ConvGeom(k, s, p)
        for tile in tiles:
            tag = name + ('' if tile == 0 else '_t%x' % tile)
            xd, wd_, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
            # FPROP (+bias)
            yd = torch.empty(y.shape, device=DEV, dtype=torch.float32)
            K.conv(lib.CONV_FPROP, geom, xd, yd, dev(pack_wt(w.detach())), bias=bd, tile=tile)
            out.append((tag + '/fprop', rel_err(yd, y), TOL_OP))
            # DGRAD
            dxd = torch.full(x.shape, float('nan'), device=DEV, dtype=torch.float32)
            K.conv(lib.CONV_DGRAD, geom, dxd, dyd, dev(pack_wd(w.detach())), tile=tile)
            out.append((tag + '/dgrad', rel_err(dxd, x.grad), TOL_OP))
            # WGRAD
            dwd = torch.zeros(w.shape, device=DEV, dtype=torch.float32)
            K.conv(lib.CONV_WGRAD, geom, xd, dyd, dwd, tile=tile)
            out.append((tag + '/wgrad', rel_err(dwd, w.grad), TOL_OP))
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
    out = []
    rng = np.random.default_rng(seed)
    for (N, H, W, F, zero_state) in [(2, 32, 32, 32, False), (3, 16, 16, 64, False), (2, 8, 8, 128, True), (2, 4, 6, 8, False)]:
        gates = (rnd(rng, N, H, W, 4 * F) * 1.5 + 0.3).requires_grad_(True)
        c = (torch.zeros(N, H, W, F, dtype=torch.float64) if zero_state else rnd(rng, N, H, W, F)).requires_grad_(True)
        g1 = (rnd(rng, 4 * F) * 0.3 + 1).requires_grad_(True)
        b1 = (rnd(rng, 4 * F) * 0.3).requires_grad_(True)
        g2 = (rnd(rng, F) * 0.3 + 1).requires_grad_(True)
        b2 = (rnd(rng, F) * 0.3).requires_grad_(True)
        cn, hn = _ref_lstm_gates(gates, c, g1, b1, g2, b2)
        dh1, dh2, dcn = rnd(rng, *hn.shape), rnd(rng, *hn.shape), rnd(rng, *cn.shape)
        ((hn * (dh1 + dh2)).sum() + (cn * dcn).sum()).backward()
        tag = 'lstm_%dx%dx%d%s' % (H, W, F, '_zero' if zero_state else '')
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
        dgates = torch.empty(N, H, W, 4 * F, device=DEV)
        dcp = torch.empty(N, H, W, F, device=DEV)
        dpar = [torch.zeros(4 * F, device=DEV), torch.zeros(4 * F, device=DEV), torch.zeros(F, device=DEV), torch.zeros(F, device=DEV)]
        K.convlstm_gates_bwd(gd, cd, p[0], p[1], p[2], p[3], stats, [dev(dh1), dev(dh2)], dev(dcn), dgates, dcp, dpar)
        out.append((tag + '/dgates', rel_err(dgates, gates.grad), 1e-4))
        out.append((tag + '/dc_prev', rel_err(dcp, c.grad), 1e-4))
        for nm, got, ref in zip(('dg1', 'db1', 'dg2', 'db2'), dpar, (g1.grad, b1.grad, g2.grad, b2.grad)):
            out.append((tag + '/' + nm, rel_err(got, ref), 1e-4))
    torch.cuda.synchronize()
    return out


ALL_CHECKS = [('conv', check_conv), ('conv_views', check_conv_views_and_epilogues), ('inorm', check_inorm),
              ('lstm', check_lstm)]


def failures(results):
    return [(n, e, t) for (n, e, t) in results if not (e <= t)]


def smoke():
    """One small hot-path invocation on cuda:0 checked against the oracle (used by __graft_entry__.smoke)."""
    res = check_conv(cases=('lstm5x5_32',)) + check_lstm()[:3]
    bad = failures(res)
    if bad:
        raise AssertionError('smoke parity failures: %r' % bad)
    return res
