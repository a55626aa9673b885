"""Thin Python wrappers over the C ABI (one function per exported kernel).

Everything here takes torch device tensors (fp32, channels-last, channel stride 1 -- channel-slice views allowed),
derives pointer/stride arguments and launches on the current stream.  No arithmetic happens in Python.
"""
import ctypes

import torch

from . import lib
from .lib import (c_f32, c_i32, c_i64, c_vp, ptr)


def _nd(t):
    """(N, D, H, W, C, sn, sd, sh, sw) of a 4-D [N,H,W,C] or 5-D [N,D,H,W,C] tensor view."""
    if t.stride(-1) != 1 and t.shape[-1] != 1:
        raise ValueError('channel stride must be 1')
    if t.dim() == 4:
        N, H, W, C = t.shape
        sn, sh, sw, _ = t.stride()
        return N, 1, H, W, C, sn, 0, sh, sw
    if t.dim() == 5:
        N, D, H, W, C = t.shape
        sn, sd, sh, sw, _ = t.stride()
        return N, D, H, W, C, sn, sd, sh, sw
    if t.dim() == 2:
        N, C = t.shape
        return N, 1, 1, 1, C, t.stride(0), 0, 0, 0
    raise ValueError('expected 2-D, 4-D or 5-D tensor')


class ConvGeom(object):
    """Kernel/stride/pad-before triple of a forward cross-correlation (3-D form; 2-D uses kd=sd=1, pd=0)."""

    def __init__(self, k, s=(1, 1, 1), p=(0, 0, 0)):
        if len(k) == 2:
            k, s, p = (1,) + tuple(k), (1,) + tuple(s), (0,) + tuple(p)
        self.k, self.s, self.p = tuple(k), tuple(s), tuple(p)

    def out_dims(self, D, H, W, pad_after):
        return tuple((i + pb + pa - k) // s + 1 for i, k, s, pb, pa in zip((D, H, W), self.k, self.s, self.p, pad_after))


def conv(mode, geom, x, y, w, bias=None, beta=0, act=0, alpha=0.0, aux=None, splitk=0, tile=0):
    """mode FPROP: y = F(x) ; DGRAD: x = F^T(y) ; WGRAD: w += x (*) y.  See include/savp_hip.h."""
    lib.require_device(x, y, w, bias, aux)
    a = lib.SavpConvArgs()
    a.mode = mode
    N, D, H, W, Cx, a.x_sn, a.x_sd, a.x_sh, a.x_sw = _nd(x)
    N2, Do, Ho, Wo, Cy, a.y_sn, a.y_sd, a.y_sh, a.y_sw = _nd(y)
    if N != N2:
        raise ValueError('batch mismatch %d vs %d' % (N, N2))
    a.N, a.D, a.H, a.W, a.Cx = N, D, H, W, Cx
    a.Do, a.Ho, a.Wo, a.Cy = Do, Ho, Wo, Cy
    a.kd, a.kh, a.kw = geom.k
    a.sd, a.sh, a.sw = geom.s
    a.pd, a.ph, a.pw = geom.p
    a.beta, a.act, a.alpha, a.splitk, a.tile = int(beta), int(act), float(alpha), int(splitk), int(tile)
    a.x, a.y, a.w = x.data_ptr(), y.data_ptr(), w.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.aux = aux.data_ptr() if aux is not None else None
    taps = geom.k[0] * geom.k[1] * geom.k[2]
    if w.numel() != taps * Cx * Cy:
        raise ValueError('weight has %d elements, expected %d' % (w.numel(), taps * Cx * Cy))
    if aux is not None:
        dst = x if mode == lib.CONV_DGRAD else y
        if aux.stride() != dst.stride() or aux.shape != dst.shape:
            raise ValueError('aux must be addressed like the destination')
    lib.check(lib.get().savp_conv(lib.stream(), ctypes.byref(a)), 'savp_conv')


ACT_IDS = {None: 0, 'none': 0, 'relu': 1, 'lrelu': 2}


def view(t):
    """SavpView of a channels-last tensor [N, spatial..., C] whose spatial dims are jointly contiguous
    (true for channel slices of contiguous buffers)."""
    lib.require_device(t)
    if t.dim() < 3:
        t = t.reshape(t.shape[0], 1, t.shape[-1])
    if t.stride(-1) != 1 and t.shape[-1] != 1:
        raise ValueError('channel stride must be 1')
    sp = t.stride(-2)
    # verify pixel-linear addressing
    exp = sp
    for d in range(t.dim() - 2, 0, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            raise ValueError('view is not pixel-linear: shape %s stride %s' % (tuple(t.shape), t.stride()))
        exp *= t.shape[d]
    v = lib.SavpView()
    v.p, v.sn, v.sp = t.data_ptr(), t.stride(0), sp
    return v


def _hw(t):
    n = 1
    for d in t.shape[1:-1]:
        n *= d
    return n


def _set_views(arr, tensors):
    for i, t in enumerate(tensors):
        arr[i] = view(t)


def instnorm_act_fwd(x, gamma, beta, outs, mean, rstd, act='relu', alpha=0.0, eps=1e-6):
    a = lib.SavpInormArgs()
    a.N, a.HW, a.C = x.shape[0], _hw(x), x.shape[-1]
    a.act, a.alpha, a.eps = ACT_IDS[act], float(alpha), float(eps)
    a.x = view(x)
    a.gamma, a.beta = gamma.data_ptr(), beta.data_ptr()
    a.nout = len(outs)
    _set_views(a.out, outs)
    a.mean, a.rstd = mean.data_ptr(), rstd.data_ptr()
    lib.check(lib.get().savp_instnorm_act_fwd(lib.stream(), ctypes.byref(a)), 'savp_instnorm_act_fwd')


def instnorm_act_bwd(x, gamma, beta, out0, mean, rstd, dys, dx, dgamma, dbeta, dx_beta=0, act='relu', alpha=0.0,
                     eps=1e-6):
    a = lib.SavpInormArgs()
    a.N, a.HW, a.C = x.shape[0], _hw(x), x.shape[-1]
    a.act, a.alpha, a.eps = ACT_IDS[act], float(alpha), float(eps)
    a.x = view(x)
    a.gamma, a.beta = gamma.data_ptr(), beta.data_ptr()
    a.nout = 1
    a.out[0] = view(out0)
    a.mean, a.rstd = mean.data_ptr(), rstd.data_ptr()
    a.ndy = len(dys)
    _set_views(a.dy, dys)
    a.dx = view(dx)
    a.dx_beta = int(dx_beta)
    a.dgamma, a.dbeta = dgamma.data_ptr(), dbeta.data_ptr()
    lib.check(lib.get().savp_instnorm_act_bwd(lib.stream(), ctypes.byref(a)), 'savp_instnorm_act_bwd')


def _lstm_args(gates, c_prev, g1, b1, g2, b2, stats, eps, forget_bias):
    a = lib.SavpLstmArgs()
    N = gates.shape[0]
    F = gates.shape[-1] // 4
    a.N, a.HW, a.F = N, _hw(gates), F
    a.eps, a.forget_bias = float(eps), float(forget_bias)
    if not gates.is_contiguous():
        raise ValueError('gates must be contiguous')
    a.gates = gates.data_ptr()
    if c_prev is not None:
        a.c_prev = view(c_prev)
    a.gamma1, a.beta1, a.gamma2, a.beta2 = g1.data_ptr(), b1.data_ptr(), g2.data_ptr(), b2.data_ptr()
    a.mean1, a.rstd1, a.mean2, a.rstd2 = [s.data_ptr() for s in stats]
    return a


def convlstm_gates_fwd(gates, c_prev, g1, b1, g2, b2, c_new, hs, stats, eps=1e-6, forget_bias=1.0):
    a = _lstm_args(gates, c_prev, g1, b1, g2, b2, stats, eps, forget_bias)
    a.c_new = c_new.data_ptr()
    a.nh = len(hs)
    _set_views(a.h, hs)
    lib.check(lib.get().savp_convlstm_gates_fwd(lib.stream(), ctypes.byref(a)), 'savp_convlstm_gates_fwd')


def convlstm_gates_bwd(gates, c_prev, g1, b1, g2, b2, stats, dhs, dc_new, dgates, dc_prev, dparams, eps=1e-6,
                       forget_bias=1.0):
    a = _lstm_args(gates, c_prev, g1, b1, g2, b2, stats, eps, forget_bias)
    a.ndh = len(dhs)
    _set_views(a.dh, dhs)
    a.dc_new = dc_new.data_ptr() if dc_new is not None else None
    a.dgates = dgates.data_ptr()
    a.dc_prev = dc_prev.data_ptr() if dc_prev is not None else None
    a.dgamma1, a.dbeta1, a.dgamma2, a.dbeta2 = [d.data_ptr() for d in dparams]
    lib.check(lib.get().savp_convlstm_gates_bwd(lib.stream(), ctypes.byref(a)), 'savp_convlstm_gates_bwd')
