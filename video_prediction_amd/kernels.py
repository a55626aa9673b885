"""Thin Python wrappers over the C ABI (one function per exported kernel).

Everything here takes torch device tensors (fp32, channels-last, channel stride 1 -- channel-slice views allowed),
derives pointer/stride arguments and launches on the current stream.  No arithmetic happens in Python.
"""
import ctypes

import torch

from . import debug as _debug
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


PRECISION = {'value': 0}      # 0 = fp32 (exact), 1 = bf16 operands / fp32 accumulate; set with set_conv_precision


def set_conv_precision(name):
    PRECISION['value'] = {'f32': 0, 'fp32': 0, 'bf16': 1}[name]


AUTOTUNE = {'enabled': False, 'cache': {}, 'log': [], 'dist': None, 'rejected': [], 'check_all': False}
CONV_CALL_LOG = None
# developer aid (tests/tools/insitu_tune.py): in-step timing of the conv launches of chosen problems and their isolated candidate
# ranking -- {'mode': 'all' | 'targets' | 'rank', 'targets': set of problem keys, 'events': [(key, e0, e1)], 'ranked': {key: [...]}}
INSITU = None


def save_tuning(path):
    """Write the (problem -> (tile, splitk)) table found by the autotuner as JSON (keys are reprs of the problem tuples)."""
    import json
    with open(path, 'w') as f:
        json.dump({repr(k): list(v) for k, v in AUTOTUNE['cache'].items()}, f, indent=0, sort_keys=True)


def load_tuning(path):
    """Pre-load a table written by save_tuning(); problems found in it are never timed again.  Returns #entries."""
    import ast
    import json
    with open(path) as f:
        d = json.load(f)
    for k, v in d.items():
        AUTOTUNE['cache'][ast.literal_eval(k)] = (int(v[0]), int(v[1]))
    sync_tuning_table()
    return len(d)


def set_tuning_group(dist_module, force=False):
    """Data-parallel runs: every replica executes the same launch sequence, so all of them meet an unknown conv problem at the
    same call; each times the candidates on its own GPU, then rank 0's choice is broadcast and used by everyone (identical
    kernels -> identical per-rank step time; without this the ranks could settle on different tiles)."""
    AUTOTUNE['dist'] = dist_module if (dist_module is not None and (dist_module.get_world_size() > 1 or force)) else None
    sync_tuning_table()


def sync_tuning_table():
    """Make every replica's (problem -> tile) cache a copy of rank 0's.  The broadcast inside conv() fires on a LOCAL cache miss,
    so caches that differ between ranks (a table file readable on some ranks only, --retune on one rank) would make the ranks
    disagree on the number of collectives; identical caches + identical launch sequences keep the misses collective."""
    d = AUTOTUNE['dist']
    if d is None:
        return
    box = [dict(AUTOTUNE['cache']) if d.get_rank() == 0 else None]
    d.broadcast_object_list(box, src=0)
    AUTOTUNE['cache'].clear()
    AUTOTUNE['cache'].update(box[0])


def enable_autotune(flag=True):
    """First use of every distinct conv problem times the tile / split-K candidates on the device and caches the best.
    (Launch-time selection only: every candidate computes the same result up to fp32 summation order.)"""
    AUTOTUNE['enabled'] = bool(flag)


def _fill_conv_args(mode, geom, x, y, w, bias, beta, act, alpha, aux, splitk, tile, precision, w16=None, stats=None, dst_gap=None,
                    norm_bwd=None, w_frag=None, w_frag_il=None):
    a = lib.SavpConvArgs()
    a.mode = mode
    N, D, H, W, Cx, a.x_sn, a.x_sd, a.x_sh, a.x_sw = _nd(x)
    N2, Do, Ho, Wo, Cy, a.y_sn, a.y_sd, a.y_sh, a.y_sw = _nd(y)
    if N != N2:
        raise ValueError('batch mismatch %d vs %d' % (N, N2))
    a.N, a.D, a.H, a.W, a.Cx = N, D, H, W, Cx
    a.Do, a.Ho, a.Wo, a.Cy = Do, Ho, Wo, Cy
    if dst_gap is not None and dst_gap[1]:
        # the destination view spans ALL physical channels; the kernel computes the logical ones (SavpConvArgs.dst_gap)
        a.dst_gap_at, a.dst_gap = int(dst_gap[0]), int(dst_gap[1])
        if mode == lib.CONV_DGRAD:
            a.Cx = Cx - a.dst_gap
        elif mode == lib.CONV_FPROP:
            a.Cy = Cy - a.dst_gap
        else:
            raise ValueError('dst_gap: FPROP / DGRAD only')
    if norm_bwd is not None:
        # the destination's channels [c0, c0 + C) are the output gradient of an instance norm (+ activation) over nb['x']: the epilogue
        # leaves that norm's backward sums in nb['ws'] (SavpConvArgs.nb_*); instnorm_act_bwd(stats=ws) then runs its apply pass alone
        nb = norm_bwd
        xv = view(nb['x'])
        a.nb_x, a.nb_x_sn, a.nb_x_sp = xv.p, xv.sn, xv.sp
        a.nb_mean, a.nb_rstd = nb['mean'].data_ptr(), nb['rstd'].data_ptr()
        a.nb_gamma, a.nb_beta = nb['gamma'].data_ptr(), nb['beta'].data_ptr()
        a.nb_ws = nb['ws'].data_ptr() if nb.get('ws') is not None else 16        # 16: plan query only (conv_stats_ok)
        a.nb_c0, a.nb_nc = int(nb.get('c0', 0)), nb['x'].shape[-1]
        a.nb_act, a.nb_alpha = ACT_IDS[nb.get('act', 'relu')], float(nb.get('alpha', 0.0))
    a.kd, a.kh, a.kw = geom.k
    a.sd, a.sh, a.sw = geom.s
    a.pd, a.ph, a.pw = geom.p
    a.beta, a.act, a.alpha, a.splitk, a.tile = int(beta), int(act), float(alpha), int(splitk), int(tile)
    a.precision = PRECISION['value'] if precision is None else int(precision)
    a.x, a.y, a.w = x.data_ptr(), y.data_ptr(), w.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.aux = aux.data_ptr() if aux is not None else None
    a.w_bf16 = w16.data_ptr() if w16 is not None else None
    # bf16 activations (ring kernel): the source / destination tensor's dtype says so; strides are in elements of that dtype
    src, dst = (y, x) if mode == lib.CONV_DGRAD else (x, y)
    a.src_bf16 = int(src.dtype == torch.bfloat16)
    # WGRAD: `out_bf16` says that the y (output-gradient) operand holds bf16
    a.out_bf16 = int((y if mode == lib.CONV_WGRAD else dst).dtype == torch.bfloat16)
    a.stats = stats.data_ptr() if stats is not None else None
    a.w_frag = w_frag.data_ptr() if w_frag is not None else None     # the gate convolution's B-fragment pack (pack_gate_weights)
    a.w_frag_il = w_frag_il.data_ptr() if w_frag_il is not None else None     # ... with interleaved gate columns (the one-launch cell)
    taps = geom.k[0] * geom.k[1] * geom.k[2]
    if w.numel() != taps * Cx * Cy:
        raise ValueError('weight has %d elements, expected %d' % (w.numel(), taps * Cx * Cy))
    if aux is not None:
        dst = x if mode == lib.CONV_DGRAD else y
        if aux.stride() != dst.stride() or aux.shape != dst.shape:
            raise ValueError('aux must be addressed like the destination')
    return a


def _tune(a, mode, dst, w, return_all=False):
    """Time candidate (tile, splitk) pairs for this problem; returns the fastest (return_all: every (ms, tile, splitk), sorted)."""
    if lib.get().savp_conv_special(ctypes.byref(a)):
        return [] if return_all else (0, 0)   # a problem-specific kernel takes the call under tile 0: nothing to choose
    torch.cuda.synchronize()             # nothing else in flight (other streams would distort the timings)
    fn = lib.get().savp_conv
    st = lib.stream()
    tiles = (0x22, 0x21, 0x12, 0x11)
    if mode == lib.CONV_WGRAD:
        cands = [(t, sk) for t in tiles for sk in (0,)]
        scratch = torch.zeros_like(w)
        real_w, real_bias = a.w, a.bias
        a.w = scratch.data_ptr()
        a.bias = None                    # tuning runs must not accumulate into the real bias gradient
    else:
        splits = (1, 2, 4, 8) if a.act == 0 else (1,)
        # 0x1xx = generic gather kernel, 0x2xx = LDS patch kernel (rejected with EINVAL where it does not apply)
        # (0x6xx = patch kernel with 8 waves per workgroup; 0x1000 / 0x2000 = its LDS budget capped at 64 / 96 KB)
        # 0x3xx / 0x7xx = LDS-DMA ring kernel with 4 / 8 waves (the only one for bf16 activations / the cell epilogue)
        algs = (0x300, 0x700) if (a.src_bf16 or a.out_bf16 or a.stats or a.nb_ws or a.dst_gap) else (0x100, 0x200, 0x600, 0x1200, 0x1600, 0x2200, 0x2600, 0x300, 0x700)
        if a.out_bf16 or a.stats:
            splits = (1,)
        cands = [(alg | t, sk) for alg in algs for t in tiles for sk in splits]
        # ring kernel with wide slabs (8 / 9 k-steps per entry; 32 x 32 wave tile only; refused where the channel count offers none)
        cands += [(alg | 0x1000 | 0x11, sk) for alg in algs if (alg & 0x300) == 0x300 for sk in splits]
        real_dst, real_beta = (a.x if mode == lib.CONV_DGRAD else a.y), a.beta
        scratch = None
        real_stats = a.stats
        if a.stats:                      # tuning runs must not accumulate into the real statistics
            stats_scratch = torch.zeros(a.N * (a.Cx if mode == lib.CONV_DGRAD else a.Cy) * 2, device=dst.device, dtype=torch.float64)
            a.stats = stats_scratch.data_ptr()
        real_nb = a.nb_ws
        if a.nb_ws:                      # ... nor into the real norm-backward sums
            nb_scratch = torch.zeros(a.N * a.nb_nc * 2, device=dst.device, dtype=torch.float64)
            a.nb_ws = nb_scratch.data_ptr()
        if a.beta:                       # never accumulate tuning runs into the real destination
            scratch = torch.empty_like(dst)
            if scratch.stride() != dst.stride():
                scratch = torch.empty(dst.untyped_storage().size() // 4, device=dst.device)  # same addressing
            if mode == lib.CONV_DGRAD:
                a.x = scratch.data_ptr()
            else:
                a.y = scratch.data_ptr()
            a.beta = 0
    best, best_t = None, 1e30
    every = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for tile, sk in cands:
        a.tile, a.splitk = tile, sk
        a.ws, a.ws_bytes = None, 0
        if mode == lib.CONV_WGRAD or sk != 1:
            _conv_scratch(a, dst.device)      # the split-K / weight-gradient slices (without scratch the candidate would run unsplit / atomically)
        if fn(st, ctypes.byref(a)) != 0:
            continue
        t = 1e30
        for _ in range(3):                   # min over three groups of four launches: robust against clock ramps
            e0.record()
            for _ in range(4):
                fn(st, ctypes.byref(a))
            e1.record()
            e1.synchronize()
            t = min(t, e0.elapsed_time(e1))
        every.append((t, tile, sk))
        if t < best_t:
            best, best_t = (tile, sk), t
    # Time alone never picks a kernel: the winner (AUTOTUNE['check_all']: every candidate) must reproduce the automatic choice's result on
    # this very problem, else it is dropped and logged in AUTOTUNE['rejected'] (the failure dump of tests/conftest.py prints that log).
    bad = []
    if every and not return_all:
        check = sorted(every) if AUTOTUNE.get('check_all') else [e for e in every if (e[1], e[2]) == best]
        check = [e for e in check if (e[1], e[2]) != (0, 0)]
        if check:
            bad = _verify_candidates(a, mode, dst, w, [(t_, sk_) for _, t_, sk_ in check])
            if best in [b[0] for b in bad]:
                ok = [e for e in sorted(every) if (e[1], e[2]) not in [b[0] for b in bad]] if AUTOTUNE.get('check_all') else []
                best = (ok[0][1], ok[0][2]) if ok else (0, 0)
    if mode == lib.CONV_WGRAD:
        a.w, a.bias = real_w, real_bias
    else:
        a.beta = real_beta
        a.stats = real_stats
        a.nb_ws = real_nb
        if mode == lib.CONV_DGRAD:
            a.x = real_dst
        else:
            a.y = real_dst
    a.ws, a.ws_bytes = None, 0
    if return_all:
        return sorted(every)
    _tune.last_rejected = bad
    return best or (0, 0)


def _verify_candidates(a, mode, dst, w, cfgs):
    """Run the automatic choice (tile 0, split-K 0) and each (tile, splitk) of `cfgs` on identical, zero-initialised destinations of the
    call's own addressing and compare everything the call writes (destination, statistics / norm-backward sums).  Returns
    [((tile, splitk), reason)] for the candidates that differ by more than summation order can explain.  `a` is the argument block as
    _tune left it (destination / statistics already redirected away from the caller's tensors); restored on return."""
    fn, st = lib.get().savp_conv, lib.stream()
    keep = (a.x, a.y, a.w, a.stats, a.nb_ws, a.tile, a.splitk, a.ws, a.ws_bytes, a.beta)
    if mode == lib.CONV_WGRAD:
        like, out_bf16 = w, False
    else:
        like, out_bf16 = dst, dst.dtype == torch.bfloat16
    ext = 1 + sum((n - 1) * s_ for n, s_ in zip(like.shape, like.stride()) if n > 0)
    nst = a.N * (a.Cx if mode == lib.CONV_DGRAD else a.Cy) * 2 if keep[3] else 0
    nnb = a.N * a.nb_nc * 2 if keep[4] else 0

    def run(cfg):
        out = torch.zeros(ext, device=like.device, dtype=like.dtype)
        sts = torch.zeros(max(nst, 1), device=like.device, dtype=torch.float64)
        nbs = torch.zeros(max(nnb, 1), device=like.device, dtype=torch.float64)
        if mode == lib.CONV_WGRAD:
            a.w = out.data_ptr()
        elif mode == lib.CONV_DGRAD:
            a.x = out.data_ptr()
        else:
            a.y = out.data_ptr()
        if nst:
            a.stats = sts.data_ptr()
        if nnb:
            a.nb_ws = nbs.data_ptr()
        a.tile, a.splitk = cfg
        a.ws, a.ws_bytes = None, 0
        if mode == lib.CONV_WGRAD or cfg[1] != 1:
            _conv_scratch(a, like.device)
        rc = fn(st, ctypes.byref(a))
        return rc, out, sts, nbs

    def differs(got, ref, tol):
        g, r = got.double(), ref.double()
        scale = float(r.abs().max())
        err = float((g - r).abs().max()) if g.numel() else 0.0
        if not (err <= tol * max(scale, 1e-30)):        # NaN lands here too
            return 'max |diff| %.3g against max |ref| %.3g (tol %.1g)' % (err, scale, tol)
        return None

    bad = []
    try:
        rc, ref, ref_st, ref_nb = run((0, 0))
        if rc != 0:
            return []
        tol = 2e-2 if out_bf16 else (1e-2 if a.precision == 1 and mode == lib.CONV_WGRAD else 2e-4)
        for cfg in cfgs:
            rc, out, sts, nbs = run(cfg)
            why = 'rc %d' % rc if rc != 0 else (differs(out, ref, tol) or (nst and differs(sts, ref_st, 1e-3)) or (nnb and differs(nbs, ref_nb, 1e-3)))
            if why:
                bad.append((cfg, why))
    finally:
        a.x, a.y, a.w, a.stats, a.nb_ws, a.tile, a.splitk, a.ws, a.ws_bytes, a.beta = keep
    return bad


def _conv_scratch(a, device):
    """Hand savp_conv the caller-owned scratch this call can use (savp_conv_workspace_bytes: split-K slices, weight-gradient partials)."""
    need = lib.get().savp_conv_workspace_bytes(ctypes.byref(a))
    if need:
        ws = scratch(device, (need + 3) // 4)
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel() * 4


def conv(mode, geom, x, y, w, bias=None, beta=0, act=0, alpha=0.0, aux=None, splitk=0, tile=0, precision=None, w16=None, stats=None,
         dst_gap=None, norm_bwd=None, defer=False, w_frag=None, w_frag_il=None):
    """mode FPROP: y = F(x) ; DGRAD: x = F^T(y) ; WGRAD: w += x (*) y.  See include/savp_hip.h.  A torch.bfloat16 source /
    destination tensor selects the ring kernel's bf16 activation paths; `stats` [N, C_dst, 2] float64 (stats_ws: zeroed by the caller)
    receives the destination's per-(sample, channel) sum / sum of squares (bf16 destination only); dst_gap = (first, count):
    `count` destination channels from `first` on are left out (neither computed nor written)."""
    lib.require_device(w, bias, aux)
    lib.require_stats(stats, (norm_bwd or {}).get('ws'))
    lib.require_device_any(x, y)
    a = _fill_conv_args(mode, geom, x, y, w, bias, beta, act, alpha, aux, splitk, tile, precision, w16, stats, dst_gap, norm_bwd, w_frag, w_frag_il)
    if AUTOTUNE['enabled'] and tile == 0 and splitk == 0 and not lib.get().savp_conv_special(ctypes.byref(a)):
        key = (mode, a.precision, a.N, a.D, a.H, a.W, a.Cx, a.Do, a.Ho, a.Wo, a.Cy, geom.k, geom.s, geom.p, a.act, a.beta,
               a.x_sw, a.y_sw, bias is not None, w16 is not None, a.src_bf16, a.out_bf16, stats is not None)
        if a.dst_gap:
            key = key + ((a.dst_gap_at, a.dst_gap),)
        if norm_bwd is not None:
            key = key + (('nb', a.nb_c0, a.nb_nc),)
        cfg = AUTOTUNE['cache'].get(key)
        if cfg is None:
            dst = x if mode == lib.CONV_DGRAD else y
            cfg = _tune(a, mode, dst, w)
            for bad_cfg, why in getattr(_tune, 'last_rejected', []):
                AUTOTUNE['rejected'].append((key, {'cfg': list(bad_cfg), 'why': why}))
            if AUTOTUNE['dist'] is not None:
                t = torch.tensor([int(cfg[0]), int(cfg[1])], dtype=torch.int32, device=dst.device)
                AUTOTUNE['dist'].broadcast(t, src=0)
                cfg = tuple(int(v) for v in t.tolist())
            AUTOTUNE['cache'][key] = cfg
            AUTOTUNE['log'].append((key, cfg))
        a.tile, a.splitk = cfg
        if INSITU is not None:
            if INSITU['mode'] == 'rank':
                if key in INSITU['targets'] and key not in INSITU['ranked']:
                    INSITU['ranked'][key] = _tune(a, mode, x if mode == lib.CONV_DGRAD else y, w, return_all=True)
                    a.tile, a.splitk = cfg
            elif INSITU['mode'] == 'all' or key in INSITU['targets']:
                if mode == lib.CONV_WGRAD or a.splitk != 1:
                    _conv_scratch(a, w.device)     # the timed launch is the launch the step makes (split-K / weight-gradient slices included)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                lib.check(lib.get().savp_conv(lib.stream(), ctypes.byref(a)), 'savp_conv')
                e1.record()
                INSITU['events'].append((key, e0, e1))
                return
    if mode == lib.CONV_WGRAD or a.splitk != 1:
        # caller-owned scratch, sized by the library's own planner for the (tile, splitk) just chosen: split-K slices of FPROP / DGRAD,
        # the per-split dW slices of the deterministic weight gradient (include/savp_hip.h SavpConvArgs.ws)
        _conv_scratch(a, w.device)
    if CONV_CALL_LOG is not None:      # profiling aid (tests/conv_shape_profile.py): launch order -> problem shape
        CONV_CALL_LOG.append((mode, a.N, a.D, a.H, a.W, a.Cx, a.Do, a.Ho, a.Wo, a.Cy, tuple(geom.k), tuple(geom.s), a.tile, a.splitk))
    if defer:                          # the filled argument block (tile / split-K chosen) for a fused-operator entry point; nothing is launched
        return a
    rc = lib.get().savp_conv(lib.stream(), ctypes.byref(a))
    if rc:
        lib.check(rc, 'savp_conv(%s)' % describe_conv(a))


def describe_conv(a):
    """One line naming a convolution problem (for error messages and logs)."""
    return ('%s N=%d in=%dx%dx%dx%d out=%dx%dx%dx%d k=%dx%dx%d s=%dx%dx%d prec=%s x=%s%s y=%s%s tile=0x%x splitk=%d act=%d beta=%d bias=%d' %
            (('fprop', 'dgrad', 'wgrad')[a.mode], a.N, a.D, a.H, a.W, a.Cx, a.Do, a.Ho, a.Wo, a.Cy, a.kd, a.kh, a.kw, a.sd, a.sh, a.sw,
             'bf16' if a.precision == 1 else 'f32', 'bf16' if a.src_bf16 else 'f32', '/sw%d' % a.x_sw, 'bf16' if a.out_bf16 else 'f32',
             '/sw%d' % a.y_sw, a.tile, a.splitk, a.act, a.beta, 1 if a.bias else 0))


def conv_stats_ok(mode, geom, x, y, w, bias=None, w16=None, dst_gap=None, norm_bwd=None):
    """True when savp_conv would honour a `stats` buffer for this FPROP / DGRAD problem (bf16 precision, ring kernel, whole tiles):
    the instance norm behind the convolution can then skip its own statistics pass (instnorm_act_fwd(stats=...)).  With norm_bwd
    (ws absent): the same question for the norm-backward statistics epilogue."""
    a = _fill_conv_args(mode, geom, x, y, w, bias, 0, 0, 0.0, None, 0, 0, None, w16, None, dst_gap, norm_bwd)
    return bool(lib.get().savp_conv_stats_ok(ctypes.byref(a)))


ACT_IDS = {None: 0, 'none': 0, 'relu': 1, 'lrelu': 2}


def view(t, any_dtype=False):
    """SavpView of a channels-last tensor [N, spatial..., C] whose spatial dims are jointly contiguous
    (true for channel slices of contiguous buffers).  any_dtype: the entry point takes a bf16 mask for this view."""
    if any_dtype:
        lib.require_device_any(t)
    else:
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


class _Acc64(object):
    """Round 6: everything several workgroups add to -- parameter gradients of the norms, the z-LSTM's dW / db, loss and KL scalars -- is a
    FLOAT64 accumulator in the C ABI (a sum of fp32 partials is exact there, so the result does not depend on arrival order).  The engine
    hands over float64 tensors (ParamGroup.grad64, the float64 loss buffer).  A caller that still passes float32 gets the old contract
    ("accumulates into the tensor") through a zeroed float64 twin that is added back after the launch."""

    def __init__(self, *tensors):
        self.pairs, self.out = [], []
        for t in tensors:
            if t is None:
                self.out.append(None)
            elif t.dtype == torch.float64:
                lib.require_stats(t)
                self.out.append(t)
            else:
                lib.require_device(t)
                d = torch.zeros(t.shape, dtype=torch.float64, device=t.device)
                self.pairs.append((t, d))
                self.out.append(d)

    def ptr(self, i):
        t = self.out[i]
        return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)

    def addr(self, i):
        t = self.out[i]
        return t.data_ptr() if t is not None else None

    def finish(self, deferred=False):
        if deferred and self.pairs:
            raise TypeError('a deferred (fused) launch takes float64 accumulators (ParamGroup.grad64)')
        for t, d in self.pairs:
            t.add_(d.to(torch.float32))


def _hw(t):
    n = 1
    for d in t.shape[1:-1]:
        n *= d
    return n


def _set_views(arr, tensors, any_dtype=False):
    for i, t in enumerate(tensors):
        arr[i] = view(t, any_dtype)


def _bf16_mask(tensors):
    """Bit k set = tensor k holds bfloat16 (only the entry points that take such a mask accept bf16 views)."""
    m = 0
    for i, t in enumerate(tensors):
        if t.dtype == torch.bfloat16:
            m |= 1 << i
        elif t.dtype != torch.float32:
            raise TypeError('expected float32 or bfloat16, got %s' % t.dtype)
    return m


class Scratch(object):
    """Caller-owned scratch of the C ABI (SavpConvArgs.ws, the ws of savp_colsum / savp_dense_fwd): one buffer per device.  Every user
    writes its part before reading it and all users are launched on the compute stream, so one buffer serves them all.  Growing
    never frees: a captured hipGraph (SAVPEngine.graph) holds the pointers it was captured with, so a retired buffer stays
    allocated for the life of the process and old replays keep reading and writing valid memory."""

    def __init__(self, device):
        self.device, self.buf, self.retired = device, None, []

    def get(self, nfloats):
        n = (int(nfloats) + 1023) & ~1023
        if self.buf is None or self.buf.numel() < n:
            if self.buf is not None:
                self.retired.append(self.buf)
            self.buf = torch.empty(max(n, 8 << 20), device=self.device, dtype=torch.float32)
        if _debug.POISON['scratch']:     # developer mode: whoever reads a scratch element it has not written reads NaN
            _debug.poison_tensor(self.buf)
            _debug.COUNTS['scratch'] += 1
        return self.buf


_SCRATCH = {}
COLSUM_WS_FLOATS = 1024 * 4 * 256        # SAVP_COLSUM_WS_FLOATS of include/savp_hip.h


def scratch(device, nfloats):
    key = str(device)
    s = _SCRATCH.get(key)
    if s is None:
        s = _SCRATCH[key] = Scratch(device)
    return s.get(nfloats)


class ZeroArena(object):
    """Pre-zeroed scratch for the atomically accumulated per-(sample, channel) reductions of the coalesced norm / ConvLSTM
    kernels: every call takes a fresh all-zero slice, the whole arena is cleared by ONE memset when it is reset (the train
    step resets it once at its start) instead of one memset per call (~700 per SAVP step, each a launch of its own)."""

    def __init__(self, device, floats=16 << 20):
        self.buf = torch.zeros(floats, device=device)
        self.off = 0
        self.hi = 0          # high-water mark of `off` since the owner last cleared it (see replayed())

    def reset(self):
        self.buf.zero_()
        self.off = 0

    def take(self, n):
        n = (int(n) + 63) & ~63
        if self.off + n > self.buf.numel():
            if n > self.buf.numel():
                raise ValueError('zero arena too small for %d floats' % n)
            self.reset()
        v = self.buf[self.off:self.off + n]
        if _debug.POISON['arena'] and not torch.cuda.is_current_stream_capturing():
            _debug.COUNTS['arena'] += 1
            if bool(v.view(torch.int32).any()):      # developer mode: "all-zero" is a contract, check it (bit pattern: -0.0 counts as dirty)
                raise RuntimeError('zero arena: slice [%d, %d) handed out dirty' % (self.off, self.off + n))
        self.off += n
        self.hi = max(self.hi, self.off)
        return v

    def replayed(self, mark):
        """A captured launch sequence that begins with reset() and took slices up to `mark` (the value of `hi` at the end of its
        capture, `hi` cleared at its start) has just been replayed: the device memory is zero from `mark` on and used below it, whatever
        the host-side offset said -- eager takes continue at `mark`.  (Without this an eager caller between two replays -- the eval
        summary of a training loop -- could be handed slices the replay had left its sums in.)"""
        self.off = int(mark)


_ARENAS = {}


def zero_arena(device):
    key = str(device)
    a = _ARENAS.get(key)
    if a is None:
        a = _ARENAS[key] = ZeroArena(device)
    return a


def stats_ws(device, N, C):
    """An all-zero FLOAT64 [N, C, 2] reduction workspace from the step's zero arena: what conv(stats=...), the norm_bwd['ws'] epilogue and the
    coalesced instance-norm kernels accumulate their per-(sample, channel) sums in.  float64 because a sum of fp32 partials is exact there:
    the statistics do not depend on the order in which the workgroups' atomics arrive, so two runs of a step give the same bits."""
    return zero_arena(device).take(N * C * 4).view(torch.float64).view(N, C, 2)


def _inorm_ws(x):
    return stats_ws(x.device, x.shape[0], x.shape[-1])


def _set_ranges(c0_arr, nc_arr, ranges):
    """ranges: None or [(first channel, channel count), ...] per view (multiples of 4); None / (0, 0) = all channels."""
    for i, r in enumerate(ranges or ()):
        if r:
            c0_arr[i], nc_arr[i] = int(r[0]), int(r[1])


def instnorm_act_fwd(x, gamma, beta, outs, mean, rstd, act='relu', alpha=0.0, eps=1e-6, out_ranges=None, stats=None, stats_shift=None,
                     defer=False):
    """out_ranges: per output view the (first channel, count) slice of the normalised tensor it receives (default: all).
    stats: [N, C, 2] sum / sum of squares of x written by the producing convolution's epilogue (conv(..., stats=...)): the
    statistics pass is skipped.  stats_shift: that convolution's bias [C] (its sums are taken around the bias); None: no bias."""
    a = lib.SavpInormArgs()
    if stats is not None:
        lib.require_device(stats_shift)
        lib.require_stats(stats)
        a.ws, a.ws_clean, a.stats_ready = stats.data_ptr(), 1, 1
        a.stats_shift = stats_shift.data_ptr() if stats_shift is not None else None
    else:
        a.ws, a.ws_clean = _inorm_ws(x).data_ptr(), 1
    a.N, a.HW, a.C = x.shape[0], _hw(x), x.shape[-1]
    a.act, a.alpha, a.eps = ACT_IDS[act], float(alpha), float(eps)
    a.x = view(x)
    a.gamma, a.beta = gamma.data_ptr(), beta.data_ptr()
    a.nout = len(outs)
    _set_views(a.out, outs, any_dtype=True)
    a.out_bf16 = _bf16_mask(outs)
    _set_ranges(a.out_c0, a.out_nc, out_ranges)
    a.mean, a.rstd = mean.data_ptr(), rstd.data_ptr()
    if defer:
        return a
    lib.check(lib.get().savp_instnorm_act_fwd(lib.stream(), ctypes.byref(a)), 'savp_instnorm_act_fwd')


def instnorm_act_bwd(x, gamma, beta, out0, mean, rstd, dys, dx, dgamma, dbeta, dx_beta=0, act='relu', alpha=0.0,
                     eps=1e-6, dy_ranges=None, stats=None, defer=False):
    """out0 is not read (the activation mask is recomputed from x, mean, rstd, gamma, beta); dy_ranges: per gradient view the
    (first channel, count) slice of the output it is the gradient of (default: all channels).  stats: [N, C, 2] sums written by the
    convolution that produced dy (conv(..., norm_bwd=...)): the statistics pass is skipped."""
    a = lib.SavpInormArgs()
    if stats is not None:
        lib.require_stats(stats)
        a.ws, a.ws_clean, a.stats_ready = stats.data_ptr(), 1, 1
    else:
        a.ws, a.ws_clean = _inorm_ws(x).data_ptr(), 1
    a.N, a.HW, a.C = x.shape[0], _hw(x), x.shape[-1]
    a.act, a.alpha, a.eps = ACT_IDS[act], float(alpha), float(eps)
    a.x = view(x)
    a.gamma, a.beta = gamma.data_ptr(), beta.data_ptr()
    a.nout = 0
    a.mean, a.rstd = mean.data_ptr(), rstd.data_ptr()
    a.ndy = len(dys)
    _set_views(a.dy, dys)
    _set_ranges(a.dy_c0, a.dy_nc, dy_ranges)
    a.dx = view(dx, any_dtype=True)
    a.dx_bf16 = _bf16_mask([dx])
    a.dx_beta = int(dx_beta)
    acc = _Acc64(dgamma, dbeta)
    a.dgamma, a.dbeta = acc.addr(0), acc.addr(1)
    if defer:
        acc.finish(deferred=True)
        return a
    lib.check(lib.get().savp_instnorm_act_bwd(lib.stream(), ctypes.byref(a)), 'savp_instnorm_act_bwd')
    acc.finish()


def _lstm_args(gates, c_prev, g1, b1, g2, b2, stats, eps, forget_bias):
    a = lib.SavpLstmArgs()
    N = gates.shape[0]
    F = gates.shape[-1] // 4
    a.N, a.HW, a.F = N, _hw(gates), F
    a.eps, a.forget_bias = float(eps), float(forget_bias)
    if not gates.is_contiguous():
        raise ValueError('gates must be contiguous')
    a.gates = gates.data_ptr()
    a.gates_bf16 = int(gates.dtype == torch.bfloat16)
    if c_prev is not None:
        a.c_prev = view(c_prev)
    if g1 is None:                      # the cell without a normaliser (SavpLstmArgs.no_norm): pointwise gate math, no parameters / statistics
        a.no_norm = 1
        return a
    a.gamma1, a.beta1, a.gamma2, a.beta2 = g1.data_ptr(), b1.data_ptr(), g2.data_ptr(), b2.data_ptr()
    a.mean1, a.rstd1, a.mean2, a.rstd2 = [s.data_ptr() for s in stats]
    return a


LSTM_RED_FLOATS = 22       # include/savp_hip.h: SavpLstmArgs.ws_stats


def lstm_ws_floats(N, HW, F):
    """Scratch size (floats) that selects the coalesced three-pass ConvLSTM kernels (include/savp_hip.h); the small
    reduction workspace comes from the zero arena."""
    return N * F * HW


def _lstm_ws(a, gates, ws, ws_stats=None):
    if ws is not None:                 # scratch of the three-pass kernels; the one-launch kernels need none
        lib.require_device(ws)
        a.ws, a.ws_floats = ws.data_ptr(), ws.numel()
    if ws_stats is None:
        ws_stats = zero_arena(gates.device).take(a.N * a.F * LSTM_RED_FLOATS)
    a.ws_stats, a.ws_stats_clean = ws_stats.data_ptr(), 1


def lstm_stats_ws(device, N, F):
    """An all-zero reduction workspace [N*F*22 floats] of the ConvLSTM gate kernels from the step's zero arena (float64 sums, see stats_ws);
    its head, viewed as float64 [N, 4F, 2], is what savp_conv's `stats` epilogue fills for convlstm_gates_fwd(stats1=...)."""
    ws = zero_arena(device).take(N * F * LSTM_RED_FLOATS)
    return ws, ws[:N * 4 * F * 4].view(torch.float64).view(N, 4 * F, 2)


def convlstm_gates_fwd(gates, c_prev, g1, b1, g2, b2, c_new, hs, stats, eps=1e-6, forget_bias=1.0, ws=None, stats1=None, defer=False):
    """stats1: the workspace returned by lstm_stats_ws whose head the gate convolution's epilogue has already filled (the
    statistics pass over the gate tensor is skipped); needed for bf16 gates."""
    a = _lstm_args(gates, c_prev, g1, b1, g2, b2, stats, eps, forget_bias)
    a.c_new = c_new.data_ptr()
    if ws is not None or stats1 is not None:
        _lstm_ws(a, gates, ws, stats1)
        a.stats1_ready = int(stats1 is not None)
    a.nh = len(hs)
    _set_views(a.h, hs, any_dtype=True)
    a.h_bf16 = _bf16_mask(hs)
    if defer:
        return a
    lib.check(lib.get().savp_convlstm_gates_fwd(lib.stream(), ctypes.byref(a)), 'savp_convlstm_gates_fwd')


def convlstm_gates_bwd(gates, c_prev, g1, b1, g2, b2, stats, dhs, dc_new, dgates, dc_prev, dparams, eps=1e-6,
                       forget_bias=1.0, ws=None, dgates_raw=None, defer=False):
    """dgates may be a bfloat16 tensor (coalesced kernels, i.e. with ws): dgates_raw is then the fp32 scratch [N, HW, 4F] the raw
    gate gradients live in between the passes."""
    a = _lstm_args(gates, c_prev, g1, b1, g2, b2, stats, eps, forget_bias)
    if dgates.dtype == torch.bfloat16:
        if dgates_raw is None or dgates_raw.dtype != torch.float32 or dgates_raw.numel() < dgates.numel():
            raise ValueError('bf16 dgates need an fp32 dgates_raw scratch of the same size')
        a.dgates_bf16 = 1
        a.dgates_raw = dgates_raw.data_ptr()
    if ws is not None:
        _lstm_ws(a, gates, ws)
    a.ndh = len(dhs)
    _set_views(a.dh, dhs)
    a.dc_new = dc_new.data_ptr() if dc_new is not None else None
    a.dgates = dgates.data_ptr()
    a.dc_prev = dc_prev.data_ptr() if dc_prev is not None else None
    acc = _Acc64(*(dparams if not a.no_norm else ()))
    if not a.no_norm:
        a.dgamma1, a.dbeta1, a.dgamma2, a.dbeta2 = [acc.addr(i) for i in range(4)]
    if defer:
        acc.finish(deferred=True)
        return a
    lib.check(lib.get().savp_convlstm_gates_bwd(lib.stream(), ctypes.byref(a)), 'savp_convlstm_gates_bwd')
    acc.finish()


# ---- one host call per fused operator (include/savp_hip.h, csrc/fused_ops.hip): the two halves are built with defer=True ------------------
def fused_ok():
    """False while a measurement hook wants to see the single launches (in-step tuner, call log): the engine then issues the halves apart."""
    return INSITU is None and CONV_CALL_LOG is None


def convlstm_cell_fwd(conv_args, lstm_args):
    c = lib.SavpConvLstmCellArgs()
    c.conv, c.gates = conv_args, lstm_args
    lib.check(lib.get().savp_convlstm_cell_fwd(lib.stream(), ctypes.byref(c)), 'savp_convlstm_cell_fwd')


def convlstm_cell_bwd(conv_args, lstm_args):
    c = lib.SavpConvLstmCellArgs()
    c.conv, c.gates = conv_args, lstm_args
    lib.check(lib.get().savp_convlstm_cell_bwd(lib.stream(), ctypes.byref(c)), 'savp_convlstm_cell_bwd')


def conv_in_act_fwd(conv_args, norm_args):
    c = lib.SavpConvNormArgs()
    c.conv, c.norm = conv_args, norm_args
    lib.check(lib.get().savp_conv_in_act_fwd(lib.stream(), ctypes.byref(c)), 'savp_conv_in_act_fwd')


def conv_in_act_bwd(conv_args, norm_args):
    c = lib.SavpConvNormArgs()
    c.conv, c.norm = conv_args, norm_args
    lib.check(lib.get().savp_conv_in_act_bwd(lib.stream(), ctypes.byref(c)), 'savp_conv_in_act_bwd')


# ---------------------------------------------------------------------------------------------------------------
# util ops
# ---------------------------------------------------------------------------------------------------------------
def _L():
    return lib.get()


def _p(t):
    return t.data_ptr() if t is not None else None


def _rows(t):
    """Leading rows R for a view [R..., HW..., C] treated as [R, HW, C]: callers pass 3-D+ tensors [R, spatial.., C]."""
    return t.shape[0]


def tile_channels(z, out, scale=1.0, beta=0):
    """out[r, p, c] (=|+=) scale * z[r, c]; z [R, C] contiguous; out view [R, spatial..., C]."""
    lib.require_device(z)
    lib.require_device_any(out)
    R, C = z.shape
    if out.dtype == torch.bfloat16:
        if beta:
            raise ValueError('tile_channels into a bf16 view overwrites (beta=0 only)')
        lib.check(_L().savp_tile_channels_bf16(lib.stream(), _p(z), R, _hw(out), C, float(scale), view(out, any_dtype=True)),
                  'savp_tile_channels_bf16')
        return
    lib.check(_L().savp_tile_channels(lib.stream(), _p(z), R, _hw(out), C, float(scale), view(out), int(beta)),
              'savp_tile_channels')


def tiled_z_pad(nz):
    """Row length of the effective weights / partial sums of csrc/tiled_z.hip for nz latent channels (8 up to nz = 8, 32 up to nz = 32)."""
    return 8 if nz <= 8 else 32


def tiled_z_weff(w_hwio, geom, z0, nz, weff):
    """Effective weights [25, Cout, tiled_z_pad(nz)] of the tiled-z gradient from the master HWIO kernel (csrc/tiled_z.hip)."""
    lib.require_device(w_hwio, weff)
    kh, kw, cin, cout = w_hwio.shape[-4:]
    if not w_hwio.is_contiguous() or weff.numel() < 25 * cout * tiled_z_pad(nz):
        raise ValueError('tiled_z_weff: contiguous HWIO kernel and a [25, Cout, %d] buffer expected' % tiled_z_pad(nz))
    lib.check(lib.get().savp_tiled_z_weff(lib.stream(), w_hwio.data_ptr(), kh, kw, geom.p[1], geom.p[2], cin, cout, int(z0), int(nz),
                                          weff.data_ptr()), 'savp_tiled_z_weff')


def tiled_z_grad(dy, weff, dz, beta=1):
    """dz [IMG, nz] (+)= gradient of a latent tiled over the plane from the conv output gradient dy [IMG, H, W, C] (fp32 / bf16,
    contiguous) and tiled_z_weff's effective weights: the per-pixel data gradient of those channels is never formed."""
    lib.require_device(weff, dz)
    lib.require_device_any(dy)
    if not dy.is_contiguous() or not dz.is_contiguous() or dy.dim() != 4:
        raise ValueError('tiled_z_grad: contiguous dy [IMG, H, W, C] and dz [IMG, nz] expected')
    img, H, W, C = dy.shape
    need = lib.get().savp_tiled_z_workspace_bytes(img, C)
    ws = scratch(dy.device, (need + 3) // 4)
    lib.check(lib.get().savp_tiled_z_grad(lib.stream(), dy.data_ptr(), int(dy.dtype == torch.bfloat16), img, H, W, C, weff.data_ptr(),
                                          dz.shape[-1], dz.data_ptr(), int(beta), ws.data_ptr(), ws.numel() * 4), 'savp_tiled_z_grad')


def tiled_z_ok(H, W, C, nz, geom):
    """Does csrc/tiled_z.hip cover this plane / kernel (else the data gradient keeps the z channels)?"""
    k, p = geom.k, geom.p
    return (H >= 4 and 4 <= W <= 32 and (W & (W - 1)) == 0 and C % 64 == 0 and 1 <= nz <= 32 and k[0] == 1 and
            p[1] <= 2 and p[2] <= 2 and k[1] - 1 - p[1] <= 2 and k[2] - 1 - p[2] <= 2 and tuple(geom.s) == (1, 1, 1))


def colsum(x, out, scale=1.0, per_row=False):
    """out += scale * sum over pixels (and rows unless per_row) of x [R, spatial..., C]."""
    lib.require_device(x, out)
    ws = scratch(x.device, COLSUM_WS_FLOATS)
    lib.check(_L().savp_colsum(lib.stream(), view(x), x.shape[0], _hw(x), x.shape[-1], float(scale), _p(out), int(per_row), _p(ws),
                               ws.numel()), 'savp_colsum')


def _view_array(tensors):
    arr = (lib.SavpView * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = view(t)
    return arr


def select(mask, a, b, outs):
    lib.require_device(a, b, *outs)
    bv = view(b) if b is not None else lib.SavpView()
    lib.check(_L().savp_select(lib.stream(), a.shape[0], _hw(a), a.shape[-1], _p(mask), view(a), bv, len(outs),
                               _view_array(outs)), 'savp_select')


def select_bwd(mask, dins, db):
    lib.check(_L().savp_select_bwd(lib.stream(), db.shape[0], _hw(db), db.shape[-1], _p(mask), len(dins), _view_array(dins),
                                   view(db)), 'savp_select_bwd')


def gather_clips(src, dst, t_start, adjoint=False):
    """src [L,B,H,W,C] time-major (a batch-slice of a contiguous [L,B2,...] buffer is allowed), dst [B,clip,H,W,C]
    contiguous: dst[b,i] = src[t_start[b]+i, b]; adjoint accumulates dst back into src."""
    lib.require_device(src, dst)
    L, B = src.shape[:2]
    clip = dst.shape[1]
    E = src[0, 0].numel()
    assert dst.is_contiguous() and dst.shape[0] == B and src[0, 0].is_contiguous() and src.stride(1) == E
    lib.check(_L().savp_gather_clips(lib.stream(), _p(src), _p(dst), _p(t_start), B, clip, E, src.stride(0), int(adjoint)),
              'savp_gather_clips')


def sigmoid_bwd(dy, y, out):
    assert out.is_contiguous()
    lib.check(_L().savp_sigmoid_bwd(lib.stream(), view(dy), view(y), _p(out), dy.shape[0], _hw(dy), dy.shape[-1]), 'savp_sigmoid_bwd')


def axpby(a, x, b, y, out):
    assert x.is_contiguous() and out.is_contiguous() and (y is None or y.is_contiguous())
    lib.check(_L().savp_axpby(lib.stream(), x.numel(), float(a), _p(x), float(b), _p(y), _p(out)), 'savp_axpby')


def fill_view(out, value=0.0):
    lib.check(_L().savp_fill_view(lib.stream(), view(out), out.shape[0], _hw(out), out.shape[-1], float(value)), 'savp_fill_view')


def fold64(src64, dst32, idx=None):
    """dst32[i] += float(src64[i]); src64[i] = 0 for the elements listed in idx (int32 device tensor; None: all) -- ParamGroup.fold64."""
    lib.require_device(dst32)
    lib.require_stats(src64)
    if idx is not None:
        _require_i32(idx)
    n = idx.numel() if idx is not None else src64.numel()
    lib.check(_L().savp_fold_f64(lib.stream(), _p(idx), n, _p(src64), _p(dst32)), 'savp_fold_f64')


def state_pred_fwd(actions, states_in, gt, W, b, sa, gen):
    """savp_model.py:411-422,655-658: the robot-state recurrence over all T steps.  actions [T,N,na] or None, states_in [T,N,ns], gt int32
    [T,N]; fills sa [T,N,na+ns] = [actions_t | state_t] and gen [T,N,ns]."""
    T, N, ns = states_in.shape
    na = actions.shape[-1] if actions is not None else 0
    for t_ in (actions, states_in, W, b, sa, gen):
        if t_ is not None:
            lib.require_device(t_)
            assert t_.is_contiguous() and t_.dtype == torch.float32
    _require_i32(gt)
    assert gt.is_contiguous() and tuple(gt.shape) == (T, N) and tuple(W.shape) == (na + ns, ns) and tuple(sa.shape) == (T, N, na + ns)
    assert tuple(gen.shape) == (T, N, ns) and (actions is None or tuple(actions.shape[:2]) == (T, N))
    lib.check(_L().savp_state_pred_fwd(lib.stream(), T, N, na, ns, _p(actions), _p(states_in), _p(gt), _p(W), _p(b), _p(sa), _p(gen)),
              'savp_state_pred_fwd')


def state_pred_bwd(gt, W, sa, dgen, dW64, db64):
    """dgen [T,N,ns]: dL/dgen_state of the loss in, total gradient out; dW64 / db64 float64, accumulated."""
    T, N, ns = dgen.shape
    na = sa.shape[-1] - ns
    lib.require_stats(dW64)
    lib.require_stats(db64)
    _require_i32(gt)
    assert dgen.is_contiguous() and sa.is_contiguous() and dW64.numel() == (na + ns) * ns and db64.numel() == ns
    lib.check(_L().savp_state_pred_bwd(lib.stream(), T, N, na, ns, _p(gt), _p(W), _p(sa), _p(dgen), _p(dW64), _p(db64)),
              'savp_state_pred_bwd')


def adam(p, g, m, v, lr_t, beta1, beta2, eps=1e-8, gscale=1.0, lr_t_dev=None):
    """lr_t_dev: optional 1-element device tensor that overrides lr_t (graph replays with a changing rate)."""
    lib.check(_L().savp_adam(lib.stream(), p.numel(), _p(p), _p(g), _p(m), _p(v), float(lr_t), float(beta1), float(beta2),
                             float(eps), float(gscale), _p(lr_t_dev) if lr_t_dev is not None else None), 'savp_adam')


# ---------------------------------------------------------------------------------------------------------------
# cdna / composite
# ---------------------------------------------------------------------------------------------------------------
def cdna_kernels_fwd(raw, kern, kh, kw, K):
    lib.check(_L().savp_cdna_kernels_fwd(lib.stream(), _p(raw), _p(kern), raw.shape[0], kh, kw, K), 'savp_cdna_kernels_fwd')


def cdna_kernels_bwd(raw, dkern, draw, kh, kw, K):
    """dkern: the FLOAT64 accumulator [N, kh * kw, K] cdna_apply_bwd filled (include/savp_hip.h: savp_cdna_kernels_bwd)."""
    lib.require_stats(dkern)             # a float32 buffer here (the pre-round-5 contract) would be read as float64: twice its size
    lib.check(_L().savp_cdna_kernels_bwd(lib.stream(), _p(raw), _p(dkern), _p(draw), raw.shape[0], kh, kw, K),
              'savp_cdna_kernels_bwd')


def _cdna_args(img, kern, kh, kw, K):
    a = lib.SavpCdnaArgs()
    a.N, a.H, a.W, a.C = img.shape
    a.K, a.kh, a.kw = K, kh, kw
    a.img = view(img)
    a.kern = _p(kern)
    return a


def cdna_apply_fwd(img, kern, out, kh, kw, K):
    a = _cdna_args(img, kern, kh, kw, K)
    a.out = view(out)
    lib.check(_L().savp_cdna_apply_fwd(lib.stream(), ctypes.byref(a)), 'savp_cdna_apply_fwd')


def cdna_apply_bwd(img, kern, dout, dimg, dkern, kh, kw, K, dimg_beta=0):
    a = _cdna_args(img, kern, kh, kw, K)
    a.dout = view(dout)
    if dimg is not None:
        a.dimg = view(dimg)
    a.dimg_beta = int(dimg_beta)
    lib.require_stats(dkern)             # FLOAT64 [N, kh * kw, K], zeroed by the caller: the kernel adds float64 atomics over twice a float32 buffer's size
    a.dkern = _p(dkern)
    lib.check(_L().savp_cdna_apply_bwd(lib.stream(), ctypes.byref(a)), 'savp_cdna_apply_bwd')


def _comp_args(logits, timgs, C, M):
    a = lib.SavpCompositeArgs()
    a.N, a.HW, a.M, a.C = logits.shape[0], _hw(logits), M, C
    assert logits.is_contiguous()
    a.logits = _p(logits)
    a.logits_stride = logits.shape[-1]
    a.timgs = view(timgs)
    return a


def composite_fwd(logits, timgs, gen, masks=None, M=None, next_inputs=None):
    """logits [N,H,W,ls] (first M columns used), timgs view [N,H,W,M*C].  next_inputs = (gt_mask [N] int32, gt_image [N,H,W,C],
    [destination views]): also writes the NEXT step's input image (ground truth where gt_mask, else this step's composite)."""
    M = M or logits.shape[-1]
    a = _comp_args(logits, timgs, gen.shape[-1], M)
    a.gen = view(gen)
    a.masks = _p(masks)
    if next_inputs is not None:
        gt_mask, gt_img, dsts = next_inputs
        lib.require_device(gt_img)
        _require_i32(gt_mask)
        if len(dsts) > 2:
            raise ValueError('composite_fwd: at most two next-input views')
        a.nnext = len(dsts)
        for i, d in enumerate(dsts):
            a.next[i] = view(d)
        a.gt_mask = gt_mask.data_ptr()
        a.gt_img = view(gt_img)
    lib.check(_L().savp_composite_fwd(lib.stream(), ctypes.byref(a)), 'savp_composite_fwd')


def composite_bwd(logits, timgs, dgen, dlogits, drow, timgs_offset, M=None):
    """Writes dlogits and the whole gradient row `drow` [N,H,W,rowc] of the mask-conv input (zeros outside the
    transformed-image channels)."""
    M = M or logits.shape[-1]
    a = _comp_args(logits, timgs, dgen.shape[-1], M)
    a.dgen = view(dgen)
    assert dlogits.is_contiguous() and dlogits.shape == logits.shape
    a.dlogits = _p(dlogits)
    a.drow = view(drow)
    a.timgs_offset, a.row_channels = int(timgs_offset), drow.shape[-1]
    lib.check(_L().savp_composite_bwd(lib.stream(), ctypes.byref(a)), 'savp_composite_bwd')


# ---------------------------------------------------------------------------------------------------------------
# small ops
# ---------------------------------------------------------------------------------------------------------------
def lstm_z_fwd(zs, W, b, hout, gates, cs, forget_bias=1.0, init=None):
    """init: (c0, h0) [nz] each -- the learned initial state, tiled over the batch (learn_initial_state); None = zero state."""
    T, B, nz = zs.shape
    c0, h0 = init if init is not None else (None, None)
    lib.require_device(zs, W, b, hout, gates, cs, c0, h0)
    lib.check(_L().savp_lstm_z_fwd_init(lib.stream(), _p(zs), _p(W), _p(b), _p(hout), _p(gates), _p(cs), T, B, nz, float(forget_bias),
                                        _p(c0), _p(h0)), 'savp_lstm_z_fwd')


def lstm_z_bwd(zs, W, hout, gates, cs, dh_out, dzs, dW, db, forget_bias=1.0, init=None, dinit=None):
    """init = (c0, h0) as in lstm_z_fwd; dinit = (dc0, dh0) [nz]: their gradients are ADDED there."""
    T, B, nz = zs.shape
    c0, h0 = init if init is not None else (None, None)
    dc0, dh0 = dinit if dinit is not None else (None, None)
    lib.require_device(c0, h0)
    acc = _Acc64(dW, db, dc0, dh0)       # float64 accumulators (include/savp_hip.h)
    lib.check(_L().savp_lstm_z_bwd_init(lib.stream(), _p(zs), _p(W), _p(hout), _p(gates), _p(cs), _p(dh_out), _p(dzs), acc.addr(0), acc.addr(1),
                                        T, B, nz, float(forget_bias), _p(c0), _p(h0), acc.addr(2), acc.addr(3)), 'savp_lstm_z_bwd')
    acc.finish()


def gru_seq_fwd(A, A2, Wg, bg, Wc, bc, hout, ru, cand, n_in, h0=None):
    """GRUCell over time (include/savp_hip.h): A [T,B,I+U] with x in [..., :I]; fills A[..., I:], A2, hout, ru, cand.  h0 [U]: the initial
    state (learn_initial_state; None: zeros)."""
    T, B, Kd = A.shape
    U = Kd - n_in
    lib.require_device(A, A2, Wg, bg, Wc, bc, hout, ru, cand, h0)
    if h0 is not None:
        lib.check(_L().savp_gru_seq_fwd_init(lib.stream(), _p(A), _p(A2), _p(Wg), _p(bg), _p(Wc), _p(bc), _p(hout), _p(ru), _p(cand), T, B, n_in,
                                             U, _p(h0)), 'savp_gru_seq_fwd_init')
        return
    lib.check(_L().savp_gru_seq_fwd(lib.stream(), _p(A), _p(A2), _p(Wg), _p(bg), _p(Wc), _p(bc), _p(hout), _p(ru), _p(cand), T, B, n_in, U),
              'savp_gru_seq_fwd')


def gru_seq_bwd(A, Wg, Wc, ru, cand, dh_out, dGg, dGc, dA, n_in, dh0=None):
    """dh0 [U] float64: the initial state's gradient, accumulated (learn_initial_state)."""
    T, B, Kd = A.shape
    U = Kd - n_in
    lib.require_device(A, Wg, Wc, ru, cand, dh_out, dGg, dGc, dA)
    if dh0 is not None:
        lib.require_stats(dh0)
        lib.check(_L().savp_gru_seq_bwd_init(lib.stream(), _p(A), _p(Wg), _p(Wc), _p(ru), _p(cand), _p(dh_out), _p(dGg), _p(dGc), _p(dA), T, B,
                                             n_in, U, _p(dh0)), 'savp_gru_seq_bwd_init')
        return
    lib.check(_L().savp_gru_seq_bwd(lib.stream(), _p(A), _p(Wg), _p(Wc), _p(ru), _p(cand), _p(dh_out), _p(dGg), _p(dGc), _p(dA), T, B, n_in, U),
              'savp_gru_seq_bwd')


def lstm_seq_fwd(A, W, b, hout, gates, cs, n_in, forget_bias=1.0):
    """BasicLSTMCell over time (see include/savp_hip.h): A [T,B,I+U] with x in [..., :I]; fills A[..., I:], hout, gates, cs."""
    T, B, K = A.shape
    U = K - n_in
    lib.require_device(A, W, b, hout, gates, cs)
    lib.check(_L().savp_lstm_seq_fwd(lib.stream(), _p(A), _p(W), _p(b), _p(hout), _p(gates), _p(cs), T, B, n_in, U,
                                     float(forget_bias)), 'savp_lstm_seq_fwd')


def lstm_seq_bwd(A, W, gates, cs, dh_out, dG, dA, n_in, forget_bias=1.0):
    T, B, K = A.shape
    U = K - n_in
    lib.require_device(A, W, gates, cs, dh_out, dG, dA)
    lib.check(_L().savp_lstm_seq_bwd(lib.stream(), _p(A), _p(W), _p(gates), _p(cs), _p(dh_out), _p(dG), _p(dA), T, B, n_in, U,
                                     float(forget_bias)), 'savp_lstm_seq_bwd')


def kl_gauss(mu1, ls1_raw, mu2, ls2_raw, kl_out=None, klw=0.0, klw_dev=None, grads=None):
    """losses.kl_loss between two Gaussians (losses.py:61-67).  grads = (dmu1, dls1_raw, dmu2, dls2_raw): accumulated into."""
    rows = mu1.numel() // mu1.shape[-1]
    g = grads or (None, None, None, None)
    lib.require_device(mu1, ls1_raw, mu2, ls2_raw)
    acc = _Acc64(kl_out)
    lib.check(_L().savp_kl_gauss(lib.stream(), mu1.numel(), rows, _p(mu1), _p(ls1_raw), _p(mu2), _p(ls2_raw),
                                 acc.addr(0), float(klw),
                                 _p(klw_dev) if klw_dev is not None else None, *[(_p(t) if t is not None else None) for t in g]),
              'savp_kl_gauss')
    acc.finish()


def reparam_fwd(mu, ls_raw, eps, ls, z, kl_out=None):
    rows = mu.numel() // mu.shape[-1]
    acc = _Acc64(kl_out)
    lib.check(_L().savp_reparam_fwd(lib.stream(), mu.numel(), rows, _p(mu), _p(ls_raw), _p(eps), _p(ls), _p(z), acc.addr(0)),
              'savp_reparam_fwd')
    acc.finish()


def reparam_bwd(mu, ls_raw, eps, dz, klw, dmu, dls_raw, klw_dev=None):
    rows = mu.numel() // mu.shape[-1]
    lib.check(_L().savp_reparam_bwd(lib.stream(), mu.numel(), rows, _p(mu), _p(ls_raw), _p(eps), _p(dz), float(klw), _p(dmu),
                                    _p(dls_raw), _p(klw_dev) if klw_dev is not None else None), 'savp_reparam_bwd')


def lp_loss(pred, target, weight, loss_out=None, dpred=None, p2=False):
    """pred/target [R, ...]: rows may be strided (pred = one batch-half of a [T,2B,...] buffer); each row contiguous."""
    rows = pred.shape[0]
    row_len = pred[0].numel()
    assert pred[0].is_contiguous() and target[0].is_contiguous() and target.shape == pred.shape
    if dpred is not None:
        assert dpred.stride() == pred.stride()
    acc = _Acc64(loss_out)
    lib.check(_L().savp_lp_loss(lib.stream(), rows, row_len, pred.stride(0), target.stride(0), int(p2), _p(pred), _p(target),
                                float(weight), acc.addr(0), _p(dpred)), 'savp_lp_loss')
    acc.finish()


def tv_loss(flows, n_channels, s1, s2, weight, loss_out=None, dflows=None):
    """base_model.py:763-769 on the flows of some images: flows view [n, H, W, >= n_channels] (channel-contiguous, any image / pixel stride);
    loss_out float64 [1] += s1 * sum |d/dy| + s2 * sum |d/dx|; dflows (same strides) += weight * gradient."""
    lib.require_device_any(flows)
    assert flows.dtype == torch.float32 and flows.dim() == 4 and flows.stride(3) == 1 and flows.stride(1) == flows.shape[2] * flows.stride(2)
    if dflows is not None:
        assert dflows.stride() == flows.stride() and dflows.dtype == torch.float32
    lib.require_stats(loss_out)
    n, H, W, _ = flows.shape
    lib.check(_L().savp_tv_loss(lib.stream(), _p(flows), n, H, W, int(n_channels), flows.stride(0), flows.stride(2), float(s1), float(s2),
                                float(weight), _p(loss_out), _p(dflows)), 'savp_tv_loss')


def lsgan_loss(logits, label, weight, loss_out=None, dlogits=None, beta=0):
    acc = _Acc64(loss_out)
    lib.check(_L().savp_lsgan_loss(lib.stream(), logits.numel(), _p(logits), float(label), float(weight), acc.addr(0), _p(dlogits),
                                   int(beta)), 'savp_lsgan_loss')
    acc.finish()


def cosine_distance(f0, f1, weight, loss_out=None, df0=None, beta=0, eps=1e-10):
    assert f0.is_contiguous() and f1.is_contiguous()
    C = f0.shape[-1]
    acc = _Acc64(loss_out)
    lib.check(_L().savp_cosine_distance(lib.stream(), f0.numel() // C, C, _p(f0), _p(f1), float(weight), float(eps), acc.addr(0),
                                        _p(df0), int(beta)), 'savp_cosine_distance')
    acc.finish()


# ---------------------------------------------------------------------------------------------------------------
# weight prep
# ---------------------------------------------------------------------------------------------------------------
def pack_weights(src, wt=None, wd=None, scale=None, wt16=None, wd16=None):
    """src HWIO [..., Cx, Cy] contiguous -> wt [Cy, taps*Cx] and/or wd [Cx, taps*Cy]; scale = device scalar tensor;
    wt16/wd16: optional bf16 copies (torch.bfloat16 tensors)."""
    Cx, Cy = src.shape[-2], src.shape[-1]
    T = src.numel() // (Cx * Cy)
    lib.check(_L().savp_pack_weights(lib.stream(), _p(src), T, Cx, Cy, _p(scale), _p(wt), _p(wd), _p(wt16), _p(wd16)),
              'savp_pack_weights')


def pack_weights_batch(entries):
    """entries: [{'src', 'wt', 'wd', 'scale', 'wt16', 'wd16'}] (pack_weights' arguments) -- one launch per 32 layers."""
    for lo in range(0, len(entries), 32):
        part = entries[lo:lo + 32]
        arr = (lib.SavpPackItem * len(part))()
        for i, e in enumerate(part):
            src = e['src']
            lib.require_device(src)
            it = arr[i]
            it.src, it.scale = _p(src), _p(e.get('scale'))
            it.wt, it.wd, it.wt_bf16, it.wd_bf16 = _p(e.get('wt')), _p(e.get('wd')), _p(e.get('wt16')), _p(e.get('wd16'))
            it.Cx, it.Cy = src.shape[-2], src.shape[-1]
            it.T = src.numel() // (it.Cx * it.Cy)
        lib.check(_L().savp_pack_weights_batch(lib.stream(), len(part), arr), 'savp_pack_weights_batch')


def gate_weights_elems(taps, Cx, Cy):
    """bf16 elements of a gate convolution's B-fragment pack (0: the shape has none)."""
    return int(lib.get().savp_gate_weights_bytes(int(taps), int(Cx), int(Cy))) // 2


def pack_gate_weights(src, out, interleave=False):
    """src HWIO fp32 [..., Cx, Cy] -> out (torch.bfloat16, gate_weights_elems elements): MFMA B-fragment order (csrc/conv_gate.hip).
    interleave: the gate columns [i | j | f | o] regrouped per 8 channels (SavpConvArgs.w_frag_il: the one-launch cell)."""
    lib.require_device(src)
    Cx, Cy = src.shape[-2], src.shape[-1]
    taps = src.numel() // (Cx * Cy)
    if out.dtype != torch.bfloat16 or out.numel() != gate_weights_elems(taps, Cx, Cy):
        raise ValueError('gate weight pack: expected %d bf16 elements' % gate_weights_elems(taps, Cx, Cy))
    lib.check(_L().savp_pack_gate_weights(lib.stream(), _p(src), taps, Cx, Cy, _p(out), int(bool(interleave))), 'savp_pack_gate_weights')


def fold_pool(inp, out, k, adjoint=False):
    C = (out.numel() if adjoint else inp.numel()) // (k * k)
    lib.check(_L().savp_fold_pool(lib.stream(), _p(inp), _p(out), k, C, int(adjoint)), 'savp_fold_pool')


def fold_bilinear(inp, out, k, Cin, F, adjoint=False):
    lib.check(_L().savp_fold_bilinear(lib.stream(), _p(inp), _p(out), k, Cin, F, int(adjoint)), 'savp_fold_bilinear')


def sn_ws_size(K, C):
    return 8 + 2 * C + 2 * K + 2 * (C + 2)       # + the float64 forward accumulators (csrc/weight_prep.hip)


def sn_fwd(W, u, ws, u_new=None):
    C = W.shape[-1]
    K = W.numel() // C
    lib.check(_L().savp_sn_fwd(lib.stream(), _p(W), K, C, _p(u), _p(ws), _p(u_new)), 'savp_sn_fwd')


def _sn_items(entries, bwd):
    arr = (lib.SavpSnItem * len(entries))()
    for i, e in enumerate(entries):
        W = e['W']
        C = W.shape[-1]
        arr[i].W, arr[i].K, arr[i].C = W.data_ptr(), W.numel() // C, C
        arr[i].u, arr[i].ws = e['u'].data_ptr(), e['ws'].data_ptr()
        arr[i].u_new = e['u_new'].data_ptr() if e.get('u_new') is not None else None
        if bwd:
            arr[i].G, arr[i].dW, arr[i].beta = e['G'].data_ptr(), e['dW'].data_ptr(), int(e.get('beta', 0))
    return arr


def sn_fwd_batch(entries):
    """entries: [{'W', 'u', 'ws', 'u_new' (optional)}] -- savp_sn_fwd for up to 16 tensors in 4 launches."""
    for lo in range(0, len(entries), 16):
        part = entries[lo:lo + 16]
        arr = _sn_items(part, False)
        lib.check(_L().savp_sn_fwd_batch(lib.stream(), len(part), arr), 'savp_sn_fwd_batch')


def sn_bwd_batch(entries):
    """entries: [{'W', 'u', 'ws', 'G', 'dW', 'beta'}] -- savp_sn_bwd for up to 16 tensors in 4 launches."""
    for lo in range(0, len(entries), 16):
        part = entries[lo:lo + 16]
        arr = _sn_items(part, True)
        lib.check(_L().savp_sn_bwd_batch(lib.stream(), len(part), arr), 'savp_sn_bwd_batch')


def sn_bwd(W, u, ws, G, dW, beta=0):
    C = W.shape[-1]
    K = W.numel() // C
    lib.check(_L().savp_sn_bwd(lib.stream(), _p(W), K, C, _p(u), _p(ws), _p(G), _p(dW), int(beta)), 'savp_sn_bwd')


def dense_fwd(x, W, bias, out, scale=None):
    """out[M,C] = scale * x[M,K] @ W[K,C] + bias for few rows (K-sliced partial sums + a reduction launch); out contiguous."""
    M, Kd = x.shape
    C = W.shape[-1]
    assert out.is_contiguous() and x.stride(1) == 1 and W.is_contiguous()
    ws = scratch(x.device, 64 * M * C)           # K-slice partial sums, written before the reduce launch reads them
    lib.check(_L().savp_dense_fwd(lib.stream(), _p(x), x.stride(0), M, Kd, C, _p(W), _p(bias), _p(scale), _p(out), _p(ws), ws.numel()),
              'savp_dense_fwd')


# ---------------------------------------------------------------------------------------------------------------
# flow warp / DNA
# ---------------------------------------------------------------------------------------------------------------
def _warp_args(img, flows, K_):
    a = lib.SavpWarpArgs()
    a.N, a.H, a.W, a.C = img.shape
    a.K = K_
    a.img = view(img)
    assert flows.is_contiguous() and flows.shape[-1] == 2 * K_
    a.flows = _p(flows)
    return a


def image_warp_fwd(img, flows, out, K_):
    a = _warp_args(img, flows, K_)
    a.out = view(out)
    lib.check(_L().savp_image_warp_fwd(lib.stream(), ctypes.byref(a)), 'savp_image_warp_fwd')


def image_warp_bwd(img, flows, dout, dflows, dimg, K_):
    a = _warp_args(img, flows, K_)
    a.dout = view(dout)
    assert dflows.is_contiguous() and (dimg is None or dimg.is_contiguous())
    a.dflows, a.dimg = _p(dflows), _p(dimg)
    lib.check(_L().savp_image_warp_bwd(lib.stream(), ctypes.byref(a)), 'savp_image_warp_bwd')


def _dna_args(img, raw, kern, kh, kw, K_):
    a = lib.SavpDnaArgs()
    a.N, a.H, a.W, a.C = img.shape
    a.K, a.kh, a.kw = K_, kh, kw
    a.img = view(img)
    assert raw.is_contiguous() and kern.is_contiguous()
    a.raw, a.kern = _p(raw), _p(kern)
    return a


def dna_apply_fwd(img, raw, kern, out, kh, kw, K_):
    a = _dna_args(img, raw, kern, kh, kw, K_)
    a.out = view(out)
    lib.check(_L().savp_dna_apply_fwd(lib.stream(), ctypes.byref(a)), 'savp_dna_apply_fwd')


def dna_apply_bwd(img, raw, kern, dout, draw, dimg, kh, kw, K_, dimg_beta=0):
    a = _dna_args(img, raw, kern, kh, kw, K_)
    a.dout = view(dout)
    a.draw = _p(draw)
    if dimg is not None:
        a.dimg = view(dimg)
    a.dimg_beta = int(dimg_beta)
    lib.check(_L().savp_dna_apply_bwd(lib.stream(), ctypes.byref(a)), 'savp_dna_apply_bwd')


# ---------------------------------------------------------------------------------------------------------------
# ConvGRU gate blocks
# ---------------------------------------------------------------------------------------------------------------
def _gru_args(pre, h, gamma, beta, mean, rstd, F, eps):
    a = lib.SavpGruArgs()
    a.N, a.HW, a.F, a.eps = pre.shape[0], _hw(pre), F, float(eps)
    assert pre.is_contiguous()
    a.pre = _p(pre)
    a.h = view(h)
    a.gamma, a.beta, a.mean, a.rstd = _p(gamma), _p(beta), _p(mean), _p(rstd)
    return a


def convgru_gates_fwd(pre, h, gamma, beta, mean, rstd, u, rh, eps=1e-6):
    a = _gru_args(pre, h, gamma, beta, mean, rstd, pre.shape[-1] // 2, eps)
    a.u, a.rh = _p(u), view(rh)
    lib.check(_L().savp_convgru_gates_fwd(lib.stream(), ctypes.byref(a)), 'savp_convgru_gates_fwd')


def convgru_out_fwd(pre, h, gamma, beta, mean, rstd, u, outs, eps=1e-6):
    a = _gru_args(pre, h, gamma, beta, mean, rstd, pre.shape[-1], eps)
    a.u = _p(u)
    a.nout = len(outs)
    _set_views(a.out, outs)
    lib.check(_L().savp_convgru_out_fwd(lib.stream(), ctypes.byref(a)), 'savp_convgru_out_fwd')


def convgru_out_bwd(pre, h, gamma, beta, mean, rstd, u, dys, dpre, du, dh, dgamma, dbeta, eps=1e-6):
    a = _gru_args(pre, h, gamma, beta, mean, rstd, pre.shape[-1], eps)
    a.u = _p(u)
    a.ndy = len(dys)
    _set_views(a.dy, dys)
    a.dpre, a.du, a.dh = _p(dpre), _p(du), view(dh)
    acc = _Acc64(dgamma, dbeta)
    a.dgamma, a.dbeta = acc.addr(0), acc.addr(1)
    lib.check(_L().savp_convgru_out_bwd(lib.stream(), ctypes.byref(a)), 'savp_convgru_out_bwd')
    acc.finish()


def convgru_gates_bwd(pre, h, gamma, beta, mean, rstd, du, drh, dpre, dh, dgamma, dbeta, eps=1e-6):
    a = _gru_args(pre, h, gamma, beta, mean, rstd, pre.shape[-1] // 2, eps)
    a.du, a.drh, a.dpre, a.dh = _p(du), view(drh), _p(dpre), view(dh)
    acc = _Acc64(dgamma, dbeta)
    a.dgamma, a.dbeta = acc.addr(0), acc.addr(1)
    lib.check(_L().savp_convgru_gates_bwd(lib.stream(), ctypes.byref(a)), 'savp_convgru_gates_bwd')
    acc.finish()


GAN_TYPES = {'LSGAN': 0, 'GAN': 1, 'SNGAN': 2}


def gan_loss(logits, label, weight, gan_loss_type='LSGAN', loss_out=None, dlogits=None, beta=0):
    """losses.gan_loss (losses.py:29-54): value accumulated into loss_out, weighted gradient into dlogits."""
    acc = _Acc64(loss_out)
    lib.check(_L().savp_gan_loss(lib.stream(), logits.numel(), GAN_TYPES[gan_loss_type], _p(logits), float(label), float(weight),
                                 acc.addr(0), _p(dlogits), int(beta)), 'savp_gan_loss')
    acc.finish()


# ---------------------------------------------------------------------------------------------------------------
# evaluation metrics / best-of-N sampling fold (csrc/metrics.hip)
# ---------------------------------------------------------------------------------------------------------------
def _tb(x):
    """(time stride, batch stride, frame length) of a time-major [T, B, ...] tensor with contiguous frames."""
    if x.dim() < 3 or not x[0, 0].is_contiguous():
        raise ValueError('expected a time-major [T, B, ...] tensor with contiguous frames')
    return x.stride(0), x.stride(1), x[0, 0].numel()


def _require_i32(*tensors):
    for t in tensors:
        if t is not None and (not t.is_cuda or t.dtype != torch.int32):
            raise RuntimeError('expected an int32 device tensor, got %s on %s' % (t.dtype, t.device))


def frame_mse_psnr(a, b, mse=None, psnr=None):
    """metrics.py:5-10 per frame: mse / psnr [T, B] (either may be None)."""
    lib.require_device(a, b, mse, psnr)
    a_st, a_sb, inner = _tb(a)
    b_st, b_sb, _ = _tb(b)
    lib.check(_L().savp_frame_mse_psnr(lib.stream(), _p(a), a_st, a_sb, _p(b), b_st, b_sb, a.shape[0], a.shape[1], inner, _p(mse),
                                       _p(psnr)), 'savp_frame_mse_psnr')


def frame_ssim(a, b, out):
    """metrics.py:13-14 (tf.image.ssim, max_val 1) per frame of [T, B, H, W, C] tensors -> out [T, B]."""
    lib.require_device(a, b, out)
    a_st, a_sb, _ = _tb(a)
    b_st, b_sb, _ = _tb(b)
    T, B, H, W, C = a.shape
    lib.check(_L().savp_frame_ssim(lib.stream(), _p(a), a_st, a_sb, _p(b), b_st, b_sb, T, B, H, W, C, _p(out)), 'savp_frame_ssim')


def eval_accumulate(metric, vmin, vsum, vmax, cond_min, cond_max):
    """base_model.py:176-190 on contiguous [T, B] tensors; cond_* int32 [B] receive the per-sequence decisions."""
    lib.require_device(metric, vmin, vsum, vmax)
    _require_i32(cond_min, cond_max)
    T, B = metric.shape
    lib.check(_L().savp_eval_accumulate(lib.stream(), _p(metric), _p(vmin), _p(vsum), _p(vmax), _p(cond_min), _p(cond_max), T, B),
              'savp_eval_accumulate')


def select_batch(cond, x, out, mode=0):
    """mode 0: out[t,b] = x[t,b] where cond[b] (base_model.py:170-171); mode 1: out += x."""
    lib.require_device(x, out)
    _require_i32(cond)
    x_st, x_sb, inner = _tb(x)
    o_st, o_sb, _ = _tb(out)
    lib.check(_L().savp_select_batch(lib.stream(), _p(cond), _p(x), x_st, x_sb, _p(out), o_st, o_sb, x.shape[0], x.shape[1], inner,
                                     int(mode)), 'savp_select_batch')


def u8_frames_to_f32(frames_u8, out_tm):
    """uint8 [B, T, H, W, C] -> float32 time-major [T, B, H, W, C] / 255 (base_dataset.py:187 + transpose_batch_time)."""
    if not (frames_u8.is_cuda and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous()):
        raise RuntimeError('expected a contiguous uint8 device tensor')
    lib.require_device(out_tm)
    B, T = frames_u8.shape[:2]
    frame = frames_u8[0, 0].numel()
    lib.check(_L().savp_u8_frames_to_f32(lib.stream(), frames_u8.data_ptr(), out_tm.data_ptr(), B, T, frame), 'savp_u8_frames_to_f32')


class KernelTimer(object):
    """Kernel-only timing of single instrumented launches (bench.py): event pairs handed to the launcher through savp_prof_arm,
    stamped by the dispatch itself (the duration rocprofv3's kernel trace reports).  arm() before the launch, taken() after it
    (False: the call ran a kernel that does not take the pair; the pair is dropped); durations_us() after a synchronise."""

    def __init__(self):
        self.free, self.used, self.cur = [], [], None

    def _event(self):
        if self.free:
            return self.free.pop()
        ev = ctypes.c_void_p()
        lib.check(_L().savp_prof_event_create(ctypes.byref(ev)), 'savp_prof_event_create')
        return ev

    def arm(self):
        self.cur = (self._event(), self._event())
        lib.check(_L().savp_prof_arm(self.cur[0], self.cur[1]), 'savp_prof_arm')

    def taken(self):
        ok = not _L().savp_prof_armed()
        if ok:
            self.used.append(self.cur)
        else:
            _L().savp_prof_arm(None, None)
            self.free.extend(self.cur)
        self.cur = None
        return ok

    def durations_us(self):
        out = []
        us = ctypes.c_float()
        for e0, e1 in self.used:
            lib.check(_L().savp_prof_elapsed_us(e0, e1, ctypes.byref(us)), 'savp_prof_elapsed_us')
            out.append(us.value)
        self.free.extend(e for pair in self.used for e in pair)
        self.used = []
        return out

    def close(self):
        for e in self.free:
            _L().savp_prof_event_destroy(e)
        self.free = []
