"""ctypes binding of libsavp_hip.so (C ABI in include/savp_hip.h).

There is no fallback: if the library has not been built (``python __graft_entry__.py build`` or
``video_prediction_amd.build.build()``) every compute entry raises.  Tensors are torch tensors used purely as HBM
allocations; kernels receive raw device pointers, element strides and the current HIP stream.
"""
import ctypes
import os

import torch

from . import debug as _debug

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SAVP_LIB') or os.path.join(_HERE, 'libsavp_hip.so')      # SAVP_LIB: developer A/B of two builds

c_i32, c_i64, c_f32, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p

CONV_FPROP, CONV_DGRAD, CONV_WGRAD = 0, 1, 2
ACT_NONE, ACT_LRELU, ACT_SIGMOID, ACT_DLRELU_FROM_OUT = 0, 1, 2, 3


class SavpConvArgs(ctypes.Structure):
    _fields_ = [
        ('mode', c_i32),
        ('N', c_i32), ('D', c_i32), ('H', c_i32), ('W', c_i32), ('Cx', c_i32),
        ('Do', c_i32), ('Ho', c_i32), ('Wo', c_i32), ('Cy', c_i32),
        ('kd', c_i32), ('kh', c_i32), ('kw', c_i32),
        ('sd', c_i32), ('sh', c_i32), ('sw', c_i32),
        ('pd', c_i32), ('ph', c_i32), ('pw', c_i32),
        ('beta', c_i32), ('act', c_i32), ('alpha', c_f32), ('splitk', c_i32), ('tile', c_i32), ('precision', c_i32),
        ('x', c_vp), ('x_sn', c_i64), ('x_sd', c_i64), ('x_sh', c_i64), ('x_sw', c_i64),
        ('y', c_vp), ('y_sn', c_i64), ('y_sd', c_i64), ('y_sh', c_i64), ('y_sw', c_i64),
        ('w', c_vp), ('bias', c_vp), ('aux', c_vp), ('w_bf16', c_vp),
        ('src_bf16', c_i32), ('out_bf16', c_i32), ('stats', c_vp),
        ('ws', c_vp), ('ws_bytes', c_i64),
        ('dst_gap_at', c_i32), ('dst_gap', c_i32),
        ('nb_x', c_vp), ('nb_x_sn', c_i64), ('nb_x_sp', c_i64), ('nb_mean', c_vp), ('nb_rstd', c_vp), ('nb_gamma', c_vp), ('nb_beta', c_vp),
        ('nb_ws', c_vp), ('nb_c0', c_i32), ('nb_nc', c_i32), ('nb_act', c_i32), ('nb_alpha', c_f32),
        ('w_frag', c_vp), ('w_frag_il', c_vp),
    ]


_lib = None


def get():
    """Return the loaded library; raise loudly when it is missing (no CPU / eager fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libsavp_hip.so not found at %s -- build it first (python -c "import __graft_entry__ as g; g.build()"). '
                'video_prediction_amd has no non-HIP fallback.' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        lib.savp_version.restype = ctypes.c_char_p
        _declare(lib)
        _lib = lib
        _forward_env_options(lib)
    if _debug.POISON['lds']:
        return _LdsPoisonProxy(_lib)
    return _lib


def get_raw():
    """The library itself, never the debug proxy."""
    lib = get()
    return lib._lib if isinstance(lib, _LdsPoisonProxy) else lib


class _LdsPoisonProxy(object):
    """debug.POISON['lds']: every entry point whose first argument is the stream is preceded by savp_debug_poison_lds on that stream, so
    each kernel of the launch sequence starts on CUs whose LDS holds NaN bit patterns."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        at = getattr(fn, 'argtypes', None)
        if not name.startswith('savp_') or name.startswith(('savp_debug_', 'savp_prof_')) or not at or at[0] is not c_vp or \
                name in ('savp_allreduce_bucket',):
            return fn
        poison = self._lib.savp_debug_poison_lds

        def call(st, *args):
            poison(st, _debug.NAN_WORD, None)
            _debug.COUNTS['lds'] += 1
            return fn(st, *args)
        return call


# Kernel-selection switches of the library (include/savp_hip.h: savp_set_option).  The library itself never reads the environment;
# for A/B runs the host forwards SAVP_<NAME>=<int> here, once, when the library is loaded.
OPTION_NAMES = ('conv_ring', 's2dgrad', 'thin', 'wgp_cfg', 'wgp_split', 'inorm_min_hw', 'colsum_2stage', 'dense_legacy', 'cdna_legacy',
                'lstm_fused', 'ring_dma', 'lstm_q', 'ring_wwarm', 'wgp_dma', 'ring_early', 'gate_kernel', 'gate_alt', 'gate_cell', 'gate_wwarm', 'splitk_reduced')


def set_option(name, value):
    check(get().savp_set_option(name.encode(), int(value)), 'savp_set_option(%s)' % name)


def get_option(name):
    v = c_i32()
    check(get().savp_get_option(name.encode(), ctypes.byref(v)), 'savp_get_option(%s)' % name)
    return v.value


def _forward_env_options(lib):
    for name in OPTION_NAMES:
        v = os.environ.get('SAVP_' + name.upper())
        if v is not None:
            check(lib.savp_set_option(name.encode(), int(v)), 'savp_set_option(%s)' % name)


EXPORTS = {}     # name -> (restype, argtypes); filled by _declare, checked by tests against include/savp_hip.h


def _sig(lib, name, argtypes, restype=c_i32):
    try:
        fn = getattr(lib, name)
    except AttributeError:
        # developer A/B against an OLDER build (SAVP_LIB=...): entry points it predates stay undeclared and raise when called.
        # The shipped library must export everything (tests/test_abi_and_host.py checks the export set against include/savp_hip.h)
        if os.environ.get('SAVP_LIB'):
            return None
        raise
    fn.argtypes = argtypes
    fn.restype = restype
    EXPORTS[name] = fn
    return fn


def _declare(lib):
    P = ctypes.POINTER
    _sig(lib, 'savp_conv', [c_vp, P(SavpConvArgs)])
    _sig(lib, 'savp_conv_workspace_bytes', [P(SavpConvArgs)], restype=c_i64)
    _sig(lib, 'savp_conv_special', [P(SavpConvArgs)])
    _sig(lib, 'savp_gate_weights_bytes', [c_i32, c_i32, c_i32], restype=c_i64)
    _sig(lib, 'savp_conv_stats_ok', [P(SavpConvArgs)])
    _sig(lib, 'savp_set_option', [ctypes.c_char_p, c_i32])
    _sig(lib, 'savp_get_option', [ctypes.c_char_p, P(c_i32)])
    _sig(lib, 'savp_allreduce_bucket', [c_vp, c_vp, c_vp, c_i64])
    _sig(lib, 'savp_tiled_z_weff', [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp])
    _sig(lib, 'savp_tiled_z_grad', [c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp, c_i64])
    _sig(lib, 'savp_tiled_z_workspace_bytes', [c_i64, c_i32], restype=c_i64)
    for name, argtypes in _EXTRA_SIGS.items():
        _sig(lib, name, argtypes)


_EXTRA_SIGS = {}


def register(name, argtypes):
    """Declare one more C-ABI entry point (used by kernels.py so that signatures live next to their wrappers)."""
    _EXTRA_SIGS[name] = argtypes
    if _lib is not None:
        _sig(_lib, name, argtypes)


def source_id():
    """16 hex digits identifying the kernel sources + shipped tuning tables of this checkout (sha256 over csrc/*, include/*.h and
    tuning_gfx950_*.json in name order).  Measurement files under profiles/ carry it, and bench.py only quotes a counter profile
    (roofline.traffic) whose id equals the running checkout's: a profile of another build cannot pass for this one."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.dirname(_HERE)
    files = []
    for d, pat in ((os.path.join(_HERE, 'csrc'), ('.hip', '.h')), (os.path.join(root, 'include'), ('.h',)), (_HERE, ('.json',))):
        for f in sorted(os.listdir(d)):
            if f.endswith(pat) and (d != _HERE or f.startswith('tuning_gfx950_')):
                files.append(os.path.join(d, f))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed with code %d' % (what, rc))


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def require_device(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('video_prediction_amd kernels need device tensors (got %s); there is no CPU path' % t.device)
        if t.dtype != torch.float32:
            raise RuntimeError('expected float32, got %s' % t.dtype)


def require_stats(*tensors):
    """float64 [N, C, 2] reduction workspaces (kernels.stats_ws)."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('video_prediction_amd kernels need device tensors (got %s); there is no CPU path' % t.device)
        if t.dtype != torch.float64 or (t.data_ptr() & 7):
            raise RuntimeError('statistics workspaces are float64 (kernels.stats_ws), got %s' % t.dtype)


def require_device_any(*tensors):
    """Activations that may be fp32 or bf16 (ring conv kernel)."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('video_prediction_amd kernels need device tensors (got %s); there is no CPU path' % t.device)
        if t.dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError('expected float32 or bfloat16, got %s' % t.dtype)


class SavpView(ctypes.Structure):
    _fields_ = [('p', c_vp), ('sn', c_i64), ('sp', c_i64)]


class SavpInormArgs(ctypes.Structure):
    _fields_ = [
        ('N', c_i32), ('HW', c_i32), ('C', c_i32), ('act', c_i32), ('alpha', c_f32), ('eps', c_f32),
        ('x', SavpView), ('gamma', c_vp), ('beta', c_vp),
        ('nout', c_i32), ('out', SavpView * 4), ('mean', c_vp), ('rstd', c_vp),
        ('ndy', c_i32), ('dy', SavpView * 4), ('dx', SavpView), ('dx_beta', c_i32),
        ('dgamma', c_vp), ('dbeta', c_vp), ('ws', c_vp), ('ws_clean', c_i32),
        ('out_c0', c_i32 * 4), ('out_nc', c_i32 * 4), ('dy_c0', c_i32 * 4), ('dy_nc', c_i32 * 4), ('out_bf16', c_i32),
        ('stats_ready', c_i32), ('stats_shift', c_vp), ('dx_bf16', c_i32),
    ]


class SavpLstmArgs(ctypes.Structure):
    _fields_ = [
        ('N', c_i32), ('HW', c_i32), ('F', c_i32), ('eps', c_f32), ('forget_bias', c_f32),
        ('gates', c_vp), ('c_prev', SavpView),
        ('gamma1', c_vp), ('beta1', c_vp), ('gamma2', c_vp), ('beta2', c_vp),
        ('c_new', c_vp), ('nh', c_i32), ('h', SavpView * 4),
        ('mean1', c_vp), ('rstd1', c_vp), ('mean2', c_vp), ('rstd2', c_vp),
        ('ndh', c_i32), ('dh', SavpView * 4), ('dc_new', c_vp), ('dgates', c_vp), ('dc_prev', c_vp),
        ('dgamma1', c_vp), ('dbeta1', c_vp), ('dgamma2', c_vp), ('dbeta2', c_vp),
        ('ws', c_vp), ('ws_floats', ctypes.c_int64), ('ws_stats', c_vp), ('ws_stats_clean', c_i32),
        ('gates_bf16', c_i32), ('stats1_ready', c_i32), ('h_bf16', c_i32), ('dgates_bf16', c_i32), ('dgates_raw', c_vp), ('no_norm', c_i32),
    ]


class SavpConvLstmCellArgs(ctypes.Structure):
    _fields_ = [('conv', SavpConvArgs), ('gates', SavpLstmArgs)]


class SavpConvNormArgs(ctypes.Structure):
    _fields_ = [('conv', SavpConvArgs), ('norm', SavpInormArgs)]


register('savp_convlstm_cell_fwd', [c_vp, ctypes.POINTER(SavpConvLstmCellArgs)])
register('savp_convlstm_cell_bwd', [c_vp, ctypes.POINTER(SavpConvLstmCellArgs)])
register('savp_conv_in_act_fwd', [c_vp, ctypes.POINTER(SavpConvNormArgs)])
register('savp_conv_in_act_bwd', [c_vp, ctypes.POINTER(SavpConvNormArgs)])
register('savp_instnorm_act_fwd', [c_vp, ctypes.POINTER(SavpInormArgs)])
register('savp_instnorm_act_bwd', [c_vp, ctypes.POINTER(SavpInormArgs)])
register('savp_convlstm_gates_fwd', [c_vp, ctypes.POINTER(SavpLstmArgs)])
register('savp_convlstm_gates_bwd', [c_vp, ctypes.POINTER(SavpLstmArgs)])


class SavpCdnaArgs(ctypes.Structure):
    _fields_ = [
        ('N', c_i32), ('H', c_i32), ('W', c_i32), ('C', c_i32), ('K', c_i32), ('kh', c_i32), ('kw', c_i32),
        ('img', SavpView), ('kern', c_vp), ('out', SavpView), ('dout', SavpView),
        ('dimg', SavpView), ('dimg_beta', c_i32), ('dkern', c_vp),
    ]


class SavpCompositeArgs(ctypes.Structure):
    _fields_ = [
        ('N', c_i32), ('HW', c_i32), ('M', c_i32), ('C', c_i32),
        ('logits', c_vp), ('logits_stride', c_i32), ('timgs', SavpView), ('gen', SavpView), ('masks', c_vp),
        ('dgen', SavpView), ('dlogits', c_vp), ('drow', SavpView), ('timgs_offset', c_i32), ('row_channels', c_i32),
        ('nnext', c_i32), ('next', SavpView * 2), ('gt_mask', c_vp), ('gt_img', SavpView),
    ]


_PV = ctypes.POINTER(SavpView)
register('savp_tile_channels', [c_vp, c_vp, c_i64, c_i32, c_i32, c_f32, SavpView, c_i32])
register('savp_tile_channels_bf16', [c_vp, c_vp, c_i64, c_i32, c_i32, c_f32, SavpView])
register('savp_colsum', [c_vp, SavpView, c_i64, c_i32, c_i32, c_f32, c_vp, c_i32, c_vp, c_i64])
register('savp_select', [c_vp, c_i32, c_i32, c_i32, c_vp, SavpView, SavpView, c_i32, _PV])
register('savp_select_bwd', [c_vp, c_i32, c_i32, c_i32, c_vp, c_i32, _PV, SavpView])
register('savp_gather_clips', [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_i64, c_i32])
register('savp_sigmoid_bwd', [c_vp, SavpView, SavpView, c_vp, c_i64, c_i32, c_i32])
register('savp_axpby', [c_vp, c_i64, c_f32, c_vp, c_f32, c_vp, c_vp])
register('savp_fill_view', [c_vp, SavpView, c_i64, c_i32, c_i32, c_f32])
register('savp_u8_frames_to_f32', [c_vp, c_vp, c_vp, c_i32, c_i32, c_i64])
register('savp_frame_mse_psnr', [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp])
register('savp_frame_ssim', [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp])
register('savp_eval_accumulate', [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32])
register('savp_select_batch', [c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32])
register('savp_adam', [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp])
register('savp_cdna_kernels_fwd', [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32])
register('savp_cdna_kernels_bwd', [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32])
register('savp_cdna_apply_fwd', [c_vp, ctypes.POINTER(SavpCdnaArgs)])
register('savp_cdna_apply_bwd', [c_vp, ctypes.POINTER(SavpCdnaArgs)])
register('savp_composite_fwd', [c_vp, ctypes.POINTER(SavpCompositeArgs)])
register('savp_composite_bwd', [c_vp, ctypes.POINTER(SavpCompositeArgs)])
register('savp_lstm_z_fwd', [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32])
register('savp_lstm_seq_fwd', [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32])
register('savp_lstm_seq_bwd', [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32])
register('savp_prof_event_create', [ctypes.POINTER(c_vp)])
register('savp_prof_event_destroy', [c_vp])
register('savp_prof_arm', [c_vp, c_vp])
register('savp_prof_armed', [])
register('savp_prof_elapsed_us', [c_vp, c_vp, ctypes.POINTER(c_f32)])
register('savp_kl_gauss', [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp])
register('savp_lstm_z_bwd', [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32])
register('savp_gru_seq_fwd', [c_vp] * 10 + [c_i32] * 4)
register('savp_gru_seq_bwd', [c_vp] * 10 + [c_i32] * 4)
register('savp_gru_seq_fwd_init', [c_vp] * 10 + [c_i32] * 4 + [c_vp])
register('savp_gru_seq_bwd_init', [c_vp] * 10 + [c_i32] * 4 + [c_vp])
register('savp_lstm_z_fwd_init', [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp])
register('savp_lstm_z_bwd_init', [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp])
register('savp_reparam_fwd', [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp])
register('savp_reparam_bwd', [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp])
register('savp_lp_loss', [c_vp, c_i64, c_i64, c_i64, c_i64, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp])
register('savp_lsgan_loss', [c_vp, c_i32, c_vp, c_f32, c_f32, c_vp, c_vp, c_i32])
register('savp_cosine_distance', [c_vp, c_i64, c_i32, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp, c_i32])
register('savp_pack_weights', [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp])
register('savp_fold_pool', [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32])
register('savp_fold_bilinear', [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32])
class SavpPackItem(ctypes.Structure):
    _fields_ = [('src', c_vp), ('scale', c_vp), ('wt', c_vp), ('wd', c_vp), ('wt_bf16', c_vp), ('wd_bf16', c_vp), ('T', c_i64),
                ('Cx', c_i32), ('Cy', c_i32)]


register('savp_pack_weights_batch', [c_vp, c_i32, ctypes.POINTER(SavpPackItem)])


class SavpSnItem(ctypes.Structure):
    _fields_ = [('W', c_vp), ('K', c_i64), ('C', c_i32), ('u', c_vp), ('ws', c_vp), ('u_new', c_vp), ('G', c_vp), ('dW', c_vp),
                ('beta', c_i32)]


register('savp_sn_fwd_batch', [c_vp, c_i32, ctypes.POINTER(SavpSnItem)])
register('savp_sn_bwd_batch', [c_vp, c_i32, ctypes.POINTER(SavpSnItem)])
register('savp_sn_fwd', [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp])
register('savp_sn_bwd', [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_i32])
register('savp_dense_fwd', [c_vp, c_vp, c_i64, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64])


class SavpWarpArgs(ctypes.Structure):
    _fields_ = [('N', c_i32), ('H', c_i32), ('W', c_i32), ('C', c_i32), ('K', c_i32), ('img', SavpView), ('flows', c_vp),
                ('out', SavpView), ('dout', SavpView), ('dflows', c_vp), ('dimg', c_vp)]


class SavpDnaArgs(ctypes.Structure):
    _fields_ = [('N', c_i32), ('H', c_i32), ('W', c_i32), ('C', c_i32), ('K', c_i32), ('kh', c_i32), ('kw', c_i32),
                ('img', SavpView), ('raw', c_vp), ('kern', c_vp), ('out', SavpView), ('dout', SavpView), ('draw', c_vp),
                ('dimg', SavpView), ('dimg_beta', c_i32)]


register('savp_image_warp_fwd', [c_vp, ctypes.POINTER(SavpWarpArgs)])
register('savp_image_warp_bwd', [c_vp, ctypes.POINTER(SavpWarpArgs)])
register('savp_dna_apply_fwd', [c_vp, ctypes.POINTER(SavpDnaArgs)])
register('savp_dna_apply_bwd', [c_vp, ctypes.POINTER(SavpDnaArgs)])


class SavpGruArgs(ctypes.Structure):
    _fields_ = [('N', c_i32), ('HW', c_i32), ('F', c_i32), ('eps', c_f32), ('pre', c_vp), ('h', SavpView),
                ('gamma', c_vp), ('beta', c_vp), ('mean', c_vp), ('rstd', c_vp), ('u', c_vp), ('rh', SavpView),
                ('nout', c_i32), ('out', SavpView * 4), ('ndy', c_i32), ('dy', SavpView * 4), ('dpre', c_vp), ('du', c_vp),
                ('dh', SavpView), ('drh', SavpView), ('dgamma', c_vp), ('dbeta', c_vp)]


for _n in ('savp_convgru_gates_fwd', 'savp_convgru_out_fwd', 'savp_convgru_out_bwd', 'savp_convgru_gates_bwd'):
    register(_n, [c_vp, ctypes.POINTER(SavpGruArgs)])
register('savp_gan_loss', [c_vp, c_i32, c_i32, c_vp, c_f32, c_f32, c_vp, c_vp, c_i32])
register('savp_fold_f64', [c_vp, c_vp, c_i64, c_vp, c_vp])
register('savp_tv_loss', [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_f32, c_f32, c_f32, c_vp, c_vp])
register('savp_state_pred_fwd', [c_vp, c_i32, c_i32, c_i32, c_i32] + [c_vp] * 7)
register('savp_state_pred_bwd', [c_vp, c_i32, c_i32, c_i32, c_i32] + [c_vp] * 6)
register('savp_pack_gate_weights', [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_i32])
register('savp_debug_poison_lds', [c_vp, ctypes.c_uint32, c_vp])
register('savp_debug_fill_u32', [c_vp, c_vp, c_i64, ctypes.c_uint32])
register('savp_debug_probe_lds', [c_vp, ctypes.c_uint32, c_vp])
