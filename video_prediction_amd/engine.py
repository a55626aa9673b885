"""Host-side engine: parameter arenas, per-step weight preparation and conv-layer objects.

Everything numerical is a HIP kernel launched through ``kernels``; this file only owns HBM layout:

* ``ParamStore`` keeps every trainable variable of one optimiser group in ONE flat fp32 arena (16-byte aligned
  slots, keyed by the reference's TF variable names), with parallel arenas for gradients and Adam moments.  A whole
  group is updated by one ``savp_adam`` launch and all-reduced as one RCCL bucket.
* ``ConvLayer`` owns the derived weight layouts of one convolution (pool-/bilinear-folded HWIO kernel, spectral-norm
  scale, k-contiguous packs for FPROP and DGRAD) and accumulates its weight gradient with ONE WGRAD launch over all
  timesteps and samples (the time axis is folded into the GEMM's K dimension).
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import kernels as K
from . import lib
from .variables import is_trainable


def _align4(n):
    return (n + 3) // 4 * 4


class Arena(object):
    """Flat fp32 device arena with named, 16-byte aligned slots."""

    def __init__(self, shapes, device):
        self.offsets = OrderedDict()
        off = 0
        for name, shape in shapes.items():
            n = int(np.prod(shape)) if len(shape) else 1
            self.offsets[name] = (off, n, tuple(shape))
            off += _align4(n)
        self.size = max(off, 4)
        self.device = device
        self.flat = torch.zeros(self.size, device=device, dtype=torch.float32)
        self._views = {}

    def view_of(self, flat, name):
        off, n, shape = self.offsets[name]
        return flat[off:off + n].view(shape)

    def __getitem__(self, name):
        v = self._views.get(name)
        if v is None:
            v = self._views[name] = self.view_of(self.flat, name)
        return v

    def __contains__(self, name):
        return name in self.offsets

    def names(self):
        return list(self.offsets.keys())

    def like(self):
        return torch.zeros_like(self.flat)


class ParamGroup(object):
    """One optimiser group: params, grads, Adam moments, Adam step counter."""

    def __init__(self, shapes, device):
        self.arena = Arena(shapes, device)
        self.p = self.arena.flat
        self.g = self.arena.like()
        self.m = self.arena.like()
        self.v = self.arena.like()
        self.t = 0
        self._gviews = {}
        # float64 twin of the gradient arena for everything that many workgroups ADD to (norm gamma / beta, the z-LSTM's kernel / bias,
        # learned initial states): the kernels accumulate there (a sum of fp32 partials is exact in float64 -- no dependence on arrival
        # order), fold64() rounds to fp32 once.  Only the variables somebody asked for (grad64) are folded.
        self.g64 = torch.zeros(self.arena.size, device=device, dtype=torch.float64)
        self._g64views, self._g64idx, self._g64names = {}, None, []

    def param(self, name):
        return self.arena[name]

    def grad(self, name):
        v = self._gviews.get(name)
        if v is None:
            v = self._gviews[name] = self.arena.view_of(self.g, name)
        return v

    def grad64(self, name):
        """The float64 accumulator of variable `name` (same shape); its content reaches grad(name) with the next fold64()."""
        v = self._g64views.get(name)
        if v is None:
            v = self._g64views[name] = self.arena.view_of(self.g64, name)
            self._g64names.append(name)
            self._g64idx = None
        return v

    def fold64(self):
        """g += float(g64), g64 = 0 over the variables handed out by grad64(); call before anything reads g (Adam, the gradient exchange,
        a test).  Safe to call repeatedly (the accumulators are cleared)."""
        if not self._g64names:
            return
        if self._g64idx is None:
            idx = []
            for name in self._g64names:
                off, n, _ = self.arena.offsets[name]
                idx.append(torch.arange(off, off + n, dtype=torch.int32))
            self._g64idx = torch.cat(idx).to(self.g.device)
        K.fold64(self.g64, self.g, self._g64idx)

    def zero_grad(self):
        self.g.zero_()

    def next_lr_t(self, lr, beta1, beta2):
        """Advance the step count; returns tf.train.AdamOptimizer's lr_t = lr*sqrt(1-b2^t)/(1-b1^t) (base_model.py:486-487)."""
        self.t += 1
        return lr * math.sqrt(1.0 - beta2 ** self.t) / (1.0 - beta1 ** self.t)

    def adam_apply(self, lr_t, beta1, beta2, gscale=1.0, eps=1e-8, lr_t_dev=None):
        K.adam(self.p, self.g, self.m, self.v, lr_t, beta1, beta2, eps=eps, gscale=gscale, lr_t_dev=lr_t_dev)

    def adam_step(self, lr, beta1, beta2, gscale=1.0, eps=1e-8):
        self.adam_apply(self.next_lr_t(lr, beta1, beta2), beta1, beta2, gscale=gscale, eps=eps)


class ParamStore(object):
    """All variables of a model: groups 'g' (scope generator/), 'd' (scope discriminator/, trainable) and 'aux'
    (non-trainable spectral-norm u vectors)."""

    def __init__(self, specs, values, device):
        shapes = {'g': OrderedDict(), 'd': OrderedDict(), 'aux': OrderedDict()}
        self.group_of = {}
        for name, (shape, _) in specs.items():
            if not is_trainable(name):
                grp = 'aux'
            elif name.startswith('discriminator/'):
                grp = 'd'
            else:
                grp = 'g'
            shapes[grp][name] = shape
            self.group_of[name] = grp
        self.groups = {k: ParamGroup(v, device) for k, v in shapes.items()}
        self.specs = specs
        self.device = device
        self.load(values)

    def load(self, values):
        for name, val in values.items():
            if name not in self.group_of:
                continue
            t = torch.as_tensor(np.asarray(val, dtype=np.float32))
            self[name].copy_(t.reshape(self[name].shape))

    def __getitem__(self, name):
        return self.groups[self.group_of[name]].param(name)

    def __contains__(self, name):
        return name in self.group_of

    def grad(self, name):
        return self.groups[self.group_of[name]].grad(name)

    def grad64(self, name):
        return self.groups[self.group_of[name]].grad64(name)

    def names(self):
        return list(self.group_of.keys())

    def chunk_of(self, group, prefix):
        """(lo, hi) element range, inside the flat arena of `group`, of the variables whose names start with `prefix`; they
        must be laid out contiguously (they are: variable_specs lists one network after the other).  Used to exchange the
        gradients of one network as soon as its backward pass has been issued (parallel.ReplicaGroup)."""
        offs = self.groups[group].arena.offsets
        lo = hi = None
        inside = done = False
        for name, (off, n, _) in offs.items():
            if name.startswith(prefix):
                if done:
                    raise ValueError('variables under %r are not contiguous in group %r' % (prefix, group))
                if not inside:
                    lo, inside = off, True
                hi = off + _align4(n)
            elif inside:
                inside, done = False, True
        if lo is None:
            return (0, 0)
        return (lo, hi)

    def to_numpy(self):
        return OrderedDict((n, self[n].detach().cpu().numpy().copy()) for n in self.specs)

    def grads_to_numpy(self):
        for grp in self.groups.values():
            grp.fold64()
        return OrderedDict((n, self.grad(n).detach().cpu().numpy().copy()) for n in self.specs if self.group_of[n] != 'aux')


# ---------------------------------------------------------------------------------------------------------------
# convolution layers
# ---------------------------------------------------------------------------------------------------------------
def same_pad_before(k, s, in_size):
    """TF SAME padding-before (ops.py:100-107)."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return total // 2


class ConvLayer(object):
    """One convolution of the SAVP graph with its derived weight layouts.

    kind:
      'conv' : plain cross-correlation (tf.nn.conv2d / conv3d / dense as 1x1), kernel used as is.
      'pool' : ops.conv_pool2d -- avg-pool folded into the kernel (k -> k+1), stride 2, SAME.
      'up'   : ops.upsample_conv2d -- bilinear x2 folded into the kernel (3 -> 6), conv2d_transpose stride 2 SAME;
               executed as the DGRAD mode of the stride-2 forward conv described by ``geom``.
    ``sn_u`` names the spectral-norm vector of a discriminator layer (kernel is divided by sigma on the fly).
    """

    def __init__(self, store, kernel_name, bias_name, kind, ksize, stride, pad, sn_u=None, cx_pad=None, cy_pad=None):
        self.store = store
        self.kernel_name, self.bias_name, self.kind = kernel_name, bias_name, kind
        W = store[kernel_name]
        self.W = W
        self.dW = store.grad(kernel_name) if store.group_of[kernel_name] != 'aux' else None
        self.bias = store[bias_name] if bias_name else None
        self.dbias = store.grad(bias_name) if bias_name else None
        dev = W.device
        if kind == 'conv':
            self.cx, self.cy = W.shape[-2], W.shape[-1]
            self.wf = W                                   # folded == master
            self.dwf = None                               # wgrad goes straight to the master grad (unless SN)
            k3 = tuple(ksize)
        elif kind == 'pool':
            k, _, cin, cout = W.shape
            self.cx, self.cy = cin, cout
            self.wf = torch.empty(k + 1, k + 1, cin, cout, device=dev)
            self.dwf = torch.zeros_like(self.wf)
            k3 = (1, k + 1, k + 1)
        elif kind == 'up':
            k, _, cin, f = W.shape
            # forward-conv description F: x-side = hi-res F channels, y-side = lo-res Cin channels
            self.cx, self.cy = f, cin
            self.wf = torch.empty(k + 3, k + 3, f, cin, device=dev)
            self.dwf = torch.zeros_like(self.wf)
            k3 = (1, k + 3, k + 3)
        else:
            raise ValueError(kind)
        if len(k3) == 2:
            k3 = (1,) + tuple(k3)
        s3 = tuple(stride) if len(stride) == 3 else (1,) + tuple(stride)
        p3 = tuple(pad) if len(pad) == 3 else (0,) + tuple(pad)
        self.geom = K.ConvGeom(k3, s3, p3)
        self.taps = k3[0] * k3[1] * k3[2]
        # optional zero-padding of the channel counts the kernel sees (activation buffers padded to multiples of 4 so
        # that every conv takes the float4 / MFMA-bf16 path: 14 -> 16 input channels of h0, 53 -> 56 of the mask conv ...)
        self.cx0, self.cy0 = self.cx, self.cy
        self.padded = False
        if (cx_pad and cx_pad != self.cx) or (cy_pad and cy_pad != self.cy):
            if kind == 'up' or sn_u:
                raise NotImplementedError('channel padding for upsample / spectral-norm layers')
            self.padded = True
            self.cx, self.cy = cx_pad or self.cx, cy_pad or self.cy
            self.wfp = torch.zeros(self.taps, self.cx, self.cy, device=dev)
            self.dwfp = torch.zeros(self.taps, self.cx, self.cy, device=dev)
            self.dw_tmp = torch.empty(self.taps, self.cx0, self.cy0, device=dev)
            if self.bias is not None:
                self.bias_master, self.dbias_master = self.bias, self.dbias
                self.bias = torch.zeros(self.cy, device=dev)
                self.dbias = torch.zeros(self.cy, device=dev)
        self.wt = torch.empty(self.cy, self.taps * self.cx, device=dev)
        self.wd = torch.empty(self.cx, self.taps * self.cy, device=dev)
        # bf16 copies of the packed weights for the bf16 MFMA mode (half the weight stream, no in-kernel conversion)
        self.wt16 = torch.empty(self.cy, self.taps * self.cx, device=dev, dtype=torch.bfloat16)
        self.wd16 = torch.empty(self.cx, self.taps * self.cy, device=dev, dtype=torch.bfloat16)
        self.sn_u_name = sn_u
        if sn_u:
            self.u = store[sn_u]
            kdim = W.numel() // self.cy
            self.sn_ws = torch.zeros(K.sn_ws_size(kdim, self.cy), device=dev)
            self.u_next = torch.empty_like(self.u)
            self.dwf = torch.zeros_like(W)                # dL/dW_bar, then sn_bwd -> master grad
        self.need_wt = self.need_wd = True
        self.wfrag = None         # enable_gate_pack(): the weights in MFMA B-fragment order for the gate convolution's own kernel (csrc/conv_gate.hip)
        self.wfrag_il = None      # ... and with interleaved gate columns: the whole cell forward in one launch (savp_convlstm_cell_fwd)
        self.prof = None          # list of (start, end) events when bench.py instruments this layer's forward launches
        self.ktimer = None        # kernels.KernelTimer: kernel-only duration of the same launches (ring kernel)

    # -- weight preparation ---------------------------------------------------------------------------------
    def sn_entry(self, update_u=False):
        """Arguments of this layer's spectral-norm forward for kernels.sn_fwd_batch (prep(..., sn_done=True) then skips its own)."""
        return {'W': self.W, 'u': self.u.reshape(-1), 'ws': self.sn_ws, 'u_new': self.u_next.reshape(-1) if update_u else None}

    def sn_bwd_entry(self):
        return {'W': self.W, 'u': self.u.reshape(-1), 'ws': self.sn_ws, 'G': self.dwf, 'dW': self.dW, 'beta': 1}

    def prep(self, update_u=False, sn_done=False, defer_pack=None):
        """defer_pack: a list -- the pack is appended to it for one kernels.pack_weights_batch over a network's layers."""
        scale = None
        if self.kind == 'pool':
            K.fold_pool(self.W, self.wf, self.W.shape[0])
        elif self.kind == 'up':
            k, _, cin, f = self.W.shape
            K.fold_bilinear(self.W, self.wf, k, cin, f)
        if self.sn_u_name:
            if not sn_done:
                K.sn_fwd(self.W, self.u.reshape(-1), self.sn_ws, self.u_next.reshape(-1) if update_u else None)
            scale = self.sn_ws[1:2]
        src = self.wf
        if self.padded:
            copy_view(self.wf.reshape(self.taps, self.cx0, self.cy0), [self.wfp[:, :self.cx0, :self.cy0]])
            if self.bias is not None:
                K.axpby(1.0, self.bias_master, 0.0, None, self.bias[:self.cy0])
            src = self.wfp
        b16 = K.PRECISION['value'] == 1
        entry = {'src': src, 'wt': self.wt if self.need_wt else None, 'wd': self.wd if self.need_wd else None, 'scale': scale,
                 'wt16': self.wt16 if (b16 and self.need_wt) else None, 'wd16': self.wd16 if (b16 and self.need_wd) else None}
        if defer_pack is not None:
            defer_pack.append(entry)
        else:
            K.pack_weights(**entry)
        if self.wfrag is not None and b16:
            K.pack_gate_weights(src, self.wfrag)
            if self.wfrag_il is not None:
                K.pack_gate_weights(src, self.wfrag_il, interleave=True)

    def enable_gate_pack(self, cell=False):
        """This layer is a ConvLSTM gate convolution (rnn_ops.py:115-126): keep its weights in B-fragment order as well, so that the bf16
        datapath's forward takes conv_gate_kernel.  No-op for shapes that kernel has no pack for."""
        if self.kind == 'conv' and not self.padded and not self.sn_u_name:
            n = K.gate_weights_elems(self.taps, self.cx, self.cy)
            if n:
                self.wfrag = torch.empty(n, device=self.W.device, dtype=torch.bfloat16)
                if cell and self.cy % 32 == 0:
                    self.wfrag_il = torch.empty(n, device=self.W.device, dtype=torch.bfloat16)

    def commit_u(self):
        """The reference's UPDATE_OP ``u.assign(u_final)`` (ops.py:1046-1048)."""
        if self.sn_u_name:
            self.u.copy_(self.u_next)

    # -- execution ----------------------------------------------------------------------------------------------
    def forward(self, x, y, beta=0, act=0, alpha=0.0, use_bias=True, stats=None, defer=False):
        """stats: see kernels.conv (bf16 destination only: the fused ConvLSTM gate convolution).  defer: return the filled
        SavpConvArgs instead of launching (for a fused-operator entry point, kernels.convlstm_cell_fwd / conv_in_act_fwd); None when this
        call is instrumented or takes the few-row dense kernel -- the caller then issues the halves apart."""
        b = self.bias if use_bias else None
        if self.kind == 'conv' and x.dim() == 2 and x.shape[0] <= 64 and not act and not beta and y.is_contiguous():
            if defer:
                return None
            # dense layer on a handful of rows: split-K kernel on the master weights (ops.py:5-16)
            K.dense_fwd(x, self.W.reshape(-1, self.cy), b, y, scale=self.sn_ws[1:2] if self.sn_u_name else None)
            return
        if defer:
            if self.ktimer is not None or self.prof is not None:
                return None
            if self.kind == 'up':
                return K.conv(lib.CONV_DGRAD, self.geom, y, x, self.wd, bias=b, beta=beta, act=act, alpha=alpha, w16=self.wd16, stats=stats,
                              defer=True)
            return K.conv(lib.CONV_FPROP, self.geom, x, y, self.wt, bias=b, beta=beta, act=act, alpha=alpha, w16=self.wt16, stats=stats,
                          defer=True, w_frag=self.wfrag, w_frag_il=self.wfrag_il)
        if self.ktimer is not None:
            self.ktimer.arm()
        if self.prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if self.kind == 'up':
            K.conv(lib.CONV_DGRAD, self.geom, y, x, self.wd, bias=b, beta=beta, act=act, alpha=alpha, w16=self.wd16, stats=stats)
        else:
            K.conv(lib.CONV_FPROP, self.geom, x, y, self.wt, bias=b, beta=beta, act=act, alpha=alpha, w16=self.wt16, stats=stats, w_frag=self.wfrag)
        if self.prof is not None:
            e1.record()
            self.prof.append((e0, e1))
        if self.ktimer is not None:
            self.ktimer.taken()

    def stats_ok(self, x, y):
        """Can forward(x, y, stats=...) leave the destination's instance-norm statistics behind (kernels.conv_stats_ok)?"""
        if K.PRECISION['value'] != 1:
            return False
        if self.kind == 'up':
            return K.conv_stats_ok(lib.CONV_DGRAD, self.geom, y, x, self.wd, bias=self.bias, w16=self.wd16)
        return K.conv_stats_ok(lib.CONV_FPROP, self.geom, x, y, self.wt, bias=self.bias, w16=self.wt16)

    def backward_data(self, dy, dx, beta=0, act=0, alpha=0.0, aux=None, skip=None, norm_bwd=None, defer=False):
        """skip = (first, count): input channels whose data gradient is not needed per pixel (left unwritten in dx).  norm_bwd: dx's
        channels [c0, c0 + C) are the output gradient of an instance norm over norm_bwd['x']; its backward sums leave with this launch
        (kernels.conv)."""
        if self.kind == 'up':
            return K.conv(lib.CONV_FPROP, self.geom, dy, dx, self.wt, beta=beta, act=act, alpha=alpha, aux=aux, w16=self.wt16, dst_gap=skip,
                          norm_bwd=norm_bwd, defer=defer)
        return K.conv(lib.CONV_DGRAD, self.geom, dx, dy, self.wd, beta=beta, act=act, alpha=alpha, aux=aux, w16=self.wd16, dst_gap=skip,
                      norm_bwd=norm_bwd, defer=defer)

    def norm_bwd_ok(self, dy, dx, norm_bwd, skip=None):
        """Can backward_data(dy, dx, norm_bwd=...) leave the norm-backward sums behind (bf16 datapath, ring kernel, whole tiles)?"""
        if K.PRECISION['value'] != 1:
            return False
        nb = dict(norm_bwd, ws=None)
        if self.kind == 'up':
            return K.conv_stats_ok(lib.CONV_FPROP, self.geom, dy, dx, self.wt, w16=self.wt16, dst_gap=skip, norm_bwd=nb)
        return K.conv_stats_ok(lib.CONV_DGRAD, self.geom, dx, dy, self.wd, w16=self.wd16, dst_gap=skip, norm_bwd=nb)

    def backward_weights(self, x, dy, feeds_instance_norm=False):
        """Accumulate the kernel (and bias) gradient from input activations x and output gradients dy; both may
        carry folded leading (time, batch) dims: [R, (D,) H, W, C].  feeds_instance_norm: the caller states that this convolution's
        output goes straight into a fused_instance_norm (see below); required for a bf16 dy."""
        target = self.dwfp if self.padded else (self.dwf if self.dwf is not None else self.dW)
        if dy.dtype == torch.bfloat16 and self.dbias is not None and not feeds_instance_norm:
            raise NotImplementedError('bias gradient from a bf16 output gradient: only for a convolution in front of an instance norm '
                                      '(feeds_instance_norm=True), whose bias gradient is identically zero')
        # A bf16 dy is the gradient an instance norm's backward wrote (SAVPGenerator.act16).  The convolution in front of a
        # fused_instance_norm (every conv_pool2d / upsample_conv2d / 3x3 head of the cell, savp_model.py:449-500,522-567,625-631) has an
        # identically zero bias gradient: the norm's input gradient gamma * rstd * (dy - mean(dy) - xhat * mean(dy * xhat)) sums to zero
        # over each (sample, channel) plane because sum(xhat) = 0.  The reference's value is that zero plus fp32 rounding noise; column
        # sums of the bf16-ROUNDED gradient would be noise 2^-9 / 2^-24 times larger (measured 2e-3 of the group's largest gradient at
        # T = 40), so with a bf16 dy the bias gradient is left at its exact value, zero.
        db = None if (feeds_instance_norm and dy.dtype == torch.bfloat16) else self.dbias
        if self.kind == 'up':
            K.conv(lib.CONV_WGRAD, self.geom, dy, x, target)
            if db is not None:
                K.colsum(dy, db)
        else:       # bias gradient = column sums of dy: fused into the WGRAD pass (include/savp_hip.h, SavpConvArgs.bias)
            K.conv(lib.CONV_WGRAD, self.geom, x, dy, target, bias=db)

    def finish_weight_grad(self, sn_done=False):
        """Map the folded / spectrally-normalised kernel gradient back to the master variable and clear it."""
        if self.padded:
            copy_view(self.dwfp[:, :self.cx0, :self.cy0], [self.dw_tmp])
            tgt = self.dwf if self.dwf is not None else self.dW
            K.axpby(1.0, self.dw_tmp.reshape(-1), 1.0, tgt.reshape(-1), tgt.reshape(-1))
            self.dwfp.zero_()
            if self.bias is not None:
                K.axpby(1.0, self.dbias[:self.cy0], 1.0, self.dbias_master, self.dbias_master)
                self.dbias.zero_()
        if self.dwf is None:
            return
        if self.sn_u_name:
            if not sn_done:
                K.sn_bwd(self.W, self.u.reshape(-1), self.sn_ws, self.dwf, self.dW, beta=1)
        elif self.kind == 'pool':
            K.fold_pool(self.dwf, self.dW, self.W.shape[0], adjoint=True)
        elif self.kind == 'up':
            k, _, cin, f = self.W.shape
            K.fold_bilinear(self.dwf, self.dW, k, cin, f, adjoint=True)
        self.dwf.zero_()


class _TensorStore(object):
    """Minimal stand-in for ParamStore around private tensors (ConvLayer only needs [], grad() and group_of)."""

    def __init__(self, tensors, grads):
        self.t, self.g = tensors, grads
        self.group_of = {k: 'g' for k in tensors}

    def __getitem__(self, name):
        return self.t[name]

    def grad(self, name):
        return self.g[name]


def prep_layers(convs, **kw):
    """prep() of a network's layers with ONE batched weight pack at the end (SAVP_PACK_BATCH=0: a pack launch per layer)."""
    if os.environ.get('SAVP_PACK_BATCH', '1') != '1':
        for c in convs:
            c.prep(**kw)
        return
    packs = []
    for c in convs:
        c.prep(defer_pack=packs, **kw)
    if packs:
        K.pack_weights_batch(packs)


class ConcatConv(object):
    """Several plain convolutions that read the SAME input with the same geometry, run as ONE convolution whose output channels are
    the concatenation of theirs (the 3x3 heads of SAVPCell.call that all read the last decoder layer: h6_scratch, h6_masks and,
    for flow / dna, the transformation head -- savp_model.py:522-544,562-567,625-631).  One launch reads the input once instead
    of once per head, forward and backward (the data gradient of the concatenation IS the sum of the heads' data gradients), and
    one weight-gradient launch serves all of them.  The master variables stay separate TF-named tensors: per step they are copied
    into the channel slices of a concatenated HWIO kernel / bias (prep), and the concatenated gradient is added back slice by slice
    (finish_weight_grad)."""

    def __init__(self, store, parts, ksize, stride, pad):
        self.parts = []
        off = 0
        W0 = store[parts[0][0]]
        self.lead = tuple(W0.shape[:-1])                           # (kh, kw, cx)
        dev = W0.device
        for kname, bname in parts:
            W = store[kname]
            if tuple(W.shape[:-1]) != self.lead:
                raise ValueError('ConcatConv parts must share kernel size and input channels')
            cy = W.shape[-1]
            self.parts.append(dict(W=W, dW=store.grad(kname), b=store[bname], db=store.grad(bname), off=off, cy=cy))
            off += cy
        self.cy = off
        self.Wcat = torch.empty(self.lead + (off,), device=dev)
        self.dWcat = torch.zeros(self.lead + (off,), device=dev)
        self.bcat = torch.zeros(off, device=dev)
        self.dbcat = torch.zeros(off, device=dev)
        shim = _TensorStore({'k': self.Wcat, 'b': self.bcat}, {'k': self.dWcat, 'b': self.dbcat})
        self.inner = ConvLayer(shim, 'k', 'b', 'conv', ksize, stride, pad)
        self.R = int(np.prod(self.lead))

    @property
    def prof(self):
        return self.inner.prof

    def prep(self, update_u=False, defer_pack=None):
        R = self.R
        for P in self.parts:
            copy_view(P['W'].reshape(R, P['cy']), [self.Wcat.reshape(R, self.cy)[:, P['off']:P['off'] + P['cy']]])
            copy_view(P['b'].reshape(1, P['cy']), [self.bcat.reshape(1, self.cy)[:, P['off']:P['off'] + P['cy']]])
        self.inner.prep(defer_pack=defer_pack)

    def forward(self, x, y, **kw):
        return self.inner.forward(x, y, **kw)

    def stats_ok(self, x, y):
        return self.inner.stats_ok(x, y)

    def backward_data(self, dy, dx, **kw):
        return self.inner.backward_data(dy, dx, **kw)

    def norm_bwd_ok(self, dy, dx, norm_bwd, skip=None):
        return self.inner.norm_bwd_ok(dy, dx, norm_bwd, skip)

    def backward_weights(self, x, dy, feeds_instance_norm=False):
        self.inner.backward_weights(x, dy, feeds_instance_norm=feeds_instance_norm)

    def finish_weight_grad(self):
        self.inner.finish_weight_grad()
        R = self.R
        for P in self.parts:
            add_views([self.dWcat.reshape(R, self.cy)[:, P['off']:P['off'] + P['cy']]], P['dW'].reshape(R, P['cy']))
            add_views([self.dbcat.reshape(1, self.cy)[:, P['off']:P['off'] + P['cy']]], P['db'].reshape(1, P['cy']))
        self.dWcat.zero_()
        self.dbcat.zero_()

    def commit_u(self):
        pass


class Tape(object):
    """Minimal reverse-mode tape: forward code appends closures, backward() runs them in reverse."""

    def __init__(self):
        self.ops = []

    def add(self, fn):
        self.ops.append(fn)

    def backward(self):
        for fn in reversed(self.ops):
            fn()
        self.ops = []


_CONST = {}


def const_i32(n, value, device):
    """Cached int32 device vector of n copies of value (masks for select used as strided copy / accumulate)."""
    key = (n, value, str(device))
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.full((n,), value, dtype=torch.int32, device=device)
    return t


def copy_view(src, dsts):
    """Strided channels-last copy src -> each dst (shape [R, spatial..., C])."""
    K.select(const_i32(src.shape[0], 1, src.device), src, None, dsts)


def add_views(srcs, dst):
    """dst += sum(srcs) on strided channels-last views."""
    K.select_bwd(const_i32(dst.shape[0], 0, dst.device), srcs, dst)
