"""TensorFlow-1.x kernel semantics restated on torch-CPU (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference owns no arithmetic: every primitive it calls is a TF kernel
(tensorflow-gpu>=1.9.0, un-vendored; /root/reference/requirements.txt:1).  This file
restates the published semantics of exactly the kernels the SAVP path calls.  All
tensors are channels-last (NHWC / NDHWC) like the reference; dtype follows the inputs
(float64 for ground truth, float32 for timing).
"""
import math

import torch
import torch.nn.functional as F


def same_pad(in_size, k, s):
    """TF 'SAME': out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0); more padding at the end.
    (restated by the reference itself in ops.py:100-107)."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    before = total // 2
    return before, total - before


def _nhwc_to_nchw(x):
    return x.permute(0, 3, 1, 2)


def _nchw_to_nhwc(x):
    return x.permute(0, 2, 3, 1)


def conv2d(x, w, strides=(1, 1), padding='SAME'):
    """tf.nn.conv2d (cross-correlation). x [N,H,W,Ci], w [kh,kw,Ci,Co] (HWIO).
    Call sites: ops.py:528, rnn_ops.py:121."""
    kh, kw = w.shape[0], w.shape[1]
    sh, sw = strides
    xc = _nhwc_to_nchw(x)
    if padding == 'SAME':
        pt, pb = same_pad(x.shape[1], kh, sh)
        pl, pr = same_pad(x.shape[2], kw, sw)
        xc = F.pad(xc, (pl, pr, pt, pb))
    elif padding != 'VALID':
        raise ValueError(padding)
    y = F.conv2d(xc, w.permute(3, 2, 0, 1).contiguous(), stride=(sh, sw))
    return _nchw_to_nhwc(y)


def conv2d_transpose(x, w, output_shape, strides, padding='SAME'):
    """tf.nn.conv2d_transpose = gradient of tf.nn.conv2d w.r.t. its input.
    x [N,h,w,Ci]; w [kh,kw,Co,Ci] (note: out before in); output [N,H,W,Co].  Call site: ops.py:584."""
    kh, kw = w.shape[0], w.shape[1]
    sh, sw = strides
    H, W = output_shape[1], output_shape[2]
    full = F.conv_transpose2d(_nhwc_to_nchw(x), w.permute(3, 2, 0, 1).contiguous(), stride=(sh, sw))
    if padding == 'SAME':
        pt, _ = same_pad(H, kh, sh)
        pl, _ = same_pad(W, kw, sw)
    elif padding == 'VALID':
        pt = pl = 0
    else:
        raise ValueError(padding)
    # the forward conv pads its (H,W) input by (pt,..),(pl,..); rows that fall into padding are dropped,
    # rows the forward conv never touched (bottom/right remainder) stay zero.
    need_h, need_w = pt + H, pl + W
    fh, fw = full.shape[2], full.shape[3]
    if fh < need_h or fw < need_w:
        full = F.pad(full, (0, max(need_w - fw, 0), 0, max(need_h - fh, 0)))
    y = full[:, :, pt:pt + H, pl:pl + W]
    return _nchw_to_nhwc(y)


def conv3d(x, w, strides=(1, 1, 1), padding='VALID'):
    """tf.nn.conv3d. x [N,D,H,W,Ci]; w [kd,kh,kw,Ci,Co].  Call site: ops.py:773."""
    xc = x.permute(0, 4, 1, 2, 3)
    if padding == 'SAME':
        pads = []
        for dim, k, s in zip(x.shape[1:4], w.shape[:3], strides):
            pads.append(same_pad(dim, k, s))
        xc = F.pad(xc, (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    elif padding != 'VALID':
        raise ValueError(padding)
    y = F.conv3d(xc, w.permute(4, 3, 0, 1, 2).contiguous(), stride=tuple(strides))
    return y.permute(0, 2, 3, 4, 1)


def depthwise_conv2d(x, w, padding='VALID'):
    """tf.nn.depthwise_conv2d, stride 1. x [N,H,W,C]; w [kh,kw,C,M]; out [N,H',W',C*M] with
    output channel index c*M+m.  Call site: savp_model.py:918."""
    kh, kw, C, M = w.shape
    xc = _nhwc_to_nchw(x)
    if padding == 'SAME':
        pt, pb = same_pad(x.shape[1], kh, 1)
        pl, pr = same_pad(x.shape[2], kw, 1)
        xc = F.pad(xc, (pl, pr, pt, pb))
    wt = w.permute(2, 3, 0, 1).reshape(C * M, 1, kh, kw)
    y = F.conv2d(xc, wt, groups=C)
    return _nchw_to_nhwc(y)


def avg_pool(x, ksize, strides, padding='VALID'):
    """tf.nn.avg_pool (VALID only needed: ops.py:789 via pool2d with FULL->VALID or global pool)."""
    if padding != 'VALID':
        raise NotImplementedError
    y = F.avg_pool2d(_nhwc_to_nchw(x), kernel_size=tuple(ksize), stride=tuple(strides))
    return _nchw_to_nhwc(y)


def pad_symmetric(x, paddings):
    """tf.pad(mode='SYMMETRIC') on H,W of an NHWC tensor: mirrors including the edge pixel."""
    (pt, pb), (pl, pr) = paddings
    H, W = x.shape[1], x.shape[2]
    iy = [pt - 1 - i for i in range(pt)] + list(range(H)) + [H - 1 - i for i in range(pb)]
    ix = [pl - 1 - i for i in range(pl)] + list(range(W)) + [W - 1 - i for i in range(pr)]
    iy = torch.tensor(iy, dtype=torch.long)
    ix = torch.tensor(ix, dtype=torch.long)
    return x.index_select(1, iy).index_select(2, ix)


def pad_constant(x, paddings):
    """tf.pad(mode='CONSTANT') with a full per-axis paddings list."""
    flat = []
    for before, after in reversed(list(paddings)):
        flat += [before, after]
    return F.pad(x, flat)


def fused_batch_norm_training(x, gamma, beta, epsilon):
    """nn.fused_batch_norm(is_training=True): normalises with the batch mean and the *biased*
    batch variance over all but the last axis.  Call site: layers/normalization.py:162."""
    C = x.shape[-1]
    flat = x.reshape(-1, C)
    mean = flat.mean(dim=0)
    var = ((flat - mean) ** 2).mean(dim=0)
    return (x - mean) * torch.rsqrt(var + epsilon) * gamma + beta


def lstm_cell(x, c, h, kernel, bias, forget_bias=1.0):
    """tf.nn.rnn_cell.LSTMCell (no peepholes/projection): gates = [x,h]@kernel + bias, order i,j,f,o;
    c' = sigmoid(f+forget_bias)*c + sigmoid(i)*tanh(j); h' = sigmoid(o)*tanh(c').  Call site: savp_model.py:356."""
    gates = torch.cat([x, h], dim=-1) @ kernel + bias
    i, j, f, o = torch.chunk(gates, 4, dim=-1)
    c_new = torch.sigmoid(f + forget_bias) * c + torch.sigmoid(i) * torch.tanh(j)
    h_new = torch.sigmoid(o) * torch.tanh(c_new)
    return h_new, (c_new, h_new)


def gru_cell(x, h, gates_kernel, gates_bias, candidate_kernel, candidate_bias):
    """tf.contrib.rnn.GRUCell (= tf.nn.rnn_cell.GRUCell, TensorFlow 1.x rnn_cell_impl.GRUCell.call; un-vendored dependency
    tensorflow>=1.9, algorithm restated from its published source):
      [r, u] = sigmoid([x, h] @ gates_kernel + gates_bias)   (r first);  c = tanh([x, r*h] @ candidate_kernel + candidate_bias);
      h' = u*h + (1-u)*c.   Call sites: savp_model.py:38-41 (encoder tail), :358-362 (_rnn_func: rnn_z)."""
    ru = torch.sigmoid(torch.cat([x, h], dim=-1) @ gates_kernel + gates_bias)
    r, u = torch.chunk(ru, 2, dim=-1)
    c = torch.tanh(torch.cat([x, r * h], dim=-1) @ candidate_kernel + candidate_bias)
    h_new = u * h + (1.0 - u) * c
    return h_new, h_new


def adam_update(p, g, m, v, lr, beta1, beta2, t, epsilon=1e-8):
    """tf.train.AdamOptimizer step t (1-based): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g^2; p -= lr_t*m/(sqrt(v)+eps).  Call site: base_model.py:486-487."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    p = p - lr_t * m / (torch.sqrt(v) + epsilon)
    return p, m, v
