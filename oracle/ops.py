"""Restatement of /root/reference/video_prediction/ops.py (hot-path symbols only) on torch-CPU.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Variables are passed explicitly (the reference creates
them with tf.get_variable); names/shapes are those of the reference.
"""
import numpy as np
import torch

from . import tf_ops


def _pair(v, n=2):
    return list(v) if isinstance(v, (tuple, list)) else [v] * n


def dense(inputs, kernel, bias=None):
    """ops.py:5-16 (spectral norm is applied by the caller via spectral_normed_weight)."""
    out = inputs @ kernel
    if bias is not None:
        out = out + bias
    return out


def pad2d_paddings(input_hw, size, strides=(1, 1), rate=(1, 1), padding='SAME'):
    """ops.py:71-126 (rate==1 branch + FULL)."""
    size = np.array(_pair(size))
    strides = np.array(_pair(strides))
    rate = np.array(_pair(rate))
    if np.any(rate > 1):
        raise NotImplementedError('dilation_rate > 1 is not on the published SAVP path')
    input_size = np.array(input_hw)
    if padding in ('SAME', 'FULL'):
        pad = np.where(input_size % strides == 0,
                       np.maximum(size - strides, 0),
                       np.maximum(size - (input_size % strides), 0))
        if padding == 'SAME':
            pad_start = pad // 2
            pad_end = pad - pad_start
        else:
            pad_start = pad
            pad_end = pad
        return [[0, 0], [int(pad_start[0]), int(pad_end[0])], [int(pad_start[1]), int(pad_end[1])], [0, 0]]
    elif padding == 'VALID':
        return [[0, 0]] * 4
    raise ValueError("Invalid padding scheme %s" % padding)


def pad2d(inputs, size, strides=(1, 1), rate=(1, 1), padding='SAME', mode='CONSTANT'):
    """ops.py:129-157."""
    paddings = pad2d_paddings(inputs.shape[1:3], size, strides=strides, rate=rate, padding=padding)
    if paddings == [[0, 0]] * 4:
        return inputs
    if mode == 'CONSTANT':
        return tf_ops.pad_constant(inputs, paddings)
    if mode == 'SYMMETRIC':
        return tf_ops.pad_symmetric(inputs, paddings[1:3])
    raise NotImplementedError(mode)


def conv2d(inputs, kernel, bias=None, strides=(1, 1), padding='SAME'):
    """ops.py:494-550 (4-D kernel branch)."""
    strides = _pair(strides)
    if padding == 'FULL':
        inputs = pad2d(inputs, kernel.shape[:2], strides=strides, padding='FULL', mode='CONSTANT')
        padding = 'VALID'
    out = tf_ops.conv2d(inputs, kernel, strides, padding)
    if bias is not None:
        out = out + bias
    return out


def deconv2d(inputs, kernel, strides=(1, 1), padding='SAME'):
    """ops.py:553-589 without bias. kernel [kh,kw,filters,Cin]."""
    strides = _pair(strides)
    kh, kw = kernel.shape[:2]
    h, w = inputs.shape[1:3]
    if padding == 'FULL':
        oh, ow = [s * (i + 1) - k for (i, k, s) in zip((h, w), (kh, kw), strides)]
    elif padding == 'SAME':
        oh, ow = [s * i for (i, s) in zip((h, w), strides)]
    elif padding == 'VALID':
        oh, ow = [s * (i - 1) + k for (i, k, s) in zip((h, w), (kh, kw), strides)]
    else:
        raise ValueError(padding)
    if padding == 'FULL':
        raise NotImplementedError
    return tf_ops.conv2d_transpose(inputs, kernel, [inputs.shape[0], oh, ow, kernel.shape[2]], strides, padding)


def get_bilinear_kernel(strides):
    """ops.py:592-600."""
    strides = np.array(_pair(strides))
    kernel_size = 2 * strides - strides % 2
    center = strides - (kernel_size % 2 == 1) - 0.5 * (kernel_size % 2 != 1)
    vertical_kernel = 1 - abs(np.arange(kernel_size[0]) - center[0]) / strides[0]
    horizontal_kernel = 1 - abs(np.arange(kernel_size[1]) - center[1]) / strides[1]
    return vertical_kernel[:, None] * horizontal_kernel[None, :]


def upsample2d(inputs, strides, padding='SAME'):
    """ops.py:603-609 (bilinear): deconv2d with a per-channel diagonal bilinear kernel."""
    k = torch.as_tensor(get_bilinear_kernel(strides).astype(np.float32), dtype=inputs.dtype)
    C = inputs.shape[-1]
    kernel = k[:, :, None, None] * torch.eye(C, dtype=inputs.dtype)[None, None]
    return deconv2d(inputs, kernel, strides=strides, padding=padding)


def upsample_kernel(kernel, strides=(2, 2)):
    """ops.py:697-704: fold bilinear upsampling into the conv kernel.
    kernel [kh,kw,Cin,F] -> kernel_up [kh+2s-1.., .., F, Cin] (conv2d_transpose layout)."""
    kh, kw, cin, f = kernel.shape
    bil = torch.as_tensor(get_bilinear_kernel(strides).astype(np.float32), dtype=kernel.dtype)
    kernel_transposed = kernel.permute(0, 1, 3, 2)
    kernel_reshaped = kernel_transposed.reshape(kh, kw, 1, f * cin)
    kernel_up_reshaped = conv2d(bil[None, :, :, None], kernel_reshaped, padding='FULL')
    return kernel_up_reshaped.reshape(kernel_up_reshaped.shape[1], kernel_up_reshaped.shape[2], f, cin)


def upsample_conv2d(inputs, kernel, bias=None, strides=(2, 2)):
    """ops.py:643-719."""
    kernel_up = upsample_kernel(kernel, strides)
    out = deconv2d(inputs, kernel_up, strides=strides, padding='SAME')
    if bias is not None:
        out = out + bias
    return out


def conv3d(inputs, kernel, bias=None, strides=(1, 1, 1), padding='SAME'):
    """ops.py:764-777."""
    out = tf_ops.conv3d(inputs, kernel, _pair(strides, 3), padding)
    if bias is not None:
        out = out + bias
    return out


def pool2d(inputs, pool_size, strides=(1, 1), padding='SAME', pool_mode='avg'):
    """ops.py:780-792 (avg; FULL or VALID)."""
    pool_size = _pair(pool_size)
    strides = _pair(strides)
    if padding == 'FULL':
        inputs = pad2d(inputs, pool_size, strides=strides, padding='FULL', mode='CONSTANT')
        padding = 'VALID'
    if pool_mode != 'avg':
        raise NotImplementedError
    if padding == 'SAME':
        # only used by the docstring identity (pool == stride, divisible input) where SAME == VALID
        assert inputs.shape[1] % strides[0] == 0 and inputs.shape[2] % strides[1] == 0 and pool_size == strides
        padding = 'VALID'
    return tf_ops.avg_pool(inputs, pool_size, strides, padding)


def pool_kernel(kernel, strides=(2, 2)):
    """ops.py:838-842: fold the avg-pool into the conv kernel. [kh,kw,Cin,F] -> [kh+s-1, kw+s-1, Cin, F]."""
    kh, kw, cin, f = kernel.shape
    kernel_reshaped = kernel.reshape(1, kh, kw, cin * f)
    kp = pool2d(kernel_reshaped, pool_size=strides, padding='FULL', pool_mode='avg')
    return kp.reshape(kp.shape[1], kp.shape[2], cin, f)


def conv_pool2d(inputs, kernel, bias=None, strides=(2, 2)):
    """ops.py:795-856."""
    strides = _pair(strides)
    if inputs.shape[1] % strides[0] or inputs.shape[2] % strides[1]:
        raise NotImplementedError("The height and width of the input should be "
                                  "an integer multiple of the respective stride.")
    kernel_pool = pool_kernel(kernel, strides)
    out = conv2d(inputs, kernel_pool, strides=strides, padding='SAME')
    if bias is not None:
        out = out + bias
    return out


def lrelu(x, alpha):
    """ops.py:895-903."""
    return torch.maximum(alpha * x, x)


def flatten(x, axis=1, end_axis=-1):
    """ops.py:936-965."""
    nd = x.dim()
    if axis < 0:
        axis += nd
    if end_axis < 0:
        end_axis += nd
    shape = list(x.shape[:axis]) + [-1] + list(x.shape[end_axis + 1:])
    return x.reshape(shape)


def tile_concat(values, axis=-1):
    """ops.py:968-1006: broadcast singleton dims, then concat."""
    nd = values[0].dim()
    if axis < 0:
        axis += nd
    bshape = [1] * nd
    for v in values:
        for d in range(nd):
            if d != axis:
                bshape[d] = max(bshape[d], v.shape[d])
    out = []
    for v in values:
        target = list(bshape)
        target[axis] = v.shape[axis]
        out.append(v.expand(target))
    return torch.cat(out, dim=axis)


def spectral_normed_weight(W, u, num_iters=1):
    """ops.py:1020-1049.  Returns (W_bar, u_final).  Gradients flow through sigma, u_final, v_final
    (no stop_gradient in the reference).  The caller performs the `u <- u_final` UPDATE_OP."""
    W_shape = W.shape
    W_reshaped = W.reshape(-1, W_shape[-1])

    def l2normalize(v, eps=1e-12):
        return v / (torch.linalg.norm(v) + eps)

    u_i = u
    v_i = None
    for _ in range(num_iters):
        v_i = l2normalize(u_i @ W_reshaped.t())
        u_i = l2normalize(v_i @ W_reshaped)
    sigma = ((v_i @ W_reshaped) @ u_i.t()).squeeze()
    W_bar = (W_reshaped / sigma).reshape(W_shape)
    return W_bar, u_i


def fused_instance_norm(x, gamma, beta, epsilon=1e-6):
    """layers/normalization.py:34-196 (NHWC / NDHWC): transpose to [1,spatial...,N*C], one
    fused_batch_norm call (biased variance), transpose back.  Equivalent per-(n,c) statement."""
    N, C = x.shape[0], x.shape[-1]
    nd = x.dim()
    perm = list(range(1, nd - 1)) + [0, nd - 1]
    xt = x.permute(perm)
    hw = list(xt.shape[:-2])
    xt = xt.reshape([1] + hw + [N * C])
    g = gamma[None, :].expand(N, C).reshape(-1)
    b = beta[None, :].expand(N, C).reshape(-1)
    y = tf_ops.fused_batch_norm_training(xt, g, b, epsilon)
    y = y.reshape(hw + [N, C])
    inv = [nd - 2] + list(range(nd - 2)) + [nd - 1]
    return y.permute(inv)
