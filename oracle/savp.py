"""Restatement of the SAVP model graph on torch-CPU (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows /root/reference/video_prediction/models/savp_model.py, rnn_ops.py, models/networks.py and
flow_ops.py.  Variables come from a dict keyed by the TF variable names (relative to the enclosing
scope); all randomness (eps, prior z, scheduled-sampling mask, discriminator clip indices) is an explicit
input because TF's random streams cannot be reproduced.

Tensors are time-major NHWC: images [T,B,H,W,C].
"""
from collections import OrderedDict

import numpy as np
import torch

from . import ops, tf_ops

RELU_SHIFT = 1e-12  # savp_model.py:18


class Scope(object):
    """Read-only view of a variable dict under a name prefix (mirrors tf.variable_scope)."""

    def __init__(self, params, prefix=''):
        self.params = params
        self.prefix = prefix

    def sub(self, name):
        return Scope(self.params, self.prefix + name + '/')

    def __getitem__(self, name):
        return self.params[self.prefix + name]

    def has(self, name):
        return (self.prefix + name) in self.params


# --------------------------------------------------------------------------------------------
# rnn_ops.py
# --------------------------------------------------------------------------------------------
def conv_lstm_cell(vs, inputs, state, filters, normalized=True, forget_bias=1.0):
    """BasicConv2DLSTMCell.call, rnn_ops.py:137-171, as constructed at savp_model.py:386-390:
    kernel 5x5, normalizer fused_instance_norm, separate_norms=False (instance), no dropout/skip."""
    c, h = state
    vs = vs.sub('basic_conv2dlstm_cell')
    tile_concat = isinstance(inputs, (list, tuple))
    if tile_concat:
        inputs, inputs_non_spatial = inputs
    args = torch.cat([inputs, h], dim=-1)                                   # :143
    concat = tf_ops.conv2d(args, vs['kernel'], (1, 1), 'SAME')              # :121
    if not normalized:
        concat = concat + vs['bias']                                        # :122-125
    if tile_concat:
        concat = concat + (inputs_non_spatial @ vs['weights'])[:, None, None, :]   # :145-146
    if normalized:
        concat = ops.fused_instance_norm(concat, vs['input_transform_forget_output/gamma'],
                                         vs['input_transform_forget_output/beta'])  # :148-149
    i, j, f, o = torch.chunk(concat, 4, dim=-1)                             # :150
    g = torch.tanh(j)
    new_c = c * torch.sigmoid(f + forget_bias) + torch.sigmoid(i) * g      # :161-162
    if normalized:
        new_c = ops.fused_instance_norm(new_c, vs['state/gamma'], vs['state/beta'])  # :163-164
    new_h = torch.tanh(new_c) * torch.sigmoid(o)                            # :165
    return new_h, (new_c, new_h)


def conv_gru_cell(vs, inputs, state, filters, normalized=True):
    """Conv2DGRUCell.call, rnn_ops.py:234-267 (separate_norms=False).  Reproduces the quirk that
    the candidate conv sees [x, h, r*h] (line :242 rebinds `inputs`, :258 concatenates again)."""
    vs = vs.sub('conv2dgru_cell')
    tile_concat = isinstance(inputs, (list, tuple))
    if tile_concat:
        inputs, inputs_non_spatial = inputs
    g = vs.sub('gates')
    inputs = torch.cat([inputs, state], dim=-1)
    concat = tf_ops.conv2d(inputs, g['kernel'], (1, 1), 'SAME')
    if not normalized:
        concat = concat + g['bias']
    if tile_concat:
        concat = concat + (inputs_non_spatial @ g['weights'])[:, None, None, :]
    if normalized:
        concat = ops.fused_instance_norm(concat, g['reset_update/gamma'], g['reset_update/beta'])
    r, u = torch.chunk(concat, 2, dim=-1)
    r, u = torch.sigmoid(r), torch.sigmoid(u)
    cs = vs.sub('candidate')
    inputs = torch.cat([inputs, r * state], dim=-1)
    candidate = tf_ops.conv2d(inputs, cs['kernel'], (1, 1), 'SAME')
    if not normalized:
        candidate = candidate + cs['bias']
    if tile_concat:
        candidate = candidate + (inputs_non_spatial @ cs['weights'])[:, None, None, :]
    if normalized:
        candidate = ops.fused_instance_norm(candidate, cs['state/gamma'], cs['state/beta'])
    c = torch.tanh(candidate)
    new_h = u * state + (1 - u) * c
    return new_h, new_h


# --------------------------------------------------------------------------------------------
# flow_ops.py
# --------------------------------------------------------------------------------------------
def image_warp(im, flow):
    """flow_ops.py:4-79: backward bilinear warp. im [N,H,W,C], flow [N,H,W,2] (x, y)."""
    N, H, W, C = im.shape
    flow_floor = torch.floor(flow)
    w = flow - flow_floor
    fx = flow_floor[..., 0].long()
    fy = flow_floor[..., 1].long()
    xw, yw = w[..., 0:1], w[..., 1:2]
    wa = (1 - xw) * (1 - yw)
    wb = (1 - xw) * yw
    wc = xw * (1 - yw)
    wd = xw * yw
    pos_x = torch.arange(W)[None, None, :].expand(N, H, W)
    pos_y = torch.arange(H)[None, :, None].expand(N, H, W)
    x0 = (pos_x + fx).clamp(0, W - 1)
    x1 = (pos_x + fx + 1).clamp(0, W - 1)
    y0 = (pos_y + fy).clamp(0, H - 1)
    y1 = (pos_y + fy + 1).clamp(0, H - 1)
    flat = im.reshape(N, H * W, C)

    def gather(yy, xx):
        idx = (yy * W + xx).reshape(N, H * W, 1).expand(N, H * W, C)
        return torch.gather(flat, 1, idx).reshape(N, H, W, C)

    return wa * gather(y0, x0) + wb * gather(y1, x0) + wc * gather(y0, x1) + wd * gather(y1, x1)


# --------------------------------------------------------------------------------------------
# savp_model.py helpers
# --------------------------------------------------------------------------------------------
def identity_kernel(kernel_size):
    """savp_model.py:968-980."""
    kh, kw = kernel_size
    kernel = np.zeros(kernel_size)

    def center_slice(k):
        if k % 2 == 0:
            return slice(k // 2 - 1, k // 2 + 1)
        return slice(k // 2, k // 2 + 1)

    kernel[center_slice(kh), center_slice(kw)] = 1.0
    kernel /= np.sum(kernel)
    return kernel


def apply_cdna_kernels(image, kernels):
    """savp_model.py:893-923 (dilation 1). image [B,H,W,C]; kernels [B,kh,kw,K] -> list of K [B,H,W,C]."""
    B, H, W, C = image.shape
    _, kh, kw, K = kernels.shape
    image_padded = ops.pad2d(image, [kh, kw], padding='SAME', mode='SYMMETRIC')          # :908
    kernels = kernels.permute(1, 2, 0, 3).reshape(kh, kw, B, K)                           # :913-914
    image_transposed = image_padded.permute(3, 1, 2, 0)                                   # :916
    outputs = tf_ops.depthwise_conv2d(image_transposed, kernels, padding='VALID')         # :918
    outputs = outputs.reshape(C, H, W, B, K).permute(4, 3, 1, 2, 0)                       # :920-921
    return list(torch.unbind(outputs, dim=0))


def apply_dna_kernels(image, kernels):
    """savp_model.py:858-890 (dilation 1). kernels [B,H,W,kh,kw,K]."""
    B, H, W, C = image.shape
    _, _, _, kh, kw, K = kernels.shape
    image_padded = ops.pad2d(image, [kh, kw], padding='SAME', mode='SYMMETRIC')
    outs = 0
    for u in range(kh):
        for v in range(kw):
            patch = image_padded[:, u:u + H, v:v + W, :]                     # [B,H,W,C]
            outs = outs + patch[..., None] * kernels[:, :, :, u, v, None, :]  # [B,H,W,C,K]
    return list(torch.unbind(outs, dim=-1))


def apply_flows(image, flows):
    """savp_model.py:955-965 (single last frame)."""
    return [image_warp(image, flows[..., i]) for i in range(flows.shape[-1])]


def layer_specs(hp, height, width):
    """savp_model.py:179-237."""
    ngf = hp.ngf
    scale_size = min(height, width)
    if scale_size >= 256:
        enc = [(ngf, False), (ngf * 2, False), (ngf * 4, True), (ngf * 8, True), (ngf * 8, True)]
        dec = [(ngf * 8, True), (ngf * 4, True), (ngf * 2, False), (ngf, False), (ngf, False)]
    elif scale_size >= 128:
        enc = [(ngf, False), (ngf * 2, True), (ngf * 4, True), (ngf * 8, True)]
        dec = [(ngf * 8, True), (ngf * 4, True), (ngf * 2, False), (ngf, False)]
    elif scale_size >= 64:
        enc = [(ngf, True), (ngf * 2, True), (ngf * 4, True)]
        dec = [(ngf * 2, True), (ngf, True), (ngf, False)]
    elif scale_size >= 32:
        enc = [(ngf, True), (ngf * 2, True)]
        dec = [(ngf, True), (ngf, False)]
    else:
        raise NotImplementedError
    total_stride = 2 ** len(enc)
    if (height % total_stride) or (width % total_stride):
        raise ValueError("The image has dimension (%d, %d), but it should be divisible "
                         "by the total stride, which is %d." % (height, width, total_stride))
    return enc, dec


def _norm_act(vs, h, hp):
    if hp.norm_layer == 'instance':
        n = vs.sub('InstanceNorm')
        h = ops.fused_instance_norm(h, n['gamma'], n['beta'])
    elif hp.norm_layer != 'none':
        raise NotImplementedError(hp.norm_layer)
    if hp.activation_layer != 'relu':
        raise NotImplementedError(hp.activation_layer)
    return torch.relu(h)


def _maybe_tile_concat(layer_fn, vs, opname, h, **kw):
    """savp_model.py:983-993."""
    v = vs.sub(opname)
    if isinstance(h, (list, tuple)):
        spatial, non_spatial = h
        return layer_fn(spatial, v['kernel'], v['bias'], **kw) + \
            ops.dense(non_spatial, vs['dense/kernel'])[:, None, None, :]
    return layer_fn(h, v['kernel'], v['bias'], **kw)


def _downsample(vs, h, hp, kernel_size):
    if hp.downsample_layer != 'conv_pool2d':
        raise NotImplementedError(hp.downsample_layer)
    return _maybe_tile_concat(ops.conv_pool2d, vs, 'conv_pool2d', h, strides=(2, 2))


def _upsample(vs, h, hp):
    if hp.upsample_layer != 'upsample_conv2d':
        raise NotImplementedError(hp.upsample_layer)
    return _maybe_tile_concat(ops.upsample_conv2d, vs, 'upsample_conv2d', h, strides=(2, 2))


def _conv_rnn(vs, inputs, state, filters, hp):
    """SAVPCell._conv_rnn_func, savp_model.py:364-391.  ablation_conv_rnn_norm (:380-384): the cell is built WITHOUT a normalizer (bias
    path of the cells) and the layer's output h -- not the state handed to the next step -- goes through normalizer_fn, whose variables
    live in its default scope `InstanceNorm` beside the cell's."""
    normalized = hp.conv_rnn_norm_layer != 'none'
    if normalized and hp.conv_rnn_norm_layer != 'instance':
        raise NotImplementedError(hp.conv_rnn_norm_layer)
    if hp.conv_rnn == 'lstm':
        cell = conv_lstm_cell
    elif hp.conv_rnn == 'gru':
        cell = conv_gru_cell
    else:
        raise NotImplementedError
    if getattr(hp, 'ablation_conv_rnn_norm', False):
        if not normalized:
            raise TypeError("ablation_conv_rnn_norm with conv_rnn_norm_layer='none': the reference calls normalizer_fn = None (:384)")
        h, state = cell(vs, inputs, state, filters, False)
        return ops.fused_instance_norm(h, vs['InstanceNorm/gamma'], vs['InstanceNorm/beta']), state
    return cell(vs, inputs, state, filters, normalized)


def _conv_rnn_layer(vs, idx, conv_rnn_h, conv_rnn_states, new_conv_rnn_states, filters, hp):
    """The `if use_conv_rnn:` block of SAVPCell.call (savp_model.py:465-482 / :501-517).  ablation_rnn: the recurrent cell is replaced by
    conv2d 5x5 -> norm_layer -> activation under scope `conv_h<idx>` and there is no state."""
    if getattr(hp, 'ablation_rnn', False):
        s = vs.sub('conv_h%d' % idx)
        h = _maybe_tile_concat(ops.conv2d, s, 'conv2d', conv_rnn_h)
        return _norm_act(s, h, hp)
    s = vs.sub('%s_h%d' % (hp.conv_rnn, idx))
    state = conv_rnn_states[len(new_conv_rnn_states)]
    h, state = _conv_rnn(s, conv_rnn_h, state, filters, hp)
    new_conv_rnn_states.append(state)
    return h


def savp_cell_zero_state(images, hp, zs=None, vs=None, n_states=0):
    """SAVPCell.zero_state, savp_model.py:263-308,344-352.

    learn_initial_state (:295-307): the conv-RNN states and the rnn_z state are variables `initial_state_<i>/initial_state`, i = position in
    nest.flatten of {'conv_rnn_states': [...], 'rnn_z_state': ...} (dict keys in sorted order, LSTM state tuples as (c, h)), zero-initialised,
    created in the scope the cell is constructed in (vs = that scope: `generator/`, savp_model.py:689-693) and tiled over the batch (:346-348);
    both unrolls of generator_fn share them (the second one reuses the scope, :730-732)."""
    T, B, H, W, C = images.shape
    enc, dec = layer_specs(hp, H, W)
    dt = images.dtype
    states = []
    h_, w_ = H, W
    for out_channels, use_conv_rnn in enc:
        h_, w_ = h_ // 2, w_ // 2
        if use_conv_rnn:
            states.append((h_, w_, out_channels))
    for out_channels, use_conv_rnn in dec:
        h_, w_ = h_ * 2, w_ * 2
        if use_conv_rnn:
            states.append((h_, w_, out_channels))
    if getattr(hp, 'ablation_rnn', False):
        states = []                                                              # `and not self.hparams.ablation_rnn` (:272,277,288)
    learn = bool(getattr(hp, 'learn_initial_state', False))
    if learn and vs is None:
        raise ValueError('learn_initial_state needs the scope the initial-state variables live in')
    counter = [0]

    def initial(shape):
        if not learn:
            return torch.zeros((B,) + shape, dtype=dt)
        v = vs['initial_state_%d/initial_state' % counter[0]]
        counter[0] += 1
        assert tuple(v.shape) == shape, (tuple(v.shape), shape)
        return v.to(dt)[None].expand((B,) + shape)                              # tf.tile(x[None], [batch_size, 1, ...]) (:346-348)

    conv_rnn_states = []
    for (sh, sw, sc) in states:
        if hp.conv_rnn == 'lstm':
            c0 = initial((sh, sw, sc))                                           # LSTMStateTuple(c, h): c first in nest.flatten
            h0 = initial((sh, sw, sc))
            conv_rnn_states.append((c0, h0))
        else:
            conv_rnn_states.append(initial((sh, sw, sc)))
    st = {'time': 0, 'gen_image': torch.zeros(B, H, W, C, dtype=dt),
          'last_images': [images[0]] * hp.last_frames, 'conv_rnn_states': conv_rnn_states}
    if zs is not None and hp.use_rnn_z and not getattr(hp, 'ablation_rnn', False):
        if hp.rnn == 'lstm':
            c0 = initial((hp.nz,))
            h0 = initial((hp.nz,))
            st['rnn_z_state'] = (c0, h0)
        else:                                                                    # GRUCell: the state is h (:288-291)
            st['rnn_z_state'] = initial((hp.nz,))
    if n_states:
        st['gen_state'] = torch.zeros(B, n_states, dtype=dt)                     # state_size['gen_state'] (:291-292), never learned (:295-297)
    return st


def savp_cell_call(vs, inputs, states, all_images, ground_truth_t, hp):
    """SAVPCell.call, savp_model.py:393-686: one timestep.

    inputs: {'images': [B,H,W,C], optional 'zs': [B,nz]}; all_images: the (sliced) [T-1,...] input sequence
    (self.inputs['images'] in the reference); ground_truth_t: bool [B] = self.ground_truth[t].
    """
    image_in = inputs['images']
    B, H, W, C = image_in.shape
    enc_specs, dec_specs = layer_specs(hp, H, W)
    conv_rnn_states = states['conv_rnn_states']
    t = states['time']

    gt = ground_truth_t.reshape(B, 1, 1, 1)
    image = torch.where(gt, image_in, states['gen_image'])                   # :406
    last_images = states['last_images'][1:] + [image]                         # :407

    # :411-422: the robot state follows the same schedule as the image (ground truth, else the cell's own prediction); actions and the
    # (stop-gradient) state join the latent in every tile-concatenated slice, state_pred sees actions and state with their gradient
    state_action, sa_z = [], []
    if 'states' in inputs:
        state = torch.where(ground_truth_t.reshape(B, 1), inputs['states'], states['gen_state'])     # :412
    if 'actions' in inputs:
        state_action.append(inputs['actions'])
        sa_z.append(inputs['actions'])
    if 'states' in inputs:
        state_action.append(state)
        sa_z.append(state.detach())                                              # tf.stop_gradient (:421-422)
    state_action_z = None
    rnn_z_state = None
    if 'zs' in inputs:
        if hp.use_rnn_z and getattr(hp, 'ablation_rnn', False):                  # :426-429: dense + tanh under scope fc_z, no state
            v = vs.sub('fc_z')
            state_action_z = torch.tanh(ops.dense(inputs['zs'], v['dense/kernel'], v['dense/bias']))
        elif hp.use_rnn_z:
            if hp.rnn == 'lstm':
                v = vs.sub('lstm_z').sub('basic_lstm_cell')
                c0, h0 = states['rnn_z_state']
                rnn_z, rnn_z_state = tf_ops.lstm_cell(inputs['zs'], c0, h0, v['kernel'], v['bias'])  # :431
            elif hp.rnn == 'gru':                                                # _rnn_func, :358-359: tf.contrib.rnn.GRUCell
                v = vs.sub('gru_z').sub('gru_cell')
                rnn_z, rnn_z_state = tf_ops.gru_cell(inputs['zs'], states['rnn_z_state'], v['gates/kernel'], v['gates/bias'],
                                                     v['candidate/kernel'], v['candidate/bias'])
            else:
                raise NotImplementedError(hp.rnn)                                 # :360-361
            state_action_z = rnn_z
        else:
            state_action_z = inputs['zs']
    if sa_z:                                                                     # concat(state_action_z, axis=-1) (:436-444)
        state_action_z = torch.cat(sa_z + ([state_action_z] if state_action_z is not None else []), dim=-1)

    def add_z(h):
        if state_action_z is None:
            # concat of zero tensors: tile_concat with a [B,0] tensor adds no channels (:436-444)
            return h
        if hp.use_tile_concat:
            return ops.tile_concat([h, state_action_z[:, None, None, :]], axis=-1)
        return [h, state_action_z]

    layers = []
    new_conv_rnn_states = []
    for i, (out_channels, use_conv_rnn) in enumerate(enc_specs):
        s = vs.sub('h%d' % i)
        if i == 0:
            h = torch.cat([image, all_images[0]], dim=-1)                     # :451
            kernel_size = (5, 5)
        else:
            h = layers[-1][-1]
            kernel_size = (3, 3)
        if hp.where_add == 'all' or (hp.where_add == 'input' and i == 0):
            h = add_z(h)
        h = _downsample(s, h, hp, kernel_size)
        h = _norm_act(s, h, hp)
        if use_conv_rnn:
            conv_rnn_h = add_z(h) if hp.where_add == 'all' else h
            conv_rnn_h = _conv_rnn_layer(vs, i, conv_rnn_h, conv_rnn_states, new_conv_rnn_states, out_channels, hp)
        layers.append((h, conv_rnn_h) if use_conv_rnn else (h,))

    num_encoder_layers = len(layers)
    for i, (out_channels, use_conv_rnn) in enumerate(dec_specs):
        s = vs.sub('h%d' % len(layers))
        if i == 0:
            h = layers[-1][-1]
        else:
            h = torch.cat([layers[-1][-1], layers[num_encoder_layers - i - 1][-1]], dim=-1)   # :491
        if hp.where_add == 'all' or (hp.where_add == 'middle' and i == 0):
            h = add_z(h)
        h = _upsample(s, h, hp)
        h = _norm_act(s, h, hp)
        if use_conv_rnn:
            conv_rnn_h = add_z(h) if hp.where_add == 'all' else h
            conv_rnn_h = _conv_rnn_layer(vs, len(layers), conv_rnn_h, conv_rnn_states, new_conv_rnn_states, out_channels, hp)
        layers.append((h, conv_rnn_h) if use_conv_rnn else (h,))
    assert len(new_conv_rnn_states) == len(conv_rnn_states)

    nl = len(layers)
    extra = {}
    kernels = flows = None
    if hp.last_frames and hp.num_transformed_images:
        if hp.last_frames != 1:
            raise NotImplementedError('last_frames != 1')
        nk = hp.last_frames * hp.num_transformed_images
        if hp.transformation == 'flow':
            s = vs.sub('h%d_flow' % nl)
            h_flow = ops.conv2d(layers[-1][-1], s['conv2d/kernel'], s['conv2d/bias'])
            h_flow = _norm_act(s, h_flow, hp)
            s = vs.sub('flows')
            flows = ops.conv2d(h_flow, s['conv2d/kernel'], s['conv2d/bias'])
            flows = flows.reshape(B, H, W, 2, nk)
        else:
            kh, kw = hp.kernel_size
            if hp.transformation == 'dna':
                s = vs.sub('h%d_dna_kernel' % nl)
                hk = ops.conv2d(layers[-1][-1], s['conv2d/kernel'], s['conv2d/bias'])
                hk = _norm_act(s, hk, hp)
                s = vs.sub('dna_kernels')
                kernels = ops.conv2d(hk, s['conv2d/kernel'], s['conv2d/bias'])
                kernels = kernels.reshape(B, H, W, kh, kw, nk)
                ident = torch.as_tensor(identity_kernel((kh, kw)), dtype=kernels.dtype)
                kernels = kernels + ident[None, None, None, :, :, None]
                spatial_axes = (3, 4)
            elif hp.transformation == 'cdna':
                s = vs.sub('cdna_kernels')
                smallest_layer = layers[num_encoder_layers - 1][-1]
                kernels = ops.dense(ops.flatten(smallest_layer), s['dense/kernel'], s['dense/bias'])  # :549
                kernels = kernels.reshape(B, kh, kw, nk)
                ident = torch.as_tensor(identity_kernel((kh, kw)), dtype=kernels.dtype)
                kernels = kernels + ident[None, :, :, None]                    # :551
                spatial_axes = (1, 2)
            else:
                raise ValueError('Invalid transformation %s' % hp.transformation)
            kernels = torch.relu(kernels - RELU_SHIFT) + RELU_SHIFT            # :558
            kernels = kernels / kernels.sum(dim=spatial_axes, keepdim=True)     # :559
            extra['kernels'] = kernels

    if hp.generate_scratch_image:
        s = vs.sub('h%d_scratch' % nl)
        h_scratch = ops.conv2d(layers[-1][-1], s['conv2d/kernel'], s['conv2d/bias'])
        h_scratch = _norm_act(s, h_scratch, hp)
        s = vs.sub('scratch_image')
        scratch_image = torch.sigmoid(ops.conv2d(h_scratch, s['conv2d/kernel'], s['conv2d/bias']))

    transformed_images = []
    if hp.last_frames and hp.num_transformed_images:
        if hp.transformation == 'flow':
            transformed_images.extend(apply_flows(last_images[-1], flows))
        elif hp.transformation == 'cdna':
            transformed_images.extend(apply_cdna_kernels(last_images[-1], kernels))
        else:
            transformed_images.extend(apply_dna_kernels(last_images[-1], kernels))
    if hp.prev_image_background:
        transformed_images.append(image)
    if hp.first_image_background and not hp.context_images_background:
        transformed_images.append(all_images[0])
    if hp.last_image_background and not hp.context_images_background:
        transformed_images.append(all_images[hp.context_frames - 1])
    if hp.last_context_image_background and not hp.context_images_background:
        transformed_images.append(all_images[t] if t < hp.context_frames else all_images[hp.context_frames - 1])
    if hp.context_images_background:
        transformed_images.extend(list(torch.unbind(all_images[:hp.context_frames], dim=0)))
    if hp.generate_scratch_image:
        transformed_images.append(scratch_image)

    if len(transformed_images) > 1:
        s = vs.sub('h%d_masks' % nl)
        h_masks = ops.conv2d(layers[-1][-1], s['conv2d/kernel'], s['conv2d/bias'])
        h_masks = _norm_act(s, h_masks, hp)
        s = vs.sub('masks')
        if hp.dependent_mask:
            h_masks = torch.cat([h_masks] + transformed_images, dim=-1)       # :632
        mask_logits = ops.conv2d(h_masks, s['conv2d/kernel'], s['conv2d/bias'])
        masks = torch.softmax(mask_logits, dim=-1)                            # :634
        masks = list(torch.split(masks, 1, dim=-1))
        extra['mask_logits'] = mask_logits
    elif len(transformed_images) == 1:
        masks = [torch.ones(B, H, W, 1, dtype=image.dtype)]
    else:
        raise ValueError("Either one of the following should be true: "
                         "last_frames and num_transformed_images, first_image_background, "
                         "prev_image_background, generate_scratch_image")

    gen_image = sum(ti * m for ti, m in zip(transformed_images, masks))        # :645-646

    outputs = {'gen_images': gen_image,
               'transformed_images': torch.stack(transformed_images, dim=-1),
               'masks': torch.stack(masks, dim=-1)}
    if hp.transformation == 'flow':
        outputs['gen_flows'] = flows
    outputs.update({'_' + k: v for k, v in extra.items()})
    new_states = {'time': t + 1, 'gen_image': gen_image, 'last_images': last_images,
                  'conv_rnn_states': new_conv_rnn_states}
    if rnn_z_state is not None:
        new_states['rnn_z_state'] = rnn_z_state
    if 'states' in inputs:                                                       # :655-658,666-667,684-685
        v = vs.sub('state_pred')
        gen_state = ops.dense(torch.cat(state_action, dim=-1), v['dense/kernel'], v['dense/bias'])
        outputs['gen_states'] = gen_state
        new_states['gen_state'] = gen_state
    return outputs, new_states


def generator_given_z_fn(vs, inputs, mode, hp, ground_truth_sampling=None):
    """savp_model.py:689-696 + tf_utils.unroll_rnn (tf_utils.py:134-141).

    ground_truth_sampling: bool [T-1-context_frames, B] (the Bernoulli draw of savp_model.py:309-330);
    None means all-False (schedule_sampling == 'none' or mode != 'train').
    """
    T1 = hp.sequence_length - 1
    images = inputs['images'][:T1]                                           # maybe_pad_or_slice
    zs = inputs.get('zs')
    B = images.shape[1]
    if ground_truth_sampling is None or mode != 'train' or hp.schedule_sampling == 'none':
        ground_truth_sampling = torch.zeros(T1 - hp.context_frames, B, dtype=torch.bool)
    ground_truth = torch.cat([torch.ones(hp.context_frames, B, dtype=torch.bool),
                              torch.as_tensor(ground_truth_sampling, dtype=torch.bool)], dim=0)   # :333-334
    cell_vs = vs.sub('rnn').sub('savp_cell')
    cond = {k: inputs[k][:T1] for k in ('actions', 'states') if k in inputs}   # maybe_pad_or_slice of every input (:690-691)
    states = savp_cell_zero_state(images, hp, zs, vs, n_states=cond['states'].shape[-1] if 'states' in cond else 0)
    outs = []
    for t in range(T1):
        step_in = {'images': images[t]}
        if zs is not None:
            step_in['zs'] = zs[t]
        for k, v in cond.items():
            step_in[k] = v[t]
        o, states = savp_cell_call(cell_vs, step_in, states, images, ground_truth[t], hp)
        outs.append(o)
    outputs = OrderedDict()
    for k in outs[0]:
        outputs[k] = torch.stack([o[k] for o in outs], dim=0)
    outputs['ground_truth_sampling_mean'] = ground_truth[hp.context_frames:].to(images.dtype).mean()
    return outputs


# --------------------------------------------------------------------------------------------
# networks.py
# --------------------------------------------------------------------------------------------
def encoder(vs, inputs, nef=64, n_layers=3, norm_layer='instance'):
    """networks.encoder, networks.py:12-32. inputs [N,H,W,C] -> [N, nef*4]."""
    paddings = [[0, 0], [1, 1], [1, 1], [0, 0]]
    s = vs.sub('layer_1')
    h = ops.conv2d(tf_ops.pad_constant(inputs, paddings), s['conv2d/kernel'], s['conv2d/bias'],
                   strides=(2, 2), padding='VALID')
    h = ops.lrelu(h, 0.2)
    for i in range(1, n_layers):
        s = vs.sub('layer_%d' % (i + 1))
        h = ops.conv2d(tf_ops.pad_constant(h, paddings), s['conv2d/kernel'], s['conv2d/bias'],
                       strides=(2, 2), padding='VALID')
        if norm_layer == 'instance':
            h = ops.fused_instance_norm(h, s['InstanceNorm/gamma'], s['InstanceNorm/beta'])
        elif norm_layer != 'none':
            raise NotImplementedError
        h = ops.lrelu(h, 0.2)
    return h.mean(dim=(1, 2))                                                # pool2d(avg, full window) + squeeze


def _sn(vs, opname, sn_state):
    """Spectral-normalised kernel for variable scope vs/opname; records u_final in sn_state."""
    v = vs.sub(opname)
    key = v.prefix + 'u'
    W_bar, u_final = ops.spectral_normed_weight(v['kernel'], v['u'])
    if sn_state is not None:
        sn_state[key] = u_final.detach()
    return W_bar


VIDEO_D_LAYERS = [  # (scope, filters multiplier, kernel, strides)  networks.py:83-102
    ('sn_conv0_0', 1, 3, (1, 1, 1)),
    ('sn_conv0_1', 2, 4, (1, 2, 2)),
    ('sn_conv1_0', 2, 3, (1, 1, 1)),
    ('sn_conv1_1', 4, 4, (1, 2, 2)),
    ('sn_conv2_0', 4, 3, (1, 1, 1)),
    ('sn_conv2_1', 8, 4, (2, 2, 2)),
    ('sn_conv3_0', 8, 3, (1, 1, 1)),
]


def video_sn_discriminator(vs, clips, ndf=64, sn_state=None):
    """networks.py:72-108. clips time-major [D,B,H,W,C]; returns list of 7 feature maps (time-major
    [D',B,H',W',C']) + logits [B,1]."""
    x = clips.permute(1, 0, 2, 3, 4)
    B = x.shape[0]
    paddings = [[0, 0], [1, 1], [1, 1], [1, 1], [0, 0]]
    layers = []
    for scope, mult, k, strides in VIDEO_D_LAYERS:
        s = vs.sub(scope)
        W = _sn(s, 'conv3d', sn_state)
        x = ops.conv3d(tf_ops.pad_constant(x, paddings), W, s['bias'], strides=strides, padding='VALID')
        x = ops.lrelu(x, 0.1)
        layers.append(x)
    s = vs.sub('sn_fc4')
    W = _sn(s, 'dense', sn_state)
    logits = ops.dense(x.reshape(B, -1), W, s['dense/bias'])
    out = [l.permute(1, 0, 2, 3, 4) for l in layers]
    out.append(logits)
    return out


IMAGE_D_LAYERS = [  # networks.py:45-64
    ('sn_conv0_0', 1, 3, 1), ('sn_conv0_1', 2, 4, 2), ('sn_conv1_0', 2, 3, 1), ('sn_conv1_1', 4, 4, 2),
    ('sn_conv2_0', 4, 3, 1), ('sn_conv2_1', 8, 4, 2), ('sn_conv3_0', 8, 3, 1),
]


def image_sn_discriminator(vs, images, ndf=64, sn_state=None):
    """networks.py:35-69. images [B,H,W,C]."""
    x = images
    B = x.shape[0]
    paddings = [[0, 0], [1, 1], [1, 1], [0, 0]]
    layers = []
    for scope, mult, k, stride in IMAGE_D_LAYERS:
        s = vs.sub(scope)
        W = _sn(s, 'conv2d', sn_state)
        x = ops.conv2d(tf_ops.pad_constant(x, paddings), W, s['conv2d/bias'], strides=(stride, stride), padding='VALID')
        x = ops.lrelu(x, 0.1)
        layers.append(x)
    s = vs.sub('sn_fc4')
    W = _sn(s, 'dense', sn_state)
    logits = ops.dense(x.reshape(B, -1), W, s['dense/bias'])
    layers.append(logits)
    return layers


# --------------------------------------------------------------------------------------------
# posterior / discriminator / generator fns
# --------------------------------------------------------------------------------------------
def basic_lstm_unroll(kernel, bias, xs, forget_bias=1.0):
    """tf.contrib.rnn.BasicLSTMCell(num_units) under tf.nn.dynamic_rnn (tf_utils.unroll_rnn, tf_utils.py:134-141), zero
    initial state.  kernel [in + units, 4*units], gate order i, j, f, o (rnn_cell_impl.BasicLSTMCell.call):
      c' = c * sigmoid(f + forget_bias) + sigmoid(i) * tanh(j);  h' = tanh(c') * sigmoid(o).   xs [T, B, in] -> hs [T, B, units]."""
    units = kernel.shape[1] // 4
    T, B = xs.shape[:2]
    c = torch.zeros(B, units, dtype=xs.dtype)
    h = torch.zeros(B, units, dtype=xs.dtype)
    out = []
    for t in range(T):
        gates = torch.cat([xs[t], h], dim=-1) @ kernel + bias
        i, j, f, o = torch.chunk(gates, 4, dim=-1)
        c = c * torch.sigmoid(f + forget_bias) + torch.sigmoid(i) * torch.tanh(j)
        h = torch.tanh(c) * torch.sigmoid(o)
        out.append(h)
    return torch.stack(out)


def _e_rnn(vs, h, hp):
    """savp_model.py:32-43 / :66-76: dense to nef*4 under scope layer_{n_layers+1}, then the `rnn` cell over time under scope
    hparams.rnn (dynamic_rnn's default scope 'rnn' -> variables '<rnn>/rnn/basic_lstm_cell/{kernel,bias}').  h [T, B, F]."""
    T, B = h.shape[:2]
    s = vs.sub('layer_%d' % (hp.n_layers + 1))
    h = ops.dense(h.reshape(T * B, -1), s['dense/kernel'], s['dense/bias']).reshape(T, B, -1)
    r = vs.sub(hp.rnn)
    if hp.rnn == 'lstm':
        return basic_lstm_unroll(r['rnn/basic_lstm_cell/kernel'], r['rnn/basic_lstm_cell/bias'], h)
    if hp.rnn == 'gru':                                                          # :38-39: tf.contrib.rnn.GRUCell, zero initial state
        g = r.sub('rnn').sub('gru_cell')
        state = torch.zeros(B, g['candidate/bias'].shape[0], dtype=h.dtype)
        out = []
        for t in range(T):
            o, state = tf_ops.gru_cell(h[t], state, g['gates/kernel'], g['gates/bias'], g['candidate/kernel'], g['candidate/bias'])
            out.append(o)
        return torch.stack(out)
    raise NotImplementedError(hp.rnn)


def _z_heads(vs, h):
    T, B = h.shape[:2]
    flat = h.reshape(T * B, -1)
    z_mu = ops.dense(flat, vs['z_mu/dense/kernel'], vs['z_mu/dense/bias']).reshape(T, B, -1)
    z_ls = ops.dense(flat, vs['z_log_sigma_sq/dense/kernel'], vs['z_log_sigma_sq/dense/bias'])
    return {'zs_mu': z_mu, 'zs_log_sigma_sq': torch.clamp(z_ls, -10, 10).reshape(T, B, -1)}


def posterior_fn(vs, inputs, hp):
    """savp_model.py:21-51; actions are tile-concatenated to the frame pairs (:24-26)."""
    images = inputs['images']
    image_pairs = torch.cat([images[:-1], images[1:]], dim=-1)
    if 'actions' in inputs:
        image_pairs = ops.tile_concat([image_pairs, inputs['actions'][..., None, None, :]], axis=-1)
    T1, B = image_pairs.shape[:2]
    flat = image_pairs.reshape((T1 * B,) + tuple(image_pairs.shape[2:]))
    h = encoder(vs, flat, nef=hp.nef, n_layers=hp.n_layers, norm_layer=hp.norm_layer).reshape(T1, B, -1)
    if hp.use_e_rnn:
        h = _e_rnn(vs, h, hp)
    return _z_heads(vs, h)


def prior_fn(vs, inputs, hp):
    """savp_model.py:54-85: encoder on the context_frames-1 context pairs, zero features for the remaining
    sequence_length - context_frames steps, dense + rnn over all T-1 steps, the two heads."""
    images = inputs['images']
    c = hp.context_frames
    image_pairs = torch.cat([images[:c - 1], images[1:c]], dim=-1)
    if 'actions' in inputs:
        # :57-59: tile_concat broadcasts size-1 dimensions only (ops.py:995-1000), so the reference fails here unless the actions
        # cover exactly the context pairs (context_frames == sequence_length); restated, not repaired
        if inputs['actions'].shape[0] != image_pairs.shape[0]:
            raise AssertionError('tile_concat: %d action steps against %d context frame pairs' % (inputs['actions'].shape[0], image_pairs.shape[0]))
        image_pairs = ops.tile_concat([image_pairs, inputs['actions'][..., None, None, :]], axis=-1)
    Tc, B = image_pairs.shape[:2]
    flat = image_pairs.reshape((Tc * B,) + tuple(image_pairs.shape[2:]))
    h = encoder(vs, flat, nef=hp.nef, n_layers=hp.n_layers, norm_layer=hp.norm_layer).reshape(Tc, B, -1)
    h = torch.cat([h, torch.zeros((hp.sequence_length - c,) + tuple(h.shape[1:]), dtype=h.dtype)], dim=0)
    h = _e_rnn(vs, h, hp)
    return _z_heads(vs, h)


def discriminator_given_video_fn(vs, targets, hp, t_sample, t_start, sn_state=None):
    """savp_model.py:88-126. targets [T-1,B,H,W,C]; t_sample,t_start int [B] (injected draws of :93,:97)."""
    L, B = targets.shape[:2]
    clip_length = hp.clip_length
    ar = torch.arange(B)
    outputs = OrderedDict()
    if hp.image_sn_gan_weight or hp.image_sn_vae_gan_weight:
        image_sample = targets[torch.as_tensor(t_sample, dtype=torch.long), ar]
        feats = image_sn_discriminator(vs.sub('image'), image_sample, ndf=hp.ndf, sn_state=sn_state)
        outputs['discrim_image_sn_logits'] = feats[-1]
        for i, f in enumerate(feats[:-1]):
            outputs['discrim_image_sn_feature%d' % i] = f
    if hp.video_sn_gan_weight or hp.video_sn_vae_gan_weight:
        ts = torch.as_tensor(t_start, dtype=torch.long)
        idx = ts[None, :] + torch.arange(clip_length)[:, None]                 # [clip,B]
        clip_sample = targets[idx, ar[None, :]]                               # [clip,B,H,W,C]
        feats = video_sn_discriminator(vs.sub('video'), clip_sample, ndf=hp.ndf, sn_state=sn_state)
        outputs['discrim_video_sn_logits'] = feats[-1]
        for i, f in enumerate(feats[:-1]):
            outputs['discrim_video_sn_feature%d' % i] = f
    if hp.images_sn_gan_weight or hp.images_sn_vae_gan_weight:
        # tf_utils.with_flat_batch(networks.image_sn_discriminator)(clip_sample)   savp_model.py:119-125
        ts = torch.as_tensor(t_start, dtype=torch.long)
        idx = ts[None, :] + torch.arange(clip_length)[:, None]
        clip_sample = targets[idx, ar[None, :]]                               # [clip,B,H,W,C]
        flat = clip_sample.reshape((clip_length * B,) + tuple(clip_sample.shape[2:]))
        feats = image_sn_discriminator(vs.sub('images'), flat, ndf=hp.ndf, sn_state=sn_state)
        feats = [f.reshape((clip_length, B) + tuple(f.shape[1:])) for f in feats]
        outputs['discrim_images_sn_logits'] = feats[-1]
        for i, f in enumerate(feats[:-1]):
            outputs['discrim_images_sn_feature%d' % i] = f
    return outputs


def discriminator_fn(vs, inputs, outputs, mode, hp, indices, sn_state=None):
    """savp_model.py:129-166.  indices: dict with keys 'enc_real','enc_fake','real','fake', each a
    (t_sample[B], t_start[B]) pair -- the four independent draws one discriminator_fn call makes."""
    if hp.nz == 0:
        d_enc_real, d_enc_fake = OrderedDict(), OrderedDict()
    else:
        if hp.use_same_discriminator:
            evs = vs
        else:
            evs = vs.sub('encoder')
        d_enc_real = discriminator_given_video_fn(evs, inputs['images'][1:], hp, *indices['enc_real'], sn_state=sn_state)
        d_enc_fake = discriminator_given_video_fn(evs, outputs['gen_images_enc'], hp, *indices['enc_fake'], sn_state=sn_state)
    d_real = discriminator_given_video_fn(vs, inputs['images'][1:], hp, *indices['real'], sn_state=sn_state)
    d_fake = discriminator_given_video_fn(vs, outputs['gen_images'], hp, *indices['fake'], sn_state=sn_state)
    out = OrderedDict()
    for suffix, d in (('_real', d_real), ('_fake', d_fake), ('_enc_real', d_enc_real), ('_enc_fake', d_enc_fake)):
        for k, v in d.items():
            out[k + suffix] = v
    return out


def generator_fn(vs, inputs, mode, hp, noise=None):
    """savp_model.py:699-768.  The gen_images_samples visualisation unroll (:745-767, num_samples draws from the prior per sequence,
    not on the train path) is run only when its draws are injected (noise['samples_prior'] / ['samples_prior_eps']).

    noise: {'eps': [T-1,B,nz], 'prior': [T-context,B,nz], 'ground_truth_sampling': bool [T-1-context,B],
            'ground_truth_sampling_enc': same for the posterior unroll}  (each unroll builds its own
    SAVPCell, hence its own Bernoulli draw: savp_model.py:693,730,732);
           'samples_prior': [T-context, S, B, nz] (or, learn_prior, 'samples_prior_eps': [T-1, S, B, nz]) and optionally
           'samples_ground_truth_sampling': bool [T-1-context, S*B] for the samples unroll.
    """
    noise = noise or {}
    if hp.nz == 0:
        return generator_given_z_fn(vs, inputs, mode, hp, noise.get('ground_truth_sampling'))
    outputs_posterior = posterior_fn(vs.sub('encoder'), inputs, hp)
    eps = noise['eps']
    zs_posterior = outputs_posterior['zs_mu'] + torch.sqrt(torch.exp(outputs_posterior['zs_log_sigma_sq'])) * eps
    if hp.learn_prior:                                                                      # :717-721 (noise['prior_eps'] [T-1,B,nz])
        outputs_prior = prior_fn(vs.sub('prior'), inputs, hp)
        zs_prior = outputs_prior['zs_mu'] + torch.sqrt(torch.exp(outputs_prior['zs_log_sigma_sq'])) * noise['prior_eps']
    else:
        outputs_prior = {}
        zs_prior = torch.cat([zs_posterior[:hp.context_frames - 1], noise['prior']], dim=0)     # :724-725
    inputs_posterior = dict(inputs)
    inputs_posterior['zs'] = zs_posterior
    inputs_prior = dict(inputs)
    inputs_prior['zs'] = zs_prior
    gen_post = generator_given_z_fn(vs, inputs_posterior, mode, hp, noise.get('ground_truth_sampling_enc'))
    gen_prior = generator_given_z_fn(vs, inputs_prior, mode, hp, noise.get('ground_truth_sampling'))
    outputs = OrderedDict()
    for k, v in gen_prior.items():
        outputs[k] = v
    for k, v in outputs_prior.items():
        outputs[k + '_prior'] = v                                                           # :735-739
    for k, v in outputs_posterior.items():
        outputs[k + '_enc'] = v
    for k, v in gen_post.items():
        outputs[k + '_enc'] = v
    if 'samples_prior' in noise or 'samples_prior_eps' in noise:                            # :745-767
        if hp.learn_prior:
            eps_s = noise['samples_prior_eps']                                              # [T-1, S, B, nz]
            zs_s = outputs_prior['zs_mu'][:, None] + torch.sqrt(torch.exp(outputs_prior['zs_log_sigma_sq']))[:, None] * eps_s
        else:
            pr = noise['samples_prior']                                                     # [T-context, S, B, nz]
            S = pr.shape[1]
            zs_s = torch.cat([zs_posterior[:hp.context_frames - 1][:, None].expand(-1, S, -1, -1), pr], dim=0)
        S = zs_s.shape[1]
        inputs_s = {k: v[:, None].expand((v.shape[0], S) + tuple(v.shape[1:])).reshape((v.shape[0], S * v.shape[1]) + tuple(v.shape[2:]))
                    for k, v in inputs.items()}                                             # tile along a new axis 1, flatten(1, 2)
        inputs_s['zs'] = zs_s.reshape((zs_s.shape[0], S * zs_s.shape[2], zs_s.shape[3]))
        gen_s = generator_given_z_fn(vs, inputs_s, mode, hp, noise.get('samples_ground_truth_sampling'))['gen_images']
        B = inputs['images'].shape[1]
        gen_s = torch.stack([gen_s[:, i * B:(i + 1) * B] for i in range(S)], dim=-1)         # tf.split(axis=1) + stack(axis=-1)
        outputs['gen_images_samples'] = gen_s
        outputs['gen_images_samples_avg'] = gen_s.mean(dim=-1)
    return outputs
