"""SAVPCell unrolled over time on the HIP kernels: forward and back-propagation through time.

Mirrors SAVPCell.call (/root/reference/video_prediction/models/savp_model.py:393-686) and its unroll
(generator_given_z_fn :689-696, tf_utils.unroll_rnn tf_utils.py:134-141) for the published SAVP family
(conv_pool2d / upsample_conv2d, instance norm, ConvLSTM, CDNA, tile_concat).

MI355X-first layout decisions:
  * every activation lives in a time-major buffer [T-1, N, H, W, C] that stays resident in HBM for the whole step
    (288 GB), so back-propagation never recomputes and every weight gradient is ONE split-K GEMM over all
    timesteps x samples instead of T-1 small ones;
  * concatenations (tile_concat with z, skip connections, [x, h_prev] of the ConvLSTM, [h_masks, transformed
    images]) are never materialised by copies: producers write straight into channel slices of the consumer's
    concat buffer, gradients are read back from the same slices;
  * the posterior and prior unrolls share weights, so they run as ONE unroll of batch 2B (all ops are per-sample).
"""
import os

import torch

from .. import kernels as K
from .. import lib
from ..engine import ConcatConv, ConvLayer, same_pad_before, copy_view, add_views, prep_layers
from ..variables import layer_specs, num_masks

CONV_STATS = os.environ.get('SAVP_CONV_STATS', '1') == '1'      # developer A/B switch of the conv-epilogue statistics
FUSED_ENTRIES = os.environ.get('SAVP_FUSED_ENTRIES', '1') == '1'      # one host call per fused operator (csrc/fused_ops.hip); 0: the halves apart
NORM_BWD_STATS = os.environ.get('SAVP_NORM_BWD_STATS', '1') == '1'      # ... and of the norm-backward sums from the DGRAD that produces dy
EPS_IN = 1e-6   # fused_instance_norm epsilon (layers/normalization.py:37)


def ceil4(n):
    return (n + 3) // 4 * 4


def ceil8(n):
    """Channel counts of convolution INPUT buffers are padded to multiples of 8 (zero channels, zero weight rows): the ring / thin kernels of
    the bf16 datapath take a reduction dimension in whole 8-channel chunks only -- with 4-channel padding KTH's first layer (2 + 32 -> 36
    channels) and its mask convolution (32 + 7 + 3 -> 44) fell to the general kernel, 35 us instead of 17 us per time step each."""
    return (n + 7) // 8 * 8


class Act(object):
    """Time-major activation buffer with an optional gradient twin."""

    def __init__(self, shape, device, grad=True, zero_grad=False, dtype=torch.float32, grad_dtype=torch.float32):
        self.v = torch.zeros(shape, device=device, dtype=dtype)            # the gradient twin is fp32 unless asked otherwise
        self.g = None
        if grad:
            self.g = torch.zeros(shape, device=device, dtype=grad_dtype) if zero_grad else \
                torch.empty(shape, device=device, dtype=grad_dtype)

    def flat(self, t):
        """[T, N, ...] -> [T*N, ...] view (time folded into the batch for the batched weight gradients)."""
        return t.reshape((t.shape[0] * t.shape[1],) + tuple(t.shape[2:]))


class Norm(object):
    def __init__(self, store, scope, T1, N, C, device):
        self.gamma, self.beta = store[scope + 'gamma'], store[scope + 'beta']
        # float64 accumulators (ParamGroup.grad64): every (sample, channel slab) workgroup of a norm-backward launch adds to them
        self.dgamma, self.dbeta = store.grad64(scope + 'gamma'), store.grad64(scope + 'beta')
        self.mean = torch.empty(T1, N, C, device=device)
        self.rstd = torch.empty(T1, N, C, device=device)


class ConcatNorm(object):
    """Instance norms of several heads that are normalised by ONE launch over their concatenated channels (instance norm is per
    channel, so concatenation changes nothing numerically): gamma / beta are gathered into one vector per step, their gradients
    are scattered back to the master variables after the backward pass."""

    def __init__(self, norms, T1, N, device):
        self.norms = norms
        self.sizes = [n.gamma.numel() for n in norms]
        C = sum(self.sizes)
        self.gamma, self.beta = torch.empty(C, device=device), torch.empty(C, device=device)
        self.dgamma, self.dbeta = torch.zeros(C, device=device, dtype=torch.float64), torch.zeros(C, device=device, dtype=torch.float64)
        self.mean = torch.empty(T1, N, C, device=device)
        self.rstd = torch.empty(T1, N, C, device=device)

    def prep(self):
        off = 0
        for n, c in zip(self.norms, self.sizes):
            K.axpby(1.0, n.gamma, 0.0, None, self.gamma[off:off + c])
            K.axpby(1.0, n.beta, 0.0, None, self.beta[off:off + c])
            off += c

    def finish(self):
        off = 0
        for n, c in zip(self.norms, self.sizes):
            n.dgamma.add_(self.dgamma[off:off + c])          # float64 accumulators on both sides
            n.dbeta.add_(self.dbeta[off:off + c])
            off += c
        self.dgamma.zero_()
        self.dbeta.zero_()


class SAVPGenerator(object):
    def __init__(self, store, hp, image_shape, N, train=True, prefix='generator/rnn/savp_cell/', cond=(0, 0)):
        """cond = (n_actions, n_states): widths of inputs['actions'] / inputs['states'] (savp_model.py:411-444): [actions_t | state_t]
        joins the latent in every tile-concatenated slice, the next state is predicted by `state_pred/dense` (:655-658)."""
        H, W, C = image_shape
        self.hp, self.store, self.N, self.H, self.W, self.C = hp, store, N, H, W, C
        self._ones = None
        self.T1 = T1 = hp.sequence_length - 1
        self.train = train
        dev = store.device
        self.dev = dev
        self._cstats = {}                # head name -> the conv's epilogue supplies the instance norm's statistics (decided at first use)
        if hp.conv_rnn not in ('lstm', 'gru') or hp.conv_rnn_norm_layer not in ('instance', 'none') or hp.norm_layer != 'instance':
            raise NotImplementedError('HIP path covers conv_rnn in (lstm, gru) with instance norm (or no normaliser inside the LSTM cell)')
        # The ConvLSTM cell WITHOUT a normaliser (rnn_ops.py:122-125: the gate convolution then has a bias; :148-165 with normalizer_fn None):
        # conv_rnn_norm_layer = 'none', or ablation_conv_rnn_norm (savp_model.py:380-384: the cell is built without one and the layer's OUTPUT
        # h -- not the state handed to the next step -- goes through normalizer_fn, variables under <cell scope>/InstanceNorm/).
        self.cell_plain = bool(hp.conv_rnn_norm_layer == 'none' or hp.ablation_conv_rnn_norm)
        self.out_norm = bool(hp.ablation_conv_rnn_norm)
        if self.out_norm and hp.conv_rnn_norm_layer == 'none':
            raise TypeError("ablation_conv_rnn_norm with conv_rnn_norm_layer='none': the reference calls normalizer_fn = None (savp_model.py:384)")
        if self.cell_plain and hp.conv_rnn != 'lstm' and not hp.ablation_rnn:
            raise NotImplementedError('the normaliser-free cell is on the HIP path for conv_rnn = lstm only')
        if hp.downsample_layer != 'conv_pool2d' or hp.upsample_layer != 'upsample_conv2d' or hp.activation_layer != 'relu':
            raise NotImplementedError('HIP path covers conv_pool2d / upsample_conv2d / relu')
        if hp.transformation not in ('cdna', 'flow', 'dna') or hp.last_frames != 1 or not hp.num_transformed_images:
            raise NotImplementedError('HIP path covers transformation in (cdna, flow, dna) with last_frames=1')
        if tuple(hp.dilation_rate) != (1, 1):
            raise NotImplementedError('dilation_rate != (1, 1)')
        if hp.nz and hp.use_rnn_z and hp.rnn not in ('lstm', 'gru') and not hp.ablation_rnn:
            raise NotImplementedError(hp.rnn)                                  # savp_model.py:360-361
        if hp.where_add not in ('input', 'all', 'middle'):
            raise ValueError('Invalid where_add %s' % hp.where_add)                      # savp_model.py:176-177
        # ablation_rnn (savp_model.py:272-291,426-429,466-474,502-509): no recurrent state anywhere -- every conv-RNN becomes conv2d 5x5
        # (+ tiled z) -> norm -> ReLU under scope conv_h<i>, the latent's cell becomes dense + tanh under scope fc_z
        self.abl_rnn = bool(hp.ablation_rnn)
        # (learn_initial_state under ablation_rnn: the reference's state_size then holds no conv_rnn_states / rnn_z_state entries, so no
        #  initial_state variables exist -- savp_model.py:269-307 -- and the flag does nothing; variables.py creates none either)
        self.nz = nz = hp.nz
        self.use_rnn_z = bool(nz and hp.use_rnn_z)
        self.tv = None                       # (weight, leading samples the loss covers, float64 [1] accumulator): set by the engine for tv_weight
        self.na, self.ns = na, ns = int(cond[0]), int(cond[1])
        self.cw = cw = na + ns                 # conditioning columns in front of the latent: state_action_z = [actions | state | z]
        self.zw = zw = cw + nz                 # width of every tiled slice
        if cw > 32:
            raise NotImplementedError('actions + states wider than 32 (csrc/state_pred.hip keeps them in registers)')
        # where the latent is tile-concatenated (savp_model.py:456-470,492-506): 'all' = the input of every down / upsample conv and of
        # every conv-RNN; 'input' = the first encoder conv only; 'middle' = the first decoder conv only
        # use_tile_concat=False (savp_model.py:458-460,470-471,495-496,506-507 -> _maybe_tile_concat_layer :983-993, rnn_ops.py:128-135,
        # 145-146): the latent enters as dense(z)[:, None, None, :] ADDED to the convolution's output instead of as tiled input channels.
        # Every such sum sits directly in front of an instance norm here (norm_layer / conv_rnn_norm_layer == 'instance' are required
        # above), and a per-(sample, channel) constant is removed exactly by the norm's mean subtraction: the outputs do not depend on
        # z, the `dense/kernel` / `weights` variables and z itself get zero gradients.  The layers therefore run without z channels and
        # those variables keep their (zero-initialised) gradients; pinned against the oracle, which computes the sums literally.
        tile = bool(hp.use_tile_concat)
        zr = zw if (hp.where_add == 'all' and tile) else 0              # tiled channels in a conv-RNN's input [x | actions state z | h]
        g = train
        # bf16 storage of tensors whose ONLY readers are convolutions of the bf16 datapath (round 4; the cell input [x | z | h] and the
        # gate gradient have been stored this way since round 3): the inputs of the down / upsample convolutions behind layer 0, the
        # last decoder layer's output (read by the 3x3 heads) and the gradient of every conv_pool / upsample / head convolution's output
        # (written by the instance norm's backward, read by that convolution's DGRAD / WGRAD).  Those kernels round their operands to
        # bf16 while staging anyway: identical numbers, half the bytes of the batched weight gradients' activation stream, and the
        # ring kernel stages a bf16 source by LDS-DMA.  SAVP_BF16_CONVIO=0 keeps fp32 tensors.
        self.act16 = bool(K.PRECISION['value'] == 1 and hp.conv_rnn == 'lstm' and os.environ.get('SAVP_BF16_CONVIO', '1') == '1')
        a16 = torch.bfloat16 if self.act16 else torch.float32
        enc_specs, dec_specs = layer_specs(hp.ngf, H, W)
        self.ne, self.nd = len(enc_specs), len(dec_specs)

        # ---- layers and their input buffers ----------------------------------------------------------------
        self.layers = []
        h_, w_ = H, W
        prev_f = None
        for i, (f, use_rnn) in enumerate(enc_specs + dec_specs):
            L = {'f': f, 'rnn': use_rnn, 'idx': i, 'dec': i >= self.ne}
            s = prefix + 'h%d/' % i
            j_dec = i - len(enc_specs)
            zc = zw if (tile and (hp.where_add == 'all' or (hp.where_add == 'input' and i == 0) or
                                  (hp.where_add == 'middle' and j_dec == 0))) else 0
            L['zc'], L['zr'] = zc, zr
            if i < self.ne:
                cx = 2 * C if i == 0 else prev_f
                k = 5 if i == 0 else 3
                L['in'] = Act((T1, N, h_, w_, ceil8(cx + zc)), dev, grad=g,      # pad channels stay zero
                              dtype=a16 if i > 0 else torch.float32)     # layer 0 holds the input image: fp32
                L['zoff_in'] = cx
                L['conv'] = ConvLayer(store, s + 'conv_pool2d/kernel', s + 'conv_pool2d/bias', 'pool', (k, k), (2, 2),
                                      (same_pad_before(k + 1, 2, h_), same_pad_before(k + 1, 2, w_)),
                                      cx_pad=ceil8(cx + zc))
                h_, w_ = h_ // 2, w_ // 2
            else:
                j = i - self.ne
                skip = 0 if j == 0 else self.layers[self.ne - j - 1]['f']
                cx = prev_f + skip
                L['in'] = Act((T1, N, h_, w_, cx + zc), dev, grad=g, dtype=a16 if (cx + zc) % 8 == 0 else torch.float32)
                L['zoff_in'] = cx
                L['skip_off'] = prev_f
                h_, w_ = h_ * 2, w_ * 2
                L['conv'] = ConvLayer(store, s + 'upsample_conv2d/kernel', s + 'upsample_conv2d/bias', 'up', (3, 3), (2, 2),
                                      (same_pad_before(6, 2, h_), same_pad_before(6, 2, w_)))
            L['hw'] = (h_, w_)
            # (a decoder layer whose input keeps an odd channel count -- conditioning widths that are no multiple of 8 -- stays fp32 on both
            #  sides: the weight gradient of an upsample conv reads the input as its `dy` operand, and only whole 4-channel groups ride
            #  the bf16-operand kernel)
            pre16 = f % 8 == 0 and (i < self.ne or L['in'].v.dtype == torch.bfloat16 or not self.act16)
            L['pre'] = Act((T1, N, h_, w_, f), dev, grad=g, grad_dtype=a16 if pre16 else torch.float32)
            L['norm'] = Norm(store, s + 'InstanceNorm/', T1, N, f, dev)
            if use_rnn and self.abl_rnn:
                r = prefix + 'conv_h%d/' % i
                L['a'] = Act((T1, N, h_, w_, ceil4(f + zr)), dev, grad=g, dtype=a16 if ceil4(f + zr) % 8 == 0 else torch.float32)
                L['pre2'] = Act((T1, N, h_, w_, f), dev, grad=g, grad_dtype=a16 if f % 8 == 0 else torch.float32)
                L['rconv'] = ConvLayer(store, r + 'conv2d/kernel', r + 'conv2d/bias', 'conv', (5, 5), (1, 1), (2, 2), cx_pad=ceil4(f + zr))
                L['n2'] = Norm(store, r + 'InstanceNorm/', T1, N, f, dev)
            elif use_rnn and hp.conv_rnn == 'gru':
                # Conv2DGRUCell (rnn_ops.py:174-267): a = [x | z | h_prev | r*h_prev]; the gates conv reads the first
                # f+zc+f channels of the same buffer (a channel-slice view), the candidate conv reads all of it
                r = prefix + 'gru_h%d/conv2dgru_cell/' % i
                L['cin1'] = f + zr + f
                L['a'] = Act((T1, N, h_, w_, f + zr + 2 * f), dev, grad=g)
                L['gates'] = Act((T1, N, h_, w_, 2 * f), dev, grad=g)
                L['cand'] = Act((T1, N, h_, w_, f), dev, grad=g)
                L['u'] = torch.empty(T1, N, h_, w_, f, device=dev)
                L['du'] = torch.empty(N, h_, w_, f, device=dev) if g else None
                L['dh_tmp'] = torch.empty(N, h_, w_, f, device=dev) if g else None
                L['rconv'] = ConvLayer(store, r + 'gates/kernel', None, 'conv', (5, 5), (1, 1), (2, 2))
                L['cconv'] = ConvLayer(store, r + 'candidate/kernel', None, 'conv', (5, 5), (1, 1), (2, 2))
                L['n1'] = Norm(store, r + 'gates/reset_update/', T1, N, 2 * f, dev)
                L['n2'] = Norm(store, r + 'candidate/state/', T1, N, f, dev)
            elif use_rnn:
                r = prefix + 'lstm_h%d/basic_conv2dlstm_cell/' % i
                # fused ConvLSTM cell of the bf16 datapath: the gate convolution's epilogue produces the statistics of the first
                # instance norm and stores the gate pre-activations as bf16 (csrc/conv_ring.hip) -> conv + 2 launches per cell
                L['fused'] = (K.PRECISION['value'] == 1 and os.environ.get('SAVP_FUSED_CELL', '1') == '1' and not self.cell_plain and
                              h_ % 8 == 0 and w_ % 8 == 0 and 16 <= f <= 256 and (f & (f - 1)) == 0 and (f + zr + f) % 8 == 0)
                # The cell's input buffer [x | z | h] and the gate gradient are held in bf16 (round 3: validated on MI355X, identical
                # numbers -- their only readers are the gate convolution's FPROP / DGRAD / WGRAD, which round to bf16 when they stage
                # their operands anyway -- and half the bytes; step time unchanged within noise, 61.04 vs 61.21 ms).  The producers
                # (instance norm of the layer's conv, tile_channels, the h' output and the gate gradient of the gate kernels) write bf16
                # through their out_bf16 / h_bf16 / dgates_bf16 flags.  SAVP_BF16_ACT=0 / SAVP_BF16_DGATES=0 keep fp32 tensors.
                a_dt = torch.bfloat16 if (L['fused'] and os.environ.get('SAVP_BF16_ACT', '1') == '1') else torch.float32
                L['a'] = Act((T1, N, h_, w_, f + zr + f), dev, grad=g, dtype=a_dt)
                dg_dt = torch.bfloat16 if (L['fused'] and g and os.environ.get('SAVP_BF16_DGATES', '1') == '1') else torch.float32
                L['gates'] = Act((T1, N, h_, w_, 4 * f), dev, grad=g, dtype=torch.bfloat16 if L['fused'] else torch.float32,
                                 grad_dtype=dg_dt)
                L['dg_raw'] = torch.empty(N, h_, w_, 4 * f, device=dev) if dg_dt == torch.bfloat16 else None
                L['c'] = Act((T1, N, h_, w_, f), dev, grad=False)
                L['dc'] = [torch.empty(N, h_, w_, f, device=dev), torch.empty(N, h_, w_, f, device=dev)] if g else None
                L['rconv'] = ConvLayer(store, r + 'kernel', (r + 'bias') if self.cell_plain else None, 'conv', (5, 5), (1, 1), (2, 2))
                if not self.cell_plain:
                    # the gate convolution's own kernel (bf16 datapath, csrc/conv_gate.hip); where a tile holds whole images (planes of <= 256
                    # pixels) also the interleaved pack: savp_convlstm_cell_fwd then runs the whole cell forward as one launch
                    L['rconv'].enable_gate_pack(cell=bool(L['fused']) and h_ * w_ <= 256)
                if not self.cell_plain:
                    L['n1'] = Norm(store, r + 'input_transform_forget_output/', T1, N, 4 * f, dev)
                    L['n2'] = Norm(store, r + 'state/', T1, N, f, dev)
                if self.out_norm:
                    L['h_raw'] = Act((T1, N, h_, w_, f), dev, grad=g)
                    L['onorm'] = Norm(store, prefix + 'lstm_h%d/InstanceNorm/' % i, T1, N, f, dev)
                # The data gradient of the gate convolution leaves the tiled-z channels of [x | z | h] out (their gradient is a per-sample
                # sum, taken once over all timesteps from region sums of the gate gradient: csrc/tiled_z.hip), which keeps its column count
                # on a tile boundary (72 / 136 / 264 -> 64 / 128 / 256; KTH's nz = 32: 96 / 160 / 288 -> 64 / 128 / 256).  bf16 datapath (the ring kernel owns the column gap).
                L['zless'] = bool(g and zr and not cw and L['fused'] and os.environ.get('SAVP_ZLESS_DGRAD', '1') == '1' and
                                  K.tiled_z_ok(h_, w_, 4 * f, zr, L['rconv'].geom))
                if L['zless']:
                    L['weff'] = torch.empty(25, 4 * f, K.tiled_z_pad(zr), device=dev)
            else:
                L['out'] = None
            self.layers.append(L)
            prev_f = f
        nl = len(self.layers)
        last = self.layers[-1]
        ngf = hp.ngf
        self.M = M = num_masks(hp)
        self.nk = nk = hp.last_frames * hp.num_transformed_images
        kh, kw = hp.kernel_size
        self.kh, self.kw = kh, kw

        # ---- heads ------------------------------------------------------------------------------------------
        self.tf = hp.transformation
        self.h_last = Act((T1, N, H, W, last['f']), dev, grad=g, dtype=a16 if last['f'] % 8 == 0 else torch.float32)
        tf_convs = []
        if self.tf == 'cdna':
            sh_, sw_ = H // 2 ** self.ne, W // 2 ** self.ne
            small_f = self.layers[self.ne - 1]['f']
            self.hsmall = Act((T1, N, sh_, sw_, small_f), dev, grad=g)
            self.cdna_dense = ConvLayer(store, prefix + 'cdna_kernels/dense/kernel', prefix + 'cdna_kernels/dense/bias', 'conv',
                                        (1, 1), (1, 1), (0, 0))
            self.cdna_raw = Act((T1, N, kh * kw * nk), dev, grad=g)
            self.cdna_kern = Act((T1, N, kh * kw, nk), dev, grad=False)
            # gradient of the normalised CDNA kernels of ONE timestep, float64: the image tiles' partial sums meet in it through float64
            # atomics (exact, order-independent: include/savp_hip.h SavpCdnaArgs.dkern); consumed by cdna_kernels_bwd right away
            self.cdna_dkern = torch.zeros(N, kh * kw, nk, device=dev, dtype=torch.float64) if g else None
            tf_convs = [self.cdna_dense]
        else:
            # flow: h_flow = relu(IN(conv3x3)); flows = conv3x3 -> 2*nk   (savp_model.py:522-530)
            # dna : h_dna_kernel = relu(IN(conv3x3)); kernels = conv3x3 -> kh*kw*nk   (:534-544)
            hs, os_, cy = (('h%d_flow/' % nl, 'flows/', 2 * nk) if self.tf == 'flow' else
                           ('h%d_dna_kernel/' % nl, 'dna_kernels/', kh * kw * nk))
            s = prefix + hs
            self.tf_conv = ConvLayer(store, s + 'conv2d/kernel', s + 'conv2d/bias', 'conv', (3, 3), (1, 1), (1, 1))
            self.tf_pre = Act((T1, N, H, W, ngf), dev, grad=g)
            self.tf_norm = Norm(store, s + 'InstanceNorm/', T1, N, ngf, dev)
            self.tf_h = Act((T1, N, H, W, ngf), dev, grad=g)
            self.tf_out = ConvLayer(store, prefix + os_ + 'conv2d/kernel', prefix + os_ + 'conv2d/bias', 'conv', (3, 3), (1, 1),
                                    (1, 1), cy_pad=ceil4(cy))
            self.tf_cy = ceil4(cy)
            self.tf_raw = Act((T1, N, H, W, self.tf_cy), dev, grad=g)
            if self.tf == 'dna':
                if self.tf_cy != cy:
                    raise NotImplementedError('dna with kh*kw*nk not a multiple of 4')
                self.dna_kern = torch.empty(T1, N, H, W, cy, device=dev)
            tf_convs = [self.tf_conv, self.tf_out]
        self.merge_heads = os.environ.get('SAVP_MERGE_HEADS', '1') == '1' and ngf % 4 == 0
        sep = not self.merge_heads          # separate pre-activation buffers only when every head has its own launch
        self.scratch = bool(hp.generate_scratch_image)          # savp_model.py:561-572; without it the head and its mask slot do not exist
        self.dep_mask = bool(hp.dependent_mask)                  # :631-632: the mask conv also reads the transformed images
        self.Cs = Cs = ceil4(C)                       # scratch conv writes a 4-aligned channel group into its maskin slot
        if self.scratch:
            s = prefix + 'h%d_scratch/' % nl
            self.scratch_conv = ConvLayer(store, s + 'conv2d/kernel', s + 'conv2d/bias', 'conv', (3, 3), (1, 1), (1, 1))
            self.scratch_pre = Act((T1, N, H, W, ngf), dev, grad=g) if sep else None
            self.scratch_norm = Norm(store, s + 'InstanceNorm/', T1, N, ngf, dev)
            self.scratch_h = Act((T1, N, H, W, ngf), dev, grad=g)
            s = prefix + 'scratch_image/'
            self.scratch_out = ConvLayer(store, s + 'conv2d/kernel', s + 'conv2d/bias', 'conv', (3, 3), (1, 1), (1, 1), cy_pad=Cs)
            self.dscratch_pre = torch.empty(T1, N, H, W, Cs, device=dev) if g else None
        s = prefix + 'h%d_masks/' % nl
        self.masks_conv = ConvLayer(store, s + 'conv2d/kernel', s + 'conv2d/bias', 'conv', (3, 3), (1, 1), (1, 1))
        self.masks_pre = Act((T1, N, H, W, ngf), dev, grad=g) if sep else None
        self.masks_norm = Norm(store, s + 'InstanceNorm/', T1, N, ngf, dev)
        # background images in the reference's order (savp_model.py:581-594): the step's input image, then frames of the INPUT video --
        # ('fixed', k) = images[k] at every step, ('last_context',) = images[min(t, context_frames - 1)]
        cf = hp.context_frames
        self.bgs = [('prev',)] if hp.prev_image_background else []
        if hp.context_images_background:
            self.bgs += [('fixed', k) for k in range(cf)]
        else:
            self.bgs += ([('fixed', 0)] if hp.first_image_background else []) + ([('fixed', cf - 1)] if hp.last_image_background else []) + \
                        ([('last_context',)] if hp.last_context_image_background else [])
        nb = len(self.bgs)
        assert M == nk + nb + int(self.scratch)
        if M < 2:
            raise NotImplementedError('a single transformed image (mask == 1 everywhere, savp_model.py:636-637)')
        # maskin = [h_masks (ngf) | nk transformed images | background images | scratch image]   (savp_model.py:632)
        self.Cmask = Cmask = ceil8(ngf + M * C + ((Cs - C) if self.scratch else 0))     # scratch slot is last: room for its padded write
        self.Ml = Ml = ceil4(M)                                    # padded logits row
        self.maskin = Act((T1, N, H, W, Cmask), dev, grad=g)
        self.o_cdna = ngf
        self.o_bg = [ngf + (nk + i) * C for i in range(nb)]
        self.o_prev = self.o_bg[self.bgs.index(('prev',))] if ('prev',) in self.bgs else None
        self.o_scratch = ngf + (nk + nb) * C
        s = prefix + 'masks/'
        self.mask_cin = Cmask if self.dep_mask else ngf            # channels of maskin the mask conv reads
        self.masks_out = ConvLayer(store, s + 'conv2d/kernel', s + 'conv2d/bias', 'conv', (3, 3), (1, 1), (1, 1),
                                   cx_pad=self.mask_cin, cy_pad=Ml)
        self.logits = Act((T1, N, H, W, Ml), dev, grad=g)
        self.masks = torch.empty(T1, N, H, W, M, device=dev)
        self.gen = Act((T1, N, H, W, C), dev, grad=g, zero_grad=True)
        self.dimg_cdna = torch.empty(N, H, W, C, device=dev) if g else None

        # ---- actions / states (savp_model.py:411-422,655-658) -----------------------------------------------------
        # saz [T1, N, zw] = what every slice tiles: [actions_t | state_t | rnn_z_t]; the state recurrence (state_t = ground truth or the
        # previous step's prediction, gen_state_t = dense([actions_t | state_t])) involves no image, so one launch runs all steps
        # (csrc/state_pred.hip) before the unroll.  The tiled copy is under stop_gradient (:421-422): only the state loss reaches state_pred.
        self.saz = torch.zeros(T1, N, zw, device=dev) if cw else None
        if ns:
            s_ = prefix + 'state_pred/dense/'
            self.spW, self.spb = store[s_ + 'kernel'], store[s_ + 'bias']
            self.dspW, self.dspb = (store.grad64(s_ + 'kernel'), store.grad64(s_ + 'bias')) if g else (None, None)
            self.sa = torch.zeros(T1, N, cw, device=dev)
            self.gen_states = Act((T1, N, ns), dev, grad=g, zero_grad=True)
        # ---- z path -------------------------------------------------------------------------------------------
        if nz:
            self.zs = Act((T1, N, nz), dev, grad=g)
            self.rnn_z = Act((T1, N, nz), dev, grad=g, zero_grad=True)
            if self.use_rnn_z and self.abl_rnn:
                self.fc_z = ConvLayer(store, prefix + 'fc_z/dense/kernel', prefix + 'fc_z/dense/bias', 'conv', (1, 1), (1, 1), (0, 0))
                self.fcz_pre = Act((T1 * N, 1, 1, nz), dev, grad=g)
            elif self.use_rnn_z and hp.rnn == 'gru':            # tf.contrib.rnn.GRUCell under scope gru_z (savp_model.py:358-359,426)
                z = prefix + 'gru_z/gru_cell/'
                self.zg = ConvLayer(store, z + 'gates/kernel', z + 'gates/bias', 'conv', (1, 1), (1, 1), (0, 0))
                self.zc = ConvLayer(store, z + 'candidate/kernel', z + 'candidate/bias', 'conv', (1, 1), (1, 1), (0, 0))
                for c_ in (self.zg, self.zc):
                    c_.need_wt = c_.need_wd = False
                self.zA = torch.zeros(T1, N, 2 * nz, device=dev)
                self.zA2 = torch.zeros(T1, N, 2 * nz, device=dev)
                self.z_ru = torch.empty(T1, N, 2 * nz, device=dev)
                self.z_cand = torch.empty(T1, N, nz, device=dev)
                if g:
                    self.z_dGg = torch.empty(T1, N, 2 * nz, device=dev)
                    self.z_dGc = torch.empty(T1, N, nz, device=dev)
                    self.z_dA = torch.empty(T1, N, 2 * nz, device=dev)
            elif self.use_rnn_z:
                z = prefix + 'lstm_z/basic_lstm_cell/'
                self.zW, self.zb = store[z + 'kernel'], store[z + 'bias']
                self.dzW, self.dzb = store.grad64(z + 'kernel'), store.grad64(z + 'bias')      # float64 accumulators (one workgroup per sample adds to them)
                self.z_gates = torch.empty(T1, N, 4 * nz, device=dev)
                self.z_cs = torch.empty(T1, N, nz, device=dev)
        self.gru = hp.conv_rnn == 'gru'
        # ---- learn_initial_state (savp_model.py:295-307,344-352): the conv-RNN states and the rnn_z state start from variables
        # `generator/initial_state_<i>/initial_state` (i = position in nest.flatten of {'conv_rnn_states': [...], 'rnn_z_state': ...}: layer
        # order, LSTM tuples as (c, h), the latent cell last), tiled over the batch; both unrolls (N = 2B) share them.  Forward: a broadcast
        # copy into step 0's state slots; backward: what step 0 hands back, summed over the batch.
        self.learn_init = bool(hp.learn_initial_state) and not self.abl_rnn
        if self.learn_init:
            k = 0

            def var(shape, acc64=False):
                nonlocal k
                name = 'generator/initial_state_%d/initial_state' % k
                k += 1
                assert tuple(store[name].shape) == tuple(shape), (name, tuple(store[name].shape), shape)
                return store[name], ((store.grad64(name) if acc64 else store.grad(name)) if g else None)
            for L in self.layers:
                if not L['rnn']:
                    continue
                hw_f = L['hw'] + (L['f'],)
                if not self.gru:
                    L['c0v'], L['c0g'] = var(hw_f)
                    L['c0'] = torch.empty((N,) + hw_f, device=dev)
                L['h0v'], L['h0g'] = var(hw_f)
            if self.use_rnn_z and hp.rnn == 'gru':               # GRUCell: the state is h alone (savp_model.py:288-291)
                self.z_h0, self.z_dh0 = var((nz,), acc64=True)
            elif self.use_rnn_z:
                self.z_c0, self.z_dc0 = var((nz,), acc64=True)    # savp_lstm_z_bwd_init adds to them from every sample's workgroup
                self.z_h0, self.z_dh0 = var((nz,), acc64=True)
        # ---- merged 3x3 heads on the last decoder layer (SAVP_MERGE_HEADS=0: one launch per head, the reference's structure) ----
        # h6_scratch, h6_masks (and h6_flow / h6_dna_kernel) all read h_last through a 3x3 conv + instance norm + relu: ONE conv with
        # concatenated output channels, ONE instance norm over them, outputs routed by channel range; backward likewise.
        head_convs = ([self.scratch_conv] if self.scratch else []) + [self.masks_conv]
        if self.merge_heads:
            parts = ([(self.scratch_conv.kernel_name, self.scratch_conv.bias_name)] if self.scratch else []) + \
                    [(self.masks_conv.kernel_name, self.masks_conv.bias_name)]
            norms = ([self.scratch_norm] if self.scratch else []) + [self.masks_norm]
            if self.tf != 'cdna':
                parts.append((self.tf_conv.kernel_name, self.tf_conv.bias_name))
                norms.append(self.tf_norm)
                tf_convs = [self.tf_out]
            self.nheads = len(parts)
            self.heads_conv = ConcatConv(store, parts, (3, 3), (1, 1), (1, 1))
            self.heads_norm = ConcatNorm(norms, T1, N, dev)
            self.heads_pre = Act((T1, N, H, W, self.nheads * ngf), dev, grad=g, grad_dtype=a16 if (self.nheads * ngf) % 8 == 0 else torch.float32)
            head_convs = [self.heads_conv]
        self.convs = [L['conv'] for L in self.layers] + [L['rconv'] for L in self.layers if L['rnn']] + \
                     [L['cconv'] for L in self.layers if L['rnn'] and self.gru and not self.abl_rnn] + \
                     ([self.fc_z] if (self.use_rnn_z and self.abl_rnn) else []) + \
                     tf_convs + head_convs + ([self.scratch_out] if self.scratch else []) + [self.masks_out]
        # only FPROP packs needed at inference
        self._routes()

    # ---------------------------------------------------------------------------------------------------------
    def _routes(self):
        """Where the output of layer i (layers[i][-1] in the reference) is written: list of (Act, channel offset)."""
        ne, nd = self.ne, self.nd
        for i, L in enumerate(self.layers):
            r = []
            if i + 1 < len(self.layers):
                r.append((self.layers[i + 1]['in'], 0))
            else:
                r.append((self.h_last, 0))
            if i < ne:
                j = ne - 1 - i              # decoder layer that takes this encoder layer as skip (j > 0)
                if 1 <= j < nd:
                    r.append((self.layers[ne + j]['in'], self.layers[ne + j]['skip_off']))
                if i == ne - 1 and self.tf == 'cdna':
                    r.append((self.hsmall, 0))
            L['routes'] = r

    prefix_root = 'generator/rnn/'       # every variable of the generator cell lives under this scope (one gradient chunk)

    def weight_layers(self):
        return self.convs

    def prep_weights(self):
        prep_layers(self.convs)
        if self.merge_heads:
            self.heads_norm.prep()

    # ---------------------------------------------------------------------------------------------------------
    def _out_views(self, L, t):
        f = L['f']
        return [buf.v[t][..., off:off + f] for buf, off in L['routes']]

    def _out_grads(self, L, t):
        f = L['f']
        return [buf.g[t][..., off:off + f] for buf, off in L['routes']]

    def forward(self, images, zs=None, gt_mask=None, collect_masks=False, actions=None, states=None):
        """images [T>=T1, N, H, W, C] device fp32 (frame t feeds step t); zs [T1, N, nz]; gt_mask int32 [T1, N]
        (1 = take the ground-truth frame: self.ground_truth of savp_model.py:333-334); actions [T1, N, na] / states [>=T1, N, ns] when
        the generator was built with cond.  Fills self.gen.v (and self.gen_states.v)."""
        T1, N, C = self.T1, self.N, self.C
        nz = self.nz
        self.images = images
        self.gt_mask = gt_mask
        cw, na = self.cw, self.na
        if cw:
            if (na and actions is None) or (self.ns and states is None):
                raise ValueError('this generator was built for inputs with actions / states (cond=%r)' % ((na, self.ns),))
            if self.ns:
                K.state_pred_fwd(actions[:T1].contiguous() if na else None, states[:T1].contiguous(), gt_mask, self.spW, self.spb, self.sa,
                                 self.gen_states.v)
                self.saz[..., :cw].copy_(self.sa)
            else:
                self.saz[..., :na].copy_(actions[:T1])
        if nz:
            self.zs.v.copy_(zs)
            if self.use_rnn_z and self.abl_rnn:           # tanh(dense(z)) (savp_model.py:426-429)
                self.fc_z.forward(self.zs.v.reshape(T1 * N, 1, 1, nz), self.fcz_pre.v)
                torch.tanh(self.fcz_pre.v.reshape(T1, N, nz), out=self.rnn_z.v)
            elif self.use_rnn_z and self.hp.rnn == 'gru':
                self.zA[..., :nz].copy_(self.zs.v)
                K.gru_seq_fwd(self.zA, self.zA2, self.zg.W, self.zg.bias, self.zc.W, self.zc.bias, self.rnn_z.v, self.z_ru, self.z_cand, nz,
                              h0=self.z_h0 if self.learn_init else None)
            elif self.use_rnn_z:
                K.lstm_z_fwd(self.zs.v, self.zW, self.zb, self.rnn_z.v, self.z_gates, self.z_cs,
                             init=(self.z_c0, self.z_h0) if self.learn_init else None)
            else:
                self.rnn_z.v.copy_(self.zs.v)
            if cw:
                self.saz[..., cw:].copy_(self.rnn_z.v)
        if self.zw:
            zw = self.zw
            zflat = (self.saz if cw else self.rnn_z.v).reshape(T1 * N, zw)
            for L in self.layers:
                b = L['in']
                if L['zc']:
                    K.tile_channels(zflat, b.flat(b.v)[..., L['zoff_in']:L['zoff_in'] + zw])
                if L['rnn'] and L['zr']:
                    a = L['a']
                    K.tile_channels(zflat, a.flat(a.v)[..., L['f']:L['f'] + zw])
        if self.learn_init:                # step 0's state slots <- the learned initial states, tiled over the batch
            for L in self.layers:
                if L['rnn']:
                    f, hoff = L['f'], L['f'] + L['zr']
                    L['a'].v[0][..., hoff:hoff + f].copy_(L['h0v'])
                    if not self.gru:
                        L['c0'].copy_(L['c0v'])
        in0, maskin = self.layers[0]['in'], self.maskin
        # the first frame feeds every step (savp_model.py:399-400): ONE launch writes it into all T1 steps' buffers -- time is the
        # kernel's sample index (source stride 0), the N*H*W pixels of a step its pixel index
        if self._ones is None:
            self._ones = torch.ones(max(N, T1), dtype=torch.int32, device=self.dev)
        NH = N * self.H
        mflat = maskin.v.reshape(T1, NH, self.W, -1)

        def frame_all_steps(k, t0=0):          # images[k] seen from steps t0 .. T1-1 (source stride 0 over time)
            return images[k].reshape(1, NH, self.W, C).expand(T1 - t0, NH, self.W, C)
        first_dsts = [in0.v.reshape(T1, NH, self.W, -1)[..., C:2 * C]]
        for bg, off in zip(self.bgs, self.o_bg):
            if bg == ('fixed', 0):
                first_dsts.append(mflat[..., off:off + C])
        K.select(self._ones, frame_all_steps(0), None, first_dsts)
        cf = self.hp.context_frames
        for bg, off in zip(self.bgs, self.o_bg):
            if bg[0] == 'fixed' and bg[1] != 0:
                K.select(self._ones, frame_all_steps(bg[1]), None, [mflat[..., off:off + C]])
            elif bg[0] == 'last_context':          # images[t] while t < context_frames, images[context_frames - 1] afterwards
                n = min(cf, T1)
                copy_view(images[:n].reshape(n, NH, self.W, C), [mflat[:n][..., off:off + C]])
                if T1 > n:
                    K.select(self._ones, frame_all_steps(cf - 1, n), None, [mflat[n:][..., off:off + C]])
        fuse_select = os.environ.get('SAVP_FUSE_SELECT', '1') == '1'

        def image_slots(t):         # where step t's input image goes: the first conv's input and the 'prev' background slot
            return [in0.v[t][..., 0:C]] + ([maskin.v[t][..., self.o_prev:self.o_prev + C]] if self.o_prev is not None else [])
        for t in range(T1):
            # image = tf.where(ground_truth[t], inputs['images'], states['gen_image'])     (savp_model.py:406); from step 1 on the
            # compositing kernel of step t-1 has already written it (composite_fwd(next_inputs=...))
            if t == 0 or not fuse_select:
                prev_gen = self.gen.v[t - 1] if t > 0 else None
                K.select(gt_mask[t], images[t], prev_gen, image_slots(t))
            for L in self.layers:
                f = L['f']
                # the conv's epilogue leaves the instance norm's statistics behind where it can (bf16 datapath, whole tiles): the
                # norm is then ONE launch (SAVP_CONV_STATS=0: the norm takes its own statistics)
                ck = ('cstats', K.PRECISION['value'])          # the answer depends on the datapath in use: decided once per precision
                if ck not in L:
                    L[ck] = (CONV_STATS and f <= 256 and (f & (f - 1)) == 0 and L['conv'].stats_ok(L['in'].v[t], L['pre'].v[t]))
                st = K.stats_ws(self.dev, N, f) if L[ck] else None
                nrm = L['norm']
                if L['rnn'] and self.abl_rnn:
                    a = L['a']
                    self._conv_in_act(L['conv'], L['in'].v[t], L['pre'].v[t], st, nrm, [a.v[t][..., 0:f]], t)
                    self._conv_norm(('abl', L['idx']), L['rconv'], a.v[t], L['pre2'].v[t], L['n2'], self._out_views(L, t), t)
                elif L['rnn'] and self.gru:
                    L['conv'].forward(L['in'].v[t], L['pre'].v[t], stats=st)
                    a = L['a']
                    hs, rs_, cin1 = f + L['zr'], f + L['zr'] + f, L['cin1']
                    K.instnorm_act_fwd(L['pre'].v[t], nrm.gamma, nrm.beta, [a.v[t][..., 0:f]], nrm.mean[t], nrm.rstd[t],
                                       act='relu', eps=EPS_IN, stats=st, stats_shift=L['conv'].bias if st is not None else None)
                    n1, n2 = L['n1'], L['n2']
                    hprev = a.v[t][..., hs:hs + f]
                    L['rconv'].forward(a.v[t][..., 0:cin1], L['gates'].v[t], use_bias=False)
                    K.convgru_gates_fwd(L['gates'].v[t], hprev, n1.gamma, n1.beta, n1.mean[t], n1.rstd[t], L['u'][t],
                                        a.v[t][..., rs_:rs_ + f], eps=EPS_IN)
                    L['cconv'].forward(a.v[t], L['cand'].v[t], use_bias=False)
                    outs = self._out_views(L, t)
                    if t + 1 < T1:
                        outs.append(a.v[t + 1][..., hs:hs + f])
                    K.convgru_out_fwd(L['cand'].v[t], hprev, n2.gamma, n2.beta, n2.mean[t], n2.rstd[t], L['u'][t], outs, eps=EPS_IN)
                elif L['rnn'] and self.cell_plain:
                    a = L['a']
                    self._conv_in_act(L['conv'], L['in'].v[t], L['pre'].v[t], st, nrm, [a.v[t][..., 0:f]], t)
                    L['rconv'].forward(a.v[t], L['gates'].v[t])                       # conv + bias
                    nxt_h = [a.v[t + 1][..., f + L['zr']:f + L['zr'] + f]] if t + 1 < T1 else []
                    hdst = ([L['h_raw'].v[t]] if self.out_norm else self._out_views(L, t)) + nxt_h
                    K.convlstm_gates_fwd(L['gates'].v[t], L['c'].v[t - 1] if t > 0 else L.get('c0'), None, None, None, None, L['c'].v[t], hdst, None)
                    if self.out_norm:
                        on = L['onorm']
                        K.instnorm_act_fwd(L['h_raw'].v[t], on.gamma, on.beta, self._out_views(L, t), on.mean[t], on.rstd[t], act='none', eps=EPS_IN)
                elif L['rnn']:
                    a = L['a']
                    self._conv_in_act(L['conv'], L['in'].v[t], L['pre'].v[t], st, nrm, [a.v[t][..., 0:f]], t)
                    stats1 = s1 = None
                    if L['fused']:
                        stats1, s1 = K.lstm_stats_ws(self.dev, N, f)
                    cp = L.get('cell_prof')                  # bench.py: HIP events around the whole cell (gate conv + gate passes)
                    if cp is not None:
                        ce0, ce1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        ce0.record()
                    outs = self._out_views(L, t)
                    if t + 1 < T1:
                        outs.append(a.v[t + 1][..., f + L['zr']:f + L['zr'] + f])
                    n1, n2 = L['n1'], L['n2']
                    gk = L.get('gate_ktimer')                # bench.py: the gate-block launch's own begin / end stamps
                    gargs = (L['gates'].v[t], L['c'].v[t - 1] if t > 0 else L.get('c0'), n1.gamma, n1.beta, n2.gamma, n2.beta, L['c'].v[t], outs,
                             [n1.mean[t], n1.rstd[t], n2.mean[t], n2.rstd[t]])
                    # the whole cell as ONE host call (savp_convlstm_cell_fwd) unless a measurement wants the two launches apart
                    ca = (L['rconv'].forward(a.v[t], L['gates'].v[t], use_bias=False, stats=s1, defer=True)
                          if (FUSED_ENTRIES and cp is None and gk is None and K.fused_ok()) else None)
                    if ca is not None:
                        ck = L.get('cell_ktimer')            # bench.py: the one-launch cell's own begin / end stamps (the kernel takes the armed pair)
                        if ck is not None:
                            ck.arm()
                        K.convlstm_cell_fwd(ca, K.convlstm_gates_fwd(*gargs, eps=EPS_IN, ws=self._lstm_ws(L), stats1=stats1, defer=True))
                        if ck is not None:
                            ck.taken()
                    else:
                        L['rconv'].forward(a.v[t], L['gates'].v[t], use_bias=False, stats=s1)
                        if gk is not None:
                            gk.arm()
                        K.convlstm_gates_fwd(*gargs, eps=EPS_IN, ws=self._lstm_ws(L), stats1=stats1)
                        if gk is not None:
                            gk.taken()
                    if cp is not None:
                        ce1.record()
                        cp.append((ce0, ce1))
                else:
                    self._conv_in_act(L['conv'], L['in'].v[t], L['pre'].v[t], st, nrm, self._out_views(L, t), t)
            tslot = maskin.v[t][..., self.o_cdna:self.o_cdna + self.nk * C]
            ngf = self.hp.ngf
            if self.merge_heads:
                # one conv + one instance norm for every 3x3 head on h_last; outputs routed by channel range
                hn = self.heads_norm
                outs = ([self.scratch_h.v[t]] if self.scratch else []) + [maskin.v[t][..., 0:ngf]] + \
                       ([self.tf_h.v[t]] if self.tf != 'cdna' else [])
                self._conv_norm('heads', self.heads_conv, self.h_last.v[t], self.heads_pre.v[t], hn, outs, t,
                                out_ranges=[(i * ngf, ngf) for i in range(self.nheads)])
            if self.tf == 'cdna':
                # CDNA kernels from the smallest layer (savp_model.py:546-559) and their application (:580, :893-923)
                self.cdna_dense.forward(self.hsmall.v[t].reshape(N, -1), self.cdna_raw.v[t])
                K.cdna_kernels_fwd(self.cdna_raw.v[t], self.cdna_kern.v[t], self.kh, self.kw, self.nk)
                K.cdna_apply_fwd(in0.v[t][..., 0:C], self.cdna_kern.v[t], tslot, self.kh, self.kw, self.nk)
            else:
                if not self.merge_heads:
                    self._conv_norm('tf', self.tf_conv, self.h_last.v[t], self.tf_pre.v[t], self.tf_norm, [self.tf_h.v[t]], t)
                self.tf_out.forward(self.tf_h.v[t], self.tf_raw.v[t])
                if self.tf == 'flow':
                    K.image_warp_fwd(in0.v[t][..., 0:C], self.tf_raw.v[t], tslot, self.nk)            # apply_flows :955-965
                else:
                    K.dna_apply_fwd(in0.v[t][..., 0:C], self.tf_raw.v[t], self.dna_kern[t], tslot, self.kh, self.kw, self.nk)
            # scratch image (savp_model.py:561-572): sigmoid fused into the conv epilogue, written into its mask-conv slot
            if self.scratch and not self.merge_heads:
                self._conv_norm('scratch', self.scratch_conv, self.h_last.v[t], self.scratch_pre.v[t], self.scratch_norm,
                                [self.scratch_h.v[t]], t)
            if self.scratch:
                self.scratch_out.forward(self.scratch_h.v[t], maskin.v[t][..., self.o_scratch:self.o_scratch + self.Cs],
                                         act=lib.ACT_SIGMOID)
            # masks (savp_model.py:623-646)
            if not self.merge_heads:
                self._conv_norm('masks', self.masks_conv, self.h_last.v[t], self.masks_pre.v[t], self.masks_norm,
                                [maskin.v[t][..., 0:self.hp.ngf]], t)
            self.masks_out.forward(maskin.v[t][..., 0:self.mask_cin], self.logits.v[t])
            nxt = (gt_mask[t + 1], images[t + 1], image_slots(t + 1)) if (fuse_select and t + 1 < T1) else None
            K.composite_fwd(self.logits.v[t], maskin.v[t][..., self.hp.ngf:self.hp.ngf + self.M * C], self.gen.v[t],
                            self.masks[t] if collect_masks else None, M=self.M, next_inputs=nxt)
        return self.gen.v

    # ---------------------------------------------------------------------------------------------------------
    def _conv_norm(self, name, conv, x, pre, nrm, outs, t, **kw):
        """conv -> instance norm + ReLU of a head; the conv's epilogue supplies the norm's statistics where it can (see forward)."""
        ok = self._cstats.get((name, K.PRECISION['value']))
        if ok is None:
            c = pre.shape[-1]
            ok = self._cstats[(name, K.PRECISION['value'])] = bool(CONV_STATS and c <= 256 and (c & (c - 1)) == 0 and conv.stats_ok(x, pre))
        st = K.stats_ws(self.dev, self.N, pre.shape[-1]) if ok else None
        self._conv_in_act(conv, x, pre, st, nrm, outs, t, **kw)

    def _conv_in_act(self, conv, x, pre, st, nrm, outs, t, **kw):
        """conv -> fused_instance_norm + ReLU as ONE host call (savp_conv_in_act_fwd) where nothing instruments the conv; st = the
        statistics slice the conv's epilogue fills for the norm (None: the norm takes its own)."""
        bias = getattr(conv, 'inner', conv).bias          # ConcatConv keeps the concatenated bias in its inner layer
        nkw = dict(act='relu', eps=EPS_IN, stats=st, stats_shift=bias if st is not None else None, **kw)
        ca = conv.forward(x, pre, stats=st, defer=True) if (FUSED_ENTRIES and K.fused_ok()) else None
        if ca is not None:
            K.conv_in_act_fwd(ca, K.instnorm_act_fwd(pre, nrm.gamma, nrm.beta, outs, nrm.mean[t], nrm.rstd[t], defer=True, **nkw))
        else:
            conv.forward(x, pre, stats=st)
            K.instnorm_act_fwd(pre, nrm.gamma, nrm.beta, outs, nrm.mean[t], nrm.rstd[t], **nkw)

    def _norm_bwd(self, key, holder, conv, dy, dx, nrm, x, skip=None):
        """The instance norm (over x, parameters nrm) whose OUTPUT gradient is the channels [0, C) that conv.backward_data(dy, dx) is
        about to write: where the ring kernel can, its epilogue leaves that norm's backward sums behind (SavpConvArgs.nb_*), and the
        norm's backward is then ONE launch.  Returns the norm_bwd dict for backward_data (None: the norm takes its own sums); its 'ws'
        is what instnorm_act_bwd(stats=...) gets.  The decision is made once per layer (holder[key])."""
        t_ = dict(x=x, mean=nrm.mean[0], rstd=nrm.rstd[0], gamma=nrm.gamma, beta=nrm.beta, c0=0, act='relu')
        key = (key, K.PRECISION['value'])
        ok = holder.get(key)
        if ok is None:
            # savp_instnorm_act_bwd(stats_ready) runs the coalesced apply pass alone: C % 4 == 0, C <= 256 and a whole number of pixel rows
            # per 256-thread workgroup (256 % (C / 4) == 0) -- the forward twin's guard (non-power-of-two ngf would otherwise raise in every backward)
            c = x.shape[-1]
            ok = holder[key] = bool(NORM_BWD_STATS and dx.dtype == torch.float32 and c <= 256 and c % 4 == 0 and 256 % (c // 4) == 0 and
                                    conv.norm_bwd_ok(dy, dx, t_, skip))
        if not ok:
            return None
        t_['ws'] = K.stats_ws(self.dev, self.N, x.shape[-1])
        return t_

    def _in_act_conv_bwd(self, L, t, y0, dys, st):
        """Backward of a ladder layer's instance norm + ReLU and of its conv_pool / upsample convolution's data path: one host call
        (savp_conv_in_act_bwd).  st: the norm-backward sums a data gradient's epilogue has already left (None: the norm takes them)."""
        nrm = L['norm']
        nargs = (L['pre'].v[t], nrm.gamma, nrm.beta, y0, nrm.mean[t], nrm.rstd[t], dys, L['pre'].g[t], nrm.dgamma, nrm.dbeta)
        if FUSED_ENTRIES and K.fused_ok():
            K.conv_in_act_bwd(L['conv'].backward_data(L['pre'].g[t], L['in'].g[t], beta=0, defer=True),
                              K.instnorm_act_bwd(*nargs, act='relu', eps=EPS_IN, stats=st, defer=True))
        else:
            K.instnorm_act_bwd(*nargs, act='relu', eps=EPS_IN, stats=st)
            L['conv'].backward_data(L['pre'].g[t], L['in'].g[t], beta=0)

    def _lstm_ws(self, L):
        """Scratch of the coalesced ConvLSTM gate kernels (one buffer shared by all layers: the launches are serial)."""
        h, w = L['hw']
        need = K.lstm_ws_floats(self.N, h * w, L['f'])
        ws = getattr(self, '_lstm_ws_buf', None)
        if ws is None or ws.numel() < need:
            ws = self._lstm_ws_buf = torch.empty(max(need, K.lstm_ws_floats(self.N, self.H * self.W // 4, 32)), device=L["gates"].v.device)
        return ws

    def backward(self, state_grad=False):
        """BPTT.  Expects self.gen.g (zero-initialised each step by the caller) to hold dL/dgen_images (and, with state_grad,
        self.gen_states.g to hold dL/dgen_states).
        Accumulates every generator-cell weight gradient into the store and returns dL/dzs [T1, N, nz] (or None)."""
        T1, N, C, nz = self.T1, self.N, self.C, self.nz
        ngf = self.hp.ngf
        in0, maskin = self.layers[0]['in'], self.maskin
        for t in range(T1 - 1, -1, -1):
            # composite + masks head
            K.composite_bwd(self.logits.v[t], maskin.v[t][..., ngf:ngf + self.M * C], self.gen.g[t], self.logits.g[t], maskin.g[t],
                            ngf, M=self.M)
            self.masks_out.backward_data(self.logits.g[t], maskin.g[t][..., 0:self.mask_cin], beta=1)
            if self.scratch:       # scratch head: d(sigmoid) then the scratch_image conv's data gradient
                K.sigmoid_bwd(maskin.g[t][..., self.o_scratch:self.o_scratch + self.Cs],
                              maskin.v[t][..., self.o_scratch:self.o_scratch + self.Cs], self.dscratch_pre[t])
                self.scratch_out.backward_data(self.dscratch_pre[t], self.scratch_h.g[t], beta=0)
            # pixel transformation head (everything behind its 3x3 feature conv)
            dslot = maskin.g[t][..., self.o_cdna:self.o_cdna + self.nk * C]
            if self.tf == 'cdna':
                K.cdna_apply_bwd(in0.v[t][..., 0:C], self.cdna_kern.v[t], dslot, self.dimg_cdna, self.cdna_dkern, self.kh,
                                 self.kw, self.nk)
                K.cdna_kernels_bwd(self.cdna_raw.v[t], self.cdna_dkern, self.cdna_raw.g[t], self.kh, self.kw, self.nk)
                self.cdna_dense.backward_data(self.cdna_raw.g[t], self.hsmall.g[t].reshape(N, -1), beta=0)
            else:
                if self.tf == 'flow':
                    K.image_warp_bwd(in0.v[t][..., 0:C], self.tf_raw.v[t], dslot, self.tf_raw.g[t], self.dimg_cdna, self.nk)
                    if self.tv is not None:          # total-variation loss of the flows (base_model.py:763-769): this step's share + gradient
                        w_tv, rows, acc = self.tv
                        s1 = 1.0 / (T1 * rows * (self.H - 1) * self.W)
                        s2 = 1.0 / (T1 * rows * self.H * (self.W - 1))
                        K.tv_loss(self.tf_raw.v[t][:rows], 2 * self.nk, s1, s2, w_tv, acc, self.tf_raw.g[t][:rows])
                else:
                    K.dna_apply_bwd(in0.v[t][..., 0:C], self.tf_raw.v[t], self.dna_kern[t], dslot, self.tf_raw.g[t],
                                    self.dimg_cdna, self.kh, self.kw, self.nk)
                self.tf_out.backward_data(self.tf_raw.g[t], self.tf_h.g[t], beta=0)
            # the 3x3 feature convs on h_last: instance norm + conv data gradients into h_last.g
            if self.merge_heads:
                hn = self.heads_norm
                dys = ([self.scratch_h.g[t]] if self.scratch else []) + [maskin.g[t][..., 0:ngf]] + \
                      ([self.tf_h.g[t]] if self.tf != 'cdna' else [])
                K.instnorm_act_bwd(self.heads_pre.v[t], hn.gamma, hn.beta, None, hn.mean[t], hn.rstd[t], dys, self.heads_pre.g[t],
                                   hn.dgamma, hn.dbeta, act='relu', eps=EPS_IN, dy_ranges=[(i * ngf, ngf) for i in range(self.nheads)])
                Ll = self.layers[-1]
                nb_last = None
                if not Ll['rnn']:          # h_last is the output of the last layer's instance norm: its backward sums leave with this DGRAD
                    nl_ = Ll['norm']
                    nb_last = self._norm_bwd('nb_heads', self._cstats, self.heads_conv, self.heads_pre.g[t], self.h_last.g[t], nl_,
                                             Ll['pre'].v[t])
                    if nb_last is not None:
                        nb_last['mean'], nb_last['rstd'] = nl_.mean[t], nl_.rstd[t]
                self.heads_conv.backward_data(self.heads_pre.g[t], self.h_last.g[t], beta=0, norm_bwd=nb_last)
            else:
                mn = self.masks_norm
                K.instnorm_act_bwd(self.masks_pre.v[t], mn.gamma, mn.beta, maskin.v[t][..., 0:ngf], mn.mean[t], mn.rstd[t],
                                   [maskin.g[t][..., 0:ngf]], self.masks_pre.g[t], mn.dgamma, mn.dbeta, act='relu', eps=EPS_IN)
                self.masks_conv.backward_data(self.masks_pre.g[t], self.h_last.g[t], beta=0)
                if self.scratch:
                    sn = self.scratch_norm
                    K.instnorm_act_bwd(self.scratch_pre.v[t], sn.gamma, sn.beta, self.scratch_h.v[t], sn.mean[t], sn.rstd[t],
                                       [self.scratch_h.g[t]], self.scratch_pre.g[t], sn.dgamma, sn.dbeta, act='relu', eps=EPS_IN)
                    self.scratch_conv.backward_data(self.scratch_pre.g[t], self.h_last.g[t], beta=1)
                if self.tf != 'cdna':
                    tn = self.tf_norm
                    K.instnorm_act_bwd(self.tf_pre.v[t], tn.gamma, tn.beta, self.tf_h.v[t], tn.mean[t], tn.rstd[t], [self.tf_h.g[t]],
                                       self.tf_pre.g[t], tn.dgamma, tn.dbeta, act='relu', eps=EPS_IN)
                    self.tf_conv.backward_data(self.tf_pre.g[t], self.h_last.g[t], beta=1)
            if not self.merge_heads:
                nb_last = None
            # decoder / encoder ladder in reverse
            for L in reversed(self.layers):
                f = L['f']
                dys = self._out_grads(L, t)
                nrm = L['norm']
                if L['rnn'] and self.abl_rnn:
                    a, n2 = L['a'], L['n2']
                    K.instnorm_act_bwd(L['pre2'].v[t], n2.gamma, n2.beta, self._out_views(L, t)[0], n2.mean[t], n2.rstd[t], dys, L['pre2'].g[t],
                                       n2.dgamma, n2.dbeta, act='relu', eps=EPS_IN)
                    L['rconv'].backward_data(L['pre2'].g[t], a.g[t], beta=0)
                    self._in_act_conv_bwd(L, t, a.v[t][..., 0:f], [a.g[t][..., 0:f]], None)
                    continue
                if L['rnn'] and self.gru:
                    a = L['a']
                    hs, rs_, cin1 = f + L['zr'], f + L['zr'] + f, L['cin1']
                    if t + 1 < T1:
                        dys.append(a.g[t + 1][..., hs:hs + f])
                    n1, n2 = L['n1'], L['n2']
                    hprev = a.v[t][..., hs:hs + f]
                    K.convgru_out_bwd(L['cand'].v[t], hprev, n2.gamma, n2.beta, n2.mean[t], n2.rstd[t], L['u'][t], dys,
                                      L['cand'].g[t], L['du'], L['dh_tmp'], n2.dgamma, n2.dbeta, eps=EPS_IN)
                    L['cconv'].backward_data(L['cand'].g[t], a.g[t], beta=0)
                    add_views([L['dh_tmp']], a.g[t][..., hs:hs + f])
                    K.convgru_gates_bwd(L['gates'].v[t], hprev, n1.gamma, n1.beta, n1.mean[t], n1.rstd[t], L['du'],
                                        a.g[t][..., rs_:rs_ + f], L['gates'].g[t], a.g[t][..., hs:hs + f], n1.dgamma, n1.dbeta,
                                        eps=EPS_IN)
                    L['rconv'].backward_data(L['gates'].g[t], a.g[t][..., 0:cin1], beta=1)
                    K.instnorm_act_bwd(L['pre'].v[t], nrm.gamma, nrm.beta, a.v[t][..., 0:f], nrm.mean[t], nrm.rstd[t],
                                       [a.g[t][..., 0:f]], L['pre'].g[t], nrm.dgamma, nrm.dbeta, act='relu', eps=EPS_IN)
                elif L['rnn'] and self.cell_plain:
                    a = L['a']
                    if self.out_norm:
                        on = L['onorm']
                        K.instnorm_act_bwd(L['h_raw'].v[t], on.gamma, on.beta, self._out_views(L, t)[0], on.mean[t], on.rstd[t], dys,
                                           L['h_raw'].g[t], on.dgamma, on.dbeta, act='none', eps=EPS_IN)
                        dys = [L['h_raw'].g[t]]
                    if t + 1 < T1:
                        dys.append(a.g[t + 1][..., f + L['zr']:f + L['zr'] + f])
                    dc_new = L['dc'][(t + 1) & 1] if t + 1 < T1 else None
                    dc_prev = L['dc'][t & 1] if (t > 0 or self.learn_init) else None
                    K.convlstm_gates_bwd(L['gates'].v[t], L['c'].v[t - 1] if t > 0 else L.get('c0'), None, None, None, None, None, dys, dc_new,
                                         L['gates'].g[t], dc_prev, None)
                    L['rconv'].backward_data(L['gates'].g[t], a.g[t], beta=0)
                    self._in_act_conv_bwd(L, t, a.v[t][..., 0:f], [a.g[t][..., 0:f]], None)
                    continue
                elif L['rnn']:
                    a = L['a']
                    if t + 1 < T1:
                        dys.append(a.g[t + 1][..., f + L['zr']:f + L['zr'] + f])
                    n1, n2 = L['n1'], L['n2']
                    dc_new = L['dc'][(t + 1) & 1] if t + 1 < T1 else None
                    dc_prev = L['dc'][t & 1] if (t > 0 or self.learn_init) else None
                    bargs = (L['gates'].v[t], L['c'].v[t - 1] if t > 0 else L.get('c0'), n1.gamma, n1.beta, n2.gamma, n2.beta,
                             [n1.mean[t], n1.rstd[t], n2.mean[t], n2.rstd[t]], dys, dc_new, L['gates'].g[t], dc_prev,
                             [n1.dgamma, n1.dbeta, n2.dgamma, n2.dbeta])
                    skip = (f, L['zr']) if L['zless'] else None
                    nb = self._norm_bwd('nbstats', L, L['rconv'], L['gates'].g[t], a.g[t], nrm, L['pre'].v[t], skip)
                    if nb is not None:
                        nb['mean'], nb['rstd'] = nrm.mean[t], nrm.rstd[t]
                    if FUSED_ENTRIES and K.fused_ok():       # gate block backward + the gate convolution's DGRAD: one host call
                        K.convlstm_cell_bwd(L['rconv'].backward_data(L['gates'].g[t], a.g[t], beta=0, skip=skip, norm_bwd=nb, defer=True),
                                            K.convlstm_gates_bwd(*bargs, eps=EPS_IN, ws=self._lstm_ws(L), dgates_raw=L.get('dg_raw'), defer=True))
                    else:
                        K.convlstm_gates_bwd(*bargs, eps=EPS_IN, ws=self._lstm_ws(L), dgates_raw=L.get('dg_raw'))
                        L['rconv'].backward_data(L['gates'].g[t], a.g[t], beta=0, skip=skip, norm_bwd=nb)
                    self._in_act_conv_bwd(L, t, a.v[t][..., 0:f], [a.g[t][..., 0:f]], nb['ws'] if nb is not None else None)
                    continue
                else:
                    y0 = self._out_views(L, t)[0]
                    st_ = nb_last['ws'] if (L is self.layers[-1] and nb_last is not None and len(dys) == 1) else None
                    self._in_act_conv_bwd(L, t, y0, dys, st_)
                    continue
                L['conv'].backward_data(L['pre'].g[t], L['in'].g[t], beta=0)
            # d image -> previous step's generated frame where it was fed back (not ground truth)
            if t > 0:
                K.select_bwd(self.gt_mask[t], [in0.g[t][..., 0:C], self.dimg_cdna] +
                             ([maskin.g[t][..., self.o_prev:self.o_prev + C]] if self.o_prev is not None else []), self.gen.g[t - 1])
        if self.learn_init:                # gradients of the learned initial states: step 0's state gradients summed over the batch
            for L in self.layers:
                if L['rnn']:
                    f, hoff = L['f'], L['f'] + L['zr']
                    L['h0g'].add_(L['a'].g[0][..., hoff:hoff + f].float().sum(0))
                    if not self.gru:
                        L['c0g'].add_(L['dc'][0].sum(0))
        # ---- weight gradients: one split-K GEMM per layer over all (t, n) ----------------------------------------
        for L in self.layers:
            b, pre = L['in'], L['pre']
            L['conv'].backward_weights(b.flat(b.v), pre.flat(pre.g), feeds_instance_norm=True)
            if L['rnn'] and self.abl_rnn:
                a, p2 = L['a'], L['pre2']
                L['rconv'].backward_weights(a.flat(a.v), p2.flat(p2.g), feeds_instance_norm=True)
            elif L['rnn'] and self.gru:
                a, gt, cd = L['a'], L['gates'], L['cand']
                L['rconv'].backward_weights(a.flat(a.v)[..., 0:L['cin1']], gt.flat(gt.g))
                L['cconv'].backward_weights(a.flat(a.v), cd.flat(cd.g))
            elif L['rnn']:
                a, gt = L['a'], L['gates']
                L['rconv'].backward_weights(a.flat(a.v), gt.flat(gt.g))
        hl = self.h_last
        if self.tf == 'cdna':
            hs = self.hsmall
            self.cdna_dense.backward_weights(hs.v.reshape(T1 * N, -1), self.cdna_raw.g.reshape(T1 * N, -1))
        else:
            if not self.merge_heads:
                self.tf_conv.backward_weights(hl.flat(hl.v), hl.flat(self.tf_pre.g), feeds_instance_norm=True)
            self.tf_out.backward_weights(hl.flat(self.tf_h.v), hl.flat(self.tf_raw.g))
        if self.merge_heads:
            self.heads_conv.backward_weights(hl.flat(hl.v), hl.flat(self.heads_pre.g), feeds_instance_norm=True)
            self.heads_norm.finish()
        else:
            if self.scratch:
                self.scratch_conv.backward_weights(hl.flat(hl.v), hl.flat(self.scratch_pre.g), feeds_instance_norm=True)
            self.masks_conv.backward_weights(hl.flat(hl.v), hl.flat(self.masks_pre.g), feeds_instance_norm=True)
        if self.scratch:
            self.scratch_out.backward_weights(hl.flat(self.scratch_h.v), self.dscratch_pre.reshape(T1 * N, self.H, self.W, self.Cs))
        self.masks_out.backward_weights(hl.flat(maskin.v)[..., 0:self.mask_cin], hl.flat(self.logits.g))
        for c in self.convs:
            c.finish_weight_grad()
        # ---- state prediction: the state loss's gradient (the caller left it in gen_states.g) through the recurrence ----------
        if self.ns and state_grad:
            K.state_pred_bwd(self.gt_mask, self.spW, self.sa, self.gen_states.g, self.dspW, self.dspb)
        # ---- z path ----------------------------------------------------------------------------------------------
        if not nz:
            return None
        drz = self.rnn_z.g
        drz.zero_()
        cw = self.cw                     # the conditioning columns of each tiled slice are inputs / under stop_gradient: skipped
        for L in self.layers:
            b = L['in']
            if L['zc']:
                K.colsum(b.flat(b.g)[..., L['zoff_in'] + cw:L['zoff_in'] + cw + nz], drz, per_row=True)
            if L['rnn'] and L['zr'] and L.get('zless'):
                # z gradient of the gate convolution from the gate gradients of all timesteps (the DGRADs left those channels out)
                gt = L['gates']
                K.tiled_z_weff(L['rconv'].W, L['rconv'].geom, L['f'], nz, L['weff'])
                K.tiled_z_grad(gt.flat(gt.g), L['weff'], drz.reshape(T1 * N, nz), beta=1)
            elif L['rnn'] and L['zr']:
                a = L['a']
                K.colsum(a.flat(a.g)[..., L['f'] + cw:L['f'] + cw + nz], drz, per_row=True)
        if self.use_rnn_z and self.abl_rnn:               # tanh(dense(z)) backward: d pre = d rnn_z * (1 - rnn_z^2)
            dpre = self.fcz_pre.g.reshape(T1, N, nz)
            torch.mul(self.rnn_z.v, self.rnn_z.v, out=dpre)
            dpre.neg_().add_(1.0).mul_(drz)
            zs4 = self.zs.v.reshape(T1 * N, 1, 1, nz)
            self.fc_z.backward_data(self.fcz_pre.g, self.zs.g.reshape(T1 * N, 1, 1, nz), beta=0)
            self.fc_z.backward_weights(zs4, self.fcz_pre.g)
            self.fc_z.finish_weight_grad()
            return self.zs.g
        if self.use_rnn_z and self.hp.rnn == 'gru':
            R = T1 * N
            K.gru_seq_bwd(self.zA, self.zg.W, self.zc.W, self.z_ru, self.z_cand, drz, self.z_dGg, self.z_dGc, self.z_dA, nz,
                          dh0=self.z_dh0 if self.learn_init else None)
            self.zg.backward_weights(self.zA.reshape(R, 1, 1, 2 * nz), self.z_dGg.reshape(R, 1, 1, 2 * nz))
            self.zc.backward_weights(self.zA2.reshape(R, 1, 1, 2 * nz), self.z_dGc.reshape(R, 1, 1, nz))
            self.zs.g.copy_(self.z_dA[..., :nz])
            return self.zs.g
        if self.use_rnn_z:
            K.lstm_z_bwd(self.zs.v, self.zW, self.rnn_z.v, self.z_gates, self.z_cs, drz, self.zs.g, self.dzW, self.dzb,
                         init=(self.z_c0, self.z_h0) if self.learn_init else None,
                         dinit=(self.z_dc0, self.z_dh0) if self.learn_init else None)
            return self.zs.g
        return drz
