"""VAE posterior encoder and spectral-norm video discriminator on the HIP kernels.

Mirrors /root/reference/video_prediction/models/networks.py (encoder :12-32, video_sn_discriminator :72-108) and
the wiring around them in savp_model.py (posterior_fn :21-51, discriminator_given_video_fn :88-126).
"""
import os

import torch

from .. import kernels as K
from .. import lib
from ..engine import ConvLayer, copy_view, prep_layers
from ..variables import VIDEO_D_LAYERS, video_discriminator_shapes, image_discriminator_shapes

EPS_IN = 1e-6


class PosteriorEncoder(object):
    """posterior_fn (savp_model.py:21-51): frame pairs -> (z_mu, z_log_sigma_sq) for every step, flat batch M = (T-1)*B;
    with use_e_rnn the features pass through dense(nef*4) + BasicLSTMCell over time before the heads (:31-43).

    prior=True builds prior_fn (:54-85) under its own scope: the convolutional encoder sees only the context_frames-1 context
    pairs, the remaining sequence_length-context_frames feature rows are zeros, and the recurrent tail is always present."""

    def __init__(self, store, hp, image_shape, B, train=True, prefix='generator/encoder/', prior=False, n_actions=0):
        H, W, C = image_shape
        self.hp, self.store = hp, store
        self.T1 = T1 = hp.sequence_length - 1
        self.prior = prior
        self.Tc = Tc = (hp.context_frames - 1) if prior else T1           # steps whose frame pair is encoded
        self.na = na = int(n_actions)                                     # inputs['actions'] tiled behind the frame pair (savp_model.py:24-26,57-59)
        if prior and na and Tc != T1:
            # prior_fn tile-concatenates T-1 action steps to context_frames-1 frame pairs: ops.tile_concat only broadcasts size-1
            # dimensions (ops.py:995-1000), the reference's graph construction fails
            raise ValueError('learn_prior with actions: %d action steps against %d context frame pairs (savp_model.py:56-59)' % (T1, Tc))
        self.B, self.M, self.R = B, Tc * B, T1 * B
        M, R = self.M, self.R
        dev = store.device
        self.dev = dev
        if hp.norm_layer != 'instance':
            raise NotImplementedError('HIP encoder covers norm_layer=instance')
        self.recurrent = bool(prior or hp.use_e_rnn)
        if self.recurrent and hp.rnn not in ('lstm', 'gru'):
            raise NotImplementedError(hp.rnn)                                  # savp_model.py:40-41
        self.gru = bool(self.recurrent and hp.rnn == 'gru')
        self.pairs = torch.zeros(max(M, 1), H, W, 2 * C + na, device=dev)
        self.C = C
        self.layers = []
        cin, h, w = 2 * C + na, H, W
        x = self.pairs
        for i in range(hp.n_layers):
            cout = hp.nef * min(2 ** i, 4)
            s = prefix + 'layer_%d/' % (i + 1)
            L = {'conv': ConvLayer(store, s + 'conv2d/kernel', s + 'conv2d/bias', 'conv', (4, 4), (2, 2), (1, 1)),
                 'x': x, 'normed': i > 0}
            h, w = (h + 2 - 4) // 2 + 1, (w + 2 - 4) // 2 + 1
            L['y'] = torch.empty(max(M, 1), h, w, cout, device=dev)            # layer output (after lrelu)
            L['dy'] = torch.empty(max(M, 1), h, w, cout, device=dev) if train else None
            if i > 0:
                L['pre'] = torch.empty(max(M, 1), h, w, cout, device=dev)
                L['dpre'] = torch.empty(max(M, 1), h, w, cout, device=dev) if train else None
                L['gamma'], L['beta'] = store[s + 'InstanceNorm/gamma'], store[s + 'InstanceNorm/beta']
                L['dgamma'], L['dbeta'] = store.grad64(s + 'InstanceNorm/gamma'), store.grad64(s + 'InstanceNorm/beta')   # float64 accumulators
                L['mean'], L['rstd'] = torch.empty(max(M, 1), cout, device=dev), torch.empty(max(M, 1), cout, device=dev)
            self.layers.append(L)
            x = L['y']
            cin = cout
        self.hw = h * w
        # pooled features of ALL T1 steps; rows >= M (prior: the steps after the context) stay zero (savp_model.py:63-64)
        self.feat = torch.zeros(R, cin, device=dev)
        self.dfeat = torch.zeros(R, cin, device=dev) if train else None
        self.pooled = self.feat[:M]
        self.dpooled = self.dfeat[:M] if train else None
        hin = cin
        if self.recurrent:
            U = self.U = hp.nef * 4
            s = prefix + 'layer_%d/' % (hp.n_layers + 1)
            self.fc = ConvLayer(store, s + 'dense/kernel', s + 'dense/bias', 'conv', (1, 1), (1, 1), (0, 0))
            self.A = torch.zeros(T1, B, 2 * U, device=dev)                  # [dense(feat)_t | h_{t-1}]
            self.hout = torch.empty(T1, B, U, device=dev)
            if train:
                self.dh = torch.empty(T1, B, U, device=dev)
                self.dA = torch.empty(T1, B, 2 * U, device=dev)
            if self.gru:
                # tf.contrib.rnn.GRUCell (savp_model.py:38-39): gates/{kernel [2U,2U], bias}, candidate/{kernel [2U,U], bias}; as with the
                # LSTM cell the kernels are 1x1 layer objects of which only the WGRAD entry is used
                s = prefix + 'gru/rnn/gru_cell/'
                self.cell_g = ConvLayer(store, s + 'gates/kernel', s + 'gates/bias', 'conv', (1, 1), (1, 1), (0, 0))
                self.cell_c = ConvLayer(store, s + 'candidate/kernel', s + 'candidate/bias', 'conv', (1, 1), (1, 1), (0, 0))
                for c_ in (self.cell_g, self.cell_c):
                    c_.need_wt = c_.need_wd = False
                self.A2 = torch.zeros(T1, B, 2 * U, device=dev)             # [dense(feat)_t | r_t * h_{t-1}]
                self.ru = torch.empty(T1, B, 2 * U, device=dev)
                self.cand = torch.empty(T1, B, U, device=dev)
                if train:
                    self.dGg = torch.empty(T1, B, 2 * U, device=dev)
                    self.dGc = torch.empty(T1, B, U, device=dev)
            else:
                s = prefix + '%s/rnn/basic_lstm_cell/' % hp.rnn
                # the cell's [2U,4U] kernel as a 1x1 layer object: only its WGRAD entry is used (dW = A^T dG over all (t,b) rows)
                self.cell = ConvLayer(store, s + 'kernel', s + 'bias', 'conv', (1, 1), (1, 1), (0, 0))
                self.cell.need_wt = self.cell.need_wd = False
                self.gates = torch.empty(T1, B, 4 * U, device=dev)
                self.cs = torch.empty(T1, B, U, device=dev)
                if train:
                    self.dG = torch.empty(T1, B, 4 * U, device=dev)
            hin = U
        self.mu_fc = ConvLayer(store, prefix + 'z_mu/dense/kernel', prefix + 'z_mu/dense/bias', 'conv', (1, 1), (1, 1), (0, 0))
        self.ls_fc = ConvLayer(store, prefix + 'z_log_sigma_sq/dense/kernel', prefix + 'z_log_sigma_sq/dense/bias', 'conv',
                               (1, 1), (1, 1), (0, 0))
        nz = hp.nz
        self.mu = torch.empty(T1, B, nz, device=dev)
        self.ls_raw = torch.empty(T1, B, nz, device=dev)
        self.ls = torch.empty(T1, B, nz, device=dev)
        self.z = torch.empty(T1, B, nz, device=dev)
        self.dmu = torch.empty(T1, B, nz, device=dev) if train else None
        self.dls = torch.empty(T1, B, nz, device=dev) if train else None
        self.kl = torch.zeros(1, device=dev, dtype=torch.float64)     # float64 accumulator (savp_reparam_fwd / savp_kl_gauss)
        self.convs = [L['conv'] for L in self.layers] + ([self.fc] if self.recurrent else []) + [self.mu_fc, self.ls_fc]

    def prep_weights(self):
        prep_layers(self.convs)

    def _head_input(self):
        return self.hout.reshape(self.R, -1) if self.recurrent else self.feat

    def forward(self, images, eps, kl=True, actions=None):
        """images [T,B,H,W,C] contiguous; eps [T1,B,nz]; actions [T1,B,na] when built with n_actions.  Returns z [T1,B,nz] = mu +
        sigma*eps (and fills mu, ls; kl=True also accumulates KL(q || N(0,1)) into self.kl -- the learned-prior KL is taken by the
        caller with kernels.kl_gauss)."""
        T1, B, C, M, R, Tc = self.T1, self.B, self.C, self.M, self.R, self.Tc
        if M:
            a = images[:Tc].reshape((M,) + tuple(images.shape[2:]))
            b = images[1:Tc + 1].reshape((M,) + tuple(images.shape[2:]))
            copy_view(a, [self.pairs[..., 0:C]])                                   # image_pairs = concat([x_t, x_t+1])  :23
            copy_view(b, [self.pairs[..., C:2 * C]])
            if self.na:                                                            # tile_concat([image_pairs, actions[..., None, None, :]])  :24-26
                if actions is None:
                    raise ValueError('this encoder was built for inputs with actions')
                K.tile_channels(actions[:Tc].reshape(M, self.na), self.pairs[..., 2 * C:2 * C + self.na])
            for L in self.layers:
                if not L['normed']:
                    L['conv'].forward(L['x'], L['y'], act=lib.ACT_LRELU, alpha=0.2)            # networks.py:18-19
                else:
                    L['conv'].forward(L['x'], L['pre'])
                    K.instnorm_act_fwd(L['pre'], L['gamma'], L['beta'], [L['y']], L['mean'], L['rstd'], act='lrelu', alpha=0.2,
                                       eps=EPS_IN)                                              # networks.py:25-27
            last = self.layers[-1]['y']
            self.pooled.zero_()
            K.colsum(last, self.pooled, scale=1.0 / self.hw, per_row=True)                     # networks.py:30-31
        if self.recurrent:
            U = self.U
            self.fc.forward(self.feat, self.A.reshape(R, 2 * U)[:, :U])                        # savp_model.py:32-33 / 66-67
            if self.gru:
                K.gru_seq_fwd(self.A, self.A2, self.cell_g.W, self.cell_g.bias, self.cell_c.W, self.cell_c.bias, self.hout, self.ru,
                              self.cand, U)                                                    # :38-43 with rnn = 'gru'
            else:
                K.lstm_seq_fwd(self.A, self.cell.W, self.cell.bias, self.hout, self.gates, self.cs, U)     # :35-43 / 69-76
        hin = self._head_input()
        self.mu_fc.forward(hin, self.mu.reshape(R, -1))
        self.ls_fc.forward(hin, self.ls_raw.reshape(R, -1))
        self.eps = eps
        self.kl.zero_()
        K.reparam_fwd(self.mu, self.ls_raw, eps, self.ls, self.z, self.kl if kl else None)  # savp_model.py:45-49,712
        return self.z

    def backward(self, dz, kl_weight, kl_weight_dev=None):
        """dz [T1,B,nz] = dL/dz (may be None); adds the gradient of kl_weight * KL(q || N(0,1)) (weight read from the 1-element
        device tensor kl_weight_dev if given), then back-propagates through the network."""
        self.reparam_backward(dz, kl_weight, kl_weight_dev)
        self.backward_network()

    def reparam_backward(self, dz, kl_weight=0.0, kl_weight_dev=None):
        """dmu / dls <- gradient of z = mu + sigma*eps (and of the standard-normal KL when weighted in).  With a learned prior the
        caller adds the Gaussian-vs-Gaussian KL terms of both networks (kernels.kl_gauss) before backward_network()."""
        K.reparam_bwd(self.mu, self.ls_raw, self.eps, dz, kl_weight or 0.0, self.dmu, self.dls, klw_dev=kl_weight_dev)

    def backward_network(self):
        M, R = self.M, self.R
        dmu2, dls2 = self.dmu.reshape(R, -1), self.dls.reshape(R, -1)
        hin = self._head_input()
        dhin = self.dh.reshape(R, -1) if self.recurrent else self.dfeat
        self.mu_fc.backward_data(dmu2, dhin, beta=0)
        self.ls_fc.backward_data(dls2, dhin, beta=1)
        self.mu_fc.backward_weights(hin, dmu2)
        self.ls_fc.backward_weights(hin, dls2)
        if self.recurrent:
            U = self.U
            if self.gru:
                K.gru_seq_bwd(self.A, self.cell_g.W, self.cell_c.W, self.ru, self.cand, self.dh, self.dGg, self.dGc, self.dA, U)
                self.cell_g.backward_weights(self.A.reshape(R, 2 * U), self.dGg.reshape(R, 2 * U))
                self.cell_c.backward_weights(self.A2.reshape(R, 2 * U), self.dGc.reshape(R, U))
            else:
                K.lstm_seq_bwd(self.A, self.cell.W, self.gates, self.cs, self.dh, self.dG, self.dA, U)
                self.cell.backward_weights(self.A.reshape(R, 2 * U), self.dG.reshape(R, 4 * U))     # dW = A^T dG, db = colsum(dG)
            dx = self.dA.reshape(R, 2 * U)[:, :U]
            self.fc.backward_data(dx, self.dfeat, beta=0)
            self.fc.backward_weights(self.feat, dx)
        if M:
            lastL = self.layers[-1]
            K.tile_channels(self.dpooled, lastL['dy'], scale=1.0 / self.hw)
            for i in range(len(self.layers) - 1, -1, -1):
                L = self.layers[i]
                if L['normed']:
                    K.instnorm_act_bwd(L['pre'], L['gamma'], L['beta'], L['y'], L['mean'], L['rstd'], [L['dy']], L['dpre'],
                                       L['dgamma'], L['dbeta'], act='lrelu', alpha=0.2, eps=EPS_IN)
                    dpre = L['dpre']
                else:
                    dpre = L['dy']       # already multiplied by lrelu' in the producing DGRAD epilogue
                if i > 0:
                    below = self.layers[i - 1]
                    if below['normed']:
                        L['conv'].backward_data(dpre, below['dy'], beta=0)
                    else:
                        L['conv'].backward_data(dpre, below['dy'], beta=0, act=lib.ACT_DLRELU_FROM_OUT, alpha=0.2, aux=below['y'])
                L['conv'].backward_weights(L['x'], dpre)
        for c in self.convs:
            c.finish_weight_grad()


class SNDiscriminator(object):
    """Spectral-norm discriminators of networks.py on batch-major clips [Nc, clip, H, W, C]:
      kind 'video'  : video_sn_discriminator (:72-108), 3-D convs over the whole clip;
      kind 'image'  : image_sn_discriminator (:35-69) on ONE sampled frame per sequence (clip length 1);
      kind 'images' : image_sn_discriminator applied to every frame of the clip (with_flat_batch, savp_model.py:119-125).
    Rows lo:hi of every method are in units of sequences."""

    def __init__(self, store, hp, image_shape, Nc, prefix, kind='video', train=True):
        H, W, C = image_shape
        self.hp, self.store, self.Nc, self.kind = hp, store, Nc, kind
        self.prefix = prefix                       # variable scope of this network (= one chunk of the 'd' gradient arena)
        dev = store.device
        self.dev = dev
        self.frames = 1 if kind == 'image' else hp.clip_length
        self.rows_per_seq = self.frames if kind == 'images' else 1
        self.clip = torch.zeros(Nc, self.frames, H, W, C, device=dev)
        self.dclip = torch.empty(Nc, self.frames, H, W, C, device=dev) if train else None
        self.layers = []
        if kind == 'video':
            layers, flat = video_discriminator_shapes(hp, image_shape)
            x = self.clip
            R = Nc
        else:
            layers, flat = image_discriminator_shapes(hp, image_shape)
            R = Nc * self.frames
            x = self.clip.reshape(R, H, W, C)
            self.x0 = x
            self.dx0 = self.dclip.reshape(R, H, W, C) if train else None
        self.R = R
        for (scope, kshape, st, odims) in layers:
            s = prefix + scope + '/'
            k = kshape[0]
            if kind == 'video':
                conv = ConvLayer(store, s + 'conv3d/kernel', s + 'bias', 'conv', (k, k, k), st, (1, 1, 1), sn_u=s + 'conv3d/u')
            else:
                conv = ConvLayer(store, s + 'conv2d/kernel', s + 'conv2d/bias', 'conv', (k, k), (st, st), (1, 1),
                                 sn_u=s + 'conv2d/u')
            L = {'conv': conv, 'x': x}
            L['y'] = torch.empty((R,) + tuple(odims) + (kshape[-1],), device=dev)
            L['dy'] = torch.empty_like(L['y']) if train else None
            self.layers.append(L)
            x = L['y']
        s = prefix + 'sn_fc4/'
        self.fc = ConvLayer(store, s + 'dense/kernel', s + 'dense/bias', 'conv', (1, 1), (1, 1), (0, 0), sn_u=s + 'dense/u')
        self.flat = flat
        self.logits = torch.empty(R, 1, device=dev)
        self.dlogits = torch.zeros(R, 1, device=dev)
        self.convs = [L['conv'] for L in self.layers] + [self.fc]

    def prep_weights(self, update_u=False):
        """Spectral norm of all layers in one batched call (4 launches for the 8 layers), then the per-layer packs."""
        batch = os.environ.get('SAVP_SN_BATCH', '1') == '1'
        if batch:
            K.sn_fwd_batch([c.sn_entry(update_u) for c in self.convs])
        prep_layers(self.convs, update_u=update_u, sn_done=batch)

    def commit_u(self):
        for c in self.convs:
            c.commit_u()

    def features(self):
        return [L['y'] for L in self.layers]

    def rows(self, lo, hi):
        return lo * self.rows_per_seq, hi * self.rows_per_seq

    def forward(self, n=None):
        """Run on the first n sequences of self.clip (default all).  lrelu(0.1) is fused into each conv epilogue."""
        n = (n or self.Nc) * self.rows_per_seq
        for L in self.layers:
            L['conv'].forward(L['x'][:n], L['y'][:n], act=lib.ACT_LRELU, alpha=0.1)          # networks.py:45-64 / 83-102
        self.fc.forward(self.layers[-1]['y'][:n].reshape(n, -1), self.logits[:n])           # networks.py:66-67 / 104-105
        return self.logits

    def backward(self, lo, hi, weights=True, data=True, feature_grads=False):
        """Back-propagate self.dlogits (and, if feature_grads, the gradients already stored in each layer's dy) through
        sequences lo..hi.  weights: accumulate dW/dbias; data: produce self.dclip[lo:hi]."""
        lo, hi = self.rows(lo, hi)
        n = hi - lo
        top = self.layers[-1]
        self.fc.backward_data(self.dlogits[lo:hi], top['dy'][lo:hi].reshape(n, -1), beta=1 if feature_grads else 0,
                              act=lib.ACT_DLRELU_FROM_OUT, alpha=0.1, aux=top['y'][lo:hi].reshape(n, -1))
        if weights:
            self.fc.backward_weights(top['y'][lo:hi].reshape(n, -1), self.dlogits[lo:hi])
        for i in range(len(self.layers) - 1, -1, -1):
            L = self.layers[i]
            dpre = L['dy'][lo:hi]                  # = dL/d(conv output before lrelu)
            if i > 0:
                below = self.layers[i - 1]
                L['conv'].backward_data(dpre, below['dy'][lo:hi], beta=1 if feature_grads else 0,
                                        act=lib.ACT_DLRELU_FROM_OUT, alpha=0.1, aux=below['y'][lo:hi])
            elif data:
                dst = self.dclip if self.kind == 'video' else self.dx0
                L['conv'].backward_data(dpre, dst[lo:hi], beta=0)
            if weights:
                L['conv'].backward_weights(L['x'][lo:hi], dpre)

    def finish_weight_grads(self):
        batch = os.environ.get('SAVP_SN_BATCH', '1') == '1'
        if batch:
            K.sn_bwd_batch([c.sn_bwd_entry() for c in self.convs])
        for c in self.convs:
            c.finish_weight_grad(sn_done=batch)


VideoDiscriminator = SNDiscriminator
