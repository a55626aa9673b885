"""Variable inventory of the SAVP graph, keyed by the reference's TF variable names.

The reference creates its variables implicitly with tf.get_variable while building the graph; the names below
follow its variable scopes (models/base_model.py:411,415; models/savp_model.py:426-633,709; models/networks.py:
17-28,45-67,83-105; rnn_ops.py:104-135; layers/normalization.py:94-142; ops.py:6-15,515-541,769-776,1027) so that
a TF-checkpoint importer is a rename table.  Initialisers: truncated normal sigma 0.02 for kernels (ops.py:9,517,
rnn_ops.py:120), zeros for biases/beta, ones for gamma, truncated normal sigma 1 for the spectral-norm ``u``
vectors (ops.py:1027), TF's default glorot-uniform for the z-LSTM kernel (savp_model.py:356-362).
"""
from collections import OrderedDict

import numpy as np

VIDEO_D_LAYERS = [  # (scope, ndf multiplier, kernel, strides(d,h,w))   networks.py:83-102
    ('sn_conv0_0', 1, 3, (1, 1, 1)),
    ('sn_conv0_1', 2, 4, (1, 2, 2)),
    ('sn_conv1_0', 2, 3, (1, 1, 1)),
    ('sn_conv1_1', 4, 4, (1, 2, 2)),
    ('sn_conv2_0', 4, 3, (1, 1, 1)),
    ('sn_conv2_1', 8, 4, (2, 2, 2)),
    ('sn_conv3_0', 8, 3, (1, 1, 1)),
]
IMAGE_D_LAYERS = [  # networks.py:45-64
    ('sn_conv0_0', 1, 3, 1), ('sn_conv0_1', 2, 4, 2), ('sn_conv1_0', 2, 3, 1), ('sn_conv1_1', 4, 4, 2),
    ('sn_conv2_0', 4, 3, 1), ('sn_conv2_1', 8, 4, 2), ('sn_conv3_0', 8, 3, 1),
]


def layer_specs(ngf, height, width):
    """SAVPCell encoder/decoder layer table (savp_model.py:179-237)."""
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


def num_masks(hp):
    """savp_model.py:240-246."""
    return (hp.last_frames * hp.num_transformed_images +
            int(bool(hp.prev_image_background)) +
            int(bool(hp.first_image_background and not hp.context_images_background)) +
            int(bool(hp.last_image_background and not hp.context_images_background)) +
            int(bool(hp.last_context_image_background and not hp.context_images_background)) +
            (hp.context_frames if hp.context_images_background else 0) +
            int(bool(hp.generate_scratch_image)))


def uses_discriminator(hp):
    return bool(hp.video_sn_gan_weight or hp.video_sn_vae_gan_weight or
                hp.image_sn_gan_weight or hp.image_sn_vae_gan_weight or
                hp.images_sn_gan_weight or hp.images_sn_vae_gan_weight)


def generator_variable_specs(hp, image_shape, cond=(0, 0)):
    """name -> (shape, init) for everything under scope 'generator/'.

    cond = (n_actions, n_states): widths of inputs['actions'] / inputs['states'] when the dataset supplies them (savp_model.py:24-26,
    413-444: tiled into the encoder's frame pairs and, beside the latent, into every tile-concatenated slice of the cell; :655-658: the
    next state is a dense layer of [actions | state])."""
    H, W, C = image_shape
    specs = OrderedDict()
    nz = hp.nz
    na, ns = int(cond[0]), int(cond[1])
    zw = nz + na + ns                 # width of state_action_z (savp_model.py:414-444): [actions | stop_gradient(state) | z]
    def rnn_cell_specs(scope, n_in, u):
        """savp_model.py:36-41,354-362: 'lstm' = BasicLSTMCell / LSTMCell(name='basic_lstm_cell') -- kernel [in + u, 4u] with get_variable's
        default initializer (glorot-uniform), zero bias; 'gru' = tf.contrib.rnn.GRUCell (default name 'gru_cell') -- gates/{kernel [in + u,
        2u], bias = 1}, candidate/{kernel [in + u, u], bias = 0} (rnn_cell_impl.GRUCell.build)."""
        if hp.rnn == 'lstm':
            specs[scope + 'basic_lstm_cell/kernel'] = ((n_in + u, 4 * u), 'glorot')
            specs[scope + 'basic_lstm_cell/bias'] = ((4 * u,), 'zeros')
        elif hp.rnn == 'gru':
            specs[scope + 'gru_cell/gates/kernel'] = ((n_in + u, 2 * u), 'glorot')
            specs[scope + 'gru_cell/gates/bias'] = ((2 * u,), 'ones')
            specs[scope + 'gru_cell/candidate/kernel'] = ((n_in + u, u), 'glorot')
            specs[scope + 'gru_cell/candidate/bias'] = ((u,), 'zeros')
        else:
            raise NotImplementedError(hp.rnn)                                       # savp_model.py:361-362

    def encoder_specs(p, recurrent):
        """networks.encoder + the optional recurrent tail + the two heads under scope p (savp_model.py:21-51 posterior_fn with
        use_e_rnn, :54-85 prior_fn which always has the tail)."""
        cin = 2 * C + na                                                           # frame pair + tiled actions (savp_model.py:23-26,56-59)
        for i in range(hp.n_layers):
            cout = hp.nef * min(2 ** i, 4)
            s = p + 'layer_%d/' % (i + 1)
            specs[s + 'conv2d/kernel'] = ((4, 4, cin, cout), 'tn0.02')
            specs[s + 'conv2d/bias'] = ((cout,), 'zeros')
            if i > 0 and hp.norm_layer == 'instance':
                specs[s + 'InstanceNorm/beta'] = ((cout,), 'zeros')
                specs[s + 'InstanceNorm/gamma'] = ((cout,), 'ones')
            cin = cout
        if recurrent:
            u = hp.nef * 4
            s = p + 'layer_%d/' % (hp.n_layers + 1)
            specs[s + 'dense/kernel'] = ((cin, u), 'tn0.02')
            specs[s + 'dense/bias'] = ((u,), 'zeros')
            # tf_utils.unroll_rnn = tf.nn.dynamic_rnn under scope hparams.rnn: '<rnn>/rnn/<cell name>/...'
            rnn_cell_specs(p + '%s/rnn/' % hp.rnn, u, u)
            cin = u
        for head in ('z_mu', 'z_log_sigma_sq'):
            specs[p + head + '/dense/kernel'] = ((cin, nz), 'tn0.02')
            specs[p + head + '/dense/bias'] = ((nz,), 'zeros')

    if nz:
        encoder_specs('generator/encoder/', bool(hp.use_e_rnn))
        if hp.learn_prior:
            encoder_specs('generator/prior/', True)

    p = 'generator/rnn/savp_cell/'
    if nz and hp.use_rnn_z and getattr(hp, 'ablation_rnn', False):
        specs[p + 'fc_z/dense/kernel'] = ((nz, nz), 'tn0.02')                       # dense + tanh instead of the cell (savp_model.py:426-429)
        specs[p + 'fc_z/dense/bias'] = ((nz,), 'zeros')
    elif nz and hp.use_rnn_z:
        rnn_cell_specs(p + '%s_z/' % hp.rnn, nz, nz)                                # scope '%s_z' % rnn (savp_model.py:426)
    tile = hp.use_tile_concat
    zc = zw if tile else 0          # channels added by tile_concat
    enc, dec = layer_specs(hp.ngf, H, W)

    def norm(scope, c):
        if hp.norm_layer == 'instance':
            specs[scope + 'InstanceNorm/beta'] = ((c,), 'zeros')
            specs[scope + 'InstanceNorm/gamma'] = ((c,), 'ones')

    ablation_rnn = bool(getattr(hp, 'ablation_rnn', False))
    cell_norm = hp.conv_rnn_norm_layer != 'none' and not getattr(hp, 'ablation_conv_rnn_norm', False)

    def conv_rnn(scope, cx, f, add_z):
        cin = cx + (zc if add_z else 0)
        if ablation_rnn:
            # savp_model.py:474-478 / :510-513: conv2d 5x5 (+ dense(z) without tile_concat) -> norm_layer -> activation, scope conv_h<i>
            specs[scope + 'conv2d/kernel'] = ((5, 5, cin, f), 'tn0.02')
            specs[scope + 'conv2d/bias'] = ((f,), 'zeros')
            if add_z and zw and not tile:
                specs[scope + 'dense/kernel'] = ((zw, f), 'tn0.02')
            norm(scope, f)
            return
        if getattr(hp, 'ablation_conv_rnn_norm', False) and hp.conv_rnn_norm_layer != 'none':
            # :380-384: the cell is built without a normalizer; normalizer_fn(h) keeps its variables in its default scope beside the cell's
            specs[scope + 'InstanceNorm/beta'] = ((f,), 'zeros')
            specs[scope + 'InstanceNorm/gamma'] = ((f,), 'ones')
        if hp.conv_rnn == 'lstm':
            s = scope + 'basic_conv2dlstm_cell/'
            specs[s + 'kernel'] = ((5, 5, cin + f, 4 * f), 'tn0.02')
            if add_z and zw and not tile:
                specs[s + 'weights'] = ((zw, 4 * f), 'tn0.02')
            if not cell_norm:
                specs[s + 'bias'] = ((4 * f,), 'zeros')
            else:
                specs[s + 'input_transform_forget_output/gamma'] = ((4 * f,), 'ones')
                specs[s + 'input_transform_forget_output/beta'] = ((4 * f,), 'zeros')
                specs[s + 'state/gamma'] = ((f,), 'ones')
                specs[s + 'state/beta'] = ((f,), 'zeros')
        elif hp.conv_rnn == 'gru':
            s = scope + 'conv2dgru_cell/'
            specs[s + 'gates/kernel'] = ((5, 5, cin + f, 2 * f), 'tn0.02')
            specs[s + 'candidate/kernel'] = ((5, 5, cin + 2 * f, f), 'tn0.02')
            if add_z and zw and not tile:
                specs[s + 'gates/weights'] = ((zw, 2 * f), 'tn0.02')
                specs[s + 'candidate/weights'] = ((zw, f), 'tn0.02')
            if not cell_norm:
                specs[s + 'gates/bias'] = ((2 * f,), 'ones')
                specs[s + 'candidate/bias'] = ((f,), 'zeros')
            else:
                specs[s + 'gates/reset_update/gamma'] = ((2 * f,), 'ones')
                specs[s + 'gates/reset_update/beta'] = ((2 * f,), 'ones')
                specs[s + 'candidate/state/gamma'] = ((f,), 'ones')
                specs[s + 'candidate/state/beta'] = ((f,), 'zeros')
        else:
            raise NotImplementedError(hp.conv_rnn)

    layer_out = []     # channels of layers[i][-1]
    prev = None
    for i, (f, use_rnn) in enumerate(enc):
        s = p + 'h%d/' % i
        cx = 2 * C if i == 0 else prev
        k = 5 if i == 0 else 3
        add_z = bool(zw) and (hp.where_add == 'all' or (hp.where_add == 'input' and i == 0))
        specs[s + 'conv_pool2d/kernel'] = ((k, k, cx + (zc if add_z else 0), f), 'tn0.02')
        specs[s + 'conv_pool2d/bias'] = ((f,), 'zeros')
        if add_z and not tile:
            specs[s + 'dense/kernel'] = ((zw, f), 'tn0.02')
        norm(s, f)
        if use_rnn:
            conv_rnn(p + '%s_h%d/' % ('conv' if ablation_rnn else hp.conv_rnn, i), f, f, bool(zw) and hp.where_add == 'all')
        layer_out.append(f)
        prev = f
    ne = len(enc)
    for i, (f, use_rnn) in enumerate(dec):
        li = ne + i
        s = p + 'h%d/' % li
        cx = prev if i == 0 else prev + layer_out[ne - i - 1]
        add_z = bool(zw) and (hp.where_add == 'all' or (hp.where_add == 'middle' and i == 0))
        specs[s + 'upsample_conv2d/kernel'] = ((3, 3, cx + (zc if add_z else 0), f), 'tn0.02')
        specs[s + 'upsample_conv2d/bias'] = ((f,), 'zeros')
        if add_z and not tile:
            specs[s + 'dense/kernel'] = ((zw, f), 'tn0.02')
        norm(s, f)
        if use_rnn:
            conv_rnn(p + '%s_h%d/' % ('conv' if ablation_rnn else hp.conv_rnn, li), f, f, bool(zw) and hp.where_add == 'all')
        layer_out.append(f)
        prev = f
    nl = len(layer_out)
    last = layer_out[-1]
    nk = hp.last_frames * hp.num_transformed_images
    if nk:
        if hp.transformation == 'flow':
            s = p + 'h%d_flow/' % nl
            specs[s + 'conv2d/kernel'] = ((3, 3, last, hp.ngf), 'tn0.02')
            specs[s + 'conv2d/bias'] = ((hp.ngf,), 'zeros')
            norm(s, hp.ngf)
            specs[p + 'flows/conv2d/kernel'] = ((3, 3, hp.ngf, 2 * nk), 'tn0.02')
            specs[p + 'flows/conv2d/bias'] = ((2 * nk,), 'zeros')
        elif hp.transformation == 'dna':
            kh, kw = hp.kernel_size
            s = p + 'h%d_dna_kernel/' % nl
            specs[s + 'conv2d/kernel'] = ((3, 3, last, hp.ngf), 'tn0.02')
            specs[s + 'conv2d/bias'] = ((hp.ngf,), 'zeros')
            norm(s, hp.ngf)
            specs[p + 'dna_kernels/conv2d/kernel'] = ((3, 3, hp.ngf, kh * kw * nk), 'tn0.02')
            specs[p + 'dna_kernels/conv2d/bias'] = ((kh * kw * nk,), 'zeros')
        elif hp.transformation == 'cdna':
            kh, kw = hp.kernel_size
            sh, sw = H // (2 ** ne), W // (2 ** ne)
            specs[p + 'cdna_kernels/dense/kernel'] = ((sh * sw * layer_out[ne - 1], kh * kw * nk), 'tn0.02')
            specs[p + 'cdna_kernels/dense/bias'] = ((kh * kw * nk,), 'zeros')
        else:
            raise ValueError('Invalid transformation %s' % hp.transformation)
    if hp.generate_scratch_image:
        s = p + 'h%d_scratch/' % nl
        specs[s + 'conv2d/kernel'] = ((3, 3, last, hp.ngf), 'tn0.02')
        specs[s + 'conv2d/bias'] = ((hp.ngf,), 'zeros')
        norm(s, hp.ngf)
        specs[p + 'scratch_image/conv2d/kernel'] = ((3, 3, hp.ngf, C), 'tn0.02')
        specs[p + 'scratch_image/conv2d/bias'] = ((C,), 'zeros')
    nm = num_masks(hp)
    if nm > 1:
        s = p + 'h%d_masks/' % nl
        specs[s + 'conv2d/kernel'] = ((3, 3, last, hp.ngf), 'tn0.02')
        specs[s + 'conv2d/bias'] = ((hp.ngf,), 'zeros')
        norm(s, hp.ngf)
        cin = hp.ngf + (nm * C if hp.dependent_mask else 0)
        specs[p + 'masks/conv2d/kernel'] = ((3, 3, cin, nm), 'tn0.02')
        specs[p + 'masks/conv2d/bias'] = ((nm,), 'zeros')
    if ns:
        specs[p + 'state_pred/dense/kernel'] = ((na + ns, ns), 'tn0.02')            # savp_model.py:655-658
        specs[p + 'state_pred/dense/bias'] = ((ns,), 'zeros')
    if getattr(hp, 'learn_initial_state', False):
        # savp_model.py:295-307: one variable per entry of nest.flatten({'conv_rnn_states': [...], 'rnn_z_state': ...}) -- dict keys sorted,
        # LSTM state tuples as (c, h) -- created where the cell is constructed (scope `generator/`, not `generator/rnn/savp_cell/`)
        shapes = []
        h_, w_ = H, W
        for f, use_rnn in enc:
            h_, w_ = h_ // 2, w_ // 2
            if use_rnn and not ablation_rnn:
                shapes += [(h_, w_, f)] * (2 if hp.conv_rnn == 'lstm' else 1)
        for f, use_rnn in dec:
            h_, w_ = h_ * 2, w_ * 2
            if use_rnn and not ablation_rnn:
                shapes += [(h_, w_, f)] * (2 if hp.conv_rnn == 'lstm' else 1)
        if nz and hp.use_rnn_z and not ablation_rnn:
            shapes += [(nz,)] * (2 if hp.rnn == 'lstm' else 1)
        for i, shp in enumerate(shapes):
            specs['generator/initial_state_%d/initial_state' % i] = (shp, 'zeros')
    return specs


def video_discriminator_shapes(hp, image_shape):
    """[(scope, kernel_shape, strides, out_dhw)] + flat size, for clips [clip_length,H,W,C] (networks.py:72-108)."""
    H, W, C = image_shape
    d, h, w, cin = hp.clip_length, H, W, C
    layers = []
    for scope, mult, k, st in VIDEO_D_LAYERS:
        cout = hp.ndf * mult
        d = (d + 2 - k) // st[0] + 1
        h = (h + 2 - k) // st[1] + 1
        w = (w + 2 - k) // st[2] + 1
        layers.append((scope, (k, k, k, cin, cout), st, (d, h, w)))
        cin = cout
    return layers, d * h * w * cin


def image_discriminator_shapes(hp, image_shape):
    H, W, C = image_shape
    h, w, cin = H, W, C
    layers = []
    for scope, mult, k, st in IMAGE_D_LAYERS:
        cout = hp.ndf * mult
        h = (h + 2 - k) // st + 1
        w = (w + 2 - k) // st + 1
        layers.append((scope, (k, k, cin, cout), st, (h, w)))
        cin = cout
    return layers, h * w * cin


def discriminator_variable_specs(hp, image_shape):
    """name -> (shape, init) under 'discriminator/' (savp_model.py:104-160)."""
    specs = OrderedDict()
    if not uses_discriminator(hp):
        return specs
    prefixes = []
    if hp.nz and not hp.use_same_discriminator:
        prefixes.append('discriminator/encoder/')
    prefixes.append('discriminator/')
    for p in prefixes:
        for sub, on in (('image/', hp.image_sn_gan_weight or hp.image_sn_vae_gan_weight),
                        ('images/', hp.images_sn_gan_weight or hp.images_sn_vae_gan_weight)):
            if not on:
                continue
            layers, flat = image_discriminator_shapes(hp, image_shape)
            for scope, kshape, st, _ in layers:
                s = p + sub + scope + '/'
                specs[s + 'conv2d/kernel'] = (kshape, 'tn0.02')
                specs[s + 'conv2d/u'] = ((1, kshape[-1]), 'tn1')
                specs[s + 'conv2d/bias'] = ((kshape[-1],), 'zeros')
            s = p + sub + 'sn_fc4/'
            specs[s + 'dense/kernel'] = ((flat, 1), 'tn0.02')
            specs[s + 'dense/u'] = ((1, 1), 'tn1')
            specs[s + 'dense/bias'] = ((1,), 'zeros')
        if hp.video_sn_gan_weight or hp.video_sn_vae_gan_weight:
            layers, flat = video_discriminator_shapes(hp, image_shape)
            for scope, kshape, st, _ in layers:
                s = p + 'video/' + scope + '/'
                specs[s + 'conv3d/kernel'] = (kshape, 'tn0.02')
                specs[s + 'conv3d/u'] = ((1, kshape[-1]), 'tn1')
                specs[s + 'bias'] = ((kshape[-1],), 'zeros')      # quirk: bias lives outside 'conv3d' (ops.py:769-776)
            s = p + 'video/sn_fc4/'
            specs[s + 'dense/kernel'] = ((flat, 1), 'tn0.02')
            specs[s + 'dense/u'] = ((1, 1), 'tn1')
            specs[s + 'dense/bias'] = ((1,), 'zeros')
    return specs


def variable_specs(hp, image_shape, mode='train', cond=(0, 0)):
    specs = generator_variable_specs(hp, image_shape, cond)
    if mode == 'train':
        specs.update(discriminator_variable_specs(hp, image_shape))
    return specs


def is_trainable(name):
    """Spectral-norm ``u`` vectors are created with trainable=False (ops.py:1027)."""
    return not name.endswith('/u')


def _trunc_normal(rng, shape, std):
    out = rng.standard_normal(size=shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(out) > 2.0
    return out * std


def init_variables(specs, seed=4, dtype=np.float32):
    """Seeded numpy initialisation of every variable in ``specs`` (used by tests, bench and training alike)."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for name, (shape, kind) in specs.items():
        if kind == 'tn0.02':
            v = _trunc_normal(rng, shape, 0.02)
        elif kind == 'tn1':
            v = _trunc_normal(rng, shape, 1.0)
        elif kind == 'zeros':
            v = np.zeros(shape)
        elif kind == 'ones':
            v = np.ones(shape)
        elif kind == 'glorot':
            limit = np.sqrt(6.0 / (shape[0] + shape[1]))
            v = rng.uniform(-limit, limit, size=shape)
        else:
            raise ValueError(kind)
        out[name] = np.ascontiguousarray(v, dtype=dtype)
    return out
