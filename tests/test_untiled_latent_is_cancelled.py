"""CPU pin of the identity the HIP path uses for use_tile_concat=False (models/savp_cell.py).

Without tile_concat the latent enters every layer as dense(z)[:, None, None, :] added to a convolution's output (reference
savp_model.py:983-993 `_maybe_tile_concat_layer`, rnn_ops.py:128-135,145-146) and that sum is followed directly by an instance norm
(savp_model.py:461-463,497-499; rnn_ops.py:148-149): a per-(sample, channel) constant, removed exactly by the norm's mean subtraction.
The oracle computes the sums literally; here its outputs are shown not to depend on z or on the dense weights, and the gradients of those
weights and of z to vanish -- which is why the product runs such a model without z channels and leaves those gradients at zero."""
import numpy as np
import torch

from oracle import savp as OS
from video_prediction_amd import variables as V
from video_prediction_amd.hparams import HParams
from video_prediction_amd.models.hparam_defaults import savp_defaults


def _is_latent_projection(k):
    """`h<i>/dense/kernel` of a down / upsample layer, `weights` of a conv-RNN (not the CDNA head's or the encoder's dense layers)"""
    return (k.endswith('dense/kernel') and '/savp_cell/h' in k) or k.endswith('weights')


def _setup(conv_rnn, where_add):
    hp = HParams(**savp_defaults())
    hp.override_from_dict(dict(context_frames=2, sequence_length=4, nz=4, ngf=8, use_tile_concat=False, conv_rnn=conv_rnn,
                               where_add=where_add, schedule_sampling='none'))
    H = W = 64
    specs = V.variable_specs(hp, (H, W, 3), mode='test')
    vals = V.init_variables(specs, seed=3)
    rng = np.random.default_rng(5)
    P = {}
    for k, v in vals.items():
        if _is_latent_projection(k):
            v = rng.standard_normal(v.shape)             # large on purpose: the cancellation must not rest on small weights
        elif k.endswith('kernel'):
            v = v * 3
        P[k] = torch.tensor(np.asarray(v), dtype=torch.float64)
    images = torch.tensor(rng.random((4, 1, H, W, 3)))
    return hp, P, images


def _gen(hp, P, images, zs):
    vs = OS.Scope(P).sub('generator')
    return OS.generator_given_z_fn(vs, {'images': images, 'zs': zs}, 'test', hp)['gen_images']


def test_outputs_do_not_depend_on_z_or_on_the_dense_weights():
    for conv_rnn, where_add in (('lstm', 'all'), ('gru', 'all'), ('lstm', 'input'), ('lstm', 'middle')):
        hp, P, images = _setup(conv_rnn, where_add)
        names = [k for k in P if _is_latent_projection(k)]
        assert names, 'no un-tiled latent projection in the variable table for %s / %s' % (conv_rnn, where_add)
        rng = np.random.default_rng(1)
        z1 = torch.tensor(rng.standard_normal((3, 1, hp.nz)))
        z2 = torch.tensor(3.0 * rng.standard_normal((3, 1, hp.nz)))
        with torch.no_grad():
            a = _gen(hp, P, images, z1)
            b = _gen(hp, P, images, z2)
            P2 = dict(P)
            for k in names:
                P2[k] = torch.zeros_like(P[k])
            c = _gen(hp, P2, images, z1)
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 1e-9 * scale, (conv_rnn, where_add, float((a - b).abs().max()))
        assert float((a - c).abs().max()) <= 1e-9 * scale, (conv_rnn, where_add, float((a - c).abs().max()))


def test_gradients_of_the_dense_weights_and_of_z_vanish():
    hp, P, images = _setup('lstm', 'all')
    names = [k for k in P if _is_latent_projection(k)]
    leaves = {k: P[k].clone().requires_grad_(True) for k in names}
    ref_name = 'generator/rnn/savp_cell/h0/conv_pool2d/kernel'
    leaves[ref_name] = P[ref_name].clone().requires_grad_(True)
    z = torch.tensor(np.random.default_rng(2).standard_normal((3, 1, hp.nz)), requires_grad=True)
    Pg = dict(P)
    Pg.update(leaves)
    out = _gen(hp, Pg, images, z)
    loss = (out - images[1:]).abs().mean()
    grads = torch.autograd.grad(loss, list(leaves.values()) + [z])
    gref = float(grads[len(names)].abs().max())
    assert gref > 0
    for k, g in zip(names, grads[:len(names)]):
        assert float(g.abs().max()) <= 1e-9 * gref, (k, float(g.abs().max()), gref)
    assert float(grads[-1].abs().max()) <= 1e-9 * gref
