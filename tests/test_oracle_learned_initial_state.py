"""learn_initial_state in the oracle and the variable table (reference savp_model.py:295-307,344-352): the conv-RNN states and the rnn_z
state start from variables `generator/initial_state_<i>/initial_state` tiled over the batch instead of zeros.  The HIP path does not cover
the option yet (it raises); these tests pin the restatement the next implementation will be checked against."""
import numpy as np
import torch

from oracle import savp as OS
from video_prediction_amd import variables as V
from video_prediction_amd.hparams import HParams
from video_prediction_amd.models.hparam_defaults import savp_defaults


def _setup(learn, conv_rnn='lstm', nz=4):
    hp = HParams(**savp_defaults())
    hp.override_from_dict(dict(context_frames=2, sequence_length=4, nz=nz, ngf=8, learn_initial_state=learn, conv_rnn=conv_rnn,
                               schedule_sampling='none'))
    specs = V.variable_specs(hp, (64, 64, 3), mode='test')
    vals = V.init_variables(specs, seed=3)
    P = {k: torch.tensor(np.asarray(v), dtype=torch.float64) for k, v in vals.items()}
    rng = np.random.default_rng(5)
    images = torch.tensor(rng.random((4, 2, 64, 64, 3)))
    zs = torch.tensor(rng.standard_normal((3, 2, nz))) if nz else None
    return hp, specs, P, images, zs


def _gen(hp, P, images, zs):
    inputs = {'images': images}
    if zs is not None:
        inputs['zs'] = zs
    return OS.generator_given_z_fn(OS.Scope(P).sub('generator'), inputs, 'test', hp)['gen_images']


def test_variable_table_follows_nest_flatten_order():
    hp, specs, _, _, _ = _setup(True)
    names = sorted((k for k in specs if '/initial_state_' in k), key=lambda k: int(k.split('initial_state_')[1].split('/')[0]))
    shapes = [specs[k][0] for k in names]
    # 64x64, ngf = 8: conv-RNN layers at 32x32x8, 16x16x16, 8x8x32, 16x16x16, 32x32x8 -- (c, h) each -- then the rnn_z (c, h)
    want = []
    for (h, f) in ((32, 8), (16, 16), (8, 32), (16, 16), (32, 8)):
        want += [(h, h, f), (h, h, f)]
    want += [(4,), (4,)]
    assert shapes == want and all(specs[k][1] == 'zeros' for k in names)
    assert all(k.startswith('generator/initial_state_') for k in names)
    hp_g, specs_g, _, _, _ = _setup(True, conv_rnn='gru', nz=0)
    assert len([k for k in specs_g if '/initial_state_' in k]) == 5           # one state per GRU layer, no rnn_z without a latent
    hp_off, specs_off, _, _, _ = _setup(False)
    assert not [k for k in specs_off if '/initial_state_' in k]


def test_zero_valued_learned_states_are_the_zero_state_and_nonzero_ones_are_used():
    hp_on, _, P_on, images, zs = _setup(True)
    hp_off, _, P_off, _, _ = _setup(False)
    assert set(P_off) == set(k for k in P_on if '/initial_state_' not in k)
    with torch.no_grad():
        a = _gen(hp_off, P_off, images, zs)
        b = _gen(hp_on, P_on, images, zs)                      # initial-state variables are zero-initialised
        assert float((a - b).abs().max()) == 0.0
        P2 = dict(P_on)
        rng = np.random.default_rng(7)
        for k in P_on:
            if '/initial_state_' in k:
                P2[k] = torch.tensor(0.5 * rng.standard_normal(tuple(P_on[k].shape)))
        c = _gen(hp_on, P2, images, zs)
    assert float((c - a).abs().max()) > 1e-4
    # every state variable reaches the output: c and h of each conv-RNN layer and of the rnn_z cell
    leaves = {k: P2[k].clone().requires_grad_(True) for k in P2 if '/initial_state_' in k}
    Pg = dict(P2)
    Pg.update(leaves)
    out = _gen(hp_on, Pg, images, zs)
    grads = torch.autograd.grad((out - images[1:]).abs().mean(), list(leaves.values()))
    assert all(float(g.abs().max()) > 0.0 for g in grads)
    # the state is tiled over the batch: both samples start from the same tensor
    st = OS.savp_cell_zero_state(images[:3], hp_on, zs, OS.Scope(P2).sub('generator'))
    c0, h0 = st['conv_rnn_states'][0]
    assert c0.shape[0] == 2 and float((c0[0] - c0[1]).abs().max()) == 0.0
    assert float((c0[0] - P2['generator/initial_state_0/initial_state']).abs().max()) == 0.0
    assert float((h0[0] - P2['generator/initial_state_1/initial_state']).abs().max()) == 0.0
    zc, zh = st['rnn_z_state']
    assert float((zc[1] - P2['generator/initial_state_10/initial_state']).abs().max()) == 0.0
    assert float((zh[1] - P2['generator/initial_state_11/initial_state']).abs().max()) == 0.0
