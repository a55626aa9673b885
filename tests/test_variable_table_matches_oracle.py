"""The variable table (video_prediction_amd/variables.py: names, shapes and initialisers in the reference's TensorFlow naming) and the oracle
(oracle/savp.py, which looks variables up by those names) are written separately; the product's ParamStore is built from the table and its
parity is judged against the oracle.  For every option combination below: the oracle's generator_fn runs on exactly the table's variables
(no missing name, no shape it cannot use) and READS every generator variable of the table (none is dead weight the product would carry,
train, save and all-reduce for nothing).  Covers options the HIP path still refuses (learn_initial_state, conv_rnn_norm_layer='none', rnn='gru', the two rnn ablations)."""
import numpy as np
import pytest
import torch

from oracle import savp as OS
from video_prediction_amd import variables as V
from video_prediction_amd.hparams import HParams
from video_prediction_amd.models.hparam_defaults import savp_defaults

CASES = {
    'savp': dict(nz=4),
    'deterministic': dict(nz=0, schedule_sampling='none'),
    'gru': dict(nz=4, conv_rnn='gru'),
    'flow': dict(nz=4, transformation='flow'),
    'dna': dict(nz=0, transformation='dna'),
    'where_add_input': dict(nz=4, where_add='input'),
    'where_add_middle': dict(nz=4, where_add='middle'),
    'no_scratch_independent_mask': dict(nz=4, generate_scratch_image=False, dependent_mask=False),
    'learn_prior_e_rnn': dict(nz=4, learn_prior=True, use_e_rnn=True, nef=8),
    'untiled_latent': dict(nz=4, use_tile_concat=False),
    'no_rnn_z': dict(nz=4, use_rnn_z=False),
    'learn_initial_state': dict(nz=4, learn_initial_state=True),
    'learn_initial_state_gru': dict(nz=4, learn_initial_state=True, conv_rnn='gru'),
    'rnn_gru': dict(nz=4, rnn='gru', use_e_rnn=True, learn_prior=True, nef=8),
    'rnn_gru_learn_initial_state': dict(nz=4, rnn='gru', learn_initial_state=True),
    'ablation_rnn': dict(nz=4, ablation_rnn=True),
    'ablation_rnn_untiled_learn_initial_state': dict(nz=4, ablation_rnn=True, use_tile_concat=False, learn_initial_state=True),
    'ablation_conv_rnn_norm': dict(nz=4, ablation_conv_rnn_norm=True),
    'ablation_conv_rnn_norm_gru_untiled': dict(nz=4, ablation_conv_rnn_norm=True, conv_rnn='gru', use_tile_concat=False),
    'conv_rnn_norm_none': dict(nz=4, conv_rnn_norm_layer='none'),
    'conv_rnn_norm_none_untiled_gru': dict(nz=4, conv_rnn_norm_layer='none', use_tile_concat=False, conv_rnn='gru'),
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_oracle_reads_exactly_the_generator_variables_of_the_table(name):
    over = dict(context_frames=2, sequence_length=4, ngf=8)
    over.update(CASES[name])
    hp = HParams(**savp_defaults())
    hp.override_from_dict(over)
    H = W = 64
    specs = V.variable_specs(hp, (H, W, 3), mode='test')
    vals = V.init_variables(specs, seed=3)
    rng = np.random.default_rng(11)
    P = {}
    for k, v in vals.items():
        v = np.asarray(v, dtype=np.float64)
        if float(np.abs(v).max()) == 0.0:                          # zero-initialised biases / states: make them matter
            v = 0.1 * rng.standard_normal(v.shape)
        P[k] = torch.tensor(v, requires_grad=True)
    B, T = 1, hp.sequence_length
    images = torch.tensor(rng.random((T, B, H, W, 3)))
    noise = {}
    if hp.nz:
        noise['eps'] = torch.tensor(rng.standard_normal((T - 1, B, hp.nz)))
        noise['prior'] = torch.tensor(rng.standard_normal((T - hp.context_frames, B, hp.nz)))
        if hp.learn_prior:
            noise['prior_eps'] = torch.tensor(rng.standard_normal((T - 1, B, hp.nz)))
    out = OS.generator_fn(OS.Scope(P).sub('generator'), {'images': images}, 'train', hp, noise)
    total = 0.0
    for k, v in out.items():
        if torch.is_tensor(v) and v.is_floating_point() and v.requires_grad:
            total = total + (v * torch.tensor(rng.standard_normal(tuple(v.shape)))).sum()
    names = [k for k in P if k.startswith('generator/')]
    grads = torch.autograd.grad(total, [P[k] for k in names], allow_unused=True)
    unread = [k for k, g in zip(names, grads) if g is None]
    assert not unread, (name, unread)
    assert all(torch.isfinite(g).all() for g in grads)
