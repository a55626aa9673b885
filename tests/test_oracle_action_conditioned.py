"""The action / state-conditioned SAVP cell in the oracle and the variable table (reference savp_model.py:24-26, 411-444, 655-661,
684-685; base_model.py:758-762): actions and the scheduled robot state join the latent in every tile-concatenated slice, the next state is
a dense layer of [actions | state], its l2 loss against the true states trains that layer only (the tiled copy is under stop_gradient).
CPU tests of the restatement the HIP path is checked against (tests/gpu_model_checks.py::check_action_conditioned)."""
import numpy as np
import pytest
import torch

from oracle import savp as OS
from oracle import train as OT
from video_prediction_amd import variables as V
from video_prediction_amd.hparams import HParams
from video_prediction_amd.models.hparam_defaults import savp_defaults


def _setup(cond, nz=4, T=4, **over):
    hp = HParams(**savp_defaults())
    hpd = dict(context_frames=2, sequence_length=T, nz=nz, ngf=8, nef=8, ndf=8, schedule_sampling='none', clip_length=2)
    hpd.update(over)
    hp.override_from_dict(hpd)
    rng = np.random.default_rng(5)
    B = 2
    inputs = {'images': torch.tensor(rng.random((T, B, 64, 64, 3)))}
    if cond[0]:
        inputs['actions'] = torch.tensor(rng.standard_normal((T - 1, B, cond[0])))
    if cond[1]:
        inputs['states'] = torch.tensor(rng.standard_normal((T, B, cond[1])))
    return hp, inputs


def _params(hp, cond, mode='test', seed=3):
    specs = V.variable_specs(hp, (64, 64, 3), mode=mode, cond=cond)
    vals = V.init_variables(specs, seed=seed)
    return specs, {k: torch.tensor(np.asarray(v), dtype=torch.float64) for k, v in vals.items()}


def test_variable_table_widens_every_tiled_slice_and_adds_state_pred():
    hp, _ = _setup((4, 3))
    base = V.variable_specs(hp, (64, 64, 3), mode='test')
    specs = V.variable_specs(hp, (64, 64, 3), mode='test', cond=(4, 3))
    p = 'generator/rnn/savp_cell/'
    assert specs[p + 'state_pred/dense/kernel'][0] == (7, 3) and specs[p + 'state_pred/dense/bias'][0] == (3,)
    assert p + 'state_pred/dense/kernel' not in base
    widened = 0
    for k, (shape, _) in specs.items():
        if k.startswith(p + 'state_pred'):
            continue
        b = base[k][0]
        if b != shape:
            # only input-channel counts change: by the 7 conditioning columns in the cell (where_add = 'all': every conv), by the 4
            # actions in the encoder's first convolution
            assert len(shape) == 4 and shape[:2] == b[:2] and shape[3] == b[3], (k, b, shape)
            assert shape[2] - b[2] == (4 if k.startswith('generator/encoder/') else 7), (k, b, shape)
            widened += 1
    assert widened == 1 + 6 + 5                       # encoder layer_1; 6 down / upsample convs; 5 ConvLSTM kernels
    # actions alone on the deterministic model: the tiled slices exist although there is no latent
    hp0, _ = _setup((4, 0), nz=0)
    s0 = V.variable_specs(hp0, (64, 64, 3), mode='test', cond=(4, 0))
    assert s0[p + 'h0/conv_pool2d/kernel'][0] == (5, 5, 6 + 4, 8) and not [k for k in s0 if 'state_pred' in k]


def test_state_recurrence_follows_the_ground_truth_schedule():
    hp, inputs = _setup((2, 3), T=6, schedule_sampling='inverse_sigmoid')
    _, P = _params(hp, (2, 3))
    T1, B = 5, 2
    rng = np.random.default_rng(2)
    zs = torch.tensor(rng.standard_normal((T1, B, hp.nz)))
    gts = torch.tensor(rng.random((T1 - 2, B)) < 0.5)
    with torch.no_grad():
        out = OS.generator_given_z_fn(OS.Scope(P).sub('generator'), dict(inputs, zs=zs), 'train', hp, gts)
    W = P['generator/rnn/savp_cell/state_pred/dense/kernel'].numpy()
    b = P['generator/rnn/savp_cell/state_pred/dense/bias'].numpy()
    gt = np.concatenate([np.ones((2, B), bool), gts.numpy()], 0)
    a, s = inputs['actions'].numpy(), inputs['states'].numpy()
    prev = np.zeros((B, 3))
    for t in range(T1):
        state = np.where(gt[t][:, None], s[t], prev)
        prev = np.concatenate([a[t], state], -1) @ W + b
        assert np.allclose(out['gen_states'][t].numpy(), prev, atol=1e-12)


def test_tiled_conditioning_is_under_stop_gradient_and_the_state_loss_trains_state_pred_only():
    hp, inputs = _setup((2, 2), state_weight=1.0, l1_weight=0.0, l2_weight=0.0, kl_weight=0.0)
    _, P = _params(hp, (2, 2), mode='train')
    rng = np.random.default_rng(1)
    noise = {'eps': torch.tensor(rng.standard_normal((3, 2, hp.nz))), 'prior': torch.tensor(rng.standard_normal((2, 2, hp.nz)))}
    _, _, info = OT.train_step(P, OT.init_opt_state(P), inputs, hp, noise, None, None, step=0)
    assert list(info['g_losses']) == ['gen_state_loss']
    nonzero = sorted(k for k, g in info['g_grads'].items() if float(g.abs().max()) > 0)
    assert nonzero == ['generator/rnn/savp_cell/state_pred/dense/bias', 'generator/rnn/savp_cell/state_pred/dense/kernel']
    # and the image losses reach every cell convolution through the action columns as well (they are inputs, not constants)
    hp2, _ = _setup((2, 2), l1_weight=1.0, kl_weight=0.0)
    _, _, info2 = OT.train_step(P, OT.init_opt_state(P), inputs, hp2, noise, None, None, step=0)
    assert 'generator/rnn/savp_cell/state_pred/dense/kernel' not in info2['g_grads']       # state_weight = 0: no gradient, Adam skips it
    k0 = info2['g_grads']['generator/rnn/savp_cell/h0/conv_pool2d/kernel']                 # [5, 5, 6 + 2 + 2 + nz, 8]
    assert float(k0[:, :, 6:10].abs().max()) > 0


def test_prior_fn_with_actions_fails_like_the_reference_unless_the_context_covers_the_sequence():
    hp, inputs = _setup((2, 0), learn_prior=True)
    _, P = _params(hp, (2, 0))
    with pytest.raises(AssertionError):
        OS.prior_fn(OS.Scope(P).sub('generator').sub('prior'), inputs, hp)
