"""Generate tests/golden/*.npz from the fp64 oracle (run in the build container: `python tests/golden/make_golden.py`).

PARITY UNPINNED: TensorFlow (the reference's arithmetic) cannot run here, so these fixtures pin the ORACLE against
regressions and give the GPU tests committed vectors that travel to the GPU box; they are not outputs of the
reference itself.  See oracle/__init__.py.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import savp as OS  # noqa: E402
from video_prediction_amd import variables as V  # noqa: E402
from video_prediction_amd.hparams import HParams  # noqa: E402
from video_prediction_amd.models.hparam_defaults import savp_defaults  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def golden_generator(nz, name, T=4, B=1, H=32, W=32, C=3):
    hp = HParams(**savp_defaults())
    hp.override_from_dict(dict(context_frames=2, sequence_length=T, nz=nz))
    specs = V.variable_specs(hp, (H, W, C), mode='test')
    vals = V.init_variables(specs, seed=4)
    rng = np.random.default_rng(0)
    images = rng.random((T, B, H, W, C))
    noise = {}
    if nz:
        r = np.random.default_rng(1)
        noise = {'eps': torch.tensor(r.standard_normal((T - 1, B, nz))), 'prior': torch.tensor(r.standard_normal((T - 2, B, nz)))}
    P = {k: torch.tensor(v, dtype=torch.float64) for k, v in vals.items()}
    with torch.no_grad():
        out = OS.generator_fn(OS.Scope(P).sub('generator'), {'images': torch.tensor(images)}, 'test', hp, noise)
    save = {'images': images.astype(np.float32), 'gen_images': out['gen_images'].numpy(),
            'masks_argmax': out['masks'].squeeze(-2).argmax(-1).numpy().astype(np.int8)}
    top2 = out['masks'].squeeze(-2).topk(2, dim=-1).values
    save['masks_margin'] = (top2[..., 0] - top2[..., 1]).numpy().astype(np.float32)     # argmax is only asked where this is > 1e-5
    if nz:
        save['gen_images_enc'] = out['gen_images_enc'].numpy()
        save['eps'] = noise['eps'].numpy()
        save['prior'] = noise['prior'].numpy()
        save['zs_mu_enc'] = out['zs_mu_enc'].numpy()
    np.savez_compressed(os.path.join(HERE, name), **save)
    print(name, {k: v.shape for k, v in save.items()})


def golden_conditioned(name='gen_cond_32x32.npz', T=5, B=2, H=32, W=32, C=3, nz=8, cond=(4, 3)):
    """The action / state-conditioned cell (savp_model.py:24-26,411-444,655-661) at BAIR's use_state widths, scheduled sampling on (the state
    follows the image's ground-truth schedule): generated frames of both unrolls, predicted states, posterior means."""
    hp = HParams(**savp_defaults())
    hp.override_from_dict(dict(context_frames=2, sequence_length=T, nz=nz, schedule_sampling='inverse_sigmoid'))
    specs = V.variable_specs(hp, (H, W, C), mode='test', cond=cond)
    vals = V.init_variables(specs, seed=4)
    for k in vals:                      # state_pred starts near zero (truncated normal 0.02): scaled so that the predicted states matter
        if 'state_pred' in k and k.endswith('kernel'):
            vals[k] = (vals[k] * 30).astype(np.float32)
    rng = np.random.default_rng(0)
    images = rng.random((T, B, H, W, C))
    actions = rng.standard_normal((T - 1, B, cond[0]))
    states = np.cumsum(0.3 * rng.standard_normal((T, B, cond[1])), axis=0)
    r = np.random.default_rng(1)
    noise = {'eps': torch.tensor(r.standard_normal((T - 1, B, nz))), 'prior': torch.tensor(r.standard_normal((T - 2, B, nz))),
             'ground_truth_sampling': torch.tensor(r.random((T - 3, B)) < 0.5), 'ground_truth_sampling_enc': torch.tensor(r.random((T - 3, B)) < 0.5)}
    P = {k: torch.tensor(v, dtype=torch.float64) for k, v in vals.items()}
    with torch.no_grad():
        out = OS.generator_fn(OS.Scope(P).sub('generator'), {'images': torch.tensor(images), 'actions': torch.tensor(actions),
                                                             'states': torch.tensor(states)}, 'train', hp, noise)
    save = {'images': images.astype(np.float32), 'actions': actions.astype(np.float32), 'states': states.astype(np.float32),
            'eps': noise['eps'].numpy(), 'prior': noise['prior'].numpy(), 'gts': noise['ground_truth_sampling'].numpy(),
            'gts_enc': noise['ground_truth_sampling_enc'].numpy()}
    for k in ('gen_images', 'gen_images_enc', 'gen_states', 'gen_states_enc', 'zs_mu_enc'):
        save[k] = out[k].numpy()
    np.savez_compressed(os.path.join(HERE, name), **save)
    print(name, {k: v.shape for k, v in save.items()})


if __name__ == '__main__':
    golden_generator(0, 'gen_det_32x32.npz')
    golden_generator(8, 'gen_savp_32x32.npz')
    golden_conditioned()
