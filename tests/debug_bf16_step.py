"""Developer aid: bf16-mode vs fp32-mode train step on the GPU (no oracle): prints every loss and the worst per-variable gradient
deviations.  python tests/debug_bf16_step.py [T] [B] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests.gpu_model_checks import make_hparams, synth, make_noise, _l2rel  # noqa: E402
from video_prediction_amd import kernels as K, variables as V  # noqa: E402
from video_prediction_amd.models.savp_model import SAVPEngine  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
hp = make_hparams(context_frames=2, sequence_length=T, clip_length=10, nz=8, lr=2e-4, beta1=0.5, beta2=0.999, l1_weight=100.0,
                  l2_weight=0.0, kl_weight=1.0, kl_anneal='none', video_sn_vae_gan_weight=0.1, video_sn_gan_weight=0.1,
                  vae_gan_feature_cdist_weight=10.0, gan_feature_cdist_weight=0.0)
specs = V.variable_specs(hp, (64, 64, 3), mode='train')
vals = V.init_variables(specs, seed=4)
rng = np.random.default_rng(9)
for k in vals:
    if k.endswith('gamma'):
        vals[k] = (1 + 0.2 * rng.standard_normal(vals[k].shape)).astype(np.float32)
    elif k.endswith('beta') or k.endswith('bias'):
        vals[k] = (0.1 * rng.standard_normal(vals[k].shape)).astype(np.float32)
    elif k.endswith('kernel') and k.startswith('generator'):
        vals[k] = (vals[k] * 3).astype(np.float32)
images = synth(hp, B, 64, 64, 3, 0)
res = {}
for prec in ('f32', 'bf16'):
    K.set_conv_precision(prec)
    eng = SAVPEngine(hp, (64, 64, 3), B, mode='train', values=vals)
    eng.use_graph = False
    eng.set_images(images.float().cuda(), time_major=True)
    for it in range(steps):
        noise = make_noise(hp, B, seed=100 + it, sampling=True)
        info = eng.train_step(noise, return_grads=True)
    torch.cuda.synchronize()
    res[prec] = dict(d_loss=float(info['d_loss']), g_loss=float(info['g_loss']),
                     g={k: float(l) for k, (l, w) in info['g_losses'].items()}, d={k: float(l) for k, (l, w) in info['d_losses'].items()},
                     dg={k: v.cpu() for k, v in info['d_grads'].items()}, gg={k: v.cpu() for k, v in info['g_grads'].items()},
                     logits={d['name']: d['D'].logits.cpu().clone() for d in eng.discs})
a, b = res['f32'], res['bf16']
print('d_loss', a['d_loss'], b['d_loss'], ' g_loss', a['g_loss'], b['g_loss'])
for k in a['d']:
    print('  %-40s %.6f %.6f' % (k, a['d'][k], b['d'][k]))
for k in a['g']:
    print('  %-40s %.6f %.6f' % (k, a['g'][k], b['g'][k]))
for nm in a['logits']:
    print('logits', nm, a['logits'][nm].flatten()[:8].tolist(), b['logits'][nm].flatten()[:8].tolist())
for key in ('dg', 'gg'):
    errs = sorted(((_l2rel(b[key][n], a[key][n]), n, float(a[key][n].norm())) for n in a[key]), reverse=True)
    print(key, 'worst:')
    for e, n, nr in errs[:12]:
        print('   %.4f  %-70s |g|=%.3e' % (e, n, nr))
