"""A/B numerics of an environment switch on the bf16 datapath, GPU only (no oracle): the same train step (same variables, images,
noise) in two processes, one per switch setting; `dump` writes losses + all gradients, `cmp` prints per-group relative L2 of B vs A.
  python tests/tools/ab_switch_check.py dump out_a.pt ;  SAVP_X=1 python tests/tools/ab_switch_check.py dump out_b.pt
  python tests/tools/ab_switch_check.py cmp out_a.pt out_b.pt [tol]        (exit 1 above tol, default 2e-2)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def dump(path):
    from tests import gpu_model_checks as G
    from video_prediction_amd import kernels as K
    from video_prediction_amd.models.savp_model import SAVPEngine
    K.set_conv_precision(os.environ.get('PREC', 'bf16'))
    B, T = int(os.environ.get('B', 2)), int(os.environ.get('T', 8))
    hp = G.make_hparams(context_frames=2, sequence_length=T, clip_length=4, nz=8, lr=2e-4, beta1=0.5, beta2=0.999, l1_weight=100.0,
                        kl_weight=1.0, kl_anneal='none', video_sn_vae_gan_weight=0.1, video_sn_gan_weight=0.1,
                        vae_gan_feature_cdist_weight=10.0, gan_feature_cdist_weight=0.0)
    images = G.synth(hp, B, 64, 64, 3, 0).float()
    noise = G.make_noise(hp, B, seed=100, sampling=True)
    eng = SAVPEngine(hp, (64, 64, 3), B, mode='train', seed=4, device='cuda:0')
    eng.set_images(images.cuda(), time_major=True)
    info = eng.train_step(noise, return_grads=True)
    torch.cuda.synchronize()
    out = {'d_loss': float(info['d_loss']), 'g_loss': float(info['g_loss']),
           'gen': eng.gen.gen.v.float().cpu(),
           'd_grads': {k: v.cpu() for k, v in info['d_grads'].items()}, 'g_grads': {k: v.cpu() for k, v in info['g_grads'].items()}}
    torch.save(out, path)
    print('dumped', path, out['d_loss'], out['g_loss'])


def cmp(a, b, tol):
    A, Bv = torch.load(a), torch.load(b)
    bad = 0
    print('d_loss %.6g vs %.6g ; g_loss %.6g vs %.6g ; gen max abs diff %.3g' % (A['d_loss'], Bv['d_loss'], A['g_loss'], Bv['g_loss'],
                                                                                  float((A['gen'] - Bv['gen']).abs().max())))
    for key in ('d_grads', 'g_grads'):
        worst, wn = 0.0, ''
        num = den = 0.0
        for n, ga in A[key].items():
            gb = Bv[key][n]
            e = float((gb.double() - ga.double()).norm() / max(float(ga.double().norm()), 1e-30))
            num += float((gb.double() - ga.double()).pow(2).sum())
            den += float(ga.double().pow(2).sum())
            if e > worst and float(ga.abs().max()) > 0:
                worst, wn = e, n
        tot = (num / max(den, 1e-60)) ** 0.5
        print('%s: whole-group rel L2 %.3e ; worst variable %.3e (%s)' % (key, tot, worst, wn))
        bad += tot > tol
    return 1 if bad else 0


if __name__ == '__main__':
    if sys.argv[1] == 'dump':
        dump(sys.argv[2])
    else:
        sys.exit(cmp(sys.argv[2], sys.argv[3], float(sys.argv[4]) if len(sys.argv) > 4 else 2e-2))
