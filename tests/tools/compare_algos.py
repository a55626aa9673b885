"""Train-step losses for the first steps with the LDS patch kernels vs the generic implicit-GEMM kernels (same seeds).
usage: ALGO=generic|auto python tests/tools/compare_algos.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests.gpu_model_checks import make_hparams
from video_prediction_amd.models.savp_model import SAVPEngine
from video_prediction_amd import kernels as K

def main():
    K.set_conv_precision('bf16')
    if os.environ.get('ALGO', 'auto') == 'generic':
        orig = K.conv
        def conv(mode, geom, x, y, w, *a, **kw):
            kw['tile'] = kw.get('tile', 0) | 0x100
            return orig(mode, geom, x, y, w, *a, **kw)
        K.conv = conv
        import video_prediction_amd.engine as E
        if hasattr(E, 'K'):
            E.K.conv = conv
    B, T = 16, 30
    hp = make_hparams(context_frames=2, sequence_length=T, batch_size=B, lr=2e-4, beta1=0.5, beta2=0.999, l1_weight=100.0,
                      l2_weight=0.0, kl_weight=1.0, video_sn_vae_gan_weight=0.1, video_sn_gan_weight=0.1,
                      vae_gan_feature_cdist_weight=10.0, gan_feature_cdist_weight=0.0)
    torch.manual_seed(0)
    eng = SAVPEngine(hp, (64, 64, 3), B, mode='train')
    g = torch.Generator(device='cuda:0'); g.manual_seed(1)
    images = torch.rand(T, B, 64, 64, 3, device='cuda:0', generator=g)
    eng.set_images(images, time_major=True)
    for i in range(6):
        info = eng.train_step()
        print('step %d d_loss %.5f g_loss %.5f' % (i, float(info['d_loss']), float(info['g_loss'])), flush=True)

if __name__ == '__main__':
    main()
