"""Quick wall-clock of the full SAVP train step on one GPU (used while optimising)."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from tests.gpu_model_checks import make_hparams
from video_prediction_amd.models.savp_model import SAVPEngine

def main():
    from video_prediction_amd import kernels as K
    K.set_conv_precision(os.environ.get('PREC', 'f32'))
    K.enable_autotune(os.environ.get('TUNE', '1') == '1')
    if os.environ.get('LOAD_TUNE'):
        print('loaded %d tuned problems' % K.load_tuning(os.environ['LOAD_TUNE']))
    B = int(os.environ.get('B', 16)); T = int(os.environ.get('T', 30)); steps = int(os.environ.get('STEPS', 3))
    hp = make_hparams(context_frames=2, sequence_length=T, batch_size=B, lr=2e-4, beta1=0.5, beta2=0.999, l1_weight=100.0,
                      l2_weight=0.0, kl_weight=1.0, video_sn_vae_gan_weight=0.1, video_sn_gan_weight=0.1,
                      vae_gan_feature_cdist_weight=10.0, gan_feature_cdist_weight=0.0)
    eng = SAVPEngine(hp, (64, 64, 3), B, mode='train')
    images = torch.rand(T, B, 64, 64, 3, device='cuda:0')
    eng.set_images(images, time_major=True)
    for _ in range(2):
        eng.train_step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        info = eng.train_step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    if os.environ.get('SAVE_TUNE'):
        K.save_tuning(os.environ['SAVE_TUNE'])
    if os.environ.get('TUNELOG'):
        for key, cfg in K.AUTOTUNE['log']:
            print('tuned', key[:1], key[2:11], '->', hex(cfg[0]), cfg[1])
    print('step %.1f ms  -> %.1f frames/s ; d_loss %.4f g_loss %.4f ; mem %.1f GB' % (dt * 1e3, B * T / dt, float(info['d_loss']), float(info['g_loss']), torch.cuda.max_memory_allocated() / 2**30))

if __name__ == '__main__':
    main()
