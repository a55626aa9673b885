"""Debug helper: per-variable gradient errors of one train step, HIP vs fp64 oracle, with the fp32 oracle as a yardstick."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests import gpu_model_checks as G
from oracle import train as OT
from video_prediction_amd import variables as V
from video_prediction_amd.models.savp_model import SAVPEngine

def main():
    B, T, H, W, C = 2, 6, 64, 64, 3
    hp = G.make_hparams(context_frames=2, sequence_length=T, clip_length=4, nz=8, lr=2e-4, beta1=0.5, beta2=0.999,
               l1_weight=100.0, l2_weight=0.0, kl_weight=1.0, kl_anneal='none', video_sn_vae_gan_weight=0.1,
               video_sn_gan_weight=0.1, vae_gan_feature_cdist_weight=10.0, gan_feature_cdist_weight=0.0)
    specs = V.variable_specs(hp, (H, W, C), mode='train')
    vals = V.init_variables(specs, seed=4)
    rng = np.random.default_rng(9)
    for k in vals:
        if k.endswith('gamma'): vals[k] = (1 + 0.2 * rng.standard_normal(vals[k].shape)).astype(np.float32)
        elif k.endswith('beta') or k.endswith('bias'): vals[k] = (0.1 * rng.standard_normal(vals[k].shape)).astype(np.float32)
        elif k.endswith('kernel') and k.startswith('generator'): vals[k] = (vals[k] * 3).astype(np.float32)
    images = G.synth(hp, B, H, W, C, 0)
    noise = G.make_noise(hp, B, seed=100, sampling=True)
    res = {}
    for dt in (torch.float64, torch.float32):
        P = {k: torch.tensor(v, dtype=dt) for k, v in vals.items()}
        st = OT.init_opt_state(P)
        nz = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in noise.items()}
        _, _, info = OT.train_step(P, st, {'images': images.to(dt)}, hp, nz, noise['d_indices_pre'], noise['d_indices_post'], step=0)
        res[dt] = info
    eng = SAVPEngine(hp, (H, W, C), B, mode='train', values=vals, device='cuda:0')
    eng.set_images(images.float().to('cuda:0'), time_major=True)
    info = eng.train_step(noise, return_grads=True)
    torch.cuda.synchronize()
    for key in ('d_grads', 'g_grads'):
        rows = []
        gmax = max(float(v.abs().max()) for v in res[torch.float64][key].values())
        for name, ref in res[torch.float64][key].items():
            got = info[key][name].double().cpu()
            r32 = res[torch.float32][key][name].double()
            scale = max(float(ref.abs().max()), 1e-12)
            rows.append((float((got - ref).abs().max()) / scale, float((r32 - ref).abs().max()) / scale, scale, name))
        rows.sort(reverse=True)
        print('==', key, 'global max |grad| = %.3e' % gmax)
        for e, e32, sc, name in rows[:14]:
            print('  hip_err=%.2e  cpu_fp32_err=%.2e  max|ref|=%.2e  %s' % (e, e32, sc, name))

if __name__ == '__main__':
    main()
