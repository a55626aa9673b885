"""Model-level GPU parity: the HIP engine (generator unroll, encoder, discriminators, losses, Adam) vs the fp64 oracle
on identical variables, inputs and injected noise."""
import numpy as np
import torch

from oracle import savp as OS
from oracle import train as OT
from video_prediction_amd import variables as V
from video_prediction_amd.hparams import HParams
from video_prediction_amd.models.hparam_defaults import savp_defaults
from video_prediction_amd.models.savp_model import SAVPEngine

DEV = 'cuda:0'


def make_hparams(**over):
    hp = HParams(**savp_defaults())
    hp.override_from_dict(over)
    return hp


def rel(got, ref):
    got = torch.as_tensor(np.asarray(got.detach().cpu() if torch.is_tensor(got) else got)).double()
    ref = torch.as_tensor(np.asarray(ref.detach().cpu() if torch.is_tensor(ref) else ref)).double()
    return float((got - ref).abs().max() / max(float(ref.abs().max()), 1e-30))


def synth(hp, B, H, W, C, seed=0, smooth=True):
    """Seeded synthetic video in [0,1] (temporally smooth so that CDNA / masks are exercised off-saturation)."""
    T = hp.sequence_length
    rng = np.random.default_rng(seed)
    x = rng.random((1, B, H, W, C))
    frames = [x]
    for _ in range(T - 1):
        x = np.clip(x + rng.normal(0, 0.05, x.shape), 0, 1) if smooth else rng.random((1, B, H, W, C))
        frames.append(x)
    return torch.tensor(np.concatenate(frames, axis=0))            # [T,B,H,W,C] fp64


def synth_cond(hp, B, cond, seed=0):
    """Seeded actions [T-1, B, na] / states [T, B, ns] (fp64, time-major) for the action / state-conditioned cell; {} when cond == (0, 0)."""
    na, ns = cond
    rng = np.random.default_rng(1000 + seed)
    T = hp.sequence_length
    out = {}
    if na:
        out['actions'] = torch.tensor(rng.standard_normal((T - 1, B, na)))
    if ns:
        out['states'] = torch.tensor(np.cumsum(0.3 * rng.standard_normal((T, B, ns)), axis=0))
    return out


def _to_dev(cond_inputs):
    return {k: v.float().to(DEV) for k, v in cond_inputs.items()}


def make_noise(hp, B, seed=1, sampling=True):
    T1 = hp.sequence_length - 1
    rng = np.random.default_rng(seed)
    noise = {}
    if hp.nz:
        noise['eps'] = torch.tensor(rng.standard_normal((T1, B, hp.nz)))
        noise['prior'] = torch.tensor(rng.standard_normal((hp.sequence_length - hp.context_frames, B, hp.nz)))
        if hp.learn_prior:
            noise['prior_eps'] = torch.tensor(rng.standard_normal((T1, B, hp.nz)))
    ns = T1 - hp.context_frames
    if sampling:
        noise['ground_truth_sampling'] = torch.tensor(rng.random((ns, B)) < 0.5)
        noise['ground_truth_sampling_enc'] = torch.tensor(rng.random((ns, B)) < 0.5)
    if T1 - hp.clip_length + 1 > 0:
        for phase in ('pre', 'post'):
            noise['d_indices_' + phase] = {k: (rng.integers(0, T1, B), rng.integers(0, T1 - hp.clip_length + 1, B))
                                           for k in ('enc_real', 'enc_fake', 'real', 'fake')}
    return noise


def check_generator_forward(nz=0, B=2, T=5, H=64, W=64, C=3, seed=0, tag=None, cond=(0, 0), **over):
    hpd = dict(context_frames=2, sequence_length=T, nz=nz, schedule_sampling='none' if nz == 0 else 'inverse_sigmoid')
    hpd.update(over)
    hp = make_hparams(**hpd)
    specs = V.variable_specs(hp, (H, W, C), mode='test', cond=cond)
    vals = V.init_variables(specs, seed=4)
    # perturb norm params / biases so that they matter
    rng = np.random.default_rng(9)
    for k in vals:
        if k.endswith('gamma'):
            vals[k] = (1 + 0.2 * rng.standard_normal(vals[k].shape)).astype(np.float32)
        elif k.endswith('beta') or k.endswith('bias'):
            vals[k] = (0.1 * rng.standard_normal(vals[k].shape)).astype(np.float32)
        elif k.endswith('initial_state'):        # learn_initial_state: zero-initialised in the reference; perturbed so that they matter
            vals[k] = (0.3 * rng.standard_normal(vals[k].shape)).astype(np.float32)
        elif k.endswith('kernel'):
            vals[k] = (vals[k] * 3).astype(np.float32)
    images = synth(hp, B, H, W, C, seed)
    noise = make_noise(hp, B, sampling=True)
    ci = synth_cond(hp, B, cond, seed)
    P = {k: torch.tensor(v, dtype=torch.float64) for k, v in vals.items()}
    with torch.no_grad():
        ref = OS.generator_fn(OS.Scope(P).sub('generator'), dict(ci, images=images), 'train', hp, noise)
    eng = SAVPEngine(hp, (H, W, C), B, mode='test', values=vals, device=DEV, cond=cond)
    eng.mode = 'train'          # honour the injected scheduled-sampling mask like mode='train' does
    eng.set_images(dict(_to_dev(ci), images=images.float().to(DEV)), time_major=True)
    eng.prep_generator_weights()
    gen = eng.forward_generator(noise, collect_masks=True)
    torch.cuda.synchronize()
    tag = tag or ('gen_fwd_nz%d_%dx%d' % (nz, H, W))
    out = []
    lo = B if nz else 0
    out.append((tag + '/gen_images', rel(gen[:, lo:], ref['gen_images']), 1e-3))
    g = eng.gen
    masks = g.masks.reshape(g.T1, g.N, H, W, 1, g.M)
    out.append((tag + '/masks', rel(masks[:, lo:], ref['masks']), 1e-3))
    # bit-exact argmax of the compositing masks except where the oracle's top-2 margin is < 1e-5
    m_ref = ref['masks'].squeeze(-2)
    top2 = m_ref.topk(2, dim=-1).values
    safe = (top2[..., 0] - top2[..., 1]) > 1e-5
    mism = (masks[:, lo:].squeeze(-2).argmax(-1).cpu() != m_ref.argmax(-1)) & safe
    out.append((tag + '/mask_argmax_mismatch_frac', float(mism.sum()) / float(safe.sum()), 0.0))
    if hp.transformation == 'cdna':
        kern_ref = ref['_kernels']                                   # [T1,B,5,5,4]
        kern = g.cdna_kern.v.reshape(g.T1, g.N, 5, 5, 4)[:, lo:]
        out.append((tag + '/cdna_kernels', rel(kern, kern_ref), 1e-3))
        kr = kern_ref.reshape(g.T1, B, 25, 4)
        t2 = kr.topk(2, dim=2).values
        safe = (t2[:, :, 0] - t2[:, :, 1]) > 1e-5
        mism = (kern.reshape(g.T1, B, 25, 4).argmax(2).cpu() != kr.argmax(2)) & safe
        out.append((tag + '/cdna_tap_argmax_mismatch', float(mism.sum()), 0.0))
    if nz:
        out.append((tag + '/gen_images_enc', rel(gen[:, :B], ref['gen_images_enc']), 1e-3))
        out.append((tag + '/zs_mu', rel(eng.enc.mu, ref['zs_mu_enc']), 1e-4))
        out.append((tag + '/zs_log_sigma_sq', rel(eng.enc.ls, ref['zs_log_sigma_sq_enc']), 1e-4))
    if nz and hp.learn_prior:
        out.append((tag + '/zs_mu_prior', rel(eng.prior.mu, ref['zs_mu_prior']), 1e-4))
        out.append((tag + '/zs_log_sigma_sq_prior', rel(eng.prior.ls, ref['zs_log_sigma_sq_prior']), 1e-4))
    if cond[1]:
        out.append((tag + '/gen_states', rel(g.gen_states.v[:, lo:], ref['gen_states']), 1e-5))
        if nz:
            out.append((tag + '/gen_states_enc', rel(g.gen_states.v[:, :B], ref['gen_states_enc']), 1e-5))
    return out


def _l2rel(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return float((got - ref).norm() / max(float(ref.norm()), 1e-30))


def check_train_step(B=2, T=6, H=64, W=64, C=3, nz=8, steps=2, seed=0, tag='train', abs_floor=2e-5, cond=(0, 0), **over):
    """One (or two) sess.run(train_op) equivalents.  Gradients are compared per variable in relative L2 against the
    fp64 oracle; the yardstick for "within fp32 tolerance" is the SAME oracle evaluated in fp32 on the CPU: the HIP
    path must be within max(20x that error, 2e-3).  (LeakyReLU/ReLU kinks make a few discriminator gradients
    discretely sensitive to fp32 rounding -- the fp32 CPU oracle shows the same 1e-2 outliers.)"""
    hpd = dict(context_frames=2, sequence_length=T, clip_length=4, nz=nz, lr=2e-4, beta1=0.5, beta2=0.999,
               l1_weight=100.0, l2_weight=0.0, kl_weight=1.0, kl_anneal='none', video_sn_vae_gan_weight=0.1,
               video_sn_gan_weight=0.1, vae_gan_feature_cdist_weight=10.0, gan_feature_cdist_weight=0.0)
    hpd.update(over)
    hp = make_hparams(**hpd)
    specs = V.variable_specs(hp, (H, W, C), mode='train', cond=cond)
    vals = V.init_variables(specs, seed=4)
    rng = np.random.default_rng(9)
    for k in vals:
        if k.endswith('gamma'):
            vals[k] = (1 + 0.2 * rng.standard_normal(vals[k].shape)).astype(np.float32)
        elif k.endswith('beta') or k.endswith('bias'):
            vals[k] = (0.1 * rng.standard_normal(vals[k].shape)).astype(np.float32)
        elif k.endswith('initial_state'):        # learn_initial_state: zero-initialised in the reference; perturbed so that they matter
            vals[k] = (0.3 * rng.standard_normal(vals[k].shape)).astype(np.float32)
        elif k.endswith('kernel') and k.startswith('generator'):
            vals[k] = (vals[k] * (30 if 'state_pred' in k else 3)).astype(np.float32)
    images = synth(hp, B, H, W, C, seed)
    ci = synth_cond(hp, B, cond, seed)
    inputs64 = dict(ci, images=images)
    inputs32 = {k: v.float() for k, v in inputs64.items()}
    P = {k: torch.tensor(v, dtype=torch.float64) for k, v in vals.items()}
    st = OT.init_opt_state(P)
    eng = SAVPEngine(hp, (H, W, C), B, mode='train', values=vals, device=DEV, cond=cond)
    eng.set_images(dict(_to_dev(ci), images=images.float().to(DEV)), time_major=True)
    out = []
    for it in range(steps):
        noise = make_noise(hp, B, seed=100 + it, sampling=True)
        info32 = None
        if it == 0:
            P32 = {k: v.float() for k, v in P.items()}
            n32 = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in noise.items()}
            _, _, info32 = OT.train_step(P32, OT.init_opt_state(P32), inputs32, hp, n32,
                                         noise.get('d_indices_pre'), noise.get('d_indices_post'), step=it)
        P_before = P
        P, st, info_ref = OT.train_step(P, st, inputs64, hp, noise, noise.get('d_indices_pre'), noise.get('d_indices_post'),
                                        step=it)
        info = eng.train_step(noise, return_grads=(it == 0))
        torch.cuda.synchronize()
        t = '%s/step%d' % (tag, it)
        ltol = 1e-3 if it == 0 else 2e-2        # later steps inherit Adam's sign-like first update of near-zero gradients
        if 'd_loss' in info_ref:
            out.append((t + '/d_loss', rel(info['d_loss'], torch.tensor(info_ref['d_loss'])), ltol))
        out.append((t + '/g_loss', rel(info['g_loss'], torch.tensor(info_ref['g_loss'])), ltol))
        for nm, (l, w) in info['g_losses'].items():
            out.append((t + '/' + nm, rel(l, torch.tensor(info_ref['g_losses'][nm])), 2 * ltol))
        if it == 0:
            for grp, key in (('d', 'd_grads'), ('g', 'g_grads')):
                if key not in info_ref:
                    continue
                gmax = max(float(v.abs().max()) for v in info_ref[key].values())
                worst_excess, worst_name, worst_err, nzero, worst_abs = 0.0, '', 0.0, 0, 0.0
                for name, gref in info_ref[key].items():
                    got = info[key][name]
                    if float(gref.abs().max()) < 1e-9 * gmax:
                        # analytically zero gradient (bias in front of an instance norm): absolute check
                        nzero += 1
                        e, tol = float(got.abs().max()) / gmax, 1e-4
                    else:
                        e = _l2rel(got, gref)
                        tol = max(20.0 * _l2rel(info32[key][name], gref), 2e-3)
                        # heavily cancelling sums (e.g. real/fake bias gradients of a discriminator at init) are judged on
                        # their absolute error relative to the largest gradient of the group
                        aerr = float((got.detach().double().cpu() - gref).abs().max())
                        if aerr <= abs_floor * gmax:
                            e = min(e, tol)
                    if e / tol > worst_excess:
                        worst_excess, worst_name, worst_err = e / tol, name, e
                        worst_abs = float((got.detach().double().cpu() - gref).abs().max()) / gmax
                out.append((t + '/%s_grads_worst_err_over_tol[%s rel %.2e abs/gmax %.2e]' % (grp, worst_name.split('/', 1)[-1][-36:], worst_err,
                                                                                          worst_abs if worst_excess else 0.0), worst_excess, 1.0))
            # Adam: m == (1-beta1) * g exactly after the first step; compare the moment arenas instead of the sign-like update
            lr = hp.lr
            tot, cnt = 0.0, 0
            for name, pref in P.items():
                d = (eng.store[name].detach().double().cpu() - pref).abs()
                tot += float(d.sum())
                cnt += d.numel()
            out.append((t + '/param_mean_abs_diff_over_lr', tot / cnt / lr, 0.05))
    return out


# bench.py's workloads (BASELINE.json configs[1], [3], [4]) as recipe_case arguments: the golden steps are taken at exactly these shapes
BENCH_CASES = {
    'c2': dict(B=16, T=30, H=64, W=64, C=3, context=2, nz=8, kl_weight=1.0),
    'c4': dict(B=16, T=40, H=64, W=64, C=1, context=10, nz=32, kl_weight=0.01),      # KTH (kth_dataset.py:26-36, hparams/kth/ours_savp)
    'c5': dict(B=8, T=30, H=128, W=128, C=3, context=2, nz=8, kl_weight=1.0),        # 128x128: the >= 128 layer table (savp_model.py:198-210)
}


def recipe_case(B, T=30, H=64, W=64, C=3, seed=0, context=2, nz=8, kl_weight=1.0):
    """The benchmarked step's inputs (hparams/bair_action_free/ours_savp recipe: T=30, clip_length=10, nz=8) at batch B, all seeded:
    (hparams, variables (init + perturbed norm parameters / biases so that they matter), images [T,B,H,W,C] fp64, noise)."""
    hp = make_hparams(context_frames=context, sequence_length=T, clip_length=10, nz=nz, lr=2e-4, beta1=0.5, beta2=0.999, l1_weight=100.0,
                      l2_weight=0.0, kl_weight=kl_weight, kl_anneal='none', video_sn_vae_gan_weight=0.1, video_sn_gan_weight=0.1,
                      vae_gan_feature_cdist_weight=10.0, gan_feature_cdist_weight=0.0)
    specs = V.variable_specs(hp, (H, W, C), mode='train')
    vals = V.init_variables(specs, seed=4)
    rng = np.random.default_rng(9)
    for k in vals:
        if k.endswith('gamma'):
            vals[k] = (1 + 0.2 * rng.standard_normal(vals[k].shape)).astype(np.float32)
        elif k.endswith('beta') or k.endswith('bias'):
            vals[k] = (0.1 * rng.standard_normal(vals[k].shape)).astype(np.float32)
        elif k.endswith('initial_state'):        # learn_initial_state: zero-initialised in the reference; perturbed so that they matter
            vals[k] = (0.3 * rng.standard_normal(vals[k].shape)).astype(np.float32)
        elif k.endswith('kernel') and k.startswith('generator'):
            vals[k] = (vals[k] * 3).astype(np.float32)
    images = synth(hp, B, H, W, C, seed)
    noise = make_noise(hp, B, seed=100, sampling=True)
    return hp, vals, images, noise


def check_train_recipe_shapes(B=2, T=30, H=64, W=64, C=3, seed=0):
    """One train step at the recipe's sequence / clip lengths (hparams/bair_action_free/ours_savp/model_hparams.json: T=30,
    clip_length=10, nz=8) with B scaled down, run on BOTH datapaths from the same variables / inputs / noise and compared with ONE
    fp64 oracle step.  fp32 mode: same yardstick as check_train_step.  bf16 mode (bench default: conv operands rounded to bf16,
    fp32 accumulate): losses within 2e-2 of max(|ref|, 0.05) (the LSGAN generator terms (D-1)^2 sit at ~1e-3 after the D update, so
    a plain relative error would only measure cancellation), the generated frames within 5e-2 absolute, per-variable gradients
    within 0.25 relative L2 = cosine >= 0.97 (measured on MI355X: 0.11 worst for D, 0.19 worst for G against the fp32 datapath;
    the spread comes from ReLU / LeakyReLU masks that flip under the 4e-3 operand rounding, the fp32 CPU oracle shows the same
    effect at 1e-2)."""
    from video_prediction_amd import kernels as K
    hp, vals, images, noise = recipe_case(B, T, H, W, C, seed)
    P = {k: torch.tensor(v, dtype=torch.float64) for k, v in vals.items()}
    P_new, _, ref = OT.train_step(P, OT.init_opt_state(P), {'images': images}, hp, noise, noise['d_indices_pre'],
                                  noise['d_indices_post'], step=0)
    P32 = {k: v.float() for k, v in P.items()}
    n32 = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in noise.items()}
    _, _, ref32 = OT.train_step(P32, OT.init_opt_state(P32), {'images': images.float()}, hp, n32, noise['d_indices_pre'],
                                noise['d_indices_post'], step=0)
    out = []
    for prec in ('f32', 'bf16'):
        K.set_conv_precision(prec)
        try:
            eng = SAVPEngine(hp, (H, W, C), B, mode='train', values=vals, device=DEV)
            eng.set_images(images.float().to(DEV), time_major=True)
            info = eng.train_step(noise, return_grads=True)
            torch.cuda.synchronize()
        finally:
            K.set_conv_precision('f32')
        t = 'recipe_T%d_clip10/%s' % (T, prec)
        ltol = 1e-3 if prec == 'f32' else 2e-2
        floor = 0.0 if prec == 'f32' else 0.05

        def lrel(got, want):
            return abs(float(got) - float(want)) / max(abs(float(want)), floor, 1e-30)
        out.append((t + '/d_loss', lrel(info['d_loss'], ref['d_loss']), ltol))
        out.append((t + '/g_loss', lrel(info['g_loss'], ref['g_loss']), ltol))
        for nm, (l, w) in info['g_losses'].items():
            # bf16 datapath: 3 * ltol = 6e-2 of max(|ref|, 0.05) per term -- besides the operand rounding, the ConvLSTM gate
            # pre-activations make their HBM round trip in bf16 (fused cell epilogue), measured 4.7e-2 on the LSGAN generator term
            # fp32 datapath: 2e-3 relative, or 20 x the distance of the fp32 CPU oracle from the fp64 one on that very term if that is
            # more (the yardstick of the gradient gates below): the LSGAN generator terms are taken AFTER the discriminator's first Adam
            # update, which moves a weight by +-lr according to the SIGN of its gradient -- an element whose fp32 gradient has the other
            # sign than the fp64 one shifts (D(fake) - 1)^2 by more than any rounding (measured on MI355X: 3.6e-3 in one run of several)
            tol32 = max(2 * ltol, 20.0 * lrel(ref32['g_losses'][nm], ref['g_losses'][nm])) if nm in ref32.get('g_losses', {}) else 2 * ltol
            out.append((t + '/' + nm, lrel(l, ref['g_losses'][nm]), tol32 if prec == 'f32' else 3 * ltol))
        gen = eng.gen.gen.v
        out.append((t + '/gen_images_enc_abs', float((gen[:, :B].double().cpu() - ref['gen_images_enc']).abs().max()),
                    1e-3 if prec == 'f32' else 5e-2))
        out.append((t + '/gen_images_abs', float((gen[:, B:].double().cpu() - ref['gen_images']).abs().max()),
                    1e-3 if prec == 'f32' else 5e-2))
        for grp, key in (('d', 'd_grads'), ('g', 'g_grads')):
            gmax = max(float(v.abs().max()) for v in ref[key].values())
            worst, wname = 0.0, ''
            for name, gref in ref[key].items():
                got = info[key][name]
                if float(gref.abs().max()) < 1e-9 * gmax:
                    e, tol = float(got.abs().max()) / gmax, (1e-4 if prec == 'f32' else 1e-2)
                else:
                    e = _l2rel(got, gref)
                    tol = max(20.0 * _l2rel(ref32[key][name], gref), 2e-3) if prec == 'f32' else 0.25
                    aerr = float((got.detach().double().cpu() - gref).abs().max())
                    if aerr <= (2e-5 if prec == 'f32' else 2e-3) * gmax:        # heavily cancelling sums: absolute yardstick
                        e = min(e, tol)
                if e / tol > worst:
                    worst, wname = e / tol, name
            out.append((t + '/%s_grads_worst_err_over_tol[%s]' % (grp, wname.split('/', 1)[-1][-36:]), worst, 1.0))
        if prec == 'f32':
            tot, cnt = 0.0, 0
            for name, pref in P_new.items():
                d = (eng.store[name].detach().double().cpu() - pref).abs()
                tot += float(d.sum())
                cnt += d.numel()
            out.append((t + '/param_mean_abs_diff_over_lr', tot / cnt / hp.lr, 0.05))
    return out


def check_config_c1():
    """BASELINE configs[0] at its own shape: ours_deterministic_l1 (hparams/bair_action_free/ours_deterministic_l1/model_hparams.json:
    nz=0, l1 only, lr 1e-3, beta1 0.9), BAIR 64x64x3, context 2 + predict 10 -> T=12, B=4; forward and one train step, fp32 datapath."""
    res = check_generator_forward(nz=0, B=4, T=12, tag='c1_det_fwd')
    res += check_train_step(B=4, T=12, nz=0, steps=1, tag='c1_det_train', lr=1e-3, beta1=0.9, beta2=0.999, l1_weight=1.0, l2_weight=0.0,
                            kl_weight=0.0, video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.0, vae_gan_feature_cdist_weight=0.0)
    return res


def check_config_c4():
    res = check_generator_forward(nz=32, B=2, T=12, C=1, tag='c4_kth_fwd', context_frames=10)
    res += check_train_step(B=1, T=12, C=1, nz=32, steps=1, tag='c4_kth_train', context_frames=10, clip_length=10, kl_weight=0.01)
    return res


def check_config_c5():
    res = check_generator_forward(nz=8, B=1, T=4, H=128, W=128, tag='c5_128_fwd')
    res += check_train_step(B=1, T=5, H=128, W=128, nz=8, steps=1, tag='c5_128_train', clip_length=4)
    return res


# Loss weights of the SAVP recipes the reference ships besides ours_savp / ours_deterministic_l1 (which C2 / C4 and C1 cover): the VALUES of
# hparams/<dataset>/<name>/model_hparams.json (bair_action_free: nz = 8, 64x64x3, context 2; kth: nz = 32, 64x64x1, context 10)
SHIPPED_RECIPES = {
    'bair_ours_gan': dict(C=3, nz=8, context_frames=2, lr=2e-4, beta1=0.5, l1_weight=100.0, l2_weight=0.0, kl_weight=0.0,
                          video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.1, vae_gan_feature_cdist_weight=0.0, gan_feature_cdist_weight=10.0),
    'bair_ours_vae_l1': dict(C=3, nz=8, context_frames=2, lr=1e-3, beta1=0.9, l1_weight=1.0, l2_weight=0.0, kl_weight=0.001,
                             video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.0, vae_gan_feature_cdist_weight=0.0),
    'bair_ours_deterministic_l2': dict(C=3, nz=0, context_frames=2, lr=1e-3, beta1=0.9, l1_weight=0.0, l2_weight=1.0, kl_weight=0.0,
                                       video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.0, vae_gan_feature_cdist_weight=0.0),
    'kth_ours_gan': dict(C=1, nz=32, context_frames=10, lr=2e-4, beta1=0.5, l1_weight=100.0, l2_weight=0.0, kl_weight=0.0,
                         video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.1, vae_gan_feature_cdist_weight=0.0, gan_feature_cdist_weight=10.0),
    'kth_ours_vae_l1': dict(C=1, nz=32, context_frames=10, lr=1e-3, beta1=0.9, l1_weight=1.0, l2_weight=0.0, kl_weight=1e-5,
                            video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.0, vae_gan_feature_cdist_weight=0.0),
}


def check_shipped_recipes(names=None):
    """One train step (losses, per-variable gradients, Adam) per shipped SAVP recipe against the fp64 oracle, at the recipe's clip
    length (10 -> T = 12) and batch 1."""
    res = []
    for name in (names or sorted(SHIPPED_RECIPES)):
        r = dict(SHIPPED_RECIPES[name])
        C = r.pop('C')
        res += check_train_step(B=1, T=12, C=C, steps=1, tag='recipe_' + name, clip_length=10, **r)
    return res


def check_model_small():
    res = []
    res += check_generator_forward(nz=0, B=2, T=5)
    res += check_generator_forward(nz=8, B=2, T=4)
    res += check_generator_forward(nz=0, B=1, T=3, H=64, W=64, C=1, tag='gen_fwd_gray')
    res += check_generator_forward(nz=8, B=1, T=4, tag='gen_fwd_flow', transformation='flow')
    res += check_generator_forward(nz=0, B=1, T=4, tag='gen_fwd_dna', transformation='dna')
    res += check_generator_forward(nz=8, B=2, T=4, tag='gen_fwd_gru', conv_rnn='gru')
    # the latent enters only the first encoder conv / only the first decoder conv (savp_model.py:456-470,492-506)
    res += check_generator_forward(nz=8, B=1, T=4, tag='gen_fwd_where_add_input', where_add='input')
    res += check_generator_forward(nz=8, B=1, T=4, tag='gen_fwd_where_add_middle', where_add='middle')
    # other background image sets of the compositing step (savp_model.py:581-594): first + last + last-context frames (9 masks) and
    # all context frames without the previous image (8 masks)
    res += check_generator_forward(nz=8, B=1, T=5, tag='gen_fwd_bg_last_frames', context_frames=3, last_image_background=True,
                                   last_context_image_background=True)
    res += check_generator_forward(nz=0, B=1, T=5, tag='gen_fwd_bg_context_images', context_frames=3, context_images_background=True,
                                   prev_image_background=False)
    # mask conv on h_masks alone (dependent_mask=False, savp_model.py:631-632) / no scratch image (:561-572, 6 masks)
    res += check_generator_forward(nz=8, B=1, T=4, tag='gen_fwd_independent_mask', dependent_mask=False)
    res += check_generator_forward(nz=8, B=1, T=4, tag='gen_fwd_no_scratch', generate_scratch_image=False)
    # latent added as dense(z) behind the convolutions instead of tiled into their inputs (savp_model.py:983-993, rnn_ops.py:145-146):
    # cancelled by the instance norms (tests/test_untiled_latent_is_cancelled.py); the oracle computes it literally
    res += check_generator_forward(nz=8, B=1, T=4, tag='gen_fwd_untiled_latent', use_tile_concat=False)
    return res


def check_train_small():
    res = []
    res += check_train_step(B=2, T=6, nz=8, steps=2, tag='train_savp')
    # image + per-frame images discriminators next to the video one (networks.py:35-69, savp_model.py:105-125)
    res += check_train_step(B=2, T=6, nz=8, steps=1, tag='train_all_discriminators', image_sn_gan_weight=0.1,
                            image_sn_vae_gan_weight=0.1, images_sn_gan_weight=0.05, images_sn_vae_gan_weight=0.05,
                            gan_feature_cdist_weight=1.0)
    res += check_train_step(B=1, T=4, nz=8, steps=1, tag='train_gru', conv_rnn='gru', video_sn_vae_gan_weight=0.0,
                            video_sn_gan_weight=0.0, vae_gan_feature_cdist_weight=0.0)
    for tf in ('flow', 'dna'):
        res += check_train_step(B=1, T=4, nz=8, steps=1, tag='train_' + tf, transformation=tf, video_sn_vae_gan_weight=0.0,
                                video_sn_gan_weight=0.0, vae_gan_feature_cdist_weight=0.0)
    for wa in ('input', 'middle'):
        res += check_train_step(B=1, T=4, nz=8, steps=1, tag='train_where_add_' + wa, where_add=wa, video_sn_vae_gan_weight=0.0,
                                video_sn_gan_weight=0.0, vae_gan_feature_cdist_weight=0.0)
    res += check_train_step(B=1, T=5, nz=8, steps=1, tag='train_bg_context_images', context_frames=3, context_images_background=True,
                            prev_image_background=False, video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.0,
                            vae_gan_feature_cdist_weight=0.0)
    res += check_train_step(B=1, T=4, nz=8, steps=1, tag='train_no_scratch_independent_mask', generate_scratch_image=False,
                            dependent_mask=False, transformation='flow', video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.0,
                            vae_gan_feature_cdist_weight=0.0)
    res += check_train_step(B=1, T=4, nz=8, steps=1, tag='train_untiled_latent', use_tile_concat=False, video_sn_vae_gan_weight=0.0,
                            video_sn_gan_weight=0.0, vae_gan_feature_cdist_weight=0.0)
    res += check_train_step(B=2, T=5, nz=0, steps=1, tag='train_det', video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.0,
                            vae_gan_feature_cdist_weight=0.0, kl_weight=0.0, l1_weight=1.0)
    return res


def check_model_bf16():
    """bf16-operand conv mode end to end (reported, loosely gated: SURVEY.md 8c asks per-op rel <= 1e-2; the recurrent
    unroll amplifies it).  Images are in [0,1]; masks/gen compared in absolute terms."""
    from video_prediction_amd import kernels as K
    K.set_conv_precision('bf16')
    try:
        res = check_generator_forward(nz=8, B=2, T=6, tag='bf16_gen_fwd')
        out = []
        for n, e, t in res:
            if n.endswith('/gen_images') or n.endswith('/gen_images_enc') or n.endswith('/masks') or n.endswith('/cdna_kernels'):
                out.append((n, e, 5e-2))
            elif 'argmax' in n:
                out.append((n + '(reported)', e, 1.0))
            else:
                out.append((n, e, 5e-2))
    finally:
        K.set_conv_precision('f32')
    return out


def check_eval_best_of_n(B=2, T=6, H=32, W=32, C=3, num_samples=4):
    """eval_outputs_and_metrics_fn (base_model.py:132-227) vs the oracle: best / mean / worst of num_samples prior samples."""
    from oracle import metrics as OM
    hp = make_hparams(context_frames=2, sequence_length=T, nz=8, schedule_sampling='none')
    specs = V.variable_specs(hp, (H, W, C), mode='test')
    vals = V.init_variables(specs, seed=4)
    rng = np.random.default_rng(5)
    for k in vals:                                     # make the latent matter at init scale
        if 'rnn_z' in k or k.endswith('gamma'):
            vals[k] = (vals[k] + 0.3 * rng.standard_normal(vals[k].shape)).astype(np.float32)
    images = synth(hp, B, H, W, C, 3)
    noises = [make_noise(hp, B, seed=20 + i, sampling=False) for i in range(num_samples)]
    P = {k: torch.tensor(v, dtype=torch.float64) for k, v in vals.items()}
    gens = []
    with torch.no_grad():
        for n in noises:
            gens.append(OS.generator_fn(OS.Scope(P).sub('generator'), {'images': images}, 'test', hp, n)['gen_images'])
    r_out, r_met = OM.eval_outputs_and_metrics(images, gens, hp.context_frames)
    eng = SAVPEngine(hp, (H, W, C), B, mode='test', values=vals, device=DEV)
    eng.set_images(images.float().to(DEV), time_major=True)
    outs, mets = eng.eval_outputs_and_metrics(num_samples, noises)
    torch.cuda.synchronize()
    res = []
    for k in r_met:
        res.append(('eval/' + k, rel(mets[k], r_met[k]), 2e-3))
    for k in r_out:
        res.append(('eval/' + k, rel(outs[k], r_out[k]), 2e-3))
    # metrics_fn (base_model.py:113-130) on the first sample
    m = eng.metrics(eng.generate(noises[0]))
    r = OM.metrics_fn(images, gens[0], hp.context_frames)
    for k in r:
        res.append(('metrics_fn/' + k, rel(m[k], r[k]), 2e-3))
    return res


def check_action_conditioned():
    """The action / state-conditioned cell (savp_model.py:24-26,411-444,655-661; base_model.py:758-762), forward (fp32 datapath vs the
    fp64 oracle) and through one train step with the state loss on: actions + states with the latent (tiled width 16: every channel
    count stays aligned), BAIR's use_state shapes (4 actions + 3 states + nz 8 = 15 tiled channels: odd channel counts everywhere),
    actions only on the deterministic model (nz = 0), and where_add = 'input'."""
    res = []
    cases = [('a4s4', (4, 4), 8, dict(state_weight=1.0)),
             ('a4s3_bair_use_state', (4, 3), 8, dict(state_weight=1.0)),
             ('a4_det', (4, 0), 0, dict()),
             ('a2s2_where_input', (2, 2), 8, dict(state_weight=0.5, where_add='input'))]
    for tag, cond, nz, over in cases:
        over_f = {k: v for k, v in over.items() if k != 'state_weight'}
        res += check_generator_forward(nz=nz, B=2, T=5, tag='gen_fwd_' + tag, cond=cond, **over_f)
        if nz:
            res += check_train_step(B=2, T=5, nz=nz, steps=1, tag='train_' + tag, cond=cond, **over)
        else:
            res += check_train_step(B=2, T=5, nz=0, steps=1, tag='train_' + tag, cond=cond, kl_weight=0.0, video_sn_vae_gan_weight=0.0,
                                    vae_gan_feature_cdist_weight=0.0, **over)
    res += check_action_conditioned_bf16((4, 4), 'a4s4')
    res += check_action_conditioned_bf16((4, 3), 'a4s3_bair_use_state')
    return res


def check_action_conditioned_bf16(cond, tag, B=2, T=5, H=64, W=64, C=3):
    """The conditioned cell on the bf16 datapath (bench default) against the SAME engine on the exact-fp32 datapath (itself held to the
    oracle above), one train step from identical variables / inputs / noise: the gates of check_train_recipe_shapes' bf16 arm (losses 2e-2 /
    6e-2 of max(|ref|, 0.05), frames 5e-2 absolute, per-variable gradients 0.25 relative L2 with the 2e-3-of-gmax absolute floor); the
    state recurrence involves no convolution and must agree to fp32 rounding."""
    from video_prediction_amd import kernels as K
    hp = make_hparams(context_frames=2, sequence_length=T, clip_length=4, nz=8, lr=2e-4, beta1=0.5, beta2=0.999, l1_weight=100.0,
                      l2_weight=0.0, kl_weight=1.0, kl_anneal='none', video_sn_vae_gan_weight=0.1, video_sn_gan_weight=0.1,
                      vae_gan_feature_cdist_weight=10.0, gan_feature_cdist_weight=0.0, state_weight=1.0)
    specs = V.variable_specs(hp, (H, W, C), mode='train', cond=cond)
    vals = V.init_variables(specs, seed=4)
    rng = np.random.default_rng(9)
    for k in vals:
        if k.endswith('gamma'):
            vals[k] = (1 + 0.2 * rng.standard_normal(vals[k].shape)).astype(np.float32)
        elif k.endswith('beta') or k.endswith('bias'):
            vals[k] = (0.1 * rng.standard_normal(vals[k].shape)).astype(np.float32)
        elif k.endswith('kernel') and k.startswith('generator'):
            vals[k] = (vals[k] * (30 if 'state_pred' in k else 3)).astype(np.float32)
    images = synth(hp, B, H, W, C, 0)
    ci = synth_cond(hp, B, cond, 0)
    noise = make_noise(hp, B, seed=100, sampling=True)
    runs = {}
    for prec in ('f32', 'bf16'):
        K.set_conv_precision(prec)
        try:
            eng = SAVPEngine(hp, (H, W, C), B, mode='train', values=vals, device=DEV, cond=cond)
            eng.set_images(dict(_to_dev(ci), images=images.float().to(DEV)), time_major=True)
            info = eng.train_step(noise, return_grads=True)
            torch.cuda.synchronize()
            runs[prec] = (info, eng.gen.gen.v.double().cpu(), eng.gen.gen_states.v.double().cpu())
        finally:
            K.set_conv_precision('f32')
    (ref, gen_r, gs_r), (got, gen_g, gs_g) = runs['f32'], runs['bf16']
    t = 'cond_bf16_%s' % tag
    out = []

    def lrel(a, b):
        return abs(float(a) - float(b)) / max(abs(float(b)), 0.05)
    out.append((t + '/d_loss', lrel(got['d_loss'], ref['d_loss']), 2e-2))
    out.append((t + '/g_loss', lrel(got['g_loss'], ref['g_loss']), 2e-2))
    for nm, (l, w) in got['g_losses'].items():
        out.append((t + '/' + nm, lrel(l, ref['g_losses'][nm][0]), 6e-2))
    out.append((t + '/gen_images_abs', float((gen_g - gen_r).abs().max()), 5e-2))
    out.append((t + '/gen_states', rel(gs_g, gs_r), 1e-5))
    for grp, key in (('d', 'd_grads'), ('g', 'g_grads')):
        gmax = max(float(v.abs().max()) for v in ref[key].values())
        worst, wname = 0.0, ''
        for name, gref in ref[key].items():
            g_ = got[key][name]
            if float(gref.abs().max()) < 1e-9 * gmax:
                e, tol = float(g_.abs().max()) / gmax, 1e-2
            else:
                e, tol = _l2rel(g_, gref), 0.25
                if float((g_.double() - gref.double()).abs().max()) <= 2e-3 * gmax:
                    e = min(e, tol)
            if e / tol > worst:
                worst, wname = e / tol, name
        out.append((t + '/%s_grads_worst_err_over_tol[%s]' % (grp, wname.split('/', 1)[-1][-36:]), worst, 1.0))
    return out


def check_generator_samples(B=1, T=4, S=2, H=64, W=64, C=3, nz=8):
    """generator_fn's visualisation unroll (savp_model.py:745-767): S draws from the prior per sequence -> gen_images_samples
    [T-1, B, H, W, C, S] and their mean, plug-in function vs the oracle with the draws injected."""
    from video_prediction_amd.models import savp_model as SM
    out = []
    for learn_prior in (False, True):
        over = dict(learn_prior=True, use_e_rnn=True, nef=16) if learn_prior else {}
        hp = make_hparams(context_frames=2, sequence_length=T, nz=nz, schedule_sampling='none', num_samples=S, **over)
        vals = V.init_variables(V.variable_specs(hp, (H, W, C), mode='test'), seed=4)
        images = synth(hp, B, H, W, C, 0)
        noise = make_noise(hp, B, sampling=False)
        rng = np.random.default_rng(77)
        if learn_prior:
            noise['samples_prior_eps'] = torch.tensor(rng.standard_normal((T - 1, S, B, nz)))
        else:
            noise['samples_prior'] = torch.tensor(rng.standard_normal((T - hp.context_frames, S, B, nz)))
        P = {k: torch.tensor(v, dtype=torch.float64) for k, v in vals.items()}
        with torch.no_grad():
            ref = OS.generator_fn(OS.Scope(P).sub('generator'), {'images': images}, 'test', hp, noise)
        eng = SAVPEngine(hp, (H, W, C), B, mode='test', values=vals, device=DEV)
        got = SM.generator_fn({'images': images.float().to(DEV)}, 'test', hp, engine=eng, noise=noise)
        torch.cuda.synchronize()
        tag = 'gen_samples_%s' % ('learned_prior' if learn_prior else 'unit_prior')
        assert tuple(got['gen_images_samples'].shape) == (T - 1, B, H, W, C, S)
        out.append((tag + '/gen_images_samples', rel(got['gen_images_samples'], ref['gen_images_samples']), 1e-3))
        out.append((tag + '/gen_images_samples_avg', rel(got['gen_images_samples_avg'], ref['gen_images_samples_avg']), 1e-3))
        out.append((tag + '/gen_images_after_the_samples', rel(got['gen_images'], ref['gen_images']), 1e-3))      # the views still hold the main unroll
        out.append((tag + '/gen_images_enc_after_the_samples', rel(got['gen_images_enc'], ref['gen_images_enc']), 1e-3))
    return out


def check_flow_tv_loss():
    """tv_weight with transformation = 'flow' (base_model.py:763-769): total variation of the predicted flows, value and gradient through
    one train step (fp32 datapath vs the fp64 oracle)."""
    return check_train_step(B=2, T=5, nz=8, steps=1, tag='train_flow_tv', transformation='flow', tv_weight=0.05)


def check_cell_options():
    """The options of SAVPCell that no shipped recipe sets, each forward (fp32 datapath vs the fp64 oracle) and through one train step:
    learn_initial_state (savp_model.py:295-307,344-352), ablation_rnn (:272-291,426-429,466-474,502-509), ablation_conv_rnn_norm
    (:380-384), conv_rnn_norm_layer = 'none' (rnn_ops.py:122-125), rnn = 'gru' for the latent's cell and the encoders' recurrent tail
    (:38-41,358-359)."""
    res = []
    cases = [('learn_init', dict(learn_initial_state=True)),
             ('learn_init_gru', dict(learn_initial_state=True, conv_rnn='gru')),
             ('learn_init_rnn_gru', dict(learn_initial_state=True, rnn='gru')),          # the latent's GRUCell from a learned state (round 6)
             ('abl_rnn', dict(ablation_rnn=True)),
             ('abl_rnn_learn_init', dict(ablation_rnn=True, learn_initial_state=True)),      # no state to learn: the flag does nothing (savp_model.py:269-307)
             ('abl_cell_norm', dict(ablation_conv_rnn_norm=True)),
             ('cell_norm_none', dict(conv_rnn_norm_layer='none')),
             ('rnn_gru', dict(rnn='gru', use_e_rnn=True, nef=16))]
    for tag, over in cases:
        res += check_generator_forward(nz=8, B=2, T=5, tag='gen_fwd_' + tag, **over)
        res += check_train_step(B=2, T=5, nz=8, steps=1, tag='train_' + tag, **over)
    return res
