"""Restatement of the SAVP losses and the sequential D-then-G Adam train step on torch-CPU.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows /root/reference/video_prediction/losses.py,
models/base_model.py:286-319 (lr / kl schedules), :402-465 (tower_fn), :486-510 (train op), :733-852 (loss fns).
Gradients come from torch autograd (double precision capable), standing in for tf.gradients.
"""
from collections import OrderedDict

import torch

from . import savp, tf_ops


# ---- losses.py ---------------------------------------------------------------------------------
def l1_loss(pred, target):
    return (target - pred).abs().mean()                                        # losses.py:6-7


def l2_loss(pred, target):
    return ((target - pred) ** 2).mean()                                       # losses.py:10-11


def normalize_tensor(t, eps=1e-10):
    return t / (torch.linalg.norm(t, dim=-1, keepdim=True) + eps)             # losses.py:14-16


def cosine_distance(t0, t1):
    t0 = normalize_tensor(t0)
    t1 = normalize_tensor(t1)
    return ((t0 - t1) ** 2).sum(dim=-1).mean() / 2.0                           # losses.py:19-22


def gan_loss(logits, labels, gan_loss_type):
    if gan_loss_type == 'LSGAN':
        return ((logits - labels) ** 2).mean()                                 # losses.py:41-44
    if gan_loss_type == 'GAN':
        lab = torch.full_like(logits, labels)
        return torch.nn.functional.binary_cross_entropy_with_logits(logits, lab)
    if gan_loss_type == 'SNGAN':
        if labels == 0.0:
            return torch.nn.functional.softplus(logits).mean()
        if labels == 1.0:
            return torch.nn.functional.softplus(-logits).mean()
        raise NotImplementedError
    raise ValueError('Unknown GAN loss type %s' % gan_loss_type)


def kl_loss(mu, log_sigma_sq, mu2=None, log_sigma2_sq=None):
    """losses.py:57-67: KL(N(mu, s) || N(0, I)), or against a second diagonal Gaussian (the learned prior, base_model.py:825-828)."""
    if mu2 is None and log_sigma2_sq is None:
        sigma_sq = torch.exp(log_sigma_sq)
        return -0.5 * (1 + log_sigma_sq - mu ** 2 - sigma_sq).sum(dim=-1).mean()   # losses.py:57-60
    return ((log_sigma2_sq - log_sigma_sq) / 2 + (torch.exp(log_sigma_sq) + (mu - mu2) ** 2) / (2 * torch.exp(log_sigma2_sq))
            - 0.5).sum(dim=-1).mean()                                              # losses.py:62-67


# ---- schedules (base_model.py:286-319) ---------------------------------------------------------
def learning_rate(hp, step):
    if any(hp.lr_boundaries):
        # tf.train.piecewise_constant: value i+1 when step > boundary i
        vals = [hp.lr * 0.1 ** i for i in range(len(hp.lr_boundaries) + 1)]
        idx = sum(1 for b in hp.lr_boundaries if step > b)
        return vals[idx]
    elif any(hp.decay_steps):
        start_step, end_step = hp.decay_steps
        if start_step == end_step:
            schedule = 0.0 if step < start_step else 1.0
        else:
            s = min(max(step, start_step), end_step)
            schedule = float(s - start_step) / float(end_step - start_step)
        return hp.lr + (hp.end_lr - hp.lr) * schedule
    return hp.lr


def kl_weight(hp, step):
    if not hp.kl_weight:
        return None
    if hp.kl_anneal == 'none':
        return hp.kl_weight
    if hp.kl_anneal == 'linear':
        start_step, end_step = hp.kl_anneal_steps
        s = min(max(step, start_step), end_step)
        return hp.kl_weight * float(s - start_step) / float(end_step - start_step)
    raise NotImplementedError(hp.kl_anneal)


# ---- loss fns (base_model.py:733-852) ----------------------------------------------------------
def generator_loss_fn(hp, inputs, outputs, kl_w):
    losses = OrderedDict()
    gen_images = outputs.get('gen_images_enc', outputs['gen_images'])
    target_images = inputs['images'][1:]
    if hp.l1_weight:
        losses['gen_l1_loss'] = (l1_loss(gen_images, target_images), hp.l1_weight)
    if hp.l2_weight:
        losses['gen_l2_loss'] = (l2_loss(gen_images, target_images), hp.l2_weight)
    if getattr(hp, 'state_weight', 0):                                                       # base_model.py:758-762
        gen_states = outputs.get('gen_states_enc', outputs['gen_states'])
        losses['gen_state_loss'] = (l2_loss(gen_states, inputs['states'][1:]), hp.state_weight)
    if getattr(hp, 'tv_weight', 0):                                                          # base_model.py:763-769
        gen_flows = outputs.get('gen_flows_enc', outputs['gen_flows'])                       # [T, B, H, W, 2, nk]
        d1 = gen_flows[..., 1:, :, :, :] - gen_flows[..., :-1, :, :, :]
        d2 = gen_flows[..., :, 1:, :, :] - gen_flows[..., :, :-1, :, :]
        # sum over the multiple transformations but take the mean for the other dimensions
        losses['gen_tv_loss'] = (d1.abs().sum(dim=(-2, -1)).mean() + d2.abs().sum(dim=(-2, -1)).mean(), hp.tv_weight)
    for infix, w, wf_l2, wf_cd, sfx, nm in (
            ('_image_sn', hp.image_sn_gan_weight, hp.gan_feature_l2_weight, hp.gan_feature_cdist_weight, '', 'gan'),
            ('_video_sn', hp.video_sn_gan_weight, hp.gan_feature_l2_weight, hp.gan_feature_cdist_weight, '', 'gan'),
            ('_images_sn', hp.images_sn_gan_weight, hp.gan_feature_l2_weight, hp.gan_feature_cdist_weight, '', 'gan'),
            ('_images_sn', hp.images_sn_vae_gan_weight, hp.vae_gan_feature_l2_weight, hp.vae_gan_feature_cdist_weight, '_enc', 'vae_gan'),
            ('_image_sn', hp.image_sn_vae_gan_weight, hp.vae_gan_feature_l2_weight, hp.vae_gan_feature_cdist_weight, '_enc', 'vae_gan'),
            ('_video_sn', hp.video_sn_vae_gan_weight, hp.vae_gan_feature_l2_weight, hp.vae_gan_feature_cdist_weight, '_enc', 'vae_gan')):
        if not w:
            continue
        losses['gen%s_%s_loss' % (infix, nm)] = (
            gan_loss(outputs['discrim%s_logits%s_fake' % (infix, sfx)], 1.0, hp.gan_loss_type), w)
        if wf_l2 or wf_cd:
            fk, rl = [], []
            i = 0
            while True:
                f = outputs.get('discrim%s_feature%d%s_fake' % (infix, i, sfx))
                r = outputs.get('discrim%s_feature%d%s_real' % (infix, i, sfx))
                if f is None or r is None:
                    break
                fk.append(f)
                rl.append(r)
                i += 1
            if wf_l2:
                losses['gen%s_%s_feature_l2_loss' % (infix, nm)] = (sum(l2_loss(f, r) for f, r in zip(fk, rl)), wf_l2)
            if wf_cd:
                losses['gen%s_%s_feature_cdist_loss' % (infix, nm)] = (
                    sum(cosine_distance(f, r) for f, r in zip(fk, rl)), wf_cd)
    if hp.kl_weight:
        losses['gen_kl_loss'] = (kl_loss(outputs['zs_mu_enc'], outputs['zs_log_sigma_sq_enc'], outputs.get('zs_mu_prior'),
                                         outputs.get('zs_log_sigma_sq_prior')), kl_w)       # base_model.py:825-828
    return losses


def discriminator_loss_fn(hp, inputs, outputs):
    losses = OrderedDict()
    for infix, w, sfx, nm in (('_image_sn', hp.image_sn_gan_weight, '', 'gan'),
                              ('_video_sn', hp.video_sn_gan_weight, '', 'gan'),
                              ('_images_sn', hp.images_sn_gan_weight, '', 'gan'),
                              ('_images_sn', hp.images_sn_vae_gan_weight, '_enc', 'vae_gan'),
                              ('_image_sn', hp.image_sn_vae_gan_weight, '_enc', 'vae_gan'),
                              ('_video_sn', hp.video_sn_vae_gan_weight, '_enc', 'vae_gan')):
        if not w:
            continue
        real = gan_loss(outputs['discrim%s_logits%s_real' % (infix, sfx)], 1.0, hp.gan_loss_type)
        fake = gan_loss(outputs['discrim%s_logits%s_fake' % (infix, sfx)], 0.0, hp.gan_loss_type)
        losses['discrim%s_%s_loss' % (infix, nm)] = (real + fake, w)
    return losses


def total_loss(losses):
    tot = 0.0
    for loss, weight in losses.values():
        tot = tot + loss * weight
    return tot


# ---- train step (base_model.py:402-465, 486-510) -----------------------------------------------
def is_d_var(name):
    return name.startswith('discriminator/')


def is_trainable(name):
    return not name.endswith('/u')


def train_step(params, opt_state, inputs, hp, noise, d_indices_pre, d_indices_post, step, mode='train'):
    """One sess.run(train_op): D Adam update, then G Adam update against the *updated* D (against the pre-update D when
    hp.joint_gan_optimization: no control dependency / read replacement, base_model.py:498-505).

    params: dict name -> tensor (leaf, requires_grad irrelevant; copied).  opt_state: dict with 'm','v' dicts and
    't_d','t_g' counters (Adam's own beta-power accumulators, one optimizer per network).
    d_indices_pre / d_indices_post: clip-index draws for the pre- and post-update discriminator_fn calls
    (base_model.py:414-419).
    Spectral-norm semantics (ops.py:1020-1049 + base_model.py:490): both discriminator_fn calls read the same
    pre-assign `u` (a single read op per run); `u <- u_final(pre-update W)` is applied before the D update.
    Returns (new_params, new_opt_state, info).
    """
    P = {k: v.detach().clone().requires_grad_(is_trainable(k)) for k, v in params.items()}
    root = savp.Scope(P)
    gvs = root.sub('generator')
    dvs = root.sub('discriminator')
    lr = learning_rate(hp, step)
    kl_w = kl_weight(hp, step)

    gen_outputs = savp.generator_fn(gvs, inputs, mode, hp, noise)
    has_d = bool(hp.video_sn_gan_weight or hp.video_sn_vae_gan_weight or
                 hp.image_sn_gan_weight or hp.image_sn_vae_gan_weight or
                 hp.images_sn_gan_weight or hp.images_sn_vae_gan_weight)
    info = OrderedDict()
    new_params = {k: v.detach().clone() for k, v in params.items()}
    m, v_ = dict(opt_state['m']), dict(opt_state['v'])
    t_d, t_g = opt_state['t_d'], opt_state['t_g']

    if has_d:
        sn_state = {}
        detached = OrderedDict((k, val.detach()) for k, val in gen_outputs.items())
        d_out = savp.discriminator_fn(dvs, inputs, detached, mode, hp, d_indices_pre, sn_state=sn_state)
        outputs = OrderedDict(list(detached.items()) + list(d_out.items()))
        d_losses = discriminator_loss_fn(hp, inputs, outputs)
        d_loss = total_loss(d_losses)
        d_names = [k for k in P if is_d_var(k) and is_trainable(k)]
        grads = torch.autograd.grad(d_loss, [P[k] for k in d_names], allow_unused=True)
        t_d += 1
        for k, g in zip(d_names, grads):
            if g is None:
                continue
            new_params[k], m[k], v_[k] = tf_ops.adam_update(new_params[k], g, m[k], v_[k], lr, hp.beta1, hp.beta2, t_d)
        info['d_loss'] = float(d_loss.detach())
        info['d_losses'] = OrderedDict((k, float(l.detach())) for k, (l, w) in d_losses.items())
        info['d_grads'] = {k: g for k, g in zip(d_names, grads) if g is not None}
        # post-update discriminator on the (attached) generator outputs, pre-assign u
        P2 = dict(P)
        if not hp.joint_gan_optimization:          # base_model.py:498-501: replace_read_ops(g_loss_post, d_vars) only when sequential
            for k in d_names:
                P2[k] = new_params[k]
        d_out_post = savp.discriminator_fn(savp.Scope(P2).sub('discriminator'), inputs, gen_outputs, mode, hp,
                                           d_indices_post, sn_state=None)
        outputs_post = OrderedDict(list(gen_outputs.items()) + list(d_out_post.items()))
        for k, u in sn_state.items():
            new_params[k] = u.reshape(new_params[k].shape)
    else:
        outputs_post = gen_outputs

    g_losses = generator_loss_fn(hp, inputs, outputs_post, kl_w)
    g_loss = total_loss(g_losses)
    g_names = [k for k in P if k.startswith('generator/') and is_trainable(k)]
    grads = torch.autograd.grad(g_loss, [P[k] for k in g_names], allow_unused=True)
    t_g += 1
    for k, g in zip(g_names, grads):
        if g is None:
            continue
        new_params[k], m[k], v_[k] = tf_ops.adam_update(new_params[k], g, m[k], v_[k], lr, hp.beta1, hp.beta2, t_g)
    info['g_loss'] = float(g_loss.detach())
    info['g_losses'] = OrderedDict((k, float(l.detach())) for k, (l, w) in g_losses.items())
    info['g_grads'] = {k: g for k, g in zip(g_names, grads) if g is not None}
    info['gen_images'] = gen_outputs['gen_images'].detach()
    if 'gen_images_enc' in gen_outputs:
        info['gen_images_enc'] = gen_outputs['gen_images_enc'].detach()
    new_state = {'m': m, 'v': v_, 't_d': t_d, 't_g': t_g}
    return new_params, new_state, info


def init_opt_state(params):
    return {'m': {k: torch.zeros_like(v) for k, v in params.items() if is_trainable(k)},
            'v': {k: torch.zeros_like(v) for k, v in params.items() if is_trainable(k)},
            't_d': 0, 't_g': 0}
