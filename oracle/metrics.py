"""CPU restatement of the reference's evaluation metrics and best-of-N sampling loop -- TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py).  PARITY UNPINNED: TensorFlow cannot run here and the reference ships no tests for this path.

* mse / psnr / ssim follow /root/reference/video_prediction/metrics.py:5-15, which delegate to the un-vendored dependency
  tensorflow-gpu>=1.9.0 (requirements.txt:1): tf.image.psnr(a, b, 1.0) and tf.image.ssim(a, b, 1.0).  Their published
  algorithm (tensorflow/python/ops/image_ops_impl.py, r1.9: psnr, _fspecial_gauss, _ssim_helper, _ssim_per_channel, ssim) is
  restated below: 11x11 Gaussian window sigma 1.5 built as a softmax of -(x^2+y^2)/(2 sigma^2), depthwise 'VALID' filtering,
  k1 = 0.01, k2 = 0.03, SSIM = mean over pixels of luminance*cs per channel, then mean over channels.
* lpips (metrics.py:17-24) needs external AlexNet weights -> out of scope; eval_diversity likewise.
* eval_outputs_and_metrics restates base_model.py:132-227 (deterministic branch :163-168, sampling fold :170-226).
"""
from collections import OrderedDict

import torch


def mse(a, b):
    """metrics.py:5-6: mean squared difference over the last three (H, W, C) axes."""
    return ((a - b) ** 2).mean(dim=(-3, -2, -1))


def psnr(a, b, max_val=1.0):
    """metrics.py:9-10 -> tf.image.psnr: 20 log10(max_val) - 10 log10(mse)."""
    m = mse(a, b)
    return 20.0 * torch.log10(torch.tensor(max_val, dtype=m.dtype)) - 10.0 * torch.log10(m)


def _fspecial_gauss(size, sigma, dtype):
    """image_ops_impl._fspecial_gauss: softmax over the flattened window of -(x^2 + y^2) / (2 sigma^2)."""
    coords = torch.arange(size, dtype=dtype) - (size - 1) / 2.0
    g = coords ** 2 * (-0.5 / (sigma * sigma))
    g = g.reshape(1, -1) + g.reshape(-1, 1)
    return torch.softmax(g.reshape(-1), dim=0).reshape(size, size)


def ssim(a, b, max_val=1.0, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    """metrics.py:13-14 -> tf.image.ssim.  a, b [..., H, W, C]; returns [...]."""
    lead = a.shape[:-3]
    H, W, C = a.shape[-3:]
    x = a.reshape(-1, H, W, C).permute(0, 3, 1, 2).reshape(-1, 1, H, W)
    y = b.reshape(-1, H, W, C).permute(0, 3, 1, 2).reshape(-1, 1, H, W)
    k = _fspecial_gauss(filter_size, filter_sigma, x.dtype).reshape(1, 1, filter_size, filter_size)
    f = lambda t: torch.nn.functional.conv2d(t, k)                     # depthwise, padding 'VALID'
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    mean0, mean1 = f(x), f(y)
    num0 = mean0 * mean1 * 2.0
    den0 = mean0 ** 2 + mean1 ** 2
    luminance = (num0 + c1) / (den0 + c1)
    num1 = f(x * y) * 2.0
    den1 = f(x ** 2 + y ** 2)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)                        # compensation = 1.0
    val = (luminance * cs).mean(dim=(-2, -1)).reshape(-1, C)           # per image, per channel
    return val.mean(dim=-1).reshape(lead)


METRIC_FNS = (('psnr', psnr), ('mse', mse), ('ssim', ssim))            # base_model.py:119-124 without lpips


def metrics_fn(images, gen_images, context_frames):
    """base_model.py:113-130: means over the future frames.  images [T,B,H,W,C], gen_images [T-1,B,H,W,C] (time-major)."""
    future = images.shape[0] - context_frames
    target, pred = images[-future:], gen_images[-future:]
    return OrderedDict((name, fn(target, pred).mean()) for name, fn in METRIC_FNS)


def eval_outputs_and_metrics(images, gen_images_samples, context_frames, deterministic=False):
    """base_model.py:132-227.  gen_images_samples: list of gen_images [T-1,B,H,W,C], one per drawn sample (a single entry
    for a deterministic model).  Returns (eval_outputs, eval_metrics) with the reference's keys (lpips / diversity omitted)."""
    future = images.shape[0] - context_frames
    target = images[-future:]
    outs, mets = OrderedDict(), OrderedDict()
    outs['eval_images'] = images
    if deterministic:                                                   # :163-168
        gen = gen_images_samples[0]
        for name, fn in METRIC_FNS:
            m = fn(target, gen[-future:])
            for sfx in ('min', 'avg', 'max'):
                mets['eval_%s/%s' % (name, sfx)] = m
        outs['eval_gen_images'] = gen
        return outs, mets
    num = len(gen_images_samples)
    gen0 = gen_images_samples[0]
    B = images.shape[1]
    st = {}
    for name, _ in METRIC_FNS:                                           # initializer :201-210
        for sfx in ('min', 'sum', 'max'):
            st['g_%s/%s' % (name, sfx)] = torch.zeros_like(gen0)
        st['%s/min' % name] = torch.full((future, B), float('inf'), dtype=gen0.dtype)
        st['%s/sum' % name] = torch.zeros(future, B, dtype=gen0.dtype)
        st['%s/max' % name] = torch.full((future, B), float('-inf'), dtype=gen0.dtype)

    def where_axis1(cond, x, y):                                         # :170-171 (cond over the batch axis)
        shape = (1, -1) + (1,) * (x.dim() - 2)
        return torch.where(cond.reshape(shape), x, y)
    for gen in gen_images_samples:                                       # accum_gen_images_and_metrics_fn :176-198
        pred = gen[-future:]
        for name, fn in METRIC_FNS:
            m = fn(target, pred)                                         # [future, B]
            crit = m.mean(dim=0)                                         # sort_criterion :173-174
            cmin = crit < st['%s/min' % name].mean(dim=0)
            cmax = crit > st['%s/max' % name].mean(dim=0)
            st['%s/min' % name] = where_axis1(cmin, m, st['%s/min' % name])
            st['%s/sum' % name] = m + st['%s/sum' % name]
            st['%s/max' % name] = where_axis1(cmax, m, st['%s/max' % name])
            st['g_%s/min' % name] = where_axis1(cmin, gen, st['g_%s/min' % name])
            st['g_%s/sum' % name] = gen + st['g_%s/sum' % name]
            st['g_%s/max' % name] = where_axis1(cmax, gen, st['g_%s/max' % name])
    for name, _ in METRIC_FNS:                                           # :215-221
        outs['eval_gen_images_%s/min' % name] = st['g_%s/min' % name]
        outs['eval_gen_images_%s/avg' % name] = st['g_%s/sum' % name] / float(num)
        outs['eval_gen_images_%s/max' % name] = st['g_%s/max' % name]
        mets['eval_%s/min' % name] = st['%s/min' % name]
        mets['eval_%s/avg' % name] = st['%s/sum' % name] / float(num)
        mets['eval_%s/max' % name] = st['%s/max' % name]
    return outs, mets
