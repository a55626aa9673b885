"""CPU checks of oracle/metrics.py (restatement of tf.image.psnr / tf.image.ssim and base_model.py:132-227)."""
import numpy as np
import torch

from oracle import metrics as M


def _ssim_direct(x, y):
    """Independent window-by-window evaluation of the published SSIM definition (11x11 Gaussian, sigma 1.5, VALID)."""
    H, W, C = x.shape
    k = M._fspecial_gauss(11, 1.5, torch.float64).numpy()
    x, y = x.numpy(), y.numpy()
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    per_c = []
    for c in range(C):
        vals = []
        for i in range(H - 10):
            for j in range(W - 10):
                px, py = x[i:i + 11, j:j + 11, c], y[i:i + 11, j:j + 11, c]
                mx, my = (k * px).sum(), (k * py).sum()
                sxy, sxx, syy = (k * px * py).sum(), (k * px * px).sum(), (k * py * py).sum()
                lum = (2 * mx * my + c1) / (mx * mx + my * my + c1)
                cs = (2 * sxy - 2 * mx * my + c2) / (sxx + syy - mx * mx - my * my + c2)
                vals.append(lum * cs)
        per_c.append(np.mean(vals))
    return float(np.mean(per_c))


def test_gaussian_window_is_a_normalised_separable_gaussian():
    k = M._fspecial_gauss(11, 1.5, torch.float64)
    assert abs(float(k.sum()) - 1.0) < 1e-12
    g = torch.exp(-(torch.arange(11, dtype=torch.float64) - 5) ** 2 / (2 * 1.5 ** 2))
    g = g / g.sum()
    assert float((k - g[:, None] * g[None, :]).abs().max()) < 1e-15


def test_ssim_psnr_mse_definitions():
    torch.manual_seed(0)
    a = torch.rand(2, 3, 16, 18, 3, dtype=torch.float64)
    b = (a + 0.05 * torch.randn_like(a)).clamp(0, 1)
    assert float((M.ssim(a, a) - 1).abs().max()) < 1e-12                      # identity
    assert float((M.ssim(a, b) - M.ssim(b, a)).abs().max()) < 1e-12           # symmetry
    assert abs(float(M.ssim(a, b)[1, 2]) - _ssim_direct(a[1, 2], b[1, 2])) < 1e-12
    m = ((a - b) ** 2).reshape(2, 3, -1).mean(-1)
    assert float((M.mse(a, b) - m).abs().max()) < 1e-15
    assert float((M.psnr(a, b) + 10 * torch.log10(m)).abs().max()) < 1e-12     # max_val = 1


def test_best_of_n_fold_picks_per_sequence_extremes():
    """base_model.py:170-198: the sample with the smallest / largest time-mean metric is kept whole, per batch element."""
    torch.manual_seed(1)
    T, B, H, W, C, ctx = 6, 3, 16, 16, 1, 2
    images = torch.rand(T, B, H, W, C, dtype=torch.float64)
    gens = [(images[1:] + s * 0.05 * torch.randn(T - 1, B, H, W, C, dtype=torch.float64)).clamp(0, 1) for s in (3, 1, 2, 0.5)]
    outs, mets = M.eval_outputs_and_metrics(images, gens, ctx)
    fut = T - ctx
    for name, fn in M.METRIC_FNS:
        per = torch.stack([fn(images[-fut:], g[-fut:]) for g in gens])       # [S, fut, B]
        crit = per.mean(1)                                                   # [S, B]
        for b in range(B):
            lo, hi = int(crit[:, b].argmin()), int(crit[:, b].argmax())
            assert torch.equal(mets['eval_%s/min' % name][:, b], per[lo, :, b])
            assert torch.equal(mets['eval_%s/max' % name][:, b], per[hi, :, b])
            assert torch.equal(outs['eval_gen_images_%s/min' % name][:, b], gens[lo][:, b])
            assert torch.equal(outs['eval_gen_images_%s/max' % name][:, b], gens[hi][:, b])
        assert float((mets['eval_%s/avg' % name] - per.mean(0)).abs().max()) < 1e-12
    det_o, det_m = M.eval_outputs_and_metrics(images, gens[:1], ctx, deterministic=True)
    assert torch.equal(det_m['eval_psnr/min'], det_m['eval_psnr/max'])
