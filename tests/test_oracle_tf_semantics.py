"""Known-answer pins of oracle/tf_ops.py and oracle/metrics.py that do not go through torch's convolution kernels.

The reference ships no tests and TensorFlow cannot run here (parity unpinned).  The strongest pin available
without TF: every TF kernel restated in oracle/tf_ops.py is compared with a naive NumPy triple loop written directly from the
TensorFlow documentation of that op (SAME padding rule, conv2d_transpose as the gradient of conv2d, SYMMETRIC pad, depthwise
channel order c*M+m, fused_batch_norm's biased variance, LSTMCell gate order i,j,f,o, Adam's lr_t) plus the hand values that
documentation prints (tf.pad SYMMETRIC example) or that follow in closed form (PSNR of a constant offset, SSIM of constant
images).  Odd sizes, stride 2 and even kernels are included because that is where SAME/transpose alignment goes wrong.
"""
import math

import numpy as np
import pytest
import torch

from oracle import metrics as OM
from oracle import tf_ops as T

RNG = np.random.default_rng(123)


def _same(in_size, k, s):
    """TF docs (tf.nn.convolution, 'SAME'): out = ceil(in / s); pad = max((out - 1) * s + k - in, 0); before = pad // 2."""
    out = (in_size + s - 1) // s
    pad = max((out - 1) * s + k - in_size, 0)
    return out, pad // 2


def naive_conv2d(x, w, s, padding):
    N, H, W, Ci = x.shape
    kh, kw, _, Co = w.shape
    if padding == 'SAME':
        Ho, pt = _same(H, kh, s[0])
        Wo, pl = _same(W, kw, s[1])
    else:
        Ho, Wo, pt, pl = (H - kh) // s[0] + 1, (W - kw) // s[1] + 1, 0, 0
    y = np.zeros((N, Ho, Wo, Co))
    for oy in range(Ho):
        for ox in range(Wo):
            for u in range(kh):
                for v in range(kw):
                    iy, ix = oy * s[0] - pt + u, ox * s[1] - pl + v
                    if 0 <= iy < H and 0 <= ix < W:
                        y[:, oy, ox, :] += x[:, iy, ix, :] @ w[u, v]
    return y


def naive_conv2d_transpose(x, w, out_hw, s):
    """tf.nn.conv2d_transpose docs: 'the transpose (gradient) of conv2d'.  x [N,h,w,Ci], w [kh,kw,Co,Ci], SAME: scatter every
    input pixel through the forward conv's index map iy = oy*s - pad_before + u."""
    N, h, ww, Ci = x.shape
    kh, kw, Co, _ = w.shape
    H, W = out_hw
    Ho, pt = _same(H, kh, s[0])
    Wo, pl = _same(W, kw, s[1])
    assert (Ho, Wo) == (h, ww)
    y = np.zeros((N, H, W, Co))
    for oy in range(h):
        for ox in range(ww):
            for u in range(kh):
                for v in range(kw):
                    iy, ix = oy * s[0] - pt + u, ox * s[1] - pl + v
                    if 0 <= iy < H and 0 <= ix < W:
                        y[:, iy, ix, :] += x[:, oy, ox, :] @ w[u, v].T
    return y


@pytest.mark.parametrize('H,W,k,s,padding', [(7, 9, (3, 3), (1, 1), 'SAME'), (7, 9, (5, 5), (2, 2), 'SAME'),
                                             (8, 6, (4, 4), (2, 2), 'SAME'), (9, 7, (6, 6), (2, 2), 'SAME'),
                                             (8, 8, (4, 4), (2, 2), 'VALID'), (7, 5, (3, 2), (2, 1), 'SAME')])
def test_conv2d_same_and_valid_against_naive_loops(H, W, k, s, padding):
    x = RNG.standard_normal((2, H, W, 3))
    w = RNG.standard_normal(k + (3, 4))
    got = T.conv2d(torch.tensor(x), torch.tensor(w), strides=s, padding=padding).numpy()
    np.testing.assert_allclose(got, naive_conv2d(x, w, s, padding), rtol=0, atol=1e-12)


@pytest.mark.parametrize('H,W,k,s', [(8, 8, 6, 2), (7, 9, 6, 2), (6, 10, 4, 2), (5, 5, 3, 1), (9, 9, 5, 2)])
def test_conv2d_transpose_same_alignment_against_naive_scatter(H, W, k, s):
    h, w_ = (H + s - 1) // s, (W + s - 1) // s
    x = RNG.standard_normal((2, h, w_, 3))
    w = RNG.standard_normal((k, k, 5, 3))
    got = T.conv2d_transpose(torch.tensor(x), torch.tensor(w), (2, H, W, 5), (s, s)).numpy()
    np.testing.assert_allclose(got, naive_conv2d_transpose(x, w, (H, W), (s, s)), rtol=0, atol=1e-12)


def test_conv2d_transpose_is_the_adjoint_of_conv2d():
    """<conv2d(a), b> == <a, conv2d_transpose(b)> for the same SAME/stride-2 geometry (the defining property in the TF docs)."""
    a = RNG.standard_normal((1, 9, 7, 2))
    w = RNG.standard_normal((6, 6, 2, 3))
    y = naive_conv2d(a, w, (2, 2), 'SAME')
    b = RNG.standard_normal(y.shape)
    bt = T.conv2d_transpose(torch.tensor(b), torch.tensor(w), (1, 9, 7, 2), (2, 2)).numpy()
    assert abs((y * b).sum() - (a * bt).sum()) < 1e-9


def test_conv3d_valid_stride_against_naive_loops():
    x = RNG.standard_normal((1, 5, 6, 6, 2))
    w = RNG.standard_normal((4, 4, 4, 2, 3))
    s = (1, 2, 2)
    got = T.conv3d(torch.tensor(x), torch.tensor(w), strides=s, padding='VALID').numpy()
    Do, Ho, Wo = (5 - 4) // 1 + 1, (6 - 4) // 2 + 1, (6 - 4) // 2 + 1
    ref = np.zeros((1, Do, Ho, Wo, 3))
    for od in range(Do):
        for oy in range(Ho):
            for ox in range(Wo):
                patch = x[:, od:od + 4, oy * 2:oy * 2 + 4, ox * 2:ox * 2 + 4, :]
                ref[:, od, oy, ox, :] = np.einsum('nduvc,duvco->no', patch, w)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)


def test_depthwise_conv2d_channel_order_is_c_times_M_plus_m():
    """tf.nn.depthwise_conv2d docs: output[b, i, j, k * channel_multiplier + q] = sum filter[di, dj, k, q] * input[b, i+di, j+dj, k]."""
    x = RNG.standard_normal((2, 6, 7, 3))
    w = RNG.standard_normal((5, 5, 3, 4))
    got = T.depthwise_conv2d(torch.tensor(x), torch.tensor(w), padding='VALID').numpy()
    ref = np.zeros((2, 2, 3, 12))
    for i in range(2):
        for j in range(3):
            for k in range(3):
                for q in range(4):
                    ref[:, i, j, k * 4 + q] = (x[:, i:i + 5, j:j + 5, k] * w[:, :, k, q]).sum(axis=(1, 2))
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)


def test_pad_symmetric_matches_the_tf_documentation_example():
    """tf.pad docs: t = [[1, 2, 3], [4, 5, 6]], paddings = [[1, 1], [2, 2]], mode SYMMETRIC ->
    [[2, 1, 1, 2, 3, 3, 2], [2, 1, 1, 2, 3, 3, 2], [5, 4, 4, 5, 6, 6, 5], [5, 4, 4, 5, 6, 6, 5]]."""
    t = torch.tensor([[1., 2., 3.], [4., 5., 6.]]).reshape(1, 2, 3, 1)
    got = T.pad_symmetric(t, ((1, 1), (2, 2))).reshape(4, 7).numpy()
    exp = np.array([[2, 1, 1, 2, 3, 3, 2], [2, 1, 1, 2, 3, 3, 2], [5, 4, 4, 5, 6, 6, 5], [5, 4, 4, 5, 6, 6, 5]], dtype=np.float32)
    assert np.array_equal(got, exp)


def test_avg_pool_valid_hand_values():
    x = torch.arange(16, dtype=torch.float64).reshape(1, 4, 4, 1)
    got = T.avg_pool(x, (2, 2), (2, 2)).reshape(2, 2).numpy()
    assert np.array_equal(got, np.array([[2.5, 4.5], [10.5, 12.5]]))
    got = T.avg_pool(x, (2, 2), (1, 1)).reshape(3, 3).numpy()
    assert got[0, 0] == 2.5 and got[2, 2] == 12.5 and got[1, 1] == 7.5


def test_fused_batch_norm_training_uses_the_biased_variance():
    x = torch.tensor([1., 2., 3., 6.], dtype=torch.float64).reshape(1, 2, 2, 1)
    # mean 3, biased variance (4+1+0+9)/4 = 3.5 (unbiased would be 14/3)
    y = T.fused_batch_norm_training(x, torch.tensor([2.0], dtype=torch.float64), torch.tensor([0.5], dtype=torch.float64), 1e-6)
    exp = (np.array([1., 2., 3., 6.]) - 3.0) / math.sqrt(3.5 + 1e-6) * 2.0 + 0.5
    np.testing.assert_allclose(y.reshape(-1).numpy(), exp, rtol=0, atol=1e-14)


def test_lstm_cell_gate_order_and_forget_bias_by_hand():
    """tf.nn.rnn_cell.LSTMCell: [x, h] @ kernel + bias, gates i, j, f, o; c' = sigmoid(f + forget_bias) c + sigmoid(i) tanh(j)."""
    nz = 1
    x, h, c = torch.tensor([[0.5]]), torch.tensor([[-0.25]]), torch.tensor([[2.0]])
    kernel = torch.tensor([[0.1, 0.2, 0.3, 0.4], [1.0, -1.0, 0.5, -0.5]])
    bias = torch.tensor([0.01, 0.02, 0.03, 0.04])
    h1, (c1, _) = T.lstm_cell(x, c, h, kernel, bias, forget_bias=1.0)
    i = 0.5 * 0.1 - 0.25 * 1.0 + 0.01
    j = 0.5 * 0.2 + 0.25 * 1.0 + 0.02
    f = 0.5 * 0.3 - 0.25 * 0.5 + 0.03
    o = 0.5 * 0.4 + 0.25 * 0.5 + 0.04
    sig = lambda v: 1.0 / (1.0 + math.exp(-v))
    c_exp = sig(f + 1.0) * 2.0 + sig(i) * math.tanh(j)
    assert abs(float(c1) - c_exp) < 1e-6 and abs(float(h1) - sig(o) * math.tanh(c_exp)) < 1e-6
    assert nz == 1


def test_gru_cell_gate_order_and_update_convention_by_hand():
    """tf.contrib.rnn.GRUCell: [r, u] = sigmoid([x, h] @ gates_kernel + gates_bias) with r FIRST; the candidate sees [x, r * h];
    h' = u * h + (1 - u) * c (the update gate keeps the OLD state, unlike some other GRU write-ups)."""
    x, h = torch.tensor([[0.5]]), torch.tensor([[-0.25]])
    gk = torch.tensor([[0.3, -0.7], [1.1, 0.4]])            # rows: x, h; columns: r, u
    gb = torch.tensor([1.0, 1.0])                            # GRUCell's gate bias is initialised to 1
    ck = torch.tensor([[0.9], [-1.3]])
    cb = torch.tensor([0.05])
    h1, state = T.gru_cell(x, h, gk, gb, ck, cb)
    sig = lambda v: 1.0 / (1.0 + math.exp(-v))
    r = sig(0.5 * 0.3 - 0.25 * 1.1 + 1.0)
    u = sig(0.5 * -0.7 - 0.25 * 0.4 + 1.0)
    c = math.tanh(0.5 * 0.9 + (r * -0.25) * -1.3 + 0.05)
    assert abs(float(h1) - (u * -0.25 + (1.0 - u) * c)) < 1e-7 and state is h1
    # zero input, zero state, zero candidate bias: the state stays zero whatever the gates say
    z = torch.zeros(2, 3)
    h2, _ = T.gru_cell(z, z, torch.randn(6, 6), torch.ones(6), torch.randn(6, 3), torch.zeros(3))
    assert float(h2.abs().max()) == 0.0


def test_adam_first_two_steps_by_hand():
    """tf.train.AdamOptimizer docs: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); m, v EMA; var -= lr_t m / (sqrt(v) + eps) -- epsilon
    OUTSIDE the bias correction (differs from torch.optim.Adam)."""
    p, m, v = torch.tensor([1.0], dtype=torch.float64), torch.zeros(1, dtype=torch.float64), torch.zeros(1, dtype=torch.float64)
    g = torch.tensor([0.5], dtype=torch.float64)
    lr, b1, b2, eps = 0.1, 0.5, 0.999, 1e-8
    p1, m1, v1 = T.adam_update(p, g, m, v, lr, b1, b2, 1, eps)
    m_e, v_e = 0.5 * 0.5, 0.001 * 0.25
    lr1 = lr * math.sqrt(1 - b2) / (1 - b1)
    assert abs(float(p1) - (1.0 - lr1 * m_e / (math.sqrt(v_e) + eps))) < 1e-15
    p2, m2, v2 = T.adam_update(p1, g, m1, v1, lr, b1, b2, 2, eps)
    m_e2, v_e2 = 0.5 * m_e + 0.5 * 0.5, 0.999 * v_e + 0.001 * 0.25
    lr2 = lr * math.sqrt(1 - b2 ** 2) / (1 - b1 ** 2)
    assert abs(float(p2) - (float(p1) - lr2 * m_e2 / (math.sqrt(v_e2) + eps))) < 1e-15


def test_psnr_closed_form():
    """tf.image.psnr docs: 20 log10(max_val) - 10 log10(mse).  A constant offset of 0.1 -> mse 0.01 -> 20 dB; 0.01 -> 40 dB."""
    a = torch.full((1, 8, 8, 3), 0.3, dtype=torch.float64)
    assert abs(float(OM.psnr(a, a + 0.1)) - 20.0) < 1e-9
    assert abs(float(OM.psnr(a, a + 0.01)) - 40.0) < 1e-9
    assert abs(float(OM.mse(a, a + 0.1)) - 0.01) < 1e-15


def _naive_ssim(a, b, max_val=1.0, size=11, sigma=1.5, k1=0.01, k2=0.03):
    """tf.image.ssim docs / Wang et al. 2004: Gaussian window 11x11 sigma 1.5, VALID windows, per channel
    ssim = (2 mu_a mu_b + c1)(2 cov + c2) / ((mu_a^2 + mu_b^2 + c1)(var_a + var_b + c2)), mean over windows and channels."""
    g = np.exp(-((np.arange(size) - size // 2) ** 2) / (2.0 * sigma ** 2))
    win = np.outer(g, g)
    win /= win.sum()
    c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
    H, W, C = a.shape
    vals = []
    for c in range(C):
        for y in range(H - size + 1):
            for x in range(W - size + 1):
                pa, pb = a[y:y + size, x:x + size, c], b[y:y + size, x:x + size, c]
                ma, mb = (win * pa).sum(), (win * pb).sum()
                va, vb = (win * pa * pa).sum() - ma * ma, (win * pb * pb).sum() - mb * mb
                cov = (win * pa * pb).sum() - ma * mb
                vals.append((2 * ma * mb + c1) * (2 * cov + c2) / ((ma * ma + mb * mb + c1) * (va + vb + c2)))
    return float(np.mean(vals))


def test_ssim_closed_form_and_naive_windows():
    a = torch.full((1, 16, 16, 1), 0.5, dtype=torch.float64)
    b = torch.full((1, 16, 16, 1), 0.6, dtype=torch.float64)
    assert abs(float(OM.ssim(a, a)) - 1.0) < 1e-12
    # constant images: variances and covariance vanish -> only the luminance term (2ab + c1) / (a^2 + b^2 + c1), c1 = 1e-4
    assert abs(float(OM.ssim(a, b)) - (2 * 0.5 * 0.6 + 1e-4) / (0.25 + 0.36 + 1e-4)) < 1e-12
    x = RNG.random((14, 13, 2))
    y = np.clip(x + 0.1 * RNG.standard_normal(x.shape), 0, 1)
    got = float(OM.ssim(torch.tensor(x)[None], torch.tensor(y)[None]))
    assert abs(got - _naive_ssim(x, y)) < 1e-10
