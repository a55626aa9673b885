"""Known-answer tests of the CPU oracle.  The reference ships no tests; its only executable statements of intent are
two docstring identities (ops.py:652-679 upsample_conv2d, ops.py:799-817 conv_pool2d), reproduced here at their
stated shapes and atol, plus algebraic properties that follow from the reference code."""
import numpy as np
import torch

from oracle import ops as O
from oracle import savp as OS
from oracle import tf_ops as TF


def test_upsample_conv2d_docstring_identity():
    # ops.py:656-679: inputs [4,8,8,64], 3x3, 32 filters, stride 2, atol 1e-5
    torch.manual_seed(0)
    x = torch.randn(4, 8, 8, 64)
    kernel = torch.randn(3, 3, 64, 32) * 0.05
    bias = torch.randn(32)
    out = O.upsample_conv2d(x, kernel, bias, strides=(2, 2))
    x_up = O.upsample2d(x, (2, 2), padding='VALID')
    out_up = O.conv2d(x_up, kernel, bias, strides=(1, 1), padding='FULL')
    same = O.pad2d_paddings([8, 8], [3, 3], strides=(1, 1), padding='SAME')
    full = O.pad2d_paddings([8, 8], [3, 3], strides=(1, 1), padding='FULL')
    crop_top = (2 - 2 % 2) // 2 + full[1][1] - same[1][1]
    crop_left = (2 - 2 % 2) // 2 + full[2][1] - same[2][1]
    out_up = out_up[:, crop_top:crop_top + 16, crop_left:crop_left + 16, :]
    assert out.shape == (4, 16, 16, 32)
    assert np.allclose(out.numpy(), out_up.numpy(), atol=1e-5)


def test_conv_pool2d_docstring_identity():
    # ops.py:803-817: inputs [4,16,16,32], 3x3, 64 filters, stride 2, atol 1e-5
    torch.manual_seed(1)
    x = torch.randn(4, 16, 16, 32)
    kernel = torch.randn(3, 3, 32, 64) * 0.05
    bias = torch.randn(64)
    out = O.conv_pool2d(x, kernel, bias, strides=(2, 2))
    conv = O.conv2d(x, kernel, bias, strides=(1, 1))
    pooled = O.pool2d(conv, pool_size=(2, 2), strides=(2, 2), pool_mode='avg')
    assert np.allclose(out.numpy(), pooled.numpy(), atol=1e-5)


def test_same_padding_arithmetic_matches_reference_restatement():
    # ops.py:100-107 vs TF's ceil(in/s) rule
    for in_size in range(1, 40):
        for k in (1, 3, 4, 5, 6):
            for s in (1, 2, 3):
                before, after = TF.same_pad(in_size, k, s)
                pads = O.pad2d_paddings([in_size, in_size], [k, k], strides=(s, s), padding='SAME')
                assert pads[1] == [before, after]


def test_conv2d_same_against_explicit_loops():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 5, 6, 2))
    w = rng.standard_normal((4, 3, 2, 3))
    y = TF.conv2d(torch.tensor(x), torch.tensor(w), (2, 1), 'SAME').numpy()
    pt, _ = TF.same_pad(5, 4, 2)
    pl, _ = TF.same_pad(6, 3, 1)
    ref = np.zeros_like(y)
    for oy in range(y.shape[1]):
        for ox in range(y.shape[2]):
            for u in range(4):
                for v in range(3):
                    iy, ix = oy * 2 - pt + u, ox - pl + v
                    if 0 <= iy < 5 and 0 <= ix < 6:
                        ref[0, oy, ox] += x[0, iy, ix] @ w[u, v]
    assert np.allclose(y, ref, atol=1e-12)


def test_conv2d_transpose_is_adjoint_of_conv2d():
    rng = np.random.default_rng(1)
    x = torch.tensor(rng.standard_normal((2, 8, 8, 3)), requires_grad=True)
    w = torch.tensor(rng.standard_normal((6, 6, 3, 5)))
    y = TF.conv2d(x, w, (2, 2), 'SAME')
    g = torch.tensor(rng.standard_normal(tuple(y.shape)))
    (y * g).sum().backward()
    xt = TF.conv2d_transpose(g, w, [2, 8, 8, 3], (2, 2), 'SAME')
    assert np.allclose(xt.numpy(), x.grad.numpy(), atol=1e-10)


def test_cdna_identity_kernel_returns_input():
    # zero dense output => kernels == identity_kernel => CDNA returns the image (savp_model.py:551,968-980)
    rng = np.random.default_rng(2)
    img = torch.tensor(rng.random((2, 8, 8, 3)))
    k = torch.as_tensor(OS.identity_kernel((5, 5)))[None, :, :, None].repeat(2, 1, 1, 4)
    k = torch.relu(k - OS.RELU_SHIFT) + OS.RELU_SHIFT
    k = k / k.sum(dim=(1, 2), keepdim=True)
    outs = OS.apply_cdna_kernels(img, k)
    for o in outs:
        assert np.allclose(o.numpy(), img.numpy(), atol=1e-9)


def test_symmetric_pad_repeats_edge():
    x = torch.arange(12.0).reshape(1, 3, 4, 1)
    p = TF.pad_symmetric(x, ((2, 2), (1, 1)))
    assert p.shape == (1, 7, 6, 1)
    assert p[0, 1, 1, 0] == x[0, 0, 0, 0] and p[0, 0, 1, 0] == x[0, 1, 0, 0]
    assert p[0, 5, 0, 0] == x[0, 2, 0, 0] and p[0, 6, 5, 0] == x[0, 1, 3, 0]


def test_convlstm_zero_weights_property():
    # zero conv kernel, gamma=1/beta=0: c' = IN(c*sigmoid(1)), h' = tanh(c')*0.5  
    rng = np.random.default_rng(3)
    F = 4
    c = torch.tensor(rng.standard_normal((2, 6, 6, F)))
    h = torch.tensor(rng.standard_normal((2, 6, 6, F)))
    x = torch.tensor(rng.standard_normal((2, 6, 6, 3)))
    P = {'basic_conv2dlstm_cell/kernel': torch.zeros(5, 5, 3 + F, 4 * F, dtype=torch.float64),
         'basic_conv2dlstm_cell/input_transform_forget_output/gamma': torch.ones(4 * F, dtype=torch.float64),
         'basic_conv2dlstm_cell/input_transform_forget_output/beta': torch.zeros(4 * F, dtype=torch.float64),
         'basic_conv2dlstm_cell/state/gamma': torch.ones(F, dtype=torch.float64),
         'basic_conv2dlstm_cell/state/beta': torch.zeros(F, dtype=torch.float64)}
    hn, (cn, _) = OS.conv_lstm_cell(OS.Scope(P), x, (c, h), F)
    sig1 = 1.0 / (1.0 + np.exp(-1.0))
    ref_c = O.fused_instance_norm(c * sig1, P['basic_conv2dlstm_cell/state/gamma'], P['basic_conv2dlstm_cell/state/beta'])
    assert np.allclose(cn.numpy(), ref_c.numpy(), atol=1e-9)
    assert np.allclose(hn.numpy(), (torch.tanh(ref_c) * 0.5).numpy(), atol=1e-9)


def test_instance_norm_is_per_sample_per_channel_biased():
    rng = np.random.default_rng(4)
    x = torch.tensor(rng.standard_normal((3, 5, 7, 4)))
    g, b = torch.tensor(rng.standard_normal(4)), torch.tensor(rng.standard_normal(4))
    y = O.fused_instance_norm(x, g, b)
    mu = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(1, 2), keepdim=True)
    assert np.allclose(y.numpy(), ((x - mu) / torch.sqrt(var + 1e-6) * g + b).numpy(), atol=1e-10)


def test_basic_lstm_unroll_matches_an_independent_lstm_cell():
    """oracle.savp.basic_lstm_unroll (BasicLSTMCell: gate order i, j, f, o, forget_bias 1 added before the sigmoid) against
    torch.nn.LSTMCell (gate order i, f, g, o, no forget bias) with the weights re-mapped."""
    import torch
    from oracle import savp as OS
    torch.manual_seed(0)
    T, B, nin, units = 5, 3, 7, 6
    kernel = torch.randn(nin + units, 4 * units, dtype=torch.float64) * 0.3
    bias = torch.randn(4 * units, dtype=torch.float64) * 0.1
    xs = torch.randn(T, B, nin, dtype=torch.float64)
    hs = OS.basic_lstm_unroll(kernel, bias, xs)
    cell = torch.nn.LSTMCell(nin, units).double()
    i, j, f, o = [kernel[:, k * units:(k + 1) * units] for k in range(4)]
    bi, bj, bf, bo = [bias[k * units:(k + 1) * units] for k in range(4)]
    w = torch.cat([i, f, j, o], dim=1)                                     # torch order: i, f, g(=j), o
    with torch.no_grad():
        cell.weight_ih.copy_(w[:nin].t()); cell.weight_hh.copy_(w[nin:].t())
        cell.bias_ih.copy_(torch.cat([bi, bf + 1.0, bj, bo])); cell.bias_hh.zero_()
        h = torch.zeros(B, units, dtype=torch.float64); c = torch.zeros_like(h)
        ref = []
        for t in range(T):
            h, c = cell(xs[t], (h, c))
            ref.append(h)
    assert float((hs - torch.stack(ref)).abs().max()) < 1e-12


def test_kl_between_gaussians_reduces_to_the_standard_normal_case():
    import torch
    from oracle import train as OT
    torch.manual_seed(1)
    mu, ls = torch.randn(4, 3, 8, dtype=torch.float64), torch.randn(4, 3, 8, dtype=torch.float64)
    z = torch.zeros_like(mu)
    assert abs(float(OT.kl_loss(mu, ls) - OT.kl_loss(mu, ls, z, z))) < 1e-12           # losses.py:57-67
    assert abs(float(OT.kl_loss(mu, ls, mu, ls))) < 1e-12                               # KL(p || p) = 0
    mu2, ls2 = torch.randn_like(mu), torch.randn_like(ls)
    p = torch.distributions.Normal(mu, torch.exp(0.5 * ls)); q = torch.distributions.Normal(mu2, torch.exp(0.5 * ls2))
    ref = torch.distributions.kl_divergence(p, q).sum(-1).mean()
    assert abs(float(OT.kl_loss(mu, ls, mu2, ls2) - ref)) < 1e-10


def test_prior_fn_and_e_rnn_shapes_and_context_dependence():
    """savp_model.py:54-85: the learned prior sees only the context frames; later frames cannot influence it."""
    import numpy as np
    import torch
    from oracle import savp as OS
    rng = np.random.default_rng(0)
    T, B, H, nz, nef = 6, 2, 16, 4, 8

    class HP(object):
        nef = 8; n_layers = 3; norm_layer = 'instance'; use_e_rnn = True; rnn = 'lstm'; context_frames = 2; sequence_length = T
    P = {}
    cin = 6
    for i, cout in enumerate((nef, 2 * nef, 4 * nef)):
        P['layer_%d/conv2d/kernel' % (i + 1)] = rng.standard_normal((4, 4, cin, cout)) * 0.1
        P['layer_%d/conv2d/bias' % (i + 1)] = np.zeros(cout)
        if i:
            P['layer_%d/InstanceNorm/gamma' % (i + 1)] = np.ones(cout); P['layer_%d/InstanceNorm/beta' % (i + 1)] = np.zeros(cout)
        cin = cout
    P['layer_4/dense/kernel'] = rng.standard_normal((4 * nef, 4 * nef)) * 0.2; P['layer_4/dense/bias'] = np.zeros(4 * nef)
    P['lstm/rnn/basic_lstm_cell/kernel'] = rng.standard_normal((8 * nef, 16 * nef)) * 0.2
    P['lstm/rnn/basic_lstm_cell/bias'] = np.zeros(16 * nef)
    for head in ('z_mu', 'z_log_sigma_sq'):
        P[head + '/dense/kernel'] = rng.standard_normal((4 * nef, nz)) * 0.2; P[head + '/dense/bias'] = np.zeros(nz)
    vs = OS.Scope({k: torch.tensor(v, dtype=torch.float64) for k, v in P.items()})
    images = torch.tensor(rng.random((T, B, H, H, 3)))
    try:
        out = OS.prior_fn(vs, {'images': images}, HP)
    except KeyError as ex:                                   # encoder variable naming differs: report, do not hide
        raise AssertionError('oracle encoder expects other variable names: %s' % ex)
    assert out['zs_mu'].shape == (T - 1, B, nz) and out['zs_log_sigma_sq'].shape == (T - 1, B, nz)
    images2 = images.clone(); images2[2:] = torch.rand_like(images2[2:])
    out2 = OS.prior_fn(vs, {'images': images2}, HP)
    assert torch.equal(out['zs_mu'], out2['zs_mu'])          # frames beyond the context do not enter
    post = OS.posterior_fn(vs, {'images': images}, HP)
    post2 = OS.posterior_fn(vs, {'images': images2}, HP)
    assert torch.equal(post['zs_mu'][0], post2['zs_mu'][0]) and not torch.equal(post['zs_mu'][-1], post2['zs_mu'][-1])
