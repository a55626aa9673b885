"""CPU pin of an identity the ConvLSTM gate convolution's backward pass can use (gap 1; not built yet).

The cell input is [x | tile(z) | h] (savp_model.py:436-444, rnn_ops.py:144-146): z is constant over the plane, so its gradient is the
sum over pixels of the data gradient's z channels -- and those 8 channels push the DGRAD's column count over a tile boundary
(72 / 136 / 264 -> 128 / 192 / 384 computed).  With SAME zero padding a tap (u, v) of a 5x5 kernel sees the tiled z only where the
shifted pixel is inside the image, i.e. on a sub-rectangle of the gate gradient; the 25 sub-rectangles are unions of 25 fixed REGIONS
(row class x column class, classes {0, 1, middle, H-2, H-1}), so

    dz[n, c] = sum_{regions r} sum_k  R[n, r, k] * Weff[r, c, k],   Weff[r] = sum_{taps (u,v) whose rectangle contains r} W[u, v, c, k]

with R = per-region sums of the gate gradient.  One pass over the gate gradient (which the batched weight gradient reads anyway) plus a
tiny GEMM replaces 8 output channels of every per-timestep DGRAD."""
import numpy as np
import torch

from oracle import tf_ops as TF


def _classes(n):
    """class index of every coordinate: 0, 1, 2 (middle), 3 (n-2), 4 (n-1); needs n >= 4 so that the classes are disjoint"""
    c = np.full(n, 2)
    c[0], c[1], c[n - 2], c[n - 1] = 0, 1, 3, 4
    return c


def _valid_classes(offset, n):
    """classes of the output coordinates y for which y + offset lies inside [0, n)"""
    ok = [(0 <= y + offset < n) for y in range(n)]
    cls = _classes(n)
    out = set(cls[y] for y in range(n) if ok[y])
    # a class is either entirely valid or entirely invalid for |offset| <= 2 -- that is what makes 25 regions enough
    for y in range(n):
        assert ok[y] == (cls[y] in out)
    return out


def test_tiled_z_gradient_from_region_sums_equals_autograd():
    torch.manual_seed(0)
    for (N, H, W, f, nz) in [(2, 8, 8, 4, 3), (1, 6, 9, 2, 2), (2, 4, 4, 3, 1)]:
        Cin, Cout, k = f + nz + f, 4 * f, 5
        x = torch.randn(N, H, W, f, dtype=torch.float64)
        h = torch.randn(N, H, W, f, dtype=torch.float64)
        z = torch.randn(N, nz, dtype=torch.float64, requires_grad=True)
        Wt = torch.randn(k, k, Cin, Cout, dtype=torch.float64)
        a = torch.cat([x, z[:, None, None, :].expand(N, H, W, nz), h], dim=-1)
        gates = TF.conv2d(a, Wt, (1, 1), 'SAME')
        dg = torch.randn_like(gates)
        (gates * dg).sum().backward()
        # region sums of the gate gradient: R[n, ry, rx, k]
        ry, rx = _classes(H), _classes(W)
        R = torch.zeros(N, 5, 5, Cout, dtype=torch.float64)
        for y in range(H):
            for xx in range(W):
                R[:, ry[y], rx[xx]] += dg[:, y, xx]
        # effective weights per region: the taps whose valid rectangle contains the region
        Wz = Wt[:, :, f:f + nz, :]                                   # [5, 5, nz, Cout]
        Weff = torch.zeros(5, 5, nz, Cout, dtype=torch.float64)
        for u in range(k):
            vy = _valid_classes(u - 2, H)
            for v in range(k):
                vx = _valid_classes(v - 2, W)
                for cy in vy:
                    for cx in vx:
                        Weff[cy, cx] += Wz[u, v]
        dz = torch.einsum('nyxk,yxck->nc', R, Weff)
        assert torch.allclose(dz, z.grad, rtol=1e-10, atol=1e-10), (N, H, W, float((dz - z.grad).abs().max()))
        # the same region sums give the z rows of the weight gradient: dW[u, v, c, k] = sum_n z[n, c] * S_uv[n, k]
        a2 = a.detach().clone().requires_grad_(False)
        Wt2 = Wt.clone().requires_grad_(True)
        (TF.conv2d(a2, Wt2, (1, 1), 'SAME') * dg).sum().backward()
        for u in range(k):
            vy = _valid_classes(u - 2, H)
            for v in range(k):
                vx = _valid_classes(v - 2, W)
                S = sum(R[:, cy, cx] for cy in vy for cx in vx)     # [N, Cout]
                dW = torch.einsum('nc,nk->ck', z.detach(), S)
                assert torch.allclose(dW, Wt2.grad[u, v, f:f + nz], rtol=1e-10, atol=1e-10)
