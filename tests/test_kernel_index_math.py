"""CPU restatements of the index algebra the special-case convolution kernels are built on (csrc/conv_s2dgrad.hip, csrc/conv_thin.hip),
checked against torch's convolutions: the tap / shift tables are easy to get wrong by one and the kernels themselves only run on a GPU."""
import numpy as np
import torch


def _conv3d_nhwc(x, w, stride, pad):
    """x [N,D,H,W,Cx], w [kd,kh,kw,Cx,Cy] -> y [N,Do,Ho,Wo,Cy] (cross-correlation, symmetric padding)."""
    y = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3), w.permute(4, 3, 0, 1, 2), stride=stride, padding=pad)
    return y.permute(0, 2, 3, 4, 1)


def test_stride2_dgrad_phase_decomposition_matches_autograd():
    """conv_s2dgrad.hip: dx[z, Y, X] = sum over depth taps a and the 2 x 2 taps of the pixel's phase of dy[z + pd - a, oy, ox] W[a, u, v],
    with u_j = py ? 2 j : 2 j + 1, patch row = ry + (py + 1 - u_j) / 2 + 1 relative to a patch whose row 0 is dy row Y0 / 2 - 1."""
    rng = np.random.default_rng(0)
    N, D, H, W, Cx, Cy, kd, pd = 1, 5, 12, 20, 3, 4, 4, 1
    x = torch.tensor(rng.standard_normal((N, D, H, W, Cx)), requires_grad=True)
    w = torch.tensor(rng.standard_normal((kd, 4, 4, Cx, Cy)))
    y = _conv3d_nhwc(x, w, (1, 2, 2), (pd, 1, 1))
    dy = torch.tensor(rng.standard_normal(tuple(y.shape)))
    (y * dy).sum().backward()
    Do, Ho, Wo = y.shape[1:4]
    dyn, wn = dy.numpy(), w.numpy()
    TR, TC = 16, 32                                                # the kernel's tile
    got = np.zeros((N, D, H, W, Cx))
    for Y0 in range(0, H, TR):
        for X0 in range(0, W, TC):
            oy0, ox0 = Y0 // 2 - 1, X0 // 2 - 1                    # dy coordinates of patch pixel (0, 0)
            for py in (0, 1):
                for px in (0, 1):
                    ut = [2 * j if py else 2 * j + 1 for j in range(2)]
                    vt = [2 * j if px else 2 * j + 1 for j in range(2)]
                    shr = [(py + 1 - u) // 2 + 1 for u in ut]
                    shc = [(px + 1 - v) // 2 + 1 for v in vt]
                    for ry in range(TR // 2):
                        for cx in range(TC // 2):
                            Y, X = Y0 + py + 2 * ry, X0 + px + 2 * cx
                            if Y >= H or X >= W:
                                continue
                            for z in range(D):
                                for a in range(kd):
                                    od = z + pd - a
                                    if od < 0 or od >= Do:
                                        continue
                                    for ju in range(2):
                                        for jv in range(2):
                                            lr, lc = ry + shr[ju], cx + shc[jv]
                                            assert 0 <= lr < TR // 2 + 2 and 0 <= lc < TC // 2 + 2      # inside the staged patch
                                            oy, ox = oy0 + lr, ox0 + lc
                                            if 0 <= oy < Ho and 0 <= ox < Wo:
                                                got[:, z, Y, X, :] += dyn[:, od, oy, ox, :] @ wn[a, ut[ju], vt[jv]].T
    assert np.allclose(got, x.grad.numpy(), atol=1e-10)


def test_thin_dgrad_is_fprop_with_mirrored_taps():
    """conv_thin.hip serves the DGRAD of a 3x3(x3) stride-1 pad-1 convolution with the FPROP kernel: slot s reads the source at
    offset s - 1 per axis and multiplies by weight tap TAPS - 1 - s of the packed WD[c_out][tap][c_src] matrix."""
    rng = np.random.default_rng(1)
    for kd in (1, 3):
        N, D, H, W, Cx, Cy = 2, (1 if kd == 1 else 4), 6, 7, 5, 3
        x = torch.tensor(rng.standard_normal((N, D, H, W, Cx)), requires_grad=True)
        w = torch.tensor(rng.standard_normal((kd, 3, 3, Cx, Cy)))
        y = _conv3d_nhwc(x, w, 1, (kd // 2, 1, 1))
        dy = torch.tensor(rng.standard_normal(tuple(y.shape)))
        (y * dy).sum().backward()
        taps = 9 * kd
        wd = w.numpy().reshape(taps, Cx, Cy).transpose(1, 0, 2)     # WD [Cx][tap][Cy] as savp_pack_weights lays it out
        dyp = np.pad(dy.numpy(), ((0, 0), (kd // 2, kd // 2), (1, 1), (1, 1), (0, 0)))
        got = np.zeros((N, D, H, W, Cx))
        for s in range(taps):
            a, u, v = s // 9, (s % 9) // 3, s % 3                   # the kernel's slot -> (plane, row, column) shift
            src = dyp[:, a:a + D, u:u + H, v:v + W, :]              # source pixel = output pixel + shift - 1
            got += np.einsum('ndhwo,co->ndhwc', src, wd[:, taps - 1 - s, :])
        assert np.allclose(got, x.grad.numpy(), atol=1e-10)


def test_thin_wgrad_row_decode_covers_every_tap_channel_once():
    """conv_thin.hip WGRAD: accumulator row m of 32-row tile i is (tap 4 (2 i + (m >> 4)) + ((m & 15) >> 2), channel m & 3)."""
    for kd in (1, 3):
        taps = 9 * kd
        ng = (taps + 3) // 4
        nt32 = (ng + 1) // 2
        seen = set()
        for i in range(nt32):
            for m in range(32):
                tap, c = 4 * (2 * i + (m >> 4)) + ((m & 15) >> 2), m & 3
                if tap < taps:
                    assert (tap, c) not in seen
                    seen.add((tap, c))
        assert seen == {(t, c) for t in range(taps) for c in range(4)}
