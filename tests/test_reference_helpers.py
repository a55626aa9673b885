"""CPU checks of the test infrastructure itself: the fp64 tap-loop convolution reference used by the tuning-table parity test
(tests/gpu_checks.py:_taps_ref) against torch's own conv3d + autograd."""
import numpy as np
import pytest
import torch


@pytest.mark.parametrize('case', [
    # N, (D,H,W), Cx, Cy, k, s, p
    (2, (1, 9, 8), 5, 7, (1, 5, 5), (1, 1, 1), (0, 2, 2)),
    (2, (1, 8, 10), 4, 6, (1, 6, 6), (1, 2, 2), (0, 2, 2)),
    (1, (5, 8, 8), 3, 4, (4, 4, 4), (1, 2, 2), (1, 1, 1)),
    (2, (6, 6, 6), 3, 5, (4, 4, 4), (2, 2, 2), (1, 1, 1)),
    (3, (1, 1, 1), 16, 5, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (1, (1, 7, 9), 2, 3, (1, 4, 4), (1, 2, 2), (0, 1, 1)),
])
def test_taps_reference_matches_torch_conv_autograd(case):
    from tests import gpu_checks as G
    from video_prediction_amd import lib
    N, dhw, Cx, Cy, k, s, p = case
    rng = np.random.default_rng(0)
    out = [-(-i // st) for i, st in zip(dhw, s)]                    # SAME output size
    pa = [max((o - 1) * st + kk - i - pb, 0) for o, st, kk, i, pb in zip(out, s, k, dhw, p)]
    x = torch.tensor(rng.standard_normal((N,) + dhw + (Cx,)), requires_grad=True)
    w = torch.tensor(rng.standard_normal(k + (Cx, Cy)), requires_grad=True)
    y = G._ref_conv(x, w, k, s, p, pa)
    dy = torch.tensor(rng.standard_normal(tuple(y.shape)))
    (y * dy).sum().backward()
    assert tuple(y.shape[1:4]) == tuple(out)
    assert torch.allclose(G._taps_ref(lib.CONV_FPROP, x.detach(), w.detach(), dy, k, s, p), y.detach(), atol=1e-10)
    assert torch.allclose(G._taps_ref(lib.CONV_DGRAD, x.detach(), w.detach(), dy, k, s, p), x.grad, atol=1e-10)
    assert torch.allclose(G._taps_ref(lib.CONV_WGRAD, x.detach(), w.detach(), dy, k, s, p), w.grad, atol=1e-10)
