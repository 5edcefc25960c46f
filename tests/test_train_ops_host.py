"""Host-side pieces of gencomm_amd.train_ops that need no GPU: weight re-layouts checked against their defining formulas."""
import itertools

import torch

from gencomm_amd import train_ops as T


def test_stride2_subpixel_weights_follow_their_definition():
    """s2_subpixel_weights: dx[ci][2u + a][2v + b] = sum_{co, ty, tx} W[co][ci][a + 1 - 2 ty][b + 1 - 2 tx] dy[co][u + ty][v + tx]
    (taps outside 0..2 are zero) -- entry by entry, and through the convolution it stands for against autograd."""
    g = torch.Generator().manual_seed(3)
    cout, cin = 5, 3
    w = torch.randn(cout, cin, 3, 3, generator=g)
    wp = T.s2_subpixel_weights(w)
    assert tuple(wp.shape) == (cin, 2, 2, cout, 2, 2) and wp.is_contiguous()
    for ci, a, b, co, ty, tx in itertools.product(range(cin), (0, 1), (0, 1), range(cout), (0, 1), (0, 1)):
        ky, kx = a + 1 - 2 * ty, b + 1 - 2 * tx
        want = float(w[co, ci, ky, kx]) if 0 <= ky <= 2 and 0 <= kx <= 2 else 0.0
        assert float(wp[ci, a, b, co, ty, tx]) == want
    # the sub-pixel convolution over dy reproduces the stride-2 layer's input gradient
    x = torch.randn(2, cin, 8, 12, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(2, cout, 4, 6, generator=g, dtype=torch.float64)
    (torch.nn.functional.conv2d(x, w.double(), stride=2, padding=1) * dy).sum().backward()
    k = wp.double().reshape(cin * 4, cout, 2, 2)                       # rows (ci, a, b) as output channels of a 2x2 convolution
    sub = torch.nn.functional.conv2d(torch.nn.functional.pad(dy, (0, 1, 0, 1)), k)   # [n, ci * 4, Ho, Wo], taps (ty, tx) reach to the right / below
    dx = torch.nn.functional.pixel_shuffle(sub, 2)
    assert torch.allclose(dx, x.grad, atol=1e-12)
