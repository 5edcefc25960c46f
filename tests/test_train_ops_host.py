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


def test_prepared_buffer_sizes_of_the_general_convolution():
    """gencomm_conv2d_prepared_floats (no GPU needed: a size query): shapes the f16-pipe kernel does not take get exactly the fp32 matrix;
    eligible shapes get it + the three-term operand units (7 KiB per (16-channel chunk padded to 4, tap, 64-row block)) + one scale per
    row of the padded blocks; 1x1 layers and transposed convolutions (1x1 GEMMs) stay on the fp32 kernel; bad dims answer -1."""
    from gencomm_amd import _lib
    f = _lib.lib().gencomm_conv2d_prepared_floats

    def want(cin, rows, taps):
        chunks = ((cin + 15) // 16 + 3) // 4 * 4
        blocks = (rows + 63) // 64
        return chunks * taps * blocks * 7168 // 4 + 64 * blocks

    assert f(8, 64, 3, 3, 0) == 8 * 64 * 9                        # fewer than 16 input channels
    assert f(64, 14, 1, 1, 0) == 64 * 14                          # fewer than 32 GEMM rows (detection heads)
    assert f(20, 64, 3, 3, 0) == 20 * 64 * 9                      # Cin not a multiple of 8
    assert f(64, 64, 5, 5, 0) == 64 * 64 * 25                     # no 5x5 form
    assert f(64, 64, 3, 3, 0) == 64 * 64 * 9 + want(64, 64, 9)
    assert f(24, 70, 3, 3, 0) == 24 * 70 * 9 + want(24, 70, 9)
    assert f(256, 128, 3, 3, 2) == 256 * 128 * 9 + want(256, 128, 9)              # input-gradient form: rows = the forward's input channels
    assert f(128, 64, 2, 2, 1) == 128 * 64 * 4                                    # ConvTranspose2d(kernel = stride = 2) runs as a 1x1 GEMM: fp32 kernel
    assert f(128, 256, 1, 1, 0) == 128 * 256                                      # 1x1 / Linear layers: fp32 kernel
    assert f(64, 256, 2, 2, 0) == 64 * 256 * 4 + want(64, 256, 4)                 # sub-pixel form of a stride-2 input gradient
    assert f(0, 64, 3, 3, 0) == -1 and f(64, 64, 3, 3, 3) == -1
