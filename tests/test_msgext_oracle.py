"""Anchors for the deformable-convolution restatement (torchvision is absent here, so the
reference's DeformConv2d cannot be imported: parity is UNPINNED, see oracle/torch_port.py).
Degenerate cases whose answer is known independently of torchvision. CPU only."""
import torch
import torch.nn.functional as F

from oracle import torch_port as O


def _rand(*s, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*s, generator=g)


def test_zero_offsets_equal_plain_convolution():
    x, w, b = _rand(2, 6, 9, 11), _rand(5, 6, 3, 3, seed=1), _rand(5, seed=2)
    off = torch.zeros(2, 18, 9, 11)
    assert torch.allclose(O.deform_conv2d_ref(x, off, w, b), F.conv2d(x, w, b, padding=1), rtol=1e-5, atol=1e-5)


def test_integer_offsets_shift_the_taps():
    """Every tap displaced by (+1 row, -2 cols): integer positions, no interpolation."""
    x, w = _rand(1, 4, 10, 12), _rand(3, 4, 3, 3, seed=3)
    off = torch.zeros(1, 18, 10, 12)
    off[:, 0::2] = 1.0
    off[:, 1::2] = -2.0
    got = O.deform_conv2d_ref(x, off, w, None)
    # brute-force definition: sample x at (y-1+ky+1, x-1+kx-2), zero outside
    H, W = 10, 12
    want = torch.zeros(1, 3, H, W)
    for y in range(H):
        for xx in range(W):
            for ky in range(3):
                for kx in range(3):
                    sy, sx = y - 1 + ky + 1, xx - 1 + kx - 2
                    if 0 <= sy < H and 0 <= sx < W:
                        want[0, :, y, xx] += w[:, :, ky, kx] @ x[0, :, sy, sx]
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-5)


def test_half_pixel_offsets_average_neighbours():
    """A constant +0.5 horizontal offset samples the mean of two horizontal neighbours; positions in
    (-1, 0) and (W-1, W) still see the one existing neighbour with weight 0.5, positions >= W are zero."""
    x, w = _rand(1, 2, 6, 7), _rand(2, 2, 3, 3, seed=4)
    H, W = 6, 7
    off = torch.zeros(1, 18, H, W)
    off[:, 1::2] = 0.5
    xp = F.pad(x, (1, 1))                              # columns -1 .. W
    xm = 0.5 * (xp[:, :, :, :-1] + xp[:, :, :, 1:])    # values at -0.5, 0.5, ..., W-0.5  (W+1 samples)
    xm = F.pad(xm, (0, 1))                             # W+0.5 lies outside: zero
    want = F.conv2d(F.pad(xm, (0, 0, 1, 1)), w)        # taps x-1+kx+0.5 -> sample index x+kx
    assert torch.allclose(O.deform_conv2d_ref(x, off, w, None), want, rtol=1e-5, atol=1e-5)
