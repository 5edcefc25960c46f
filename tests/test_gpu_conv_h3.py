"""The general convolution on the f16 matrix pipe with three-term operands (csrc/conv_h3_kernels.h: the default kernel of
gencomm_conv2d_fwd for 3x3 / 2x2 shapes with Cin >= 16, Cin % 8 == 0 and >= 32 GEMM rows) against float64 torch on the same
inputs, beside the exact-fp32 kernel it replaces (GENCOMM_MODE_ARITH = 1): its error must stay at the level of fp32 accumulation --
products are accurate to 2^-26 -- on ordinary activations, on gradient-sized inputs (the running power-of-two scale has to lift
them), on inputs whose magnitude grows along the channel axis (the scale has to drop mid-sum and the accumulators follow), and on
weights whose rows differ by many octaves (per-row weight scale).  Reference call sites: base_bev_backbone.py:40-92 (3x3 stride 1 / 2,
ConvTranspose2d with kernel == stride), downsample_conv.py:17-24, the 1x1 heads of heter_model_baseline_w_gencomm_stage1.py:137-142."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _both(modes, fn):
    """fn() under the default arithmetic (must run conv2d_h3_kernel) and under exact fp32 (must not)."""
    from gencomm_amd import _lib
    out = []
    for arith in ("split", "f32"):
        modes(arith=arith)
        with _lib.kernel_log() as kl:
            out.append(fn())
        ran = any("conv2d_h3" in k for k in kl.counts)
        assert ran == (arith == "split"), (arith, dict(kl.counts))
    modes(arith="split")
    return out


def _check(y3, y32, ref, what):
    scale = ref.abs().max().item()
    e3 = (y3.double().cpu() - ref).abs().max().item() / scale
    e32 = (y32.double().cpu() - ref).abs().max().item() / scale
    assert torch.isfinite(y3).all(), what
    assert e3 <= max(2.0 * e32, 4e-7) and e3 <= 3e-6, (what, e3, e32)
    return e3, e32


@pytest.mark.parametrize("shape", [
    (2, 64, 64, 40, 52, 1),      # backbone block, 8-row tiles off (few tiles)
    (4, 64, 64, 128, 64, 1),     # 8-row tiles on: two accumulator groups per wave
    (1, 24, 70, 19, 37, 1),      # Cin = 1.5 chunks (zero-padded tail), ragged output rows / columns, 70 rows = 3 partial blocks
    (2, 128, 256, 16, 20, 1),    # many chunks
    (2, 64, 128, 33, 47, 2),     # stride 2, odd sizes
    (1, 384, 256, 32, 48, 1),    # shrink convolution: 24 chunks
    (3, 16, 32, 9, 9, 1),        # the smallest eligible shape
])
def test_conv3x3_three_term_vs_float64(modes, shape):
    from gencomm_amd import train_ops as T
    N, Cin, Cout, H, W, stride = shape
    g = torch.Generator().manual_seed(Cin * 7 + Cout + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y3, y32 = _both(modes, lambda: T.conv2d(xd, wd, bd, 1, stride))
    _check(y3, y32, ref, shape)


@pytest.mark.parametrize("case", ["gradient_sized", "growing_along_channels", "row_scales", "zeros_then_data", "huge"])
def test_three_term_ranges(modes, case):
    from gencomm_amd import train_ops as T
    g = torch.Generator().manual_seed(len(case))
    N, Cin, Cout, H, W = 2, 96, 64, 24, 40
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 30.0
    if case == "gradient_sized":
        x = x * 3e-7
    elif case == "growing_along_channels":          # every 16-channel chunk is 64 x larger than the one before: the scale drops five times mid-sum
        x = x * (64.0 ** (torch.arange(Cin) // 16).float()).view(1, Cin, 1, 1) * 1e-6
    elif case == "row_scales":                      # output rows 2^-20 .. 2^20 apart
        w = w * (2.0 ** (torch.arange(Cout).float() - 32.0) ** 1).clamp(2.0 ** -20, 2.0 ** 20).view(Cout, 1, 1, 1)
    elif case == "zeros_then_data":                 # the first chunks are exactly zero (no scale yet), one sample is all zero
        x[:, :48] = 0.0
        x[1] = 0.0
    elif case == "huge":
        x = x * 1e30
        w = w * 1e-30
    xd, wd = x.to(DEV), w.to(DEV)
    y3, y32 = _both(modes, lambda: T.conv2d(xd, wd, None, 1))
    if case == "row_scales":                        # judged per output channel: each row has its own magnitude
        ref = F.conv2d(x.double(), w.double(), None, padding=1)
        for co in (0, 17, 40, 63):
            _check(y3[:, co], y32[:, co], ref[:, co], (case, co))
    else:
        _check(y3, y32, F.conv2d(x.double(), w.double(), None, padding=1), case)
    if case == "zeros_then_data":
        assert float(y3[1].abs().max()) == 0.0


@pytest.mark.parametrize("cin,cout,hw", [(64, 256, (40, 52)), (128, 64, (130, 252)), (256, 14, (32, 32))])
def test_conv1x1_stays_on_the_exact_fp32_kernel(cin, cout, hw):
    """1 x 1 layers (Linear layers, heads) are NOT taken by the three-term kernel (per staged pixel a ninth of a 3x3 layer's matrix work:
    measured slower, profiles/r5_h3_1x1_ab.txt): the kernel log must not show it, the result is the fp32 kernel's."""
    from gencomm_amd import _lib, train_ops as T
    g = torch.Generator().manual_seed(cin + cout)
    n, (H, W) = 2, hw
    x = torch.randn(n, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double())
    with _lib.kernel_log() as kl:
        y = T.conv2d(x.to(DEV), w.to(DEV), b.to(DEV), 0)
    assert not any("conv2d_h3" in k for k in kl.counts), dict(kl.counts)
    assert torch.allclose(y.double().cpu(), ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("shape", [(2, 64, 64, 32, 48), (1, 128, 64, 20, 36), (2, 256, 128, 16, 16)])
def test_input_gradients_three_term_vs_float64(modes, shape):
    """The input-gradient convolutions of the training step: stride 1 (flipped / transposed weights laid out by the prepare launch) and
    stride 2 (2 x 2 sub-pixel form, GEMM rows = 4 x channels) against float64 autograd."""
    from gencomm_amd import train_ops as T
    N, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(H + W + Cin)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    for stride in (1, 2):
        x = torch.randn(N, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x, w.double(), None, stride=stride, padding=1)
        dy = torch.randn(y.shape, generator=g, dtype=torch.float64) * 1e-4        # gradient-sized
        y.backward(dy)
        dyd, wd = dy.float().to(DEV), w.to(DEV)
        d3, d32 = _both(modes, lambda: T.conv2d_dgrad_strided(dyd, wd, 1, stride, (H, W)))
        assert d3.shape == x.grad.shape
        _check(d3, d32, x.grad, (shape, stride))


def test_deblock_transposed_convolution_stays_on_the_exact_fp32_kernel():
    """ConvTranspose2d with kernel == stride (base_bev_backbone.py:75-83) runs as a 1 x 1 GEMM with Cout s^2 rows + pixel shuffle: fp32 kernel,
    prepared buffer without the three-term form."""
    from gencomm_amd import _lib
    from gencomm_amd.runtime import conv2d_prepare, ptr, stream_ptr
    g = torch.Generator().manual_seed(3)
    N, Cin, Cout, H, W, s = 2, 128, 64, 20, 28, 2
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, s, s, generator=g) / Cin ** 0.5
    ref = F.conv_transpose2d(x.double(), w.double(), None, stride=s)
    xd, wd = x.to(DEV), w.to(DEV)
    ss = torch.stack([torch.ones(Cout), torch.zeros(Cout)]).to(DEV)
    prepared = conv2d_prepare(wd, Cin, Cout, s, s, 1, xd.device)
    assert prepared.numel() == Cin * Cout * s * s
    y = torch.empty(N, Cout, H * s, W * s, device=DEV)
    with _lib.kernel_log() as kl:
        _lib.check(_lib.lib().gencomm_conv2d_fwd(ptr(xd), ptr(prepared), ptr(ss[0]), ptr(ss[1]), ptr(y), N, Cin, H, W, Cout, 1, 1, 1, 0, 0, s, Cout, 0,
                                                 stream_ptr(xd.device)), "gencomm_conv2d_fwd")
    assert not any("conv2d_h3" in k for k in kl.counts)
    assert torch.allclose(y.double().cpu(), ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("shape,case", [
    ((4, 64, 64, 128, 256), "plain"),            # the stage-1 leg's 64-channel level: aligned 128-bit loads, 8 tiles per workgroup
    ((2, 128, 256, 32, 64), "gradient_sized"),   # dY ~ 1e-7: the running scale lifts it
    ((2, 72, 40, 22, 40), "growing"),            # |dY| and |X| grow down the image: both scales drop mid-sum and the 144 accumulators follow
    ((1, 64, 64, 17, 70), "ragged"),             # partial tiles in both directions, scalar loads
    ((2, 64, 64, 24, 32), "zeros_first"),        # the first tiles of a workgroup are exactly zero (no scale yet)
])
def test_wide_weight_gradient_three_term_vs_float64(modes, shape, case):
    """wgrad3x3_h3_kernel (csrc/wgrad_h3_kernels.h: the 3x3 stride-1 weight gradient on the f16 pipe, six matrix instructions per product
    block, horizontal tap shifts formed from one aligned record with v_alignbit) against float64, beside the exact-fp32 kernel it replaces."""
    from gencomm_amd import _lib, train_ops as T
    N, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(H * W + Cin)
    x = torch.randn(N, Cin, H, W, generator=g)
    dy = torch.randn(N, Cout, H, W, generator=g)
    if case == "gradient_sized":
        dy = dy * 1e-7
    elif case == "growing":
        ramp = (10.0 ** torch.linspace(-5, 1, H)).view(1, 1, H, 1)
        x, dy = x * ramp, dy * ramp * 1e-3
    elif case == "zeros_first":
        x[:, :, :8] = 0.0
        dy[:, :, :6] = 0.0
    ref_w = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy.double(), stride=1, padding=1)
    ref_b = dy.double().sum((0, 2, 3))
    xd, dyd = x.to(DEV), dy.to(DEV)
    res = []
    for arith in ("split", "f32"):
        modes(arith=arith)
        with _lib.kernel_log() as kl:
            dw, db = T.conv2d_wgrad(dyd, xd, 3, 1, True, 1)
        assert any("wgrad3x3_h3" in k for k in kl.counts) == (arith == "split"), dict(kl.counts)
        res.append((dw.double().cpu(), db.double().cpu()))
    modes(arith="split")
    scale = ref_w.abs().max().item()
    e3, e32 = (res[0][0] - ref_w).abs().max().item() / scale, (res[1][0] - ref_w).abs().max().item() / scale
    assert torch.isfinite(res[0][0]).all() and e3 <= max(2.0 * e32, 4e-7) and e3 <= 3e-6, (shape, case, e3, e32)
    for o in range(0, Cout, 7):                    # per output channel as well (each row has its own magnitude)
        err = (res[0][0][o] - ref_w[o]).abs().max().item()
        assert err <= 3e-6 * ref_w[o].abs().max().item() + 1e-12 * scale, (shape, case, o, err)
    assert ((res[0][1] - ref_b).abs() <= 1e-6 * dy.double().abs().sum((0, 2, 3)) + 1e-30).all()


def test_random_shapes_three_term_convolution_and_weight_gradient_vs_float64(modes):
    """Thirty seeded random shapes (channel counts off the 16 / 32 / 64 grids, maps off the tile grids, both strides) through the three-term
    convolution, its input gradient and the three-term weight gradient against float64: the same error bound as the exact-fp32 kernels."""
    import random
    from gencomm_amd import train_ops as T
    rnd = random.Random(20261005)
    g = torch.Generator().manual_seed(77)
    for it in range(30):
        N = rnd.randint(1, 3)
        Cin = rnd.choice([16, 24, 32, 40, 64, 72, 136])
        Cout = rnd.choice([32, 40, 64, 96, 200])
        H, W = rnd.randint(5, 70), rnd.randint(5, 70)
        stride = rnd.choice([1, 1, 2])
        mag = 10.0 ** rnd.uniform(-6, 3)
        x = torch.randn(N, Cin, H, W, generator=g) * mag
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * (10.0 ** rnd.uniform(-3, 1))
        ref = F.conv2d(x.double(), w.double(), None, stride=stride, padding=1)
        y = T.conv2d(x.to(DEV), w.to(DEV), None, 1, stride)
        scale = ref.abs().max().item()
        err = (y.double().cpu() - ref).abs().max().item()
        assert torch.isfinite(y).all() and err <= 3e-6 * scale, ("fwd", it, (N, Cin, Cout, H, W, stride), err / scale)
        if stride == 1:
            dy = torch.randn(ref.shape, generator=g) * (10.0 ** rnd.uniform(-7, 0))
            xr = x.double().requires_grad_(True)
            wr = w.double().requires_grad_(True)
            F.conv2d(xr, wr, None, stride=1, padding=1).backward(dy.double())
            dx = T.conv2d_dgrad(dy.to(DEV), w.to(DEV), 1)
            sx = xr.grad.abs().max().item()
            assert (dx.double().cpu() - xr.grad).abs().max().item() <= 3e-6 * sx, ("dgrad", it, (N, Cin, Cout, H, W))
            if Cin >= 32:
                dw, _ = T.conv2d_wgrad(dy.to(DEV), x.to(DEV), 3, 1, False, 1)
                sw = wr.grad.abs().max().item()
                assert (dw.double().cpu() - wr.grad).abs().max().item() <= 3e-6 * sw, ("wgrad", it, (N, Cin, Cout, H, W))
