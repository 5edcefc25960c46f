"""Direct tests of the training-path kernels added in round 3, each against float64 torch on the same inputs:
weight gradients on the fp32 matrix cores (conv_wgrad_mfma_kernel through gencomm_conv2d_wgrad: what torch autograd's
convolution_backward computes in the reference's training runs), the elementwise glue of the sampler chain and of the Enhancer's
backward (gencomm_lincomb_fwd, gencomm_ew_slice_fwd, gencomm_nc_scale_fwd, gencomm_nc_dot_fwd), the depthwise kernels, and the
module-level workspaces under two concurrent HIP streams."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape", [(2, 8, 8, 37, 50, 3), (1, 16, 8, 16, 24, 3), (2, 16, 8, 20, 36, 1), (1, 66, 8, 18, 26, 3), (1, 8, 64, 18, 26, 3),
                                   (3, 8, 8, 200, 64, 3),
                                   # enough tiles for several tiles per workgroup (the prefetching tile loop), the last workgroup short
                                   (4, 8, 8, 200, 672, 3), (4, 16, 8, 200, 352, 3), (4, 16, 8, 200, 704, 1)])
def test_weight_gradient_on_the_matrix_cores_vs_float64(shape):
    from gencomm_amd import train_ops as T
    N, Cin, Cout, H, W, K = shape
    g = torch.Generator().manual_seed(H * W + Cin)
    x = torch.randn(N, Cin, H, W, generator=g)
    dy = torch.randn(N, Cout, H, W, generator=g) * torch.logspace(-3, 0, Cout).view(1, Cout, 1, 1)   # gradients of very different size per channel
    ref_w = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, K, K), dy.double(), padding=K // 2)
    ref_b = dy.double().sum((0, 2, 3))
    dw, db = T.conv2d_wgrad(dy.to(DEV), x.to(DEV), K, K // 2, True)
    for o in range(Cout):   # per output channel: the scale differs by 1e3
        err = (dw[o].double().cpu() - ref_w[o]).abs().max().item()
        assert err <= 3e-6 * ref_w[o].abs().max().item() + 1e-9, (shape, o, err)
    # the bias gradient is a cancelling sum: its error is judged against the sum of magnitudes
    assert ((db.double().cpu() - ref_b).abs() <= 1e-6 * dy.double().abs().sum((0, 2, 3))).all()


@pytest.mark.parametrize("shape", [(2, 64, 64, 22, 40, 1, 1), (1, 128, 64, 17, 70, 1, 1), (2, 40, 72, 12, 33, 1, 1),      # stride 1, "same" padding
                                   (2, 64, 64, 32, 48, 2, 1), (1, 64, 128, 19, 37, 2, 1), (2, 128, 256, 16, 24, 2, 0)])   # stride 2 (ZeroPad2d(1) folded / pad 0)
def test_wide_3x3_weight_gradient_vs_float64(shape):
    """The backbone's 64..256-channel 3x3 layers: 64 x 64-channel implicit-GEMM weight gradient (wgrad3x3_wide_kernel), stride 1 and 2,
    channel counts that are not multiples of 64, maps that are not multiples of the 32-column tile."""
    from gencomm_amd import train_ops as T
    N, Cin, Cout, H, W, st, pad = shape
    g = torch.Generator().manual_seed(H * W + Cin + st)
    x = torch.randn(N, Cin, H, W, generator=g)
    Ho, Wo = (H + 2 * pad - 3) // st + 1, (W + 2 * pad - 3) // st + 1
    dy = torch.randn(N, Cout, Ho, Wo, generator=g) * torch.logspace(-2, 0, Cout).view(1, Cout, 1, 1)
    ref_w = torch.nn.grad.conv2d_weight(x.double(), (Cout, Cin, 3, 3), dy.double(), stride=st, padding=pad)
    ref_b = dy.double().sum((0, 2, 3))
    dw, db = T.conv2d_wgrad(dy.to(DEV), x.to(DEV), 3, pad, True, st)
    for o in range(Cout):
        err = (dw[o].double().cpu() - ref_w[o]).abs().max().item()
        assert err <= 3e-6 * ref_w[o].abs().max().item() + 1e-9, (shape, o, err)
    assert ((db.double().cpu() - ref_b).abs() <= 1e-6 * dy.double().abs().sum((0, 2, 3))).all()


def test_narrow_linear_weight_gradient_over_many_points_vs_float64():
    """PillarVFE's Linear 10 -> 64 over (pillars x 32 points) as a 1x1 convolution: routed to the split-K GEMM kernel."""
    from gencomm_amd import train_ops as T
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 10, 2300, 32, generator=g)
    dy = torch.randn(1, 64, 2300, 32, generator=g)
    ref = torch.einsum("nohw,nihw->oi", dy.double(), x.double())
    dw, _ = T.conv2d_wgrad(dy.to(DEV), x.to(DEV), 1, 0, False)
    err = (dw[:, :, 0, 0].double().cpu() - ref).abs().max().item()
    assert err <= 3e-6 * ref.abs().max().item(), err


@pytest.mark.parametrize("C,M,P", [(64, 1500, 32), (5, 77, 32), (3, 40, 20), (4, 33, 7)])
def test_slot_max_matches_torch_max_first_occurrence(C, M, P):
    from gencomm_amd import _lib
    from gencomm_amd.runtime import ptr, stream_ptr
    g = torch.Generator().manual_seed(C + M)
    x = torch.randint(-3, 4, (C, M, P), generator=g).float()          # many ties: the first maximal slot must win
    xd = x.to(DEV)
    out, arg = torch.empty(M, C, device=DEV), torch.empty(M, C, dtype=torch.uint8, device=DEV)
    _lib.check(_lib.lib().gencomm_slot_max_fwd(ptr(xd), ptr(out), ptr(arg), C, M, P, stream_ptr(xd.device)), "gencomm_slot_max_fwd")
    ref = x.max(dim=2)
    assert torch.equal(out.cpu(), ref.values.t().contiguous())
    first = (x == ref.values.unsqueeze(2)).float().argmax(dim=2)
    assert torch.equal(arg.cpu().long(), first.t().contiguous())


def test_sampler_and_enhancer_glue_kernels_vs_torch():
    from gencomm_amd import train_ops as T
    from gencomm_amd.autograd import _lincomb
    g = torch.Generator(device=DEV).manual_seed(5)
    n, hid, H, W = 2, 24, 9, 13
    HW = H * W
    x, y, z = (torch.randn(n, hid, H, W, device=DEV, generator=g) for _ in range(3))
    out = _lincomb(torch.empty_like(x), x, 0.3, y, -1.7, z, 2.5)
    assert torch.allclose(out, 0.3 * x - 1.7 * y + 2.5 * z, rtol=1e-6, atol=1e-6)
    assert torch.equal(_lincomb(x.clone(), x, 1.0), x)
    odd = torch.randn(1027, device=DEV, generator=g)          # tail that is not a multiple of four
    assert torch.allclose(_lincomb(torch.empty_like(odd), odd, 2.0, odd, 1.0), 3.0 * odd, rtol=1e-6)
    v = torch.randn(n, 2 * hid, H, W, device=DEV, generator=g)
    h1, h2 = torch.empty_like(x), torch.empty_like(x)
    T.ew_slice(T.EW_GELU_SPLIT, v, o0=h1, o1=h2, n=n, nch=hid, HW=HW)
    ref = F.gelu(v.double())
    assert torch.allclose(h1.double(), ref[:, :hid], atol=5e-7) and torch.allclose(h2.double(), ref[:, hid:], atol=5e-7)
    gt = torch.empty_like(x)
    T.ew_slice(T.EW_GELU_GATE, x, y, o0=gt, n=n, nch=hid, HW=HW)
    assert torch.allclose(gt.double(), F.gelu(x.double()) * y.double(), atol=2e-6)
    # gate backward against autograd: g = GELU(u) * GELU(v2) with upstream dg
    u, dg = x, z
    ud, vd = u.double().requires_grad_(True), v.double().requires_grad_(True)
    hd = F.gelu(vd)
    (F.gelu(ud) * hd[:, hid:] * dg.double()).sum().backward()
    du, dv = torch.empty_like(u), torch.zeros_like(v)
    T.ew_slice(T.EW_GATE_BWD, u, h2, dg, v, o0=du, o1=dv, n=n, nch=hid, HW=HW, o1_ct=2 * hid, o1_c0=hid)
    assert torch.allclose(du.double(), ud.grad, atol=3e-6) and torch.allclose(dv[:, hid:].double(), vd.grad[:, hid:], atol=3e-6)
    assert float(dv[:, :hid].abs().max()) == 0.0                 # the other half is not touched
    T.ew_slice(T.EW_GELU_BWD, v, y, o0=dv, n=n, nch=hid, HW=HW, o0_ct=2 * hid, o0_c0=0)
    vd2 = v.double().requires_grad_(True)
    (F.gelu(vd2)[:, :hid] * y.double()).sum().backward()
    assert torch.allclose(dv[:, :hid].double(), vd2.grad[:, :hid], atol=3e-6)
    part = T.copy_slice(v, 5, 7)
    assert torch.equal(part, v[:, 5:12])
    big = torch.zeros(n, 40, H, W, device=DEV)
    T.copy_slice(v, 0, 10, big, 30)
    assert torch.equal(big[:, 30:], v[:, :10]) and float(big[:, :30].abs().max()) == 0.0
    a, b = torch.randn(n, hid, device=DEV, generator=g), torch.randn(n, hid, device=DEV, generator=g)
    assert torch.allclose(T.nc_scale(x, a, b), x * a[:, :, None, None] + b[:, :, None, None], rtol=1e-6, atol=1e-6)
    assert torch.allclose(T.nc_dot(x, y).double(), (x.double() * y.double()).sum((2, 3)), rtol=1e-6, atol=1e-5)
    assert torch.allclose(T.nc_dot(x, None).double(), x.double().sum((2, 3)), rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("cin,cout", [(40, 6), (64, 256), (128, 64), (24, 70), (72, 300)])
def test_linear_over_a_large_map(cin, cout):
    """1 x 1 convolutions (the Enhancer's Linear layers, conv_kernels.h) on a map of 32 760 pixels: channel counts below / across the
    64-row tile and the 32-channel chunk, bias, residual and a write into a channel slice, against float64."""
    from gencomm_amd import train_ops as T
    g = torch.Generator().manual_seed(cin + cout)
    n, H, W = 2, 130, 252
    x = torch.randn(n, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g)
    res = torch.randn(n, cout, H, W, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double())
    y = T.conv2d(x.to(DEV), w.to(DEV), b.to(DEV), 0)
    assert torch.allclose(y.double().cpu(), ref, rtol=1e-5, atol=2e-5)
    y = T.conv2d(x.to(DEV), w.to(DEV), b.to(DEV), 0, residual=res.to(DEV))
    assert torch.allclose(y.double().cpu(), ref + res.double(), rtol=1e-5, atol=2e-5)
    buf = torch.full((n, cout + 5, H, W), -3.0, device=DEV)
    T.conv2d(x.to(DEV), w.to(DEV), None, 0, out=buf, out_coff=2)
    assert torch.allclose(buf[:, 2:2 + cout].double().cpu(), F.conv2d(x.double(), w.double()), rtol=1e-5, atol=2e-5)
    assert bool((buf[:, :2] == -3.0).all()) and bool((buf[:, 2 + cout:] == -3.0).all())


@pytest.mark.parametrize("shape", [(2, 40, 72), (3, 19, 37), (1, 200, 704)])
def test_conv3x3_on_16_channel_slices_forward_and_input_gradient(shape):
    """gencomm_conv3x3_c16_fwd (ABI v8): FRFN.partial_conv3 at C = 64 on the UNet's 8-channel exact-fp32 kernel, reading the first 16
    channels of a wider tensor and writing the first 16 of another; transposed = its input gradient from the forward weight."""
    from gencomm_amd import train_ops as T
    n, H, W = shape
    g = torch.Generator().manual_seed(H)
    x = torch.randn(n, 24, H, W, generator=g)
    w = torch.randn(16, 16, 3, 3, generator=g) / 12
    out = torch.full((n, 20, H, W), -5.0, device=DEV)
    T.conv3x3_c16(x.to(DEV), w.to(DEV), out)
    ref = F.conv2d(x[:, :16].double(), w.double(), padding=1)
    assert torch.allclose(out[:, :16].double().cpu(), ref, rtol=1e-5, atol=2e-5) and bool((out[:, 16:] == -5.0).all())
    xd = x[:, :16].double().requires_grad_(True)
    dy = torch.randn(n, 16, H, W, generator=g)
    (F.conv2d(xd, w.double(), padding=1) * dy.double()).sum().backward()
    dx = torch.empty(n, 16, H, W, device=DEV)
    T.conv3x3_c16(dy.to(DEV), w.to(DEV), dx, transposed=True)
    assert torch.allclose(dx.double().cpu(), xd.grad, rtol=1e-5, atol=2e-5)


def test_linear_at_the_metric_geometry():
    """4 x 200 x 706 pixels, 16 -> 64 channels: ragged pixel tiles at the row ends."""
    from gencomm_amd import train_ops as T
    g = torch.Generator().manual_seed(9)
    n, cin, cout, H, W = 4, 16, 64, 200, 706       # HW = 141200 = 1103 full tiles + 16 pixels
    x = torch.randn(n, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 1, 1, generator=g) / 4
    b = torch.randn(cout, generator=g)
    y = T.conv2d(x.to(DEV), w.to(DEV), b.to(DEV), 0)
    assert torch.allclose(y.double().cpu(), F.conv2d(x.double(), w.double(), b.double()), rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("hw", [(9, 13), (8, 12), (33, 64)])   # ragged (one pixel per lane) and HW % 4 == 0 (four pixels per lane, 128-bit accesses)
def test_enhancer_gate_ops_that_recompute_gelu_from_linear1_output(hw):
    """ABI v8: ops 5 / 6 of gencomm_ew_slice_fwd read x2 = GELU(v[:, hid:]) from v itself, every op in both the scalar and the
    four-pixel form; GELU' = Phi + x phi from the library's erf form against float64 autograd."""
    from gencomm_amd import train_ops as T
    g = torch.Generator(device=DEV).manual_seed(hw[1])
    n, hid, (H, W) = 2, 12, hw
    HW = H * W
    u, dg = (torch.randn(n, hid, H, W, device=DEV, generator=g) * s for s in (1.5, 1.0))
    v = torch.randn(n, 2 * hid, H, W, device=DEV, generator=g) * 2.0
    v[0, 0, 0, :4] = torch.tensor([-9.0, 9.0, 0.0, -0.75], device=DEV)      # saturated tails, zero, GELU's minimum
    gt = torch.empty_like(u)
    T.ew_slice(T.EW_GELU2_GATE, u, d=v, o0=gt, n=n, nch=hid, HW=HW, o1_ct=2 * hid, o1_c0=hid)
    assert torch.allclose(gt.double(), F.gelu(u.double()) * F.gelu(v.double())[:, hid:], atol=3e-6)
    ud, vd = u.double().requires_grad_(True), v.double().requires_grad_(True)
    (F.gelu(ud) * F.gelu(vd)[:, hid:] * dg.double()).sum().backward()
    du, dv = torch.empty_like(u), torch.zeros_like(v)
    T.ew_slice(T.EW_GATE_BWD2, u, None, dg, v, o0=du, o1=dv, n=n, nch=hid, HW=HW, o1_ct=2 * hid, o1_c0=hid)
    assert torch.allclose(du.double(), ud.grad, atol=3e-6) and torch.allclose(dv[:, hid:].double(), vd.grad[:, hid:], atol=3e-6)
    assert float(dv[:, :hid].abs().max()) == 0.0
    # the older ops on the same shape (four-pixel form when HW % 4 == 0)
    h1, h2 = torch.empty_like(u), torch.empty_like(u)
    T.ew_slice(T.EW_GELU_SPLIT, v, o0=h1, o1=h2, n=n, nch=hid, HW=HW)
    ref = F.gelu(v.double())
    assert torch.allclose(h1.double(), ref[:, :hid], atol=5e-7) and torch.allclose(h2.double(), ref[:, hid:], atol=5e-7)
    du3, dv3 = torch.empty_like(u), torch.zeros_like(v)
    T.ew_slice(T.EW_GATE_BWD, u, h2, dg, v, o0=du3, o1=dv3, n=n, nch=hid, HW=HW, o1_ct=2 * hid, o1_c0=hid)
    assert torch.allclose(du3.double(), ud.grad, atol=3e-6) and torch.allclose(dv3[:, hid:].double(), vd.grad[:, hid:], atol=3e-6)
    T.ew_slice(T.EW_GELU_BWD, v, dg, o0=dv3, n=n, nch=hid, HW=HW, o0_ct=2 * hid, o0_c0=0)
    vd2 = v.double().requires_grad_(True)
    (F.gelu(vd2)[:, :hid] * dg.double()).sum().backward()
    assert torch.allclose(dv3[:, :hid].double(), vd2.grad[:, :hid], atol=3e-6)
    out = T.gelu_bwd(v, torch.ones_like(v))
    vd3 = v.double().requires_grad_(True)
    F.gelu(vd3).sum().backward()
    assert torch.allclose(out.double(), vd3.grad, atol=5e-7)


@pytest.mark.parametrize("shape", [(2, 6, 9, 13), (2, 5, 37, 704), (1, 4, 33, 8)])
def test_depthwise_kernels_on_gelu_of_a_channel_slice(shape):
    """ABI v8 (gencomm_dwconv3x3_act_{fwd,wgrad}): the layer's input is GELU of the first C channels of a 2 C-channel tensor."""
    from gencomm_amd import train_ops as T
    n, C, H, W = shape
    g = torch.Generator().manual_seed(W + 1)
    v, dy = torch.randn(n, 2 * C, H, W, generator=g) * 2.0, torch.randn(n, C, H, W, generator=g)
    w, b = torch.randn(C, 1, 3, 3, generator=g), torch.randn(C, generator=g)
    wd = w.double().requires_grad_(True)
    yd = F.conv2d(F.gelu(v.double())[:, :C], wd, b.double(), padding=1, groups=C)
    (yd * dy.double()).sum().backward()
    y = T.dwconv3x3(v.to(DEV), w.to(DEV), b.to(DEV), gelu_in=True)
    assert y.shape == (n, C, H, W) and torch.allclose(y.double().cpu(), yd.detach(), atol=4e-6)
    dw, db = T.dwconv3x3_wgrad(v.to(DEV), dy.to(DEV), gelu_in=True)
    scale = max(1.0, float(n * H * W) ** 0.5 / 16)
    assert torch.allclose(dw.double().cpu(), wd.grad, rtol=1e-5, atol=2e-5 * scale) and torch.allclose(db.double().cpu(), dy.double().sum((0, 2, 3)), rtol=1e-5, atol=1e-5 * scale)


@pytest.mark.parametrize("shape", [(2, 12, 9, 13), (1, 8, 16, 24), (2, 4, 7, 5),
                                   (2, 3, 37, 704), (1, 2, 70, 264), (3, 5, 8, 256), (1, 4, 33, 4)])   # W % 4 == 0: the sliding-window kernels
def test_depthwise_kernels_vs_torch(shape):
    from gencomm_amd import train_ops as T
    n, C, H, W = shape
    g = torch.Generator().manual_seed(W)
    x, dy = torch.randn(n, C, H, W, generator=g), torch.randn(n, C, H, W, generator=g)
    w, b = torch.randn(C, 1, 3, 3, generator=g), torch.randn(C, generator=g)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yd = F.conv2d(xd, wd, b.double(), padding=1, groups=C)
    (yd * dy.double()).sum().backward()
    y = T.dwconv3x3(x.to(DEV), w.to(DEV), b.to(DEV))
    assert torch.allclose(y.double().cpu(), yd.detach(), atol=2e-6)
    dx = T.dwconv3x3(dy.to(DEV), w.to(DEV), None, flip=True)
    assert torch.allclose(dx.double().cpu(), xd.grad, atol=2e-6)
    dw, db = T.dwconv3x3_wgrad(x.to(DEV), dy.to(DEV))
    scale = max(1.0, float(n * H * W) ** 0.5 / 16)   # fp32 sums of n H W products
    assert torch.allclose(dw.double().cpu(), wd.grad, rtol=1e-5, atol=2e-5 * scale) and torch.allclose(db.double().cpu(), dy.double().sum((0, 2, 3)), rtol=1e-5, atol=1e-5 * scale)


def test_modules_on_two_concurrent_streams_match_sequential_runs():
    """The module-level API keeps one scratch workspace per (device, HIP stream): two streams running GenComm -> Enhancer -> AttFusion
    at the same time on different scenes must not see each other."""
    from gencomm_amd import AttFusion, Enhancer, GenComm, normalize_pairwise_tfm, synth
    C, H, W, T, rl = 64, 24, 40, 3, [2, 1]
    gen, enh, fus = GenComm(synth.default_gencomm_cfg(C, T)).eval().to(DEV), Enhancer(C, [8, 8], 4).eval().to(DEV), AttFusion(C)
    synth.fill_params_(gen, 3)
    synth.fill_params_(enh, 4)
    scenes = []
    for s in range(2):
        inp = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_inputs(rl, C, H, W, 50 + s, max_shift=8.0).items()}
        scenes.append((inp, normalize_pairwise_tfm(inp["pairwise_t_matrix"], H * 0.8, W * 0.8, 1)))

    def run(inp, aff, seed):
        pred = gen(inp["feat"], inp["cond"], inp["record_len"], seed=seed)["pred_feature"]
        return fus(enh(pred, aff, rl), rl, aff)

    with torch.no_grad():
        ref = [run(inp, aff, 7 + i).clone() for i, (inp, aff) in enumerate(scenes)]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(device=DEV) for _ in scenes]
        outs = [None, None]
        for rep in range(3):
            for i, (inp, aff) in enumerate(scenes):
                with torch.cuda.stream(streams[i]):
                    outs[i] = run(inp, aff, 7 + i)
        torch.cuda.synchronize()
        again = [run(inp, aff, 7 + i) for i, (inp, aff) in enumerate(scenes)]
        torch.cuda.synchronize()
    for i in range(2):
        # not bit-identical by construction: the Enhancer's global average pool is a float atomic sum (enh_front_h_kernel epilogue), whose
        # order changes from run to run -- also between two sequential runs; a workspace shared by the streams would be wrong by O(1)
        d_conc, d_seq = float((outs[i] - ref[i]).abs().max()), float((again[i] - ref[i]).abs().max())
        print(f"scene {i}: max |concurrent - sequential| {d_conc:.2e}, max |sequential - sequential| {d_seq:.2e}, max |ref| {float(ref[i].abs().max()):.2f}")
        assert d_conc <= 2e-5 * float(ref[i].abs().max())


@pytest.mark.parametrize("shape", [(2, 64, 64, 32, 48), (1, 40, 72, 12, 20), (2, 128, 32, 16, 36)])
def test_stride2_input_gradient_subpixel_form_vs_float64(shape):
    """dx of a 3x3 stride-2 pad-1 convolution through the 2x2 sub-pixel convolution over dy (one launch, 16 tap-products per output
    quad) against float64 autograd; even input sizes take this path, odd ones the zero-stuffed one."""
    from gencomm_amd import train_ops as T
    N, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(H + W + Cin)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    for (h, wd) in ((H, W), (H - 1, W - 1)):
        x = torch.randn(N, Cin, h, wd, generator=g, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x, w.double(), None, stride=2, padding=1)
        dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
        y.backward(dy)
        dx = T.conv2d_dgrad_strided(dy.float().to(DEV), w.to(DEV), 1, 2, (h, wd))
        err = (dx.double().cpu() - x.grad).abs().max().item()
        assert dx.shape == x.grad.shape and err <= 3e-6 * x.grad.abs().max().item(), (shape, h, wd, err)


@pytest.mark.parametrize("C,H,W,n,spread", [(12, 20, 36, 2, 0.8), (8, 33, 18, 1, 7.0), (20, 16, 16, 2, 2.5)])
def test_deformable_sampling_gradients_vs_oracle_autograd(C, H, W, n, spread):
    """gencomm_dcn_scatter_bwd (offset gradients per (pixel, tap) + input gradient through LDS tiles with one global atomic per touched
    cell) against float64 autograd through the oracle's DCNv1 restatement: small offsets (everything lands in the workgroup's LDS
    region), offsets of several pixels (corners beyond the region take the direct global atomic; samples outside the map) and a map
    that is not a multiple of the 16 x 16 tile; channel counts that are not a multiple of the 8-channel chunk."""
    from gencomm_amd import train_ops as T
    from oracle import torch_port as O
    g = torch.Generator().manual_seed(C * H + W)
    x = torch.randn(n, C, H, W, generator=g)
    off = torch.randn(n, 18, H, W, generator=g) * spread
    dcol = torch.randn(n, C * 9, H, W, generator=g)
    xd, od = x.double().requires_grad_(True), off.double().requires_grad_(True)
    # identity "weight": output channel (c, k) of the deformable conv = sampled column (c, k)
    wid = torch.zeros(C * 9, C, 3, 3, dtype=torch.float64)
    for c in range(C):
        for k in range(9):
            wid[c * 9 + k, c, k // 3, k % 3] = 1.0
    col = O.deform_conv2d_ref(xd, od, wid, None, 1)
    (col * dcol.double()).sum().backward()
    got_col = T.dcn_sample(x.to(DEV), off.to(DEV))
    assert torch.allclose(got_col.double().cpu(), col.detach(), atol=2e-5)
    dx, doff = T.dcn_scatter_bwd(x.to(DEV), off.to(DEV), dcol.to(DEV))
    ex = (dx.double().cpu() - xd.grad).abs().max().item()
    eo = (doff.double().cpu() - od.grad).abs().max().item()
    assert ex <= 1e-5 * max(1.0, xd.grad.abs().max().item()), (ex, xd.grad.abs().max().item())
    assert eo <= 1e-4 * max(1.0, od.grad.abs().max().item()), (eo, od.grad.abs().max().item())
