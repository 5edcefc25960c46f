"""Dense conv stacks around the hot path: BaseBEVBackbone, DownsampleConv, 1x1 heads.
CPU: the oracle restatement against the reference's golden vectors + checkpoint key names.
GPU: the HIP implicit-GEMM kernel (through the C ABI) against the same vectors and against the oracle on
shapes with ragged tiles. Tolerance: rtol 1e-4 / atol 2e-5 (fp32 MFMA accumulation order over K up to 2304)."""
import numpy as np
import pytest
import torch

from helpers import assert_close, load_case, sub
from gencomm_amd import synth

CFG = {"layer_nums": [1, 2, 2], "layer_strides": [2, 2, 2], "num_filters": [64, 128, 256],
       "upsample_strides": [1, 2, 4], "num_upsample_filter": [128, 128, 128]}
SHRINK = {"kernal_size": [3], "stride": [2], "padding": [1], "dim": [128], "input_dim": 384}
RTOL, ATOL = 1e-4, 2e-5


def _modules(g, device="cpu"):
    from gencomm_amd.bev_backbone import BaseBEVBackbone, DownsampleConv, HipConv2d
    bb = BaseBEVBackbone(dict(CFG), 64).eval()
    sh = DownsampleConv(dict(SHRINK)).eval()
    heads = torch.nn.ModuleList([HipConv2d(128, 2, 1), HipConv2d(128, 14, 1), HipConv2d(128, 4, 1)]).eval()
    for k, m in enumerate((bb, sh, heads)):
        synth.fill_params_(m, int(g["weight_seed"]) + k)
    synth.fill_bn_stats_(bb, int(g["bn_seed"]))
    x = torch.from_numpy(np.maximum(synth.noise_stream(int(g["data_seed"]), 0, tuple(int(v) for v in g["in_shape"])), 0.0).astype(np.float32))
    return bb.to(device), sh.to(device), heads.to(device), x.to(device)


def test_oracle_backbone_matches_reference_golden():
    from oracle import torch_port as O
    g = load_case("backbone")
    bb, sh, heads, x = _modules(g)
    assert sorted(bb.state_dict().keys()) == list(g["backbone_keys"])
    assert sorted(sh.state_dict().keys()) == list(g["shrink_keys"])
    with torch.no_grad():
        y = O.bev_backbone_forward({k: v for k, v in bb.state_dict().items()}, "", x, CFG)
        z = O.downsample_conv_forward({k: v for k, v in sh.state_dict().items()}, "", y, SHRINK)
        hs = [torch.nn.functional.conv2d(z, h.weight, h.bias) for h in heads]
    assert tuple(y.shape) == tuple(g["backbone_shape"]) and tuple(z.shape) == tuple(g["shrink_shape"])
    assert_close(sub(y, 7), g["backbone"], 1e-5, 1e-6, "backbone")
    assert_close(sub(z, 3), g["shrink"], 1e-5, 1e-6, "shrink")
    for name, h in zip(("cls", "reg", "dir"), hs):
        assert_close(h.numpy(), g[name], 1e-5, 1e-6, name)


@pytest.mark.gpu
def test_hip_backbone_vs_reference_golden():
    g = load_case("backbone")
    bb, sh, heads, x = _modules(g, "cuda:0")
    with torch.no_grad():
        d = bb({"spatial_features": x})
        assert sorted(d.keys()) == ["spatial_features", "spatial_features_2d"]
        y = d["spatial_features_2d"]
        ms = bb.get_multiscale_feature(x)
        z = sh(y)
        hs = [h(z) for h in heads]
        y2 = bb.decode_multiscale_feature(ms)
    assert tuple(y.shape) == tuple(g["backbone_shape"]) and tuple(z.shape) == tuple(g["shrink_shape"])
    assert_close(sub(y, 7), g["backbone"], RTOL, ATOL, "backbone")
    assert_close(np.concatenate([sub(f, 11) for f in ms]), g["ms_feat"], RTOL, ATOL, "multiscale features")
    assert torch.equal(y, y2)
    assert_close(sub(z, 3), g["shrink"], RTOL, ATOL, "shrink")
    for name, h in zip(("cls", "reg", "dir"), hs):
        assert_close(h.cpu().numpy(), g[name], RTOL, ATOL, name)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,k,stride,pad,H,W", [
    (5, 3, 3, 1, 1, 7, 9),        # channels below a chunk / a tile, ragged pixel tiles
    (20, 70, 3, 2, 1, 11, 37),    # Cout crosses a 64-row tile, odd size with stride 2
    (33, 18, 3, 1, 0, 9, 20),     # no padding (offset conv of the message extractor shape family)
    (40, 6, 1, 1, 0, 5, 50),      # 1x1, Cin not a multiple of the 32-channel chunk
])
def test_hip_conv2d_vs_torch_fp32(cin, cout, k, stride, pad, H, W):
    """Numerics of the general kernel against plain torch fp32 on CPU (the oracle of a single conv IS F.conv2d)."""
    from gencomm_amd.bev_backbone import conv2d_hip
    torch.manual_seed(cin * 100 + cout)
    conv = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=pad)
    bn = torch.nn.BatchNorm2d(cout, eps=1e-3).eval()
    synth.fill_bn_stats_(bn, 3)
    x = torch.randn(3, cin, H, W)
    with torch.no_grad():
        ref = torch.relu(bn(conv(x)))
        ref_plain = conv(x)
        got = conv2d_hip(x.cuda(), conv.cuda(), bn.cuda(), relu=True).cpu()
        got_plain = conv2d_hip(x.cuda(), conv, None, relu=False).cpu()
    assert_close(got.numpy(), ref.numpy(), RTOL, ATOL, "conv+bn+relu")
    assert_close(got_plain.numpy(), ref_plain.numpy(), RTOL, ATOL, "conv")


@pytest.mark.gpu
@pytest.mark.parametrize("s", [1, 2, 4])
def test_hip_conv_transpose_into_concat_slice(s):
    from gencomm_amd.bev_backbone import conv2d_hip
    torch.manual_seed(s)
    ct = torch.nn.ConvTranspose2d(24, 10, s, stride=s, bias=False)
    bn = torch.nn.BatchNorm2d(10, eps=1e-3).eval()
    synth.fill_bn_stats_(bn, 4)
    x = torch.randn(2, 24, 6, 21)
    with torch.no_grad():
        ref = torch.relu(bn(ct(x)))
        buf = torch.full((2, 17, 6 * s, 21 * s), -7.0, device="cuda:0")
        conv2d_hip(x.cuda(), ct.cuda(), bn.cuda(), relu=True, out=buf, out_coff=4)
    assert_close(buf[:, 4:14].cpu().numpy(), ref.numpy(), RTOL, ATOL, "conv transpose slice")
    assert bool((buf[:, :4] == -7.0).all()) and bool((buf[:, 14:] == -7.0).all())


def test_training_mode_batchnorm_is_refused():
    from gencomm_amd.bev_backbone import conv2d_hip
    conv, bn = torch.nn.Conv2d(4, 4, 3, padding=1), torch.nn.BatchNorm2d(4).train()
    with pytest.raises((NotImplementedError, RuntimeError)):
        conv2d_hip(torch.zeros(1, 4, 8, 8), conv, bn)


@pytest.mark.gpu
def test_hip_naive_compressor_vs_torch_fp32():
    from gencomm_amd.bev_backbone import NaiveCompressor
    nc = NaiveCompressor(48, 4).eval()
    synth.fill_params_(nc, 11)
    synth.fill_bn_stats_(nc, 12)
    x = torch.randn(2, 48, 14, 22)
    with torch.no_grad():
        ref = nc.decoder(nc.encoder(x))          # the torch layers the module owns == naive_compress.py:30-35
        got = nc.cuda()(x.cuda()).cpu()
    assert_close(got.numpy(), ref.numpy(), RTOL, ATOL, "compressor")


@pytest.mark.gpu
def test_grad_mode_keeps_the_autograd_graph_through_the_conv_stacks():
    """Backbone -> shrink -> heads in grad mode: the HIP forward must not cut the graph (the reference's detection loss
    back-propagates through these frozen or trainable layers into fusion / enhancer / gencomm). Gradients against the
    same layers as plain torch modules."""
    import copy
    g = load_case("backbone")
    bb, sh, heads, x = _modules(g, "cuda:0")
    x = x[:, :, :32, :48].contiguous().requires_grad_(True)
    y = bb({"spatial_features": x})["spatial_features_2d"]
    out = heads[0](sh(y))
    assert out.requires_grad and y.requires_grad
    out.square().mean().backward()
    got = {"x": x.grad.clone(), "w0": bb.blocks[0][1].weight.grad.clone(), "bn": bb.blocks[1][2].weight.grad.clone(),
           "de": bb.deblocks[2][0].weight.grad.clone(), "head": heads[0].weight.grad.clone(), "sh_b": sh.layers[0].double_conv[0].bias.grad.clone()}
    # the same computation with the torch layers the modules own
    x2 = x.detach().clone().requires_grad_(True)
    for m in (bb, sh, heads):
        m.zero_grad()
    feats, h = [], x2
    for blk in bb.blocks:
        h = blk(h)
        feats.append(h)
    y2 = torch.cat([bb.deblocks[i](f) for i, f in enumerate(feats)], dim=1)
    z2 = sh.layers[0].double_conv(y2)
    out2 = torch.nn.functional.conv2d(z2, heads[0].weight, heads[0].bias)
    out2.square().mean().backward()
    want = {"x": x2.grad, "w0": bb.blocks[0][1].weight.grad, "bn": bb.blocks[1][2].weight.grad,
            "de": bb.deblocks[2][0].weight.grad, "head": heads[0].weight.grad, "sh_b": sh.layers[0].double_conv[0].bias.grad}
    assert_close(out.detach().cpu().numpy(), out2.detach().cpu().numpy(), 1e-3, 1e-4, "forward (HIP vs torch/MIOpen)")
    for k in got:
        scale = float(want[k].abs().max())
        assert float((got[k] - want[k]).abs().max()) <= 2e-3 * scale + 1e-7, (k, float((got[k] - want[k]).abs().max()), scale)


@pytest.mark.gpu
def test_training_mode_batch_statistics_forward_backward_and_running_stats():
    """Stage 1 of the reference trains the backbone with BatchNorm2d in TRAINING mode (batch statistics, base_bev_backbone.py:47-52,
    eps 1e-3, momentum 0.01). HIP path (convolution, batch statistics, normalisation, their backward kernels; stride-2 and transposed
    convolutions included) against the same nn.Sequential layers run by torch in train mode: forward, every gradient, and the
    running-statistics update."""
    import copy
    g = load_case("backbone")
    bb, sh, heads, x = _modules(g, "cuda:0")
    bb.train(); sh.train(); heads.train()
    ref_bb = copy.deepcopy(bb)
    x = x[:, :, :32, :48].contiguous()
    xa = x.clone().requires_grad_(True)
    y = bb({"spatial_features": xa})["spatial_features_2d"]
    out = heads[0](sh(y))
    out.square().mean().backward()
    # the same layers through torch (train mode: batch statistics, running stats updated)
    xb = x.clone().requires_grad_(True)
    feats, h = [], xb
    for blk in ref_bb.blocks:
        h = blk(h)
        feats.append(h)
    y2 = torch.cat([ref_bb.deblocks[i](f) for i, f in enumerate(feats)], dim=1)
    z2 = torch.nn.functional.relu(torch.nn.functional.conv2d(
        torch.nn.functional.relu(torch.nn.functional.conv2d(y2, sh.layers[0].double_conv[0].weight.detach(), sh.layers[0].double_conv[0].bias.detach(),
                                                            stride=sh.layers[0].double_conv[0].stride, padding=sh.layers[0].double_conv[0].padding)),
        sh.layers[0].double_conv[2].weight.detach(), sh.layers[0].double_conv[2].bias.detach(), padding=sh.layers[0].double_conv[2].padding))
    out2 = torch.nn.functional.conv2d(z2, heads[0].weight.detach(), heads[0].bias.detach())
    out2.square().mean().backward()
    assert_close(y.detach().cpu().numpy(), y2.detach().cpu().numpy(), 2e-3, 2e-4, "backbone forward, training-mode BatchNorm (HIP vs torch/MIOpen)")
    got, want = dict(bb.named_parameters()), dict(ref_bb.named_parameters())
    worst = 0.0
    for k in want:
        assert got[k].grad is not None, k
        scale = float(want[k].grad.abs().max())
        err = float((got[k].grad - want[k].grad).abs().max())
        assert err <= 5e-3 * scale + 1e-7, (k, err, scale)
        worst = max(worst, err / (scale + 1e-30))
    scale = float(xb.grad.abs().max())
    assert float((xa.grad - xb.grad).abs().max()) <= 5e-3 * scale + 1e-8
    for (k, b1), (_, b2) in zip(bb.named_buffers(), ref_bb.named_buffers()):
        if b1.is_floating_point():
            assert_close(b1.cpu().numpy(), b2.cpu().numpy(), 1e-4, 1e-6, "running statistic " + k)
        else:
            assert int(b1) == int(b2), k
    print(f"backbone training mode: {len(want)} parameter gradients + input, worst relative error vs torch {worst:.2e}")


@pytest.mark.gpu
@pytest.mark.parametrize("k,stride,cin,cout,H,W", [
    (3, 1, 32, 48, 19, 37),       # split 3x3 kernel: two 16-channel chunks, ragged pixel tiles, Cout below a 64-row tile
    (3, 2, 64, 130, 22, 41),      # stride 2, Cout crosses two tiles
    (3, 1, 24, 40, 15, 21),       # Cin % 16 == 8: the last chunk holds one octet
    (3, 1, 8, 66, 12, 33),        # a single half-filled chunk (the dgrad of conv_in)
    (1, 1, 64, 320, 13, 29),      # split 1x1 kernel (>= 256 outputs): 64-pixel tiles, ragged
    (1, 1, 128, 256, 9, 11),      # few workgroups: the 32-pixel tile variant
])
@pytest.mark.parametrize("case", ["unit", "tiny", "huge", "late_large_channels", "all_zero_then_data"])
def test_split_conv_kernels_hold_fp32_grade_accuracy_over_the_whole_input_range(k, stride, cin, cout, H, W, case):
    """The f16-pipe convolutions (conv3x3_f16s_kernel / conv1x1_f16s_kernel) carry a running power-of-two activation scale. Against
    float64 on CPU: |error| <= 4e-6 of sum |w| |x| (fp32 accumulation order + 22-bit products) for inputs of magnitude 1, 1e-6 (gradients)
    and 1e5 (beyond fp16's range), for channels that grow by 1e4 in the LAST chunk (the scale has to drop and the accumulators are
    rescaled), and for leading all-zero chunks; the exact-fp32 mode must agree too."""
    from gencomm_amd import _lib
    from gencomm_amd.bev_backbone import conv2d_hip
    torch.manual_seed(k * 1000 + cin + cout)
    conv = torch.nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2)
    x = torch.randn(2, cin, H, W)
    if case == "tiny":
        x *= 1e-6
    elif case == "huge":
        x *= 1e5
    elif case == "late_large_channels":
        x[:, -16:] *= 1e4
    elif case == "all_zero_then_data":
        x[:, :16] = 0.0
    with torch.no_grad():
        ref = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double(), stride=stride, padding=k // 2)
        mag = torch.nn.functional.conv2d(x.abs().double(), conv.weight.abs().double(), None, stride=stride, padding=k // 2) + conv.bias.abs().double().view(1, -1, 1, 1)
        for mode in (3, 1, 0):   # 3: the two-term split kernels (opt-in); 1, 0: exact fp32 (the default for these layers since round 3)
            with _lib.mode(_lib.MODE_ARITH, mode):
                got = conv2d_hip(x.cuda(), conv.cuda(), None, relu=False).cpu().double()
            assert torch.isfinite(got).all(), (case, mode)
            worst = float(((got - ref).abs() / mag).max())
            assert worst <= 4e-6, (case, "split" if mode == 3 else "exact fp32", worst)


@pytest.mark.gpu
def test_eval_fold_cache_follows_running_statistics_moved_by_the_hip_training_kernels():
    """eval forward (builds the folded scale / shift cache) -> train-mode forward under no_grad (the HIP kernel moves
    running_mean / running_var through raw pointers, no parameter changes) -> eval forward must use the NEW statistics."""
    import torch.nn as nn
    from gencomm_amd.bev_backbone import conv2d_hip
    torch.manual_seed(3)
    dev = "cuda:0"
    conv = nn.Conv2d(16, 32, 3, padding=1, bias=False).to(dev)
    bn = nn.BatchNorm2d(32, eps=1e-3, momentum=0.01).to(dev)
    ref_conv, ref_bn = nn.Conv2d(16, 32, 3, padding=1, bias=False).to(dev), nn.BatchNorm2d(32, eps=1e-3, momentum=0.01).to(dev)
    ref_conv.load_state_dict(conv.state_dict())
    x = torch.randn(2, 16, 24, 40, device=dev) * 3 + 1

    def both(train):
        for m in (conv, bn, ref_conv, ref_bn):
            m.train(train)
        with torch.no_grad():
            return conv2d_hip(x, conv, bn, relu=True), torch.relu(ref_bn(ref_conv(x)))

    for train in (False, True, False, True, True, False):
        got, want = both(train)
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-4), (train, float((got - want).abs().max()))
    assert torch.allclose(bn.running_mean, ref_bn.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var, ref_bn.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == 3
