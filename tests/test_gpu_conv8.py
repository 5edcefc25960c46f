"""The 8 -> 8 channel 3x3 convolution kernels of the UNet, one layer at a time through the C ABI
(gencomm_conv8_fwd), and the whole path with the 64x16-tile kernels forced onto the small golden cases.

Two kernels implement these layers (csrc/unet_kernels.h conv8_kernel: exact fp32 on v_mfma_f32_4x4x1;
csrc/conv8h_kernels.h conv8h_kernel: exact three-term operand splits on v_mfma_f32_16x16x32_f16 + one bf8 instruction,
products to 2^-26). Both must meet the same bar: rtol 1e-4 / atol 1e-5 against the reference's golden vectors, and for a
single layer 2e-5 absolute against a float64 convolution of O(1) data. Launches with several workgroups per CU
are part of the matrix on purpose: a scheduling-dependent fault of an early conv8h_kernel only showed there.
"""
import numpy as np
import pytest
import torch

from helpers import assert_close, build_inputs, build_modules, eval_noise, load_case, sub

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _conv8(x, w, b, split):
    from gencomm_amd import _lib
    from gencomm_amd.runtime import ptr, stream_ptr
    n, _, H, W = x.shape
    y = torch.full_like(x, float("nan"))
    st = torch.zeros(n, 8, 2, dtype=torch.float64, device=DEV)
    scratch = torch.zeros(4096, device=DEV)
    _lib.check(_lib.lib().gencomm_conv8_fwd(ptr(x), ptr(w), ptr(b), ptr(y), ptr(st), ptr(scratch), n, H, W, split,
                                            stream_ptr(DEV)), "gencomm_conv8_fwd")
    torch.cuda.synchronize()
    return y, st


@pytest.mark.parametrize("split", [0, 1])
@pytest.mark.parametrize("shape", [(1, 32, 64), (3, 18, 26), (2, 50, 130), (64, 64, 128), (4, 200, 704), (16, 100, 352)])
def test_single_layer_vs_float64(modes, shape, split):
    modes(tile_want=1)  # 64x16 tiles whatever the size (ragged widths take the scalar staging)
    n, H, W = shape
    g = torch.Generator(device=DEV).manual_seed(100 + n + H)
    x = torch.randn(n, 8, H, W, generator=g, device=DEV)
    w = torch.randn(8, 8, 3, 3, generator=g, device=DEV) * 0.2
    b = torch.randn(8, generator=g, device=DEV)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    y0, st0 = _conv8(x, w, b, split)
    assert float((y0.double() - ref).abs().max()) < 2e-5
    s_ref = torch.stack([ref.sum(dim=(2, 3)), (ref * ref).sum(dim=(2, 3))], dim=-1)
    assert float(((st0 - s_ref).abs() / (s_ref.abs() + 1.0)).max()) < 1e-5
    for _ in range(2):  # bit-identical from launch to launch (statistics: f64 atomics, order-dependent in the last bits)
        y1, _ = _conv8(x, w, b, split)
        assert torch.equal(y0, y1)


def test_f16_pipe_layer_is_at_least_as_accurate_as_the_exact_fp32_kernel(modes):
    """the f16-pipe kernel (three-term operands, six matrix instructions per product block) against
    the exact-fp32 v_mfma_f32_4x4x1 kernel on the same layer, both measured against a float64 convolution: the rms and
    the worst error of the f16-pipe result must not exceed the exact kernel's (each product is formed to 2^-26, the
    exact kernel's fp32 FMA chain rounds 72 times at 2^-24)."""
    modes(tile_want=1)
    g = torch.Generator(device=DEV).manual_seed(2024)
    rows = []
    for (n, H, W, xs, ws) in [(4, 200, 704, 1.0, 0.2), (16, 100, 352, 1.0, 0.2), (2, 64, 128, 30.0, 0.05), (2, 64, 128, 0.01, 3.0)]:
        x = torch.randn(n, 8, H, W, generator=g, device=DEV) * xs
        w = torch.randn(8, 8, 3, 3, generator=g, device=DEV) * ws
        b = torch.randn(8, generator=g, device=DEV) * xs * ws
        ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
        err = {}
        for split in (0, 1):
            y, _ = _conv8(x, w, b, split)
            e = (y.double() - ref).abs()
            err[split] = (float(e.pow(2).mean().sqrt()), float(e.max()))
        rows.append(((n, H, W, xs, ws), err))
        print(f"{(n, H, W)} x~{xs} w~{ws}: exact-fp32 kernel rms {err[0][0]:.3e} max {err[0][1]:.3e} | f16-pipe three-term rms {err[1][0]:.3e} max {err[1][1]:.3e}")
    for cfg, err in rows:
        # measured: 0.60x the exact kernel's rms for O(1) and O(30) activations; 1.04x for activations of 0.01, whose second
        # terms fall into fp16's subnormal range (|x| < 2^-3: absolute operand accuracy 2^-28 instead of 24 relative bits)
        assert err[1][0] <= err[0][0] * (1.10 if cfg[3] < 0.1 else 1.0) + 1e-12, cfg
        assert err[1][1] <= err[0][1] * 1.25 + 1e-12, cfg   # the maximum over 1e7 outputs is itself a noisy statistic


def _f16_terms(v):
    """numpy float64 arrays -> (t1, t2, t3): the three fp16 terms of an fp32 value, as float64 (t1 + t2 + t3 == v)"""
    v = np.asarray(v, dtype=np.float32)
    t1 = v.astype(np.float16).astype(np.float32)
    r1 = v - t1
    t2 = r1.astype(np.float16).astype(np.float32)
    t3 = r1 - t2
    return t1.astype(np.float64), t2.astype(np.float64), t3.astype(np.float64)


def test_third_term_of_the_weights_reaches_the_result_exactly(modes):
    """The same isolation for w3 = w - w1 - w2: weights of the even input channels are full 24-bit values, those of the odd
    channels the NEGATED two-term part of the same values; all activations 1, so the output is the sum of the weights' third
    terms (the kernel scales the weights by a power of two before splitting; the split commutes with it)."""
    modes(tile_want=1)
    rng = np.random.default_rng(6)
    v = (rng.random((8, 4, 3, 3)) + 1.0).astype(np.float32)   # [1, 2): third terms on fp16's subnormal grid after any power-of-two scale >= 1
    v = np.where(rng.random(v.shape) < 0.5, v, -v).astype(np.float32)
    t1, t2, t3 = _f16_terms(v)
    assert np.count_nonzero(t3) > 50
    w = np.zeros((8, 8, 3, 3), dtype=np.float32)
    w[:, 0::2] = v
    w[:, 1::2] = -(t1 + t2).astype(np.float32)               # exactly representable: two fp16 terms of one binade pair
    assert np.array_equal(w[:, 1::2].astype(np.float64), -(t1 + t2))
    n, H, W = 1, 32, 72
    x = torch.ones(n, 8, H, W, device=DEV)
    want = torch.nn.functional.conv2d(torch.ones(n, 4, H, W, dtype=torch.float64), torch.from_numpy(t3), None, padding=1)
    y, _ = _conv8(x, torch.from_numpy(w).to(DEV), torch.zeros(8, device=DEV), 1)
    got = y.double().cpu()
    assert float(want.abs().max()) > 1e-7
    assert torch.allclose(got, want, rtol=0, atol=2.0 ** -40), float((got - want).abs().max())


def _bf8(v):
    """round float64 array to OCP e5m2 (round to nearest even; the values used here stay in the normal range)"""
    v = np.asarray(v, dtype=np.float64)
    m, e = np.frexp(v)                       # v = m 2^e, |m| in [0.5, 1): 3 significant bits -> grid 2^(e-3), subnormal grid 2^-16
    g = np.maximum(e - 3, -16)
    return np.ldexp(np.round(np.ldexp(v, -g)), g)


@pytest.mark.parametrize("term", range(6), ids=["hi_w1", "lo_w1", "hi_w2", "lo_w2", "hi_w3", "t_wb"])
def test_each_term_of_the_six_instruction_product_alone(modes, term):
    """The diagnostic instantiation issues ONE of the six terms; the expected output is the float64 convolution of that
    activation term with that weight term (terms computed on the host with numpy float16 / an e5m2 rounding). Covers the
    staging of every plane (main rows, remainder rows, halo columns, tile seams, partial last tile), the three fp16 weight
    tables, the bf8 weight table, the 2^20 scale of the third term and the K order of both matrix instructions."""
    modes(tile_want=1)
    rng = np.random.default_rng(40 + term)
    n, H, W = 2, 40, 136
    x = (rng.uniform(0.25, 4.0, (n, 8, H, W)) * rng.choice([-1.0, 1.0], (n, 8, H, W))).astype(np.float32)   # |x| >= 2^-12: third terms in bf8's range
    w = (rng.standard_normal((8, 8, 3, 3)) * 0.3).astype(np.float32)
    scale = 2.0 ** (14 - np.frexp(np.abs(w).max())[1])          # the kernel's power-of-two weight scale
    xt = _f16_terms(x)
    wt = _f16_terms((w.astype(np.float64) * scale).astype(np.float32))
    wb = _bf8(w.astype(np.float64) * scale * 2.0 ** -20) * 2.0 ** 20
    ax = [xt[0], xt[1], xt[0], xt[1], xt[0], _bf8(xt[2] * 2.0 ** 20) * 2.0 ** -20][term]
    aw = [wt[0], wt[0], wt[1], wt[1], wt[2], wb][term] / scale
    if term == 5:
        assert np.array_equal(_bf8(xt[2] * 2.0 ** 20), xt[2] * 2.0 ** 20) and np.count_nonzero(xt[2]) > 1000   # third terms are powers of two: exact in bf8
    want = torch.nn.functional.conv2d(torch.from_numpy(ax), torch.from_numpy(aw), None, padding=1)
    y, _ = _conv8(torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV), torch.zeros(8, device=DEV), 0x100 | (1 << term))
    got = y.double().cpu()
    scale_out = float(want.abs().max())
    err = float((got - want).abs().max())
    print(f"term {term}: max |expected| {scale_out:.3e}, max abs err {err:.3e}")
    assert scale_out > 0 and err <= 2e-6 * scale_out, (term, err, scale_out)   # fp32 accumulation of <= 72 products of one term


def test_split_operands_cover_the_fp16_range(modes):
    """Tiny and large weights / activations: the power-of-two weight scale keeps low parts normal; activations far below
    1 lose only absolute accuracy (fp16 subnormal steps of 6e-8); activations at and beyond the fp16 range (65504) are
    range-reduced by an exact power of two from the device-side max|x| (common.h act_scale) -- same relative accuracy."""
    modes(tile_want=1)
    g = torch.Generator(device=DEV).manual_seed(7)
    for wscale, xscale in [(1e-3, 1.0), (30.0, 1.0), (0.2, 1e-3), (0.2, 200.0), (0.2, 2e4), (0.2, 1e5), (0.2, 3e7), (1e-3, 1e30)]:
        x = torch.randn(2, 8, 32, 64, generator=g, device=DEV) * xscale
        w = torch.randn(8, 8, 3, 3, generator=g, device=DEV) * wscale
        b = torch.zeros(8, device=DEV)
        ref = torch.nn.functional.conv2d(x.double(), w.double(), None, padding=1)
        y, _ = _conv8(x, w, b, 1)
        assert torch.isfinite(y).all(), (wscale, xscale)
        tol = 1e-5 * wscale * max(xscale, 1.0) * 10 + 2e-6 * float(ref.abs().max())
        assert float((y.double() - ref).abs().max()) < tol, (wscale, xscale)


@pytest.mark.parametrize("sampler", ["latent", "direct"])
def test_large_inputs_never_overflow_the_split(modes, sampler):
    """A drop-in fp32 replacement must stay finite and accurate where the reference does: BEV features up to 1e5 (beyond
    the fp16 range of an unscaled hi/lo split) and message channels up to 1e3 through the f16-pipe kernels (64x16 tiles
    forced), both sampler structures, against the CPU oracle relative to the map's magnitude."""
    from gencomm_amd import AttFusion, Enhancer, GenComm, normalize_pairwise_tfm, synth
    from oracle import torch_port as O
    modes(tile_want=1, sampler=sampler)
    C, H, W, T, rl = 32, 48, 128, 3, [2, 1]
    n = sum(rl)
    cfg = synth.default_gencomm_cfg(C, T)
    gen, enh = GenComm(cfg).eval(), Enhancer(C, [8, 8], 4).eval()
    synth.fill_params_(gen, 61)
    synth.fill_params_(enh, 62)
    inp = {k: torch.from_numpy(v) for k, v in synth.make_inputs(rl, C, H, W, 63, max_shift=8.0).items()}
    feat, cond = inp["feat"] * 2.5e4, inp["cond"] * 3e2
    assert float(feat.abs().max()) > 65504.0
    n0, sn = (torch.from_numpy(a) for a in synth.make_eval_noise(64, n, C, H, W, T))
    ref = O.path_forward({k: v.detach() for k, v in gen.state_dict().items()}, {k: v.detach() for k, v in enh.state_dict().items()},
                         cfg, feat, cond, inp["record_len"], inp["pairwise_t_matrix"], H * 0.8, W * 0.8, n0, sn)
    gen, enh = gen.to(DEV), enh.to(DEV)
    affine = normalize_pairwise_tfm(inp["pairwise_t_matrix"], H * 0.8, W * 0.8, 1)
    with torch.no_grad():
        pred = gen(feat.to(DEV), cond.to(DEV), inp["record_len"], noise=(n0.to(DEV), sn.to(DEV)))["pred_feature"]
        # the Enhancer / fusion on an input of that magnitude as well
        big = ref["pred_feature"] * (1e5 / float(ref["pred_feature"].abs().max()))
        enh_big = enh(big.to(DEV), affine, inp["record_len"])
        fused_big = AttFusion(C)(enh_big, inp["record_len"], affine)
    ref_enh = O.enhancer_forward({k: v.detach().cpu() for k, v in enh.state_dict().items()}, big, inp["record_len"])
    ref_fus = O.att_fusion(ref_enh, inp["record_len"], O.normalize_pairwise_tfm(inp["pairwise_t_matrix"], H * 0.8, W * 0.8, 1.0))
    for name, got, want in (("pred_feature", pred, ref["pred_feature"]), ("enhanced(1e5)", enh_big, ref_enh), ("fused(1e5)", fused_big, ref_fus)):
        got = got.cpu()
        assert torch.isfinite(got).all(), name
        err = (got - want).abs()
        tol = 1e-5 * float(want.abs().max()) + 1e-4 * want.abs()
        print(f"large inputs [{sampler}] {name}: max |ref| {float(want.abs().max()):.3e}, max abs err {float(err.max()):.3e}, worst err/tol {float((err / tol).max()):.3f}")
        assert (err <= tol).all(), (name, float((err / tol).max()))


def test_absurd_groupnorm_gain_saturates_instead_of_overflowing(modes):
    """GroupNorm outputs are bounded by |gamma| sqrt(group size) + |beta|; a gain that pushes them past 65504 makes the
    f16-pipe kernels saturate (MODE.FP16_OVFL), never produce inf / NaN from finite inputs."""
    from gencomm_amd import GenComm, synth
    modes(tile_want=1)
    C, H, W, T = 16, 32, 64, 2
    gen = GenComm(synth.default_gencomm_cfg(C, T)).eval()
    synth.fill_params_(gen, 5)
    with torch.no_grad():
        for name, p in gen.denoiser.named_parameters():
            if "norm" in name and name.endswith("weight"):
                p.mul_(3e4)
    gen = gen.to(DEV)
    inp = synth.make_inputs([2], C, H, W, 6)
    with torch.no_grad():
        out = gen(torch.from_numpy(inp["feat"]).to(DEV), torch.from_numpy(inp["cond"]).to(DEV), [2], seed=3)["pred_feature"]
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("mode", ["f32", "split"])
@pytest.mark.parametrize("name", ["tiny", "ragged", "mid", "shipped"])
def test_golden_path_with_64x16_tiles_forced(modes, name, mode):
    modes(tile_want=1)
    modes(arith=mode)
    g = load_case(name)
    _, gen, _ = build_modules(g, "cuda:0")
    inp = build_inputs(g, "cuda:0")
    with torch.no_grad():
        pred = gen(inp["feat"], inp["cond"], inp["record_len"], noise=eval_noise(g, "cuda:0"))["pred_feature"]
    assert_close(sub(pred, int(g["stride"])), g["pred_feature"], 1e-4, 1e-5, f"pred_feature ({mode}, forced tiles)")


def test_full_size_unet_call_modes_agree_and_repeat(modes):
    from gencomm_amd import GenComm, synth
    gen = GenComm(synth.default_gencomm_cfg(64, 20)).eval()
    synth.fill_params_(gen, 0)
    gen = gen.to(DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(8, 66, 200, 704, generator=g, device=DEV)
    t = torch.full((8,), 7.0, device=DEV)
    ys = {}
    for mode in ("f32", "split", "split"):
        modes(arith=mode)
        with torch.no_grad():
            y = gen.denoiser(x, t, T=20).clone()
        if mode in ys:
            assert float((y - ys[mode]).abs().max()) < 1e-6  # GroupNorm statistics: f64 atomics in any order
        ys[mode] = y
    assert float((ys["f32"] - ys["split"]).abs().max()) < 2e-5


@pytest.mark.parametrize("name", ["tiny", "mid"])
def test_golden_path_direct_sampler_with_64x16_tiles_forced(modes, name):
    """The literal sampler (conv_in ... conv_out + update per step, explicit noise in conv_out's epilogue) through
    conv_in_h_kernel / conv8h_kernel / conv_out_h_kernel<POST=1>."""
    modes(tile_want=1)
    modes(sampler="direct")
    g = load_case(name)
    _, gen, _ = build_modules(g, "cuda:0")
    inp = build_inputs(g, "cuda:0")
    with torch.no_grad():
        pred = gen(inp["feat"], inp["cond"], inp["record_len"], noise=eval_noise(g, "cuda:0"))["pred_feature"]
    assert_close(sub(pred, int(g["stride"])), g["pred_feature"], 1e-4, 1e-5, "pred_feature (direct sampler, forced tiles)")


@pytest.mark.parametrize("shape", [(32, 22, 46, 5, [5, 1]), (256, 16, 24, 2, [2]), (16, 34, 68, 3, [1, 3])])
def test_forced_tiles_vs_oracle_on_odd_shapes(modes, shape):
    """The 64x16-tile f16-pipe kernels on shapes no fixture has: a width that is not a multiple of 4 (scalar staging,
    fp32 conv_in / conv_out / sampler fall-backs mixed with conv8h layers), 5 agents in a scene, C = 256 (V2X-Real:
    16 channel blocks in conv_out_h, 32 chunks in conv_in_h, C/4 = 64 partial conv on the fp32 kernel), and a map whose
    half-resolution level is 17 x 34 (partial tiles at both levels) -- each stage against the CPU oracle."""
    from gencomm_amd import AttFusion, Enhancer, GenComm, normalize_pairwise_tfm, synth
    from oracle import torch_port as O
    modes(tile_want=1)
    C, H, W, T, rl = shape
    n = sum(rl)
    cfg = synth.default_gencomm_cfg(C, T)
    gen, enh = GenComm(cfg).eval(), Enhancer(C, [8, 8], 4).eval()
    synth.fill_params_(gen, 31)
    synth.fill_params_(enh, 32)
    inp = {k: torch.from_numpy(v) for k, v in synth.make_inputs(rl, C, H, W, 33, max_shift=6.0).items()}
    n0, sn = (torch.from_numpy(a) for a in synth.make_eval_noise(34, n, C, H, W, T))
    ref = O.path_forward({k: v.detach() for k, v in gen.state_dict().items()}, {k: v.detach() for k, v in enh.state_dict().items()},
                         cfg, inp["feat"], inp["cond"], inp["record_len"], inp["pairwise_t_matrix"], H * 0.8, W * 0.8, n0, sn)
    gen, enh = gen.to(DEV), enh.to(DEV)
    with torch.no_grad():
        affine = normalize_pairwise_tfm(inp["pairwise_t_matrix"], H * 0.8, W * 0.8, 1)
        pred = gen(inp["feat"].to(DEV), inp["cond"].to(DEV), inp["record_len"], noise=(n0.to(DEV), sn.to(DEV)))["pred_feature"]
        assert_close(pred.cpu().numpy(), ref["pred_feature"].numpy(), 1e-4, 1e-5, "pred_feature")
        e = enh(ref["pred_feature"].to(DEV), affine, inp["record_len"])
        assert_close(e.cpu().numpy(), ref["enhanced"].numpy(), 1e-4, 1e-5, "enhanced (oracle input)")
        f = AttFusion(C)(ref["enhanced"].to(DEV), inp["record_len"], affine)
        assert_close(f.cpu().numpy(), ref["fused"].numpy(), 1e-4, 1e-5, "fused (oracle input)")


# ---------------------------------------------------------------------- 64x8 tiles (GENCOMM_MODE_TILE8; conv8h8_kernels.h)
@pytest.mark.parametrize("sampler", ["latent", "direct"])
@pytest.mark.parametrize("name", ["tiny", "ragged", "mid", "shipped"])
def test_golden_path_with_64x8_tiles_forced(modes, name, sampler):
    modes(tile_want=1, tile8=10 ** 9, sampler=sampler)
    g = load_case(name)
    _, gen, _ = build_modules(g, "cuda:0")
    inp = build_inputs(g, "cuda:0")
    with torch.no_grad():
        pred = gen(inp["feat"], inp["cond"], inp["record_len"], noise=eval_noise(g, "cuda:0"))["pred_feature"]
    assert_close(sub(pred, int(g["stride"])), g["pred_feature"], 1e-4, 1e-5, f"pred_feature ({sampler}, 64x8 tiles)")


@pytest.mark.parametrize("shape", [(8, 200, 704), (3, 100, 352), (2, 36, 68), (1, 8, 64), (2, 10, 60), (2, 18, 132)])
def test_64x8_tiles_equal_64x16_tiles(modes, shape):
    """Same arithmetic per pixel in both tilings: only the order of the f64 statistics atomics may differ."""
    from gencomm_amd import GenComm, synth
    n, H, W = shape
    gen = GenComm(synth.default_gencomm_cfg(64, 20)).eval()
    synth.fill_params_(gen, 5)
    gen = gen.to(DEV)
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(n, 66, H, W, generator=g, device=DEV)
    t = torch.full((n,), 4.0, device=DEV)
    ys = []
    for tile8 in (0, 10 ** 9, 0, 10 ** 9):
        modes(tile_want=1, tile8=tile8)
        with torch.no_grad():
            ys.append(gen.denoiser(x, t, T=20).clone())
    assert float((ys[0] - ys[2]).abs().max()) < 1e-6
    assert float((ys[1] - ys[3]).abs().max()) < 1e-6
    assert float((ys[0] - ys[1]).abs().max()) < 1e-6
    assert torch.isfinite(ys[1]).all()


def test_64x8_tiles_vs_oracle_with_partial_tiles(modes):
    """Height 34 -> 5 row tiles of 8 with 2 live rows in the last; half-resolution level 17 x 36 (3 row tiles, 1 live row)."""
    from gencomm_amd import Enhancer, GenComm, synth
    from oracle import torch_port as O
    modes(tile_want=1, tile8=10 ** 9)
    C, H, W, T, rl = 16, 34, 72, 3, [1, 3]
    n = sum(rl)
    cfg = synth.default_gencomm_cfg(C, T)
    gen, enh = GenComm(cfg).eval(), Enhancer(C, [8, 8], 4).eval()
    synth.fill_params_(gen, 41)
    synth.fill_params_(enh, 42)
    inp = {k: torch.from_numpy(v) for k, v in synth.make_inputs(rl, C, H, W, 43, max_shift=6.0).items()}
    n0, sn = (torch.from_numpy(a) for a in synth.make_eval_noise(44, n, C, H, W, T))
    ref = O.path_forward({k: v.detach() for k, v in gen.state_dict().items()}, {k: v.detach() for k, v in enh.state_dict().items()},
                         cfg, inp["feat"], inp["cond"], inp["record_len"], inp["pairwise_t_matrix"], H * 0.8, W * 0.8, n0, sn)
    gen = gen.to(DEV)
    with torch.no_grad():
        pred = gen(inp["feat"].to(DEV), inp["cond"].to(DEV), inp["record_len"], noise=(n0.to(DEV), sn.to(DEV)))["pred_feature"]
    assert_close(pred.cpu().numpy(), ref["pred_feature"].numpy(), 1e-4, 1e-5, "pred_feature (64x8 tiles)")


# ---------------------------------------------------------------------- persistent form (GENCOMM_MODE_PERSIST; conv8hp_kernel)
@pytest.mark.parametrize("shape,mask", [((16, 200, 704), 31), ((16, 200, 704), 2), ((13, 200, 704), 31), ((16, 104, 708), 31), ((6, 400, 704), 13)])
def test_persistent_kernels_equal_one_tile_per_workgroup(modes, shape, mask):
    """conv8hp_kernel walks several tiles per workgroup with the next tile's loads one tile ahead: the same tile function and products;
    what may differ is the order in which a product block's six terms meet (the low-register matrix phase) and the order of the f64
    statistics atomics.  Full-resolution launches of these shapes have more tiles than resident slots (the condition for the persistent
    form); partial tiles in both directions, an agent count that does not divide the slots, every variant mask on its own."""
    from gencomm_amd import GenComm, _lib, synth
    n, H, W = shape
    gen = GenComm(synth.default_gencomm_cfg(64, 20)).eval()
    synth.fill_params_(gen, 5)
    gen = gen.to(DEV)
    g = torch.Generator(device=DEV).manual_seed(13)
    x = torch.randn(n, 66, H, W, generator=g, device=DEV)
    t = torch.full((n,), 4.0, device=DEV)
    ys, logs = [], []
    for persist in (0, mask, 0, mask):
        modes(persist=persist)
        with torch.no_grad(), _lib.kernel_log() as kl:
            ys.append(gen.denoiser(x, t, T=20).clone())
        logs.append(dict(kl.counts))
    assert any("persistent" in k for k in logs[1]) and not any("persistent" in k for k in logs[0]), logs[1]
    scale = float(ys[0].abs().max())
    assert float((ys[0] - ys[2]).abs().max()) <= 1e-6 * scale
    assert float((ys[1] - ys[3]).abs().max()) <= 1e-6 * scale
    assert float((ys[0] - ys[1]).abs().max()) <= 2e-6 * scale, float((ys[0] - ys[1]).abs().max()) / scale
    assert torch.isfinite(ys[1]).all()
