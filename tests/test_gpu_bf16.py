"""bf16 denoise mode (BASELINE.json config 5; GENCOMM_MODE_ARITH = 2, csrc/conv8b_kernels.h): the UNet's 8-channel maps are
stored as bf16 and multiplied by single bf16 MFMAs. Accuracy is that of bf16 storage and is REPORTED here against the fp32
oracle, not held to the fp32 tolerance: relative rms error of the T-step result below 2 % (a bf16 value carries 8 significant
bits: 2^-9 = 0.2 % per stored value, ~26 stored maps per UNet call, T calls), no element further off than 15 % of the map's
largest magnitude; bit-reproducible up to the order of the statistics atomics; unsupported shapes are refused, not mis-run."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(C, H, W, T, rl, seed):
    from gencomm_amd import Enhancer, GenComm, synth
    cfg = synth.default_gencomm_cfg(C, T)
    gen, enh = GenComm(cfg).eval(), Enhancer(C, [8, 8], 4).eval()
    synth.fill_params_(gen, seed)
    synth.fill_params_(enh, seed + 1)
    inp = {k: torch.from_numpy(v) for k, v in synth.make_inputs(rl, C, H, W, seed + 2, max_shift=8.0).items()}
    n0, sn = (torch.from_numpy(a) for a in synth.make_eval_noise(seed + 3, sum(rl), C, H, W, T))
    return cfg, gen, enh, inp, n0, sn


@pytest.mark.parametrize("C,H,W,T,rl", [(64, 64, 128, 20, [4]), (128, 64, 128, 3, [2, 1]), (32, 48, 64, 5, [1, 2])])
def test_bf16_denoise_vs_fp32_oracle(modes, C, H, W, T, rl):
    from gencomm_amd import AttFusion, normalize_pairwise_tfm
    from oracle import torch_port as O
    cfg, gen, enh, inp, n0, sn = _setup(C, H, W, T, rl, 400 + C)
    ref = O.path_forward({k: v.detach() for k, v in gen.state_dict().items()}, {k: v.detach() for k, v in enh.state_dict().items()}, cfg,
                         inp["feat"], inp["cond"], inp["record_len"], inp["pairwise_t_matrix"], H * 0.8, W * 0.8, n0, sn)
    gen, enh = gen.to(DEV), enh.to(DEV)
    affine = normalize_pairwise_tfm(inp["pairwise_t_matrix"], H * 0.8, W * 0.8, 1)
    outs = {}
    for mode in ("split", "bf16", "bf16"):
        modes(arith=mode)
        with torch.no_grad():
            pred = gen(inp["feat"].to(DEV), inp["cond"].to(DEV), inp["record_len"], noise=(n0.to(DEV), sn.to(DEV)))["pred_feature"]
            fused = AttFusion(C)(enh(pred, affine, inp["record_len"]), inp["record_len"], affine)
        torch.cuda.synchronize()
        if mode in outs:
            assert float((pred.cpu() - outs[mode][0]).abs().max()) < 1e-2 * float(outs[mode][0].abs().max())  # repeatable (atomics order only)
        outs[mode] = (pred.cpu(), fused.cpu())
    for name, k, want in (("pred_feature", 0, ref["pred_feature"]), ("fused", 1, ref["fused"])):
        got = outs["bf16"][k]
        assert torch.isfinite(got).all()
        err = (got - want).abs()
        rel_rms = float(err.pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
        rel_max = float(err.max() / want.abs().max())
        f32_rms = float((outs["split"][k] - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
        print(f"bf16 denoise C={C} {H}x{W} T={T} {name}: relative rms error {rel_rms:.3e} (fp32 mode: {f32_rms:.1e}), max abs error / max |ref| {rel_max:.3e}")
        assert rel_rms < 2e-2 and rel_max < 0.15, (name, rel_rms, rel_max)


def test_bf16_mode_full_size_unet_call_vs_fp32_mode(modes):
    from gencomm_amd import GenComm, synth
    gen = GenComm(synth.default_gencomm_cfg(64, 20)).eval()
    synth.fill_params_(gen, 0)
    gen = gen.to(DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(4, 66, 200, 704, generator=g, device=DEV)
    t = torch.full((4,), 7.0, device=DEV)
    ys = {}
    for mode in ("split", "bf16"):
        modes(arith=mode)
        with torch.no_grad():
            ys[mode] = gen.denoiser(x, t, T=20).clone()
    rel = float((ys["bf16"] - ys["split"]).pow(2).mean().sqrt() / ys["split"].pow(2).mean().sqrt())
    print(f"bf16 vs fp32 mode, one full-size UNet call (4 x 64 x 200 x 704): relative rms difference {rel:.3e}")
    assert torch.isfinite(ys["bf16"]).all() and rel < 1e-2


def test_bf16_mode_refuses_what_it_does_not_cover(modes):
    from gencomm_amd import GenComm, _lib, synth
    modes(arith="bf16")
    gen = GenComm(synth.default_gencomm_cfg(16, 3)).eval().to(DEV)
    with pytest.raises(_lib.GenCommHipError):   # half-resolution width 18 is not a multiple of 4
        gen(torch.zeros(1, 16, 16, 36, device=DEV), torch.zeros(1, 2, 16, 36, device=DEV), [1], seed=1)
    gen.train()
    with pytest.raises(_lib.GenCommHipError):   # no backward in this mode
        out = gen(torch.zeros(1, 16, 16, 32, device=DEV), torch.zeros(1, 2, 16, 32, device=DEV), [1], seed=1)["pred_feature"]
        out.sum().backward()
