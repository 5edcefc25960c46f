"""General DiffusionUNet widths (ch = 16, ch_mult [1, 2], one res-block per level: outside the accelerated family) against the
reference's own modules -- tests/golden/unet_wide.npz, written by oracle/make_golden.py `wide`: the same state_dict keys and shapes, every
single UNet call at the elementwise tolerance, the whole eval forward with the reference's injected noise; and the guards (no gradients,
no AttnBlocks on this path)."""
import json

import numpy as np
import pytest
import torch

from helpers import load_case

DEV = "cuda:0"


def _gen(g):
    from gencomm_amd import GenComm, synth
    gen = GenComm(json.loads(str(g["cfg"]))).eval()
    synth.fill_params_(gen, int(g["weight_seed"]))
    return gen


def test_wide_unet_has_the_reference_state_dict():
    from gencomm_amd.unet_generic import GenericDiffusionUNet
    g = load_case("unet_wide")
    gen = _gen(g)
    assert isinstance(gen.denoiser, GenericDiffusionUNet)
    sd = gen.state_dict()
    assert sorted(sd.keys()) == list(g["keys"])
    ref_shapes = json.loads(str(g["shapes"]))
    assert {k: list(v.shape) for k, v in sd.items()} == ref_shapes


@pytest.mark.gpu
def test_wide_unet_calls_and_eval_forward_vs_reference_golden():
    from gencomm_amd import synth
    g = load_case("unet_wide")
    C, H, W, T = (int(g[k]) for k in ("C", "H", "W", "T"))
    rl = [int(v) for v in g["record_len"]]
    n = sum(rl)
    gen = _gen(g).to(DEV)
    inp = synth.make_inputs(rl, C, H, W, int(g["data_seed"]))
    feat, cond = torch.from_numpy(inp["feat"]).to(DEV), torch.from_numpy(inp["cond"]).to(DEV)
    with torch.no_grad():
        for t in range(T):
            got = gen.denoiser(torch.cat([cond, feat], dim=1), torch.full((n,), float(t), device=DEV)).cpu().numpy()
            ref = g[f"unet_out_t{t}"]
            err = np.abs(got - ref)
            assert (err <= 1e-5 + 1e-4 * np.abs(ref)).all(), (t, err.max())
        n0, sn = (torch.from_numpy(a).to(DEV) for a in synth.make_eval_noise(int(g["noise_seed"]), n, C, H, W, T))
        pred = gen(feat, cond, torch.tensor(rl), noise=(n0, sn))["pred_feature"].cpu().numpy()
    ref = g["pred_feature"]
    err = np.abs(pred - ref)
    print(f"wide UNet, T = {T} chain: max abs err {err.max():.3e}, worst err / tol {(err / (1e-5 + 1e-4 * np.abs(ref))).max():.3f}")
    assert (err <= 1e-5 + 1e-4 * np.abs(ref)).all()
    # in-kernel noise: deterministic per seed, different across seeds
    with torch.no_grad():
        a = gen(feat, cond, torch.tensor(rl), seed=5)["pred_feature"]
        b = gen(feat, cond, torch.tensor(rl), seed=5)["pred_feature"]
        c = gen(feat, cond, torch.tensor(rl), seed=6)["pred_feature"]
    assert torch.equal(a, b) and float((a - c).abs().max()) > 1e-3


@pytest.mark.gpu
def test_wide_unet_guards():
    from gencomm_amd import GenComm, synth
    g = load_case("unet_wide")
    gen = _gen(g).to(DEV).train()
    C, H, W = int(g["C"]), int(g["H"]), int(g["W"])
    x, c = torch.randn(1, C, H, W, device=DEV), torch.randn(1, 2, H, W, device=DEV)
    with pytest.raises(NotImplementedError, match="general-width"):
        gen(x, c, [1])
    cfg = json.loads(str(g["cfg"]))
    cfg["model"]["attn_resolutions"] = [64]
    with pytest.raises(NotImplementedError, match="AttnBlocks"):
        GenComm(cfg)


def test_both_unet_families_survive_deepcopy_and_pickle():
    """copy.deepcopy / pickle rebuild a module with cls.__new__(cls): the width-dispatching __new__ must accept that."""
    import copy
    import pickle
    from gencomm_amd import GenComm, synth
    from gencomm_amd.unet import DiffusionUNet
    from gencomm_amd.unet_generic import GenericDiffusionUNet
    narrow = GenComm(synth.default_gencomm_cfg(16, 3))
    wide = _gen(load_case("unet_wide"))
    for gen, cls in ((narrow, DiffusionUNet), (wide, GenericDiffusionUNet)):
        for clone in (copy.deepcopy(gen), pickle.loads(pickle.dumps(gen))):
            assert type(clone.denoiser) is cls
            a, b = gen.state_dict(), clone.state_dict()
            assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
