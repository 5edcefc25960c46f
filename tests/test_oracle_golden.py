"""Pin the CPU oracle (oracle/torch_port.py) against golden vectors produced by running the
reference's OWN modules (oracle/make_golden.py). CPU only.

Tolerances: the oracle and the reference execute the same ATen CPU kernels in the same order, so
they agree to float32 rounding; rtol 1e-5 / atol 1e-6 on tensors, exact-to-1-ulp on the schedule.
"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, assert_close, build_inputs, build_modules, eval_noise, load_case, sub, train_noise
from oracle import torch_port as O

CASES = ["tiny", "ragged", "mid", "shipped"]
RTOL, ATOL = 1e-5, 1e-6


def _sd(module):
    return {k: v.detach() for k, v in module.state_dict().items()}


@pytest.mark.parametrize("name", CASES)
def test_schedule_buffers_match_reference(name):
    g = load_case(name)
    sched = O.make_schedule(int(g["T"]))
    _, gen, _ = build_modules(g)
    for k, v in sched.items():
        ref = g["sched/" + k]
        np.testing.assert_array_max_ulp(v.numpy(), ref, maxulp=1)
        # the product module registers the same buffers (state_dict contract)
        np.testing.assert_array_max_ulp(getattr(gen, k).numpy(), ref, maxulp=1)


def test_T3_betas_known_values():
    # for T=3 betas = [0.005, 0.021656, 0.05]
    b = O.make_schedule(3)["betas"].numpy()
    np.testing.assert_allclose(b, [0.005, 0.021656, 0.05], rtol=2e-5)


@pytest.mark.parametrize("name", CASES)
def test_path_matches_reference(name):
    g = load_case(name)
    cfg, gen, enh = build_modules(g)
    inp = build_inputs(g)
    n0, sn = eval_noise(g)
    H, W, px = int(g["H"]), int(g["W"]), float(g["px_m"])
    out = O.path_forward(_sd(gen), _sd(enh), cfg, inp["feat"], inp["cond"], inp["record_len"],
                         inp["pairwise_t_matrix"], H * px, W * px, n0, sn)
    st = int(g["stride"])
    np.testing.assert_allclose(out["affine"].numpy(), g["affine"], rtol=0, atol=1e-12)
    assert tuple(out["pred_feature"].shape) == tuple(g["shape/pred_feature"])
    assert tuple(out["fused"].shape) == tuple(g["shape/fused"])
    assert_close(sub(out["pred_feature"], st), g["pred_feature"], RTOL, ATOL, "pred_feature")
    assert_close(sub(out["enhanced"], st), g["enhanced"], RTOL, ATOL, "enhanced")
    assert_close(sub(out["fused"], max(1, st // 2)), g["fused"], RTOL, ATOL, "fused")
    for k in ("pred_feature", "enhanced", "fused"):
        assert abs(out[k].abs().double().mean().item() - float(g["absmean/" + k])) < 1e-5
    # fusion without the enhancer in between (shells without an `enhancer:` yaml key)
    with torch.no_grad():
        f2 = O.att_fusion(out["pred_feature"], inp["record_len"], out["affine"])
    assert_close(sub(f2, max(1, st // 2)), g["fused_noenh"], RTOL, ATOL, "fused_noenh")


def test_unet_single_calls_match_reference():
    g = load_case("tiny")
    cfg, gen, _ = build_modules(g)
    inp = build_inputs(g)
    sd = _sd(gen)
    n = inp["feat"].shape[0]
    with torch.no_grad():
        for t in range(int(g["T"])):
            tt = torch.full((n,), t, dtype=torch.long)
            y = O.unet_forward(sd, "denoiser", torch.cat([inp["cond"], inp["feat"]], 1), tt.float(), cfg["model"])
            assert_close(y.numpy(), g[f"unet_out_t{t}"], RTOL, ATOL, f"unet t={t}")


def test_attnblock_variant_matches_reference():
    g = load_case("attn")
    cfg, gen, _ = build_modules(g, attn_resolutions=g["attn_resolutions"])
    assert len(gen.state_dict()) == int(g["n_keys"])
    inp = build_inputs(g)
    sd = _sd(gen)
    n = inp["feat"].shape[0]
    with torch.no_grad():
        for t in range(int(g["T"])):
            tt = torch.full((n,), t, dtype=torch.long)
            y = O.unet_forward(sd, "denoiser", torch.cat([inp["cond"], inp["feat"]], 1), tt.float(), cfg["model"])
            assert_close(y.numpy(), g[f"unet_out_t{t}"], RTOL, ATOL, f"unet(attn) t={t}")


def test_train_branch_and_grads_match_reference():
    """Training branch = same maths with per-agent RNG order (cond_diff.py:342-360); gradients
    flow through all T UNet calls (autograd over the oracle)."""
    g = load_case("tiny")
    cfg, gen, _ = build_modules(g)
    inp = build_inputs(g)
    n0, sn = train_noise(g)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and k.startswith("denoiser")) for k, v in gen.state_dict().items()}
    pred = O.gencomm_forward(sd, cfg, inp["feat"], inp["cond"], inp["record_len"], n0, sn, per_agent=True)
    assert_close(pred.detach().numpy(), g["pred_feature_train"], RTOL, ATOL, "pred_feature_train")
    loss = (pred ** 2).mean()
    assert abs(loss.item() - float(g["train_loss"])) < 1e-6
    loss.backward()
    for k in ("conv_in.weight", "conv_out.bias", "mid.block_1.norm1.weight"):
        assert_close(sd["denoiser." + k].grad.numpy(), g["grad/" + k], 1e-4, 1e-7, "grad " + k)


def test_state_dict_keys_match_reference():
    """Checkpoint contract: key names and shapes of GenComm(C=128,T=3) and
    Enhancer(128) equal the reference's, dumped by oracle/make_golden.py."""
    from gencomm_amd import Enhancer, GenComm, synth
    with open(os.path.join(GOLDEN, "state_dict_keys_C128_T3.json")) as f:
        ref = json.load(f)
    gen = GenComm(synth.default_gencomm_cfg(128, 3))
    enh = Enhancer(128, [8, 8], 4)
    assert {k: list(v.shape) for k, v in gen.state_dict().items()} == ref["gencomm"]
    assert list(gen.state_dict().keys()) == list(ref["gencomm"].keys())
    assert {k: list(v.shape) for k, v in enh.state_dict().items()} == ref["enhancer"]
    assert list(enh.state_dict().keys()) == list(ref["enhancer"].keys())
    assert sum(p.numel() for p in gen.parameters()) == ref["gencomm_param_count"] == 43176
    assert sum(p.numel() for p in enh.parameters()) == ref["enhancer_param_count"] == 564924
    assert "lvlb_weights" not in gen.state_dict()  # persistent=False, cond_diff.py:257
