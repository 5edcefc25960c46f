"""HeterModelLate (BASELINE.json configs[0]: PointPillars ego-only, no fusion -- the reference's single-agent pre-training model,
heter_model_late.py) with its ResNetBEVBackbone. CPU: the oracle restatement against the golden vector the reference's OWN model
produced on CPU (tests/golden/late.npz, oracle/make_golden.py `late`), checkpoint keys and plugin resolution. GPU: the HIP model
against the same golden vector, and one training step."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, assert_close, load_case
from gencomm_amd import synth


def _spec():
    with open(os.path.join(GOLDEN, "late_state_dict_keys.json")) as f:
        return json.load(f)


def _model_and_data(device="cpu"):
    from gencomm_amd.heter_model_late import HeterModelLate
    g, spec = load_case("late"), _spec()
    model = HeterModelLate(copy.deepcopy(spec["args"])).eval()
    synth.fill_params_(model, int(g["weight_seed"]))
    synth.fill_bn_stats_(model, int(g["bn_seed"]))
    pil = synth.make_pillars(int(g["M"]), 1, int(g["nx"]), int(g["ny"]), int(g["data_seed"]), voxel_size=[0.4, 0.4, 4.0],
                             pc_range=spec["args"]["lidar_range"])
    data = {"inputs_m1": {k: torch.from_numpy(pil[k]).to(device) for k in ("voxel_features", "voxel_coords", "voxel_num_points")}}
    return g, spec, model.to(device), data


def test_checkpoint_keys_match_the_reference_and_the_plugin_resolves():
    import importlib
    g, spec, model, _ = _model_and_data()
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == spec["state_dict"]
    lib = importlib.import_module("gencomm_amd.heter_model_late")        # train_utils.py:269-287 with the package prefix swapped
    assert any(name.lower() == "hetermodellate" for name in lib.__dict__)


def test_oracle_matches_reference_golden():
    from oracle import torch_port as O
    g, spec, model, data = _model_and_data()
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    d = data["inputs_m1"]
    with torch.no_grad():
        out = O.late_model_forward(sd, spec["args"], d["voxel_features"], d["voxel_coords"], d["voxel_num_points"])
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        assert_close(out[k].numpy(), g[k], 1e-4, 1e-5, "late oracle " + k)


@pytest.mark.gpu
def test_hip_late_model_vs_reference_golden():
    g, spec, model, data = _model_and_data("cuda:0")
    with torch.no_grad():
        out = model(data)
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        assert list(out[k].shape) == list(g[k].shape)
        assert_close(out[k].cpu().numpy(), g[k], 2e-4, 5e-5, "late HIP " + k)      # 40 fp32 conv layers deep


@pytest.mark.gpu
def test_hip_late_model_training_step():
    """Pre-training a new agent type = training this model (BatchNorm with batch statistics): one step, finite gradients everywhere."""
    g, spec, model, data = _model_and_data("cuda:0")
    model.train()
    out = model(data)
    (out["cls_preds"].square().mean() + out["reg_preds"].square().mean() + out["dir_preds"].square().mean()).backward()
    unused = [n for n, p in model.named_parameters() if p.grad is None]
    assert all(n.startswith("layers_m1.resnet.layer0") for n in unused), unused     # level 0 of `layers` is never used (heter_model_late.py:93-101)
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n
