"""Model shells (SURVEY.md 8b plugin resolution + constructor/forward contract) against the reference's own stage-1
shell run on CPU (tests/golden/shell.npz, made by oracle/make_golden.py `shell`): checkpoint keys, output dict keys and
shapes, and -- on the GPU -- the numbers of every stage from the pillars to the detection heads.
The deformable conv inside MessageExtractorv2 is third-party (torchvision) arithmetic: the fixture ran the oracle's
restatement there (parity unpinned for that op, see oracle/torch_port.py)."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, assert_close, load_case, shell_noise, sub
from gencomm_amd import synth


def _spec():
    with open(os.path.join(GOLDEN, "shell_state_dict_keys.json")) as f:
        return json.load(f)


def _resolve(core_method):
    """opencood/tools/train_utils.py:269-287, with the package prefix swapped."""
    import importlib
    lib = importlib.import_module("gencomm_amd." + core_method)
    target = core_method.replace("_", "")
    model = None
    for name, cls in lib.__dict__.items():
        if name.lower() == target.lower():
            model = cls
    return model


@pytest.mark.parametrize("core_method", ["heter_model_baseline_w_gencomm_stage1", "heter_model_baseline_w_gencomm_stage2",
                                         "heter_model_baseline_w_gencomm"])
def test_plugin_resolution_finds_a_class(core_method):
    assert _resolve(core_method) is not None


def test_shell_checkpoint_keys_match_reference():
    spec = _spec()
    model = _resolve("heter_model_baseline_w_gencomm_stage1")(copy.deepcopy(spec["args"]))
    got = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert got == spec["state_dict"]


def test_stage2_accepts_both_key_spellings_and_freezes_fixed_modules():
    args = copy.deepcopy(_spec()["args"])
    cls = _resolve("heter_model_baseline_w_gencomm_stage2")
    m = cls(copy.deepcopy(args))
    args2 = copy.deepcopy(args)
    args2["diffcomm"] = args2.pop("gencomm")
    del args2["enhancer"]  # stage2.py:156 would crash here; the build only freezes an enhancer that exists
    m2 = cls(args2)
    assert not hasattr(m2, "enhancer")
    trainable = sorted({n.split(".")[0] for n, p in m.named_parameters() if p.requires_grad})
    assert trainable == []  # single modality == ego modality: every module is fixed (stage2.py:69-71)
    assert all(not b.training for b in m.backbone_m1.modules() if isinstance(b, torch.nn.BatchNorm2d))


def test_unsupported_blocks_fail_loudly():
    args = copy.deepcopy(_spec()["args"])
    cls = _resolve("heter_model_baseline_w_gencomm_stage1")
    bad = copy.deepcopy(args); bad["fusion_method"] = "v2xvit"
    with pytest.raises(NotImplementedError):
        cls(bad)
    bad = copy.deepcopy(args); bad["m1"]["core_method"] = "second"
    with pytest.raises(NotImplementedError):
        cls(bad)
    with_comp = copy.deepcopy(args); with_comp["compressor"] = {"input_dim": 128, "compress_ratio": 2}
    m = cls(with_comp)
    assert sorted({n.split(".")[0] for n, p in m.named_parameters() if p.requires_grad}) == ["compressor"]  # stage1.py:163-172


@pytest.mark.gpu
def test_shell_forward_vs_reference_golden():
    g = load_case("shell")
    spec = _spec()
    dev = "cuda:0"
    model = _resolve("heter_model_baseline_w_gencomm_stage1")(copy.deepcopy(spec["args"])).eval()
    synth.fill_params_(model, int(g["weight_seed"]))
    synth.fill_bn_stats_(model, int(g["bn_seed"]))
    model = model.to(dev)
    rl = [int(v) for v in g["record_len"]]
    pil = synth.make_pillars(int(g["M"]), sum(rl), int(g["nx"]), int(g["ny"]), int(g["data_seed"]), voxel_size=[0.4, 0.4, 4.0],
                             pc_range=spec["args"]["lidar_range"])
    ptm = synth.make_pairwise_t_matrix(rl, 5, int(g["pose_seed"]), max_shift=float(g["max_shift"]))
    data = {"agent_modality_list": ["m1"] * sum(rl), "record_len": torch.tensor(rl), "pairwise_t_matrix": torch.from_numpy(ptm).to(dev),
            "inputs_m1": {k: torch.from_numpy(pil[k]).to(dev) for k in ("voxel_features", "voxel_coords", "voxel_num_points")}}
    with torch.no_grad(), shell_noise(model.gencomm, int(g["noise_seed"]), sum(rl), 128, 16, 32, dev):
        out = model(data)
    assert sorted(out.keys()) == list(g["out_keys"])
    for k, shp in zip(("gt_feature", "pred_feature", "cls_preds", "reg_preds", "dir_preds", "message"), g["shapes"]):
        assert list(out[k].shape) == list(shp), k
    tol = dict(rtol=2e-4, atol=5e-5)  # 30+ fp32 layers deep
    assert_close(out["message"].cpu().numpy(), g["message"], what="message", **tol)
    assert_close(sub(out["gt_feature"], 5), g["gt_feature"], what="gt_feature", **tol)
    assert_close(sub(out["pred_feature"], 5), g["pred_feature"], what="pred_feature", **tol)
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        assert_close(out[k].cpu().numpy(), g[k], what=k, **tol)
