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
    bad = copy.deepcopy(args); bad["fusion_method"] = "v2vnet"
    with pytest.raises(NotImplementedError):
        cls(bad)
    bad = copy.deepcopy(args); bad["m1"]["core_method"] = "lift_splat_shoot"; bad["m1"]["sensor_type"] = "camera"
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


def _v2xvit_args():
    """The `v2xvit` block of opv2v/GenComm_yamls/gencomm/stage1/m1_v2xvit.yaml:137-171 (stored with the V2X-ViT fixture)."""
    g = load_case("v2xvit")
    return json.loads(str(g["args"]))


def test_v2xvit_shell_constructs_with_reference_checkpoint_keys():
    """`fusion_method: v2xvit` (every *_v2xvit.yaml): the shell builds, and its fusion_net carries the reference's 134 keys."""
    args = copy.deepcopy(_spec()["args"])
    args["fusion_method"] = "v2xvit"
    args["v2xvit"] = _v2xvit_args()
    for core in ("heter_model_baseline_w_gencomm_stage1", "heter_model_baseline_w_gencomm"):
        m = _resolve(core)(copy.deepcopy(args))
        with open(os.path.join(GOLDEN, "v2xvit_state_dict_keys.json")) as f:
            ref = json.load(f)
        got = {k[len("fusion_net."):]: list(v.shape) for k, v in m.state_dict().items() if k.startswith("fusion_net.")}
        assert got == ref


@pytest.mark.gpu
def test_v2xvit_shell_forward_runs_on_the_hip_path():
    """Stage-1 shell with V2X-ViT fusion end to end on the GPU (pillars -> backbone -> GenComm -> Enhancer -> V2X-ViT -> heads):
    the same inputs as the golden shell case; the fused map must equal V2XViTFusion applied to the Enhancer output by the
    oracle (everything upstream of the fusion is covered by test_shell_forward_vs_reference_golden)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import v2xvit_port as V
    from torch_port import normalize_pairwise_tfm as npt_oracle
    g = load_case("shell")
    spec = _spec()
    dev = "cuda:0"
    args = copy.deepcopy(spec["args"])
    args["fusion_method"] = "v2xvit"
    args["v2xvit"] = _v2xvit_args()
    model = _resolve("heter_model_baseline_w_gencomm_stage1")(args).eval()
    synth.fill_params_(model, int(g["weight_seed"]))
    synth.fill_bn_stats_(model, int(g["bn_seed"]))
    model = model.to(dev)
    rl = [int(v) for v in g["record_len"]]
    pil = synth.make_pillars(int(g["M"]), sum(rl), int(g["nx"]), int(g["ny"]), int(g["data_seed"]), voxel_size=[0.4, 0.4, 4.0],
                             pc_range=spec["args"]["lidar_range"])
    ptm = synth.make_pairwise_t_matrix(rl, 5, int(g["pose_seed"]), max_shift=float(g["max_shift"]))
    data = {"agent_modality_list": ["m1"] * sum(rl), "record_len": torch.tensor(rl), "pairwise_t_matrix": torch.from_numpy(ptm).to(dev),
            "inputs_m1": {k: torch.from_numpy(pil[k]).to(dev) for k in ("voxel_features", "voxel_coords", "voxel_num_points")}}
    seen = {}
    hook = model.fusion_net.register_forward_hook(lambda mod, inp, out: seen.update(x=inp[0].detach().cpu(), aff=inp[2], out=out.detach().cpu()))
    with torch.no_grad(), shell_noise(model.gencomm, int(g["noise_seed"]), sum(rl), 128, 16, 32, dev):
        out = model(data)
    hook.remove()
    assert torch.isfinite(out["cls_preds"]).all() and out["cls_preds"].shape[0] == len(rl)
    sd = {k[len("fusion_net."):] if False else k: v.detach().cpu() for k, v in model.fusion_net.state_dict().items()}
    ref = V.v2xvit_fusion(sd, args["v2xvit"], seen["x"], rl, seen["aff"].cpu())
    assert_close(seen["out"].numpy(), ref.numpy(), 1e-4, 1e-5, "V2X-ViT fused map inside the shell")


@pytest.mark.gpu
def test_where2comm_shell_forward_runs_on_the_hip_path():
    """Stage-1 shell with `fusion_method: where2comm` (MoreModality/Diffcomm/*/m1_diffcomm_where2comm.yaml:130-131) end to end on the GPU:
    the fused map must equal Where2commFusion applied to the Enhancer output by the oracle."""
    from oracle import torch_port as O
    g = load_case("shell")
    spec = _spec()
    dev = "cuda:0"
    args = copy.deepcopy(spec["args"])
    args["fusion_method"], args["where2comm"] = "where2comm", 128
    model = _resolve("heter_model_baseline_w_gencomm_stage1")(args).eval()
    assert sorted(k for k in model.state_dict() if k.startswith("fusion_net."))[0] == "fusion_net.mha_fusion.attn.in_proj_bias"
    synth.fill_params_(model, int(g["weight_seed"]))
    synth.fill_bn_stats_(model, int(g["bn_seed"]))
    model = model.to(dev)
    rl = [int(v) for v in g["record_len"]]
    pil = synth.make_pillars(int(g["M"]), sum(rl), int(g["nx"]), int(g["ny"]), int(g["data_seed"]), voxel_size=[0.4, 0.4, 4.0],
                             pc_range=spec["args"]["lidar_range"])
    ptm = synth.make_pairwise_t_matrix(rl, 5, int(g["pose_seed"]), max_shift=float(g["max_shift"]))
    data = {"agent_modality_list": ["m1"] * sum(rl), "record_len": torch.tensor(rl), "pairwise_t_matrix": torch.from_numpy(ptm).to(dev),
            "inputs_m1": {k: torch.from_numpy(pil[k]).to(dev) for k in ("voxel_features", "voxel_coords", "voxel_num_points")}}
    seen = {}
    hook = model.fusion_net.register_forward_hook(lambda mod, inp, out: seen.update(x=inp[0].detach().cpu(), aff=inp[2], out=out.detach().cpu()))
    with torch.no_grad(), shell_noise(model.gencomm, int(g["noise_seed"]), sum(rl), 128, 16, 32, dev):
        out = model(data)
    hook.remove()
    assert torch.isfinite(out["cls_preds"]).all() and out["cls_preds"].shape[0] == len(rl)
    sd = {k: v.detach().cpu() for k, v in model.fusion_net.state_dict().items()}
    with torch.no_grad():
        ref = O.where2comm_fusion(sd, seen["x"], rl, seen["aff"].cpu())
    assert_close(seen["out"].numpy(), ref.numpy(), 1e-4, 1e-5, "Where2comm fused map inside the shell")


def _second_shell_args():
    """m3 of opv2v/GenComm_yamls/gencomm/stage1/m3_att.yaml:101-133 (SECOND encoder, 0.1 m voxels, stride-1 first backbone block)
    on the golden shell's lidar range."""
    args = copy.deepcopy(_spec()["args"])
    rng_ = args["lidar_range"]
    args["m1"] = {"core_method": "second", "sensor_type": "lidar",
                  "encoder_args": {"voxel_size": [0.1, 0.1, 0.1], "lidar_range": rng_, "mean_vfe": {"num_point_features": 4},
                                   "spconv": {"num_features_in": 4, "num_features_out": 64}, "map2bev": {"feature_num": 128}},
                  "backbone_args": {"layer_nums": [3, 5, 8], "layer_strides": [1, 2, 2], "num_filters": [64, 128, 256],
                                    "upsample_strides": [1, 2, 4], "num_upsample_filter": [128, 128, 128], "inplanes": 128},
                  "shrink_header": {"kernal_size": [3], "stride": [2], "padding": [1], "dim": [128], "input_dim": 384}}
    return args


def test_second_shell_constructs_with_encoder_keys():
    m = _resolve("heter_model_baseline_w_gencomm_stage1")(_second_shell_args())
    keys = [k for k in m.state_dict() if k.startswith("encoder_m1.")]
    assert "encoder_m1.spconv_block.conv_input.0.weight" in keys and "encoder_m1.spconv_block.conv_out.1.running_var" in keys
    assert list(m.state_dict()["encoder_m1.spconv_block.conv4.0.0.weight"].shape) == [64, 3, 3, 3, 64]
    assert list(m.state_dict()["backbone_m1.blocks.0.1.weight"].shape)[:2] == [64, 128]      # inplanes 128 = 64 channels x 2 height slices


@pytest.mark.gpu
def test_second_shell_forward_runs_on_the_hip_path():
    """Stage-1 shell with the SECOND encoder end to end on the GPU (voxels -> sparse 3-D backbone -> BEV backbone -> GenComm ->
    Enhancer -> AttFusion -> heads); the encoder's output inside the shell must equal the dense-volume oracle's."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import second_port as S
    from test_second import _voxels
    g = load_case("shell")
    dev = "cuda:0"
    args = _second_shell_args()
    model = _resolve("heter_model_baseline_w_gencomm_stage1")(args).eval()
    synth.fill_params_(model, int(g["weight_seed"]))
    synth.fill_bn_stats_(model, int(g["bn_seed"]))
    rl = [2, 1]
    vf, vc, vn = _voxels(np.random.RandomState(11), [4000, 2500, 3000], 512, 256, 40)
    enc_sd = {k[len("encoder_m1."):]: v.detach().clone() for k, v in model.state_dict().items() if k.startswith("encoder_m1.")}
    ref = S.second_forward(enc_sd, "", vf, vc, vn, [512, 256, 40])
    model = model.to(dev)
    ptm = synth.make_pairwise_t_matrix(rl, 5, int(g["pose_seed"]), max_shift=float(g["max_shift"]))
    data = {"agent_modality_list": ["m1"] * sum(rl), "record_len": torch.tensor(rl), "pairwise_t_matrix": torch.from_numpy(ptm).to(dev),
            "inputs_m1": {"voxel_features": vf.to(dev), "voxel_coords": vc.to(dev), "voxel_num_points": vn.to(dev)}}
    seen = {}
    hook = model.encoder_m1.register_forward_hook(lambda mod, inp, out: seen.update(out=out.detach().cpu()))
    with torch.no_grad(), shell_noise(model.gencomm, int(g["noise_seed"]), sum(rl), 128, 16, 32, dev):
        out = model(data)
    hook.remove()
    assert list(seen["out"].shape) == [3, 128, 32, 64]
    assert_close(seen["out"].numpy(), ref.numpy(), 1e-4, 1e-5, "SECOND output inside the shell")
    assert list(out["cls_preds"].shape) == [2, 2, 16, 32] and list(out["pred_feature"].shape) == [3, 128, 16, 32]
    for k in ("cls_preds", "reg_preds", "dir_preds", "pred_feature", "message"):
        assert torch.isfinite(out[k]).all(), k


def test_stage2_shell_has_the_reference_keys_and_frozen_set():
    """tests/golden/shell2.npz: the reference's own stage-2 shell (two lidar modalities, `diffcomm:` key spelling, `trick`): same
    state_dict keys, and the same parameters frozen by model_train_init (stage2.py:180-185: everything but the new agent type's
    message extractor)."""
    g = load_case("shell2")
    args = json.loads(str(g["args"]))
    model = _resolve("heter_model_baseline_w_gencomm_stage2")(copy.deepcopy(args))
    assert sorted(model.state_dict().keys()) == list(g["keys"])
    assert sorted(n for n, p in model.named_parameters() if not p.requires_grad) == list(g["frozen"])


@pytest.mark.gpu
def test_stage2_shell_forward_vs_reference_golden():
    g = load_case("shell2")
    args = json.loads(str(g["args"]))
    dev = "cuda:0"
    model = _resolve("heter_model_baseline_w_gencomm_stage2")(copy.deepcopy(args)).eval()
    synth.fill_params_(model, int(g["weight_seed"]))
    synth.fill_bn_stats_(model, int(g["bn_seed"]))
    model = model.to(dev)
    rl, mods = [int(v) for v in g["record_len"]], [str(m) for m in g["mods"]]
    ptm = synth.make_pairwise_t_matrix(rl, 5, int(g["pose_seed"]), max_shift=float(g["max_shift"]))
    data = {"agent_modality_list": mods, "record_len": torch.tensor(rl), "pairwise_t_matrix": torch.from_numpy(ptm).to(dev)}
    for i, m in enumerate(("m1", "m2")):
        pil = synth.make_pillars(int(g["M"]), mods.count(m), int(g["nx"]), int(g["ny"]), int(g["data_seed"]) + i, voxel_size=[0.4, 0.4, 4.0],
                                 pc_range=args["lidar_range"])
        data[f"inputs_{m}"] = {k: torch.from_numpy(pil[k]).to(dev) for k in ("voxel_features", "voxel_coords", "voxel_num_points")}
    with torch.no_grad(), shell_noise(model.gencomm, int(g["noise_seed"]), sum(rl), 128, 16, 32, dev):
        out = model(data)
    assert sorted(out.keys()) == list(g["out_keys"])
    tol = dict(rtol=2e-4, atol=5e-5)
    assert_close(out["message"].cpu().numpy(), g["message"], what="message", **tol)
    assert_close(sub(out["gt_feature"], 5), g["gt_feature"], what="gt_feature", **tol)
    assert_close(sub(out["pred_feature"], 5), g["pred_feature"], what="pred_feature", **tol)
    for k in ("cls_preds", "reg_preds", "dir_preds"):
        assert_close(out[k].cpu().numpy(), g[k], what=k, **tol)


@pytest.mark.parametrize("order", [["m1"] * 4, ["m1", "m2", "m1", "m1"], ["m2", "m1"], ["m1", "m1", "m2", "m2", "m1"], ["m1", "m1", "m1"]])
def test_agent_assembly_equals_select_and_stack(order):
    """heter_model.assemble_agents (slices / the tensor itself) against the reference's per-agent select + stack (stage1.py:213-224):
    same values and the same gradients; a modality batch longer than its agents in the list is not aliased."""
    from gencomm_amd.heter_model import assemble_agents
    g = torch.Generator().manual_seed(len(order))
    extra = 1 if order == ["m1", "m1", "m1"] else 0
    per = {m: torch.randn(order.count(m) + extra, 3, 2, 5, generator=g).requires_grad_() for m in sorted(set(order))}
    counting, rows = {m: 0 for m in per}, []
    for m in order:
        rows.append(per[m][counting[m]])
        counting[m] += 1
    want = torch.stack(rows)
    got = assemble_agents(per, order)
    assert torch.equal(got, want)
    w = torch.randn(want.shape, generator=g)
    g_want = torch.autograd.grad((want * w).sum(), list(per.values()))
    g_got = torch.autograd.grad((got * w).sum(), list(per.values()))
    assert all(torch.equal(a, b) for a, b in zip(g_got, g_want))
    if len(set(order)) == 1 and not extra:
        assert got is per[order[0]]     # one modality: no copy
