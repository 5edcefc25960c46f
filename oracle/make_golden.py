#!/usr/bin/env python3
"""ORACLE tooling (test infrastructure): generate the golden vectors in ``tests/golden``.

Runs ONLY where the reference checkout is mounted at /root/reference (the build container).
It imports the reference's own hot-path modules -- nothing of the reference is copied -- with
import-only stubs for packages that are absent here and are dead code on this path
(``icecream``, ``timm.models.layers``, ``shapely.geometry``, ``pyquaternion``),
feeds them the deterministic synthetic weights / inputs / noise of ``gencomm_amd.synth`` and stores
inputs' *seeds* and the reference's *outputs* as small ``.npz`` fixtures.

    python oracle/make_golden.py            # rewrites tests/golden/*.npz

``torch.randn`` / ``torch.randn_like`` are patched while the reference runs so that the k-th draw
of a forward returns ``synth.noise_stream(seed, k, shape)``; the draw order is the reference's own.
"""
from __future__ import annotations

import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.nn as nn

from gencomm_amd import synth

REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")


def _install_stubs() -> None:
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class DropPath(nn.Module):  # instantiated then overwritten by nn.Identity (enhancer.py:343-344)
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    _install_auto_stubs()
    stub("icecream", ic=lambda *a, **k: None)
    stub("timm")
    stub("timm.models")
    stub("timm.models.layers", DropPath=DropPath, PatchEmbed=object, Mlp=object,
         trunc_normal_=lambda *a, **k: None, lecun_normal_=lambda *a, **k: None,
         to_2tuple=lambda x: (x, x))


def _install_auto_stubs() -> None:
    """Import-time stand-ins for third-party packages the model SHELLS import but the hot path never calls
    (torchvision, cv2, spconv, matplotlib ...): any attribute is a do-nothing class. The one member that is
    executed -- torchvision.ops.DeformConv2d in MessageExtractorv2 -- is bound to the oracle's restatement of the
    published DCNv1 formula (third-party arithmetic, PARITY UNPINNED; see oracle/torch_port.py)."""
    import importlib.abc
    import importlib.machinery

    auto = ("torchvision", "cv2", "open3d", "spconv", "matplotlib", "efficientnet_pytorch", "numba", "tensorboardX",
            "swanlab", "wandb", "h5py", "termcolor", "easydict", "cumm", "seaborn", "shapely", "pyquaternion")

    class _Auto(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            if k[0].islower():  # sub-module or function: another callable stand-in module
                m = _Auto(self.__name__ + "." + k)
                m.__path__ = []
                setattr(self, k, m)
                return m
            return type(k, (), {"__init__": lambda self, *a, **kw: None, "__call__": lambda self, *a, **kw: None})

        def __call__(self, *a, **kw):
            return None

    class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path, target=None):
            if name.split(".")[0] in auto and name not in sys.modules:
                try:
                    if importlib.machinery.PathFinder.find_spec(name, path):
                        return None
                except Exception:
                    pass
                return importlib.machinery.ModuleSpec(name, self, is_package=True)

        def create_module(self, spec):
            m = _Auto(spec.name)
            m.__path__ = []
            return m

        def exec_module(self, m):
            pass

    sys.meta_path.append(_Finder())
    import torch_port as O  # oracle/ is on sys.path (this script lives there)

    class DeformConv2d(nn.Module):
        def __init__(self, in_channels, out_channels, kernel_size=3, padding=1):
            super().__init__()
            self.padding = padding
            self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
            self.bias = nn.Parameter(torch.empty(out_channels))

        def forward(self, x, offset):
            return O.deform_conv2d_ref(x, offset, self.weight, self.bias, self.padding)

    ops = types.ModuleType("torchvision.ops")
    ops.DeformConv2d = DeformConv2d
    import torchvision  # the auto stub
    sys.modules["torchvision.ops"] = ops
    torchvision.ops = ops


class PatchedNoise:
    """Route the reference's torch.randn / randn_like through synth.noise_stream, in call order."""

    def __init__(self, seed: int):
        self.seed, self.k = seed, 0

    def _draw(self, shape):
        v = torch.from_numpy(synth.noise_stream(self.seed, self.k, tuple(shape)))
        self.k += 1
        return v

    def __enter__(self):
        self._randn, self._randn_like = torch.randn, torch.randn_like
        torch.randn = lambda *size, **kw: self._draw(size[0] if len(size) == 1 and not isinstance(size[0], int) else size)
        torch.randn_like = lambda x, **kw: self._draw(x.shape)
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._randn_like


def sub(a: np.ndarray, stride: int) -> np.ndarray:
    return np.ascontiguousarray(a.reshape(-1)[::stride])


CASES = [
    # name, C, H, W, record_len, T, px_m, stride (1 = store whole tensors), extras
    dict(name="tiny", C=8, H=16, W=24, record_len=[2, 1], T=3, px_m=1.6, stride=1, unet_calls=True, train=True),
    dict(name="ragged", C=16, H=18, W=26, record_len=[1, 3, 2], T=4, px_m=1.6, stride=1),
    dict(name="mid", C=64, H=20, W=36, record_len=[4], T=20, px_m=0.4, stride=7),
    dict(name="shipped", C=128, H=64, W=128, record_len=[2], T=3, px_m=1.6, stride=61),
]
WEIGHT_SEED, DATA_SEED, NOISE_SEED = 0, 1, 2


def run_case(case: dict) -> None:
    from opencood.models.gencomm_modules.cond_diff import GenComm
    from opencood.models.gencomm_modules.enhancer import Enhancer
    from opencood.models.fuse_modules.fusion_in_one import AttFusion
    from opencood.utils.transformation_utils import normalize_pairwise_tfm

    C, H, W, T = case["C"], case["H"], case["W"], case["T"]
    rl = case["record_len"]
    n = sum(rl)
    cfg = synth.default_gencomm_cfg(C, T)
    gen = GenComm(cfg).eval()
    enh = Enhancer(C, [8, 8], 4).eval()
    fus = AttFusion(C).eval()
    synth.fill_params_(gen, WEIGHT_SEED)
    synth.fill_params_(enh, WEIGHT_SEED + 1)

    bev_h_m, bev_w_m = H * case["px_m"], W * case["px_m"]
    inp = synth.make_inputs(rl, C, H, W, DATA_SEED, max_shift=0.15 * bev_w_m)
    feat, cond = torch.from_numpy(inp["feat"]), torch.from_numpy(inp["cond"])
    record_len = torch.from_numpy(inp["record_len"])
    ptm = torch.from_numpy(inp["pairwise_t_matrix"])

    rec = dict(C=C, H=H, W=W, T=T, record_len=np.asarray(rl), px_m=case["px_m"], stride=case["stride"],
               weight_seed=WEIGHT_SEED, data_seed=DATA_SEED, noise_seed=NOISE_SEED,
               max_shift=0.15 * bev_w_m)
    st = case["stride"]
    with torch.no_grad():
        for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
                  "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                  "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
            rec["sched/" + k] = getattr(gen, k).numpy()
        if case.get("unet_calls"):
            for t in range(T):
                tt = torch.full((n,), t, dtype=torch.long)
                rec[f"unet_out_t{t}"] = gen.denoiser(torch.cat([cond, feat], dim=1), tt.float()).numpy()
        with PatchedNoise(NOISE_SEED) as pn:
            out = gen(feat, cond, record_len)
            assert pn.k == 3 + T, pn.k
        pred = out["pred_feature"]
        # the model shell normalises with the BEV extent in metres and discrete_ratio 1
        # (heter_model_baseline_w_gencomm_stage1.py:96-98, :177)
        affine = normalize_pairwise_tfm(ptm.clone(), bev_h_m, bev_w_m, 1)
        enhd = enh(pred, affine, record_len)
        fused = fus(enhd, record_len, affine)
        fused_noenh = fus(pred, record_len, affine)
    rec["affine"] = affine.numpy()
    rec["pred_feature"] = sub(pred.numpy(), st)
    rec["enhanced"] = sub(enhd.numpy(), st)
    rec["fused"] = sub(fused.numpy(), max(1, st // 2))
    rec["fused_noenh"] = sub(fused_noenh.numpy(), max(1, st // 2))
    for k in ("pred_feature", "enhanced", "fused"):
        rec["absmean/" + k] = np.float64({"pred_feature": pred, "enhanced": enhd, "fused": fused}[k].abs().double().mean().item())
    rec["shape/pred_feature"] = np.asarray(pred.shape)
    rec["shape/fused"] = np.asarray(fused.shape)

    if case.get("train"):
        gen.train()
        with PatchedNoise(NOISE_SEED) as pn:
            out = gen(feat, cond, record_len)
            assert pn.k == n * (T + 1), pn.k
        ptrain = out["pred_feature"]
        loss = (ptrain ** 2).mean()
        loss.backward()
        rec["pred_feature_train"] = ptrain.detach().numpy()
        rec["train_loss"] = np.float64(loss.item())
        rec["grad/conv_in.weight"] = gen.denoiser.conv_in.weight.grad.numpy()
        rec["grad/conv_out.bias"] = gen.denoiser.conv_out.bias.grad.numpy()
        rec["grad/mid.block_1.norm1.weight"] = gen.denoiser.mid.block_1.norm1.weight.grad.numpy()
        gen.eval()

    path = os.path.join(OUT, f"{case['name']}.npz")
    np.savez_compressed(path, **rec)
    print(f"{case['name']}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB) "
          f"|pred|={rec['absmean/pred_feature']:.4f} |fused|={rec['absmean/fused']:.4f}")


def run_attn_case() -> None:
    """AttnBlock exists only when the nominal resolution counter (128, halved per level,
    unet.py:211,:237) is in attn_resolutions; [64] with ch_mult [1,1] gives 5 blocks at H/2 x W/2."""
    from opencood.models.gencomm_modules.cond_diff import GenComm
    C, H, W, T, n = 8, 16, 16, 3, 2
    cfg = synth.default_gencomm_cfg(C, T)
    cfg["model"]["attn_resolutions"] = [64]
    gen = GenComm(cfg).eval()
    synth.fill_params_(gen, WEIGHT_SEED)
    inp = synth.make_inputs([n], C, H, W, DATA_SEED)
    feat, cond = torch.from_numpy(inp["feat"]), torch.from_numpy(inp["cond"])
    rec = dict(C=C, H=H, W=W, T=T, record_len=np.asarray([n]), weight_seed=WEIGHT_SEED, data_seed=DATA_SEED,
               attn_resolutions=np.asarray([64]), n_keys=len(gen.state_dict()))
    with torch.no_grad():
        for t in range(T):
            tt = torch.full((n,), t, dtype=torch.long)
            rec[f"unet_out_t{t}"] = gen.denoiser(torch.cat([cond, feat], dim=1), tt.float()).numpy()
    path = os.path.join(OUT, "attn.npz")
    np.savez_compressed(path, **rec)
    print(f"attn: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), state_dict keys {rec['n_keys']}")


def run_pillar_case() -> None:
    """PointPillars front half: the reference's own PillarVFE + PointPillarScatter
    (torch-only modules, opencood/models/sub_modules/{pillar_vfe,point_pillar_scatter}.py) in eval mode."""
    from opencood.models.sub_modules.pillar_vfe import PillarVFE
    from opencood.models.sub_modules.point_pillar_scatter import PointPillarScatter
    nx, ny, B, M = 88, 50, 2, 1500
    vs, rng_ = [0.4, 0.4, 4.0], [-17.6, -10.0, -3.0, 17.6, 10.0, 1.0]
    vfe = PillarVFE({"use_norm": True, "with_distance": False, "use_absolute_xyz": True, "num_filters": [64]},
                    num_point_features=4, voxel_size=vs, point_cloud_range=rng_).eval()
    synth.fill_params_(vfe, WEIGHT_SEED + 5)
    r = np.random.RandomState(9)
    with torch.no_grad():  # non-trivial running statistics
        vfe.pfn_layers[0].norm.running_mean.copy_(torch.from_numpy(r.normal(0, 0.5, 64).astype(np.float32)))
        vfe.pfn_layers[0].norm.running_var.copy_(torch.from_numpy(r.uniform(0.5, 2.0, 64).astype(np.float32)))
    sc = PointPillarScatter({"num_features": 64, "grid_size": np.array([nx, ny, 1])})
    pil = synth.make_pillars(M, B, nx, ny, DATA_SEED + 5, voxel_size=vs, pc_range=rng_)
    bd = {"voxel_features": torch.from_numpy(pil["voxel_features"]), "voxel_num_points": torch.from_numpy(pil["voxel_num_points"]),
          "voxel_coords": torch.from_numpy(pil["voxel_coords"])}
    with torch.no_grad():
        bd = vfe(bd)
        pf = bd["pillar_features"].clone()
        bd = sc(bd)
    sp = bd["spatial_features"]
    nz = (sp.abs().sum(1) > 0)
    rec = dict(nx=nx, ny=ny, B=B, M=M, voxel_size=np.asarray(vs), pc_range=np.asarray(rng_), weight_seed=WEIGHT_SEED + 5,
               data_seed=DATA_SEED + 5, bn_seed=9, pillar_features=pf.numpy(), spatial_shape=np.asarray(sp.shape),
               occupied_index=torch.nonzero(nz.flatten()).flatten().numpy().astype(np.int64),
               spatial_sample=sub(sp.numpy(), 13), spatial_absmean=np.float64(sp.abs().double().mean().item()))
    path = os.path.join(OUT, "pillars.npz")
    np.savez_compressed(path, **rec)
    print(f"pillars: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), occupied cells {int(nz.sum())}")


BACKBONE_CFG = {"layer_nums": [1, 2, 2], "layer_strides": [2, 2, 2], "num_filters": [64, 128, 256],
                "upsample_strides": [1, 2, 4], "num_upsample_filter": [128, 128, 128]}
SHRINK_CFG = {"kernal_size": [3], "stride": [2], "padding": [1], "dim": [128], "input_dim": 384}


def run_backbone_case() -> None:
    """Dense conv stacks around the hot path: the reference's own BaseBEVBackbone and DownsampleConv
    (opencood/models/sub_modules/{base_bev_backbone,downsample_conv}.py) in eval mode + three 1x1 heads."""
    from opencood.models.sub_modules.base_bev_backbone import BaseBEVBackbone
    from opencood.models.sub_modules.downsample_conv import DownsampleConv
    bb = BaseBEVBackbone(dict(BACKBONE_CFG), 64).eval()
    sh = DownsampleConv(dict(SHRINK_CFG)).eval()
    heads = torch.nn.ModuleList([torch.nn.Conv2d(128, 2, 1), torch.nn.Conv2d(128, 14, 1), torch.nn.Conv2d(128, 4, 1)]).eval()
    for k, m in enumerate((bb, sh, heads)):
        synth.fill_params_(m, WEIGHT_SEED + 20 + k)
    synth.fill_bn_stats_(bb, WEIGHT_SEED + 30)
    x = torch.from_numpy(np.maximum(synth.noise_stream(DATA_SEED + 20, 0, (2, 64, 48, 80)), 0.0).astype(np.float32))
    with torch.no_grad():
        d = bb({"spatial_features": x})
        y = d["spatial_features_2d"]
        assert sorted(d.keys()) == ["spatial_features", "spatial_features_2d"]  # the per-stride entries stay in a local dict
        z = sh(y)
        hs = [h(z) for h in heads]
    rec = dict(weight_seed=WEIGHT_SEED + 20, bn_seed=WEIGHT_SEED + 30, data_seed=DATA_SEED + 20, in_shape=np.asarray(x.shape),
               backbone_shape=np.asarray(y.shape), backbone=sub(y.numpy(), 7), backbone_absmean=np.float64(y.abs().double().mean().item()),
               ms_feat=np.concatenate([sub(f.detach().numpy(), 11) for f in bb.get_multiscale_feature(x)]),
               shrink_shape=np.asarray(z.shape), shrink=sub(z.numpy(), 3),
               cls=hs[0].numpy(), reg=hs[1].numpy(), dir=hs[2].numpy(),
               backbone_keys=np.asarray(sorted(bb.state_dict().keys())), shrink_keys=np.asarray(sorted(sh.state_dict().keys())))
    path = os.path.join(OUT, "backbone.npz")
    np.savez_compressed(path, **rec)
    print(f"backbone: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), out {tuple(y.shape)} -> {tuple(z.shape)}")


def shell_args(T: int = 3) -> dict:
    """A reduced stage-1 yaml `model.args` block (structure of opv2v/GenComm_yamls/gencomm/stage1/m1_att.yaml):
    128 x 64 pillars of 0.4 m, backbone [1,1,2] layers, feature 128 x 16 x 32 at the GenComm input."""
    rng_ = [-25.6, -12.8, -3, 25.6, 12.8, 1]
    return {
        "ego_modality": "m1", "lidar_range": rng_,
        "m1": {"core_method": "point_pillar", "sensor_type": "lidar",
               "encoder_args": {"voxel_size": [0.4, 0.4, 4], "lidar_range": rng_,
                                "pillar_vfe": {"use_norm": True, "with_distance": False, "use_absolute_xyz": True, "num_filters": [64]},
                                "point_pillar_scatter": {"num_features": 64}},
               "backbone_args": {"layer_nums": [1, 1, 2], "layer_strides": [2, 2, 2], "num_filters": [64, 128, 256],
                                 "upsample_strides": [1, 2, 4], "num_upsample_filter": [128, 128, 128]},
               "shrink_header": {"kernal_size": [3], "stride": [2], "padding": [1], "dim": [128], "input_dim": 384}},
        "enhancer": {"in_ch": 128}, "message_extractor": {"in_ch": 128, "out_ch": 2},
        "fusion_method": "att", "att": {"feat_dim": 128}, "in_head": 128, "anchor_number": 2,
        "dir_args": {"dir_offset": 0.7853, "num_bins": 2, "anchor_yaw": [0, 90]}, "gmatch": True,
        "gencomm": synth.default_gencomm_cfg(128, T),
    }


def run_shell_case() -> None:
    """The reference's own stage-1 model shell (heter_model_baseline_w_gencomm_stage1.py) end to end on CPU:
    PointPillar encoder -> BaseBEVBackbone -> DownsampleConv -> MessageExtractorv2 -> GenComm -> Enhancer ->
    AttFusion -> heads, on 2 scenes (3 + 1 agents). Everything is reference code except torchvision's DeformConv2d
    (absent here), which runs the oracle's DCNv1 restatement -- so `message`, and what follows it, is pinned only up
    to that third-party op (PARITY UNPINNED for the deformable conv itself)."""
    import copy
    import json
    from opencood.models.heter_model_baseline_w_gencomm_stage1 import HeterModelBaselineWGenComm
    args = shell_args()
    model = HeterModelBaselineWGenComm(copy.deepcopy(args)).eval()
    synth.fill_params_(model, WEIGHT_SEED + 40)
    synth.fill_bn_stats_(model, WEIGHT_SEED + 41)
    rl = [3, 1]
    nx, ny = 128, 64
    pil = synth.make_pillars(5000, sum(rl), nx, ny, DATA_SEED + 40, voxel_size=[0.4, 0.4, 4.0], pc_range=args["lidar_range"])
    ptm = synth.make_pairwise_t_matrix(rl, 5, DATA_SEED + 41, max_shift=8.0)
    data = {"agent_modality_list": ["m1"] * sum(rl), "record_len": torch.tensor(rl), "pairwise_t_matrix": torch.from_numpy(ptm),
            "inputs_m1": {"voxel_features": torch.from_numpy(pil["voxel_features"]), "voxel_coords": torch.from_numpy(pil["voxel_coords"]),
                          "voxel_num_points": torch.from_numpy(pil["voxel_num_points"])}}
    with torch.no_grad(), PatchedNoise(NOISE_SEED + 40):
        out = model(data)
    keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    rec = dict(weight_seed=WEIGHT_SEED + 40, bn_seed=WEIGHT_SEED + 41, data_seed=DATA_SEED + 40, pose_seed=DATA_SEED + 41,
               noise_seed=NOISE_SEED + 40, record_len=np.asarray(rl), M=5000, nx=nx, ny=ny, max_shift=8.0,
               out_keys=np.asarray(sorted(out.keys())),
               message=out["message"].numpy(), gt_feature=sub(out["gt_feature"].numpy(), 5), pred_feature=sub(out["pred_feature"].numpy(), 5),
               cls_preds=out["cls_preds"].numpy(), reg_preds=out["reg_preds"].numpy(), dir_preds=out["dir_preds"].numpy(),
               shapes=np.asarray([list(out[k].shape) for k in ("gt_feature", "pred_feature", "cls_preds", "reg_preds", "dir_preds", "message")]))
    np.savez_compressed(os.path.join(OUT, "shell.npz"), **rec)
    with open(os.path.join(OUT, "shell_state_dict_keys.json"), "w") as f:
        json.dump({"args": args, "state_dict": keys}, f, indent=0)
    print(f"shell: wrote shell.npz ({os.path.getsize(os.path.join(OUT, 'shell.npz')) / 1024:.0f} KiB), {len(keys)} state_dict keys, "
          f"|cls| mean {out['cls_preds'].abs().mean().item():.4f}")


def late_args() -> dict:
    """A reduced `model.args` block of opv2v/Single/m1_pointpillar_pretrain.yaml:99-146 (heter_model_late: PointPillars ego-only, no
    fusion -- BASELINE.json configs[0]): 128 x 64 pillars of 0.4 m, light ResNet backbone [3] + multiscale ResNet layers [3, 5, 8]."""
    rng_ = [-25.6, -12.8, -3, 25.6, 12.8, 1]
    return {"ego_modality": "m1", "lidar_range": rng_, "anchor_number": 2, "dir_args": {"dir_offset": 0.7853, "num_bins": 2, "anchor_yaw": [0, 90]},
            "m1": {"core_method": "point_pillar", "sensor_type": "lidar",
                   "encoder_args": {"voxel_size": [0.4, 0.4, 4], "lidar_range": rng_,
                                    "pillar_vfe": {"use_norm": True, "with_distance": False, "use_absolute_xyz": True, "num_filters": [64]},
                                    "point_pillar_scatter": {"num_features": 64}},
                   "backbone_args": {"layer_nums": [3], "layer_strides": [2], "num_filters": [64]},
                   "aligner_args": {"core_method": "identity"},
                   "layers_args": {"layer_nums": [3, 5, 8], "layer_strides": [2, 2, 2], "num_filters": [64, 128, 256],
                                   "upsample_strides": [1, 2, 4], "num_upsample_filter": [128, 128, 128]},
                   "shrink_header": {"kernal_size": [3], "stride": [1], "padding": [1], "dim": [256], "input_dim": 384},
                   "head_args": {"in_head": 256}}}


def run_late_case() -> None:
    """The reference's own single-agent model (heter_model_late.py: PointPillar encoder -> ResNetBEVBackbone -> multiscale ResNet
    layers -> deblocks -> DownsampleConv -> heads) on CPU, one agent, synthetic pillars: BASELINE.json configs[0] as a plumbing check."""
    import copy
    import json
    from opencood.models.heter_model_late import HeterModelLate
    args = late_args()
    model = HeterModelLate(copy.deepcopy(args)).eval()
    synth.fill_params_(model, WEIGHT_SEED + 80)
    synth.fill_bn_stats_(model, WEIGHT_SEED + 81)
    nx, ny = 128, 64
    pil = synth.make_pillars(2500, 1, nx, ny, DATA_SEED + 80, voxel_size=[0.4, 0.4, 4.0], pc_range=args["lidar_range"])
    data = {"inputs_m1": {"voxel_features": torch.from_numpy(pil["voxel_features"]), "voxel_coords": torch.from_numpy(pil["voxel_coords"]),
                          "voxel_num_points": torch.from_numpy(pil["voxel_num_points"])}}
    with torch.no_grad():
        out = model(data)
    keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "late.npz"), weight_seed=WEIGHT_SEED + 80, bn_seed=WEIGHT_SEED + 81, data_seed=DATA_SEED + 80, M=2500,
                        nx=nx, ny=ny, cls_preds=out["cls_preds"].numpy(), reg_preds=out["reg_preds"].numpy(), dir_preds=out["dir_preds"].numpy())
    with open(os.path.join(OUT, "late_state_dict_keys.json"), "w") as f:
        json.dump({"args": args, "state_dict": keys}, f, indent=0)
    print(f"late: {len(keys)} state_dict keys, cls {tuple(out['cls_preds'].shape)} |cls| mean {out['cls_preds'].abs().mean().item():.4f}")


POSTPROC_PARAMS = {
    "core_method": "VoxelPostprocessor", "gt_range": [-35.2, -20.0, -3, 35.2, 20.0, 1],
    "anchor_args": {"cav_lidar_range": [-35.2, -20.0, -3, 35.2, 20.0, 1], "l": 3.9, "w": 1.6, "h": 1.56, "r": [0, 90],
                    "feature_stride": 2, "num": 2, "vw": 0.4, "vh": 0.4, "vd": 4, "W": 176, "H": 100, "D": 1},
    "target_args": {"pos_threshold": 0.6, "neg_threshold": 0.45, "score_threshold": 0.2},
    "order": "hwl", "max_num": 150, "nms_thresh": 0.15,
    "dir_args": {"dir_offset": 0.7853, "num_bins": 2, "anchor_yaw": [0, 90]},
}


def run_postproc_case() -> None:
    """Detection tail: the reference's own VoxelPostprocessor (generate_anchor_box, post_process and the
    box_utils helpers it calls) on synthetic head outputs, for an identity and a non-identity agent-to-ego transform.
    shapely is absent: `common_utils.convert_format/compute_iou` (the only shapely users on this path) are bound to the
    oracle's convex-clipping IoU; the greedy NMS loop itself is the reference's. Also: the compiled reference
    `bbox_overlaps` (oracle/_ref) on random boxes."""
    import build_ref
    import detect_port as D
    import json
    assert build_ref.build(), "oracle/_ref/box_overlaps could not be built"
    bo = build_ref.load_box_overlaps()
    import opencood.utils as ou
    sys.modules["opencood.utils.box_overlaps"] = bo
    ou.box_overlaps = bo
    from opencood.data_utils.post_processor.voxel_postprocessor import VoxelPostprocessor
    from opencood.utils import box_utils, common_utils
    common_utils.convert_format = lambda boxes: np.asarray(boxes, dtype=np.float64)[:, :4, :2]
    common_utils.compute_iou = lambda box, boxes: D.quad_iou_one_to_many(box, boxes)
    pp = VoxelPostprocessor(json.loads(json.dumps(POSTPROC_PARAMS)), train=False)
    anchors = pp.generate_anchor_box()
    H, W, A = anchors.shape[:3]
    rec = dict(anchors=anchors, H=H, W=W, A=A, params=json.dumps(POSTPROC_PARAMS))
    th = 0.3
    T2 = np.array([[np.cos(th), -np.sin(th), 0, 1.5], [np.sin(th), np.cos(th), 0, -0.7], [0, 0, 1, 0.1], [0, 0, 0, 1]], dtype=np.float32)
    for tag, seed, T in (("a", DATA_SEED + 50, np.eye(4, dtype=np.float32)), ("b", DATA_SEED + 51, T2)):
        cls, reg, dirp = synth.make_detection_maps(H, W, A, seed)
        data = {"ego": {"transformation_matrix": torch.from_numpy(T), "anchor_box": torch.from_numpy(anchors)}}
        out = {"ego": {"cls_preds": torch.from_numpy(cls), "reg_preds": torch.from_numpy(reg), "dir_preds": torch.from_numpy(dirp)}}
        boxes, scores = pp.post_process(data, out)
        rec.update({f"seed_{tag}": seed, f"T_{tag}": T, f"boxes_{tag}": boxes.numpy(), f"scores_{tag}": scores.numpy(),
                    f"n_above_thr_{tag}": int((torch.sigmoid(torch.from_numpy(cls)) > 0.2).sum())})
        print(f"postproc {tag}: {rec[f'n_above_thr_{tag}']} candidates -> {len(scores)} boxes after NMS + range mask")
    r = np.random.RandomState(DATA_SEED + 52)
    def rboxes(n):
        xy, wh = r.uniform(0, 50, (n, 2)), r.uniform(0.5, 20, (n, 2))
        return np.concatenate([xy, xy + wh], 1).astype(np.float32)
    qa, qb = rboxes(257), rboxes(61)
    rec.update(ov_boxes=qa, ov_query=qb, ov=bo.bbox_overlaps(qa, qb))
    np.savez_compressed(os.path.join(OUT, "postproc.npz"), **rec)
    print(f"postproc: wrote postproc.npz ({os.path.getsize(os.path.join(OUT, 'postproc.npz')) / 1024:.0f} KiB)")


def make_eval_frames(seed: int, n_frames: int = 6):
    """Synthetic detection frames for the AP fixture: ground-truth BEV boxes (corners (K, 8, 3)), detections = jittered
    ground truth (some well inside IoU 0.7, some between the thresholds) + false positives, scores; one frame with no
    detection (det_boxes None), one with no ground truth."""
    r = np.random.RandomState(seed)

    def corners(c, size, yaw):
        l, w = size
        loc = np.array([[l / 2, w / 2], [l / 2, -w / 2], [-l / 2, -w / 2], [-l / 2, w / 2]])
        R = np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]])
        xy = loc @ R.T + c
        top = np.concatenate([xy, np.full((4, 1), 1.0)], 1)
        bot = np.concatenate([xy, np.full((4, 1), -1.0)], 1)
        return np.concatenate([bot, top], 0).astype(np.float32)

    frames = []
    for f in range(n_frames):
        K = 0 if f == 4 else r.randint(6, 14)
        gts, dets, scores = [], [], []
        for _ in range(K):
            c, size, yaw = r.uniform(-60, 60, 2), (r.uniform(3.5, 5.5), r.uniform(1.6, 2.2)), r.uniform(-np.pi, np.pi)
            gts.append(corners(c, size, yaw))
            u = r.rand()
            if u < 0.8:  # detected, with jitter that lands on either side of the 0.5 / 0.7 thresholds
                j = r.choice([0.05, 0.3, 0.8])
                dets.append(corners(c + r.normal(0, j, 2), (size[0] * r.uniform(0.9, 1.1), size[1] * r.uniform(0.9, 1.1)), yaw + r.normal(0, 0.05)))
                scores.append(r.uniform(0.3, 0.99))
                if r.rand() < 0.2:  # duplicate detection of the same object (must become a false positive)
                    dets.append(corners(c + r.normal(0, 0.05, 2), size, yaw))
                    scores.append(r.uniform(0.2, 0.9))
        for _ in range(r.randint(1, 5)):  # false positives
            dets.append(corners(r.uniform(-60, 60, 2), (4.5, 1.9), r.uniform(-np.pi, np.pi)))
            scores.append(r.uniform(0.2, 0.7))
        gt = np.stack(gts) if gts else np.zeros((0, 8, 3), np.float32)
        if f == 2:
            frames.append((None, None, gt))
        else:
            frames.append((np.stack(dets), np.asarray(scores, dtype=np.float32), gt))
    return frames


def run_eval_case() -> None:
    """AP harness (SURVEY 8d parity metric): the reference's own caluclate_tp_fp / calculate_ap / voc_ap
    (opencood/utils/eval_utils.py) on synthetic frames. shapely is absent: `common_utils.convert_format/compute_iou` are bound
    to the oracle's float64 convex-clipping IoU (PARITY UNPINNED for that call, see oracle/eval_port.py); the matching, the
    cumulative sums and the precision envelope are the reference's."""
    import detect_port as D
    from opencood.utils import common_utils, eval_utils
    common_utils.convert_format = lambda boxes: np.asarray(boxes, dtype=np.float64)[:, :4, :2]
    common_utils.compute_iou = lambda box, boxes: D.quad_iou_one_to_many(box, np.asarray(boxes).reshape(-1, 4, 2))
    seed = DATA_SEED + 60
    frames = make_eval_frames(seed)
    stat = {0.3: {'tp': [], 'fp': [], 'gt': 0, 'score': []}, 0.5: {'tp': [], 'fp': [], 'gt': 0, 'score': []}, 0.7: {'tp': [], 'fp': [], 'gt': 0, 'score': []}}
    for det, score, gt in frames:
        for thr in (0.3, 0.5, 0.7):
            eval_utils.caluclate_tp_fp(None if det is None else torch.from_numpy(det), None if det is None else torch.from_numpy(score),
                                       torch.from_numpy(gt), stat, thr)
    rec = {"seed": seed, "n_frames": len(frames)}
    for thr in (0.3, 0.5, 0.7):
        k = str(thr)
        rec["tp_" + k], rec["fp_" + k] = np.array(stat[thr]["tp"]), np.array(stat[thr]["fp"])
        rec["score_" + k], rec["gt_" + k] = np.array(stat[thr]["score"], dtype=np.float64), stat[thr]["gt"]
    import copy
    for gs in (True, False):  # inference.py:231-234 order; the second call cumulates the stored lists in place, so copy
        for thr in (0.3, 0.5, 0.7):
            ap, mrec, mpre = eval_utils.calculate_ap(copy.deepcopy(stat), thr, gs)
            rec[f"ap_{thr}_{int(gs)}"], rec[f"mrec_{thr}_{int(gs)}"], rec[f"mpre_{thr}_{int(gs)}"] = ap, np.array(mrec), np.array(mpre)
    np.savez_compressed(os.path.join(OUT, "eval.npz"), **rec)
    print("eval: AP@0.3/0.5/0.7 global-sort", [round(float(rec[f'ap_{t}_1']), 4) for t in (0.3, 0.5, 0.7)],
          "per-frame order", [round(float(rec[f'ap_{t}_0']), 4) for t in (0.3, 0.5, 0.7)])


V2XVIT_ARGS = {"transformer": {"encoder": {
    "num_blocks": 1, "depth": 3, "use_roi_mask": True, "use_RTE": False, "RTE_ratio": 0,
    "cav_att_config": {"dim": 128, "use_hetero": True, "use_RTE": False, "RTE_ratio": 0, "heads": 8, "dim_head": 32, "dropout": 0.3},
    "pwindow_att_config": {"dim": 128, "heads": [16, 8, 4], "dim_head": [16, 32, 64], "dropout": 0.3, "window_size": [4, 8, 16],
                           "relative_pos_embedding": True, "fusion_method": "split_attn128"},
    "feed_forward": {"mlp_dim": 128, "dropout": 0.3},
    "sttf": {"voxel_size": [0.4, 0.4, 4], "downsample_rate": 4}}}}  # opv2v/GenComm_yamls/gencomm/stage1/m1_v2xvit.yaml:137-171


def run_v2xvit_case() -> None:
    """V2XViTFusion (fusion_in_one.py:355-407 + sub_modules/v2xvit_basic.py, hmsa.py, mswin.py, split_attn.py,
    base_transformer.py) with the shipped m1_v2xvit.yaml transformer block, eval mode, the reference's own modules."""
    import json
    from opencood.models.fuse_modules.fusion_in_one import V2XViTFusion
    from opencood.utils.transformation_utils import normalize_pairwise_tfm
    C, H, W, rl, L = 128, 16, 32, [3, 1, 2], 5
    net = V2XViTFusion(json.loads(json.dumps(V2XVIT_ARGS))).eval()
    seed_w, seed_d = WEIGHT_SEED + 70, DATA_SEED + 70
    synth.fill_params_(net, seed_w)
    inp = synth.make_inputs(rl, C, H, W, seed_d, max_shift=4.0)
    x = torch.from_numpy(inp["feat"])
    record_len = torch.from_numpy(inp["record_len"])
    ptm = torch.from_numpy(inp["pairwise_t_matrix"])
    affine = normalize_pairwise_tfm(ptm, H * 0.8, W * 0.8, 1)
    with torch.no_grad():
        out = net(x, record_len, affine)
    keys = {k: list(v.shape) for k, v in net.state_dict().items()}
    with open(os.path.join(OUT, "v2xvit_state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)
    np.savez_compressed(os.path.join(OUT, "v2xvit.npz"), C=C, H=H, W=W, record_len=np.array(rl), max_cav=L, weight_seed=seed_w, data_seed=seed_d,
                        max_shift=4.0, stride=3, fused=sub(out.numpy(), 3), fused_shape=np.array(out.shape), args=json.dumps(V2XVIT_ARGS))
    print(f"v2xvit: {len(keys)} state_dict tensors, {sum(int(np.prod(v)) for v in keys.values())} values, out {tuple(out.shape)} "
          f"mean |out| {float(out.abs().mean()):.4f} finite {bool(torch.isfinite(out).all())}")


def run_where2comm_case() -> None:
    """Where2commFusion (fusion_in_one.py:466-519 + where2comm_attn.py:64-102: nn.MultiheadAttention over the agents per pixel + FFN),
    the reference's own module: forward, and the gradients of sum(out * probe) w.r.t. the input and every parameter."""
    import json
    sys.modules.setdefault("turtle", types.SimpleNamespace(update=None))     # `from turtle import update` (unused; needs tkinter)
    from opencood.models.fuse_modules.fusion_in_one import Where2commFusion
    from opencood.utils.transformation_utils import normalize_pairwise_tfm
    C, H, W, rl = 128, 16, 32, [3, 1, 2]
    net = Where2commFusion(C).eval()
    seed_w, seed_d = WEIGHT_SEED + 80, DATA_SEED + 80
    synth.fill_params_(net, seed_w)
    inp = synth.make_inputs(rl, C, H, W, seed_d, max_shift=4.0)
    x = torch.from_numpy(inp["feat"]).requires_grad_(True)
    record_len = torch.from_numpy(inp["record_len"])
    affine = normalize_pairwise_tfm(torch.from_numpy(inp["pairwise_t_matrix"]), H * 0.8, W * 0.8, 1)
    out = net(x, record_len, affine)
    probe = torch.from_numpy(synth.noise_stream(seed_d, 99, tuple(out.shape)))
    (out * probe).sum().backward()
    keys = {k: list(v.shape) for k, v in net.state_dict().items()}
    with open(os.path.join(OUT, "where2comm_state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)
    grads = {"g_" + k.replace(".", "__"): p.grad.numpy() for k, p in net.named_parameters() if p.numel() <= 4096}
    grads["g_in_proj_weight_sub"] = net.mha_fusion.attn.in_proj_weight.grad.numpy()[::7, ::5]
    np.savez_compressed(os.path.join(OUT, "where2comm.npz"), C=C, H=H, W=W, record_len=np.array(rl), weight_seed=seed_w, data_seed=seed_d,
                        max_shift=4.0, stride=3, fused=sub(out.detach().numpy(), 3), fused_shape=np.array(out.shape),
                        dx=sub(x.grad.numpy(), 3), **grads)
    print(f"where2comm: {len(keys)} state_dict tensors, out {tuple(out.shape)} mean |out| {float(out.abs().mean()):.4f} "
          f"mean |dx| {float(x.grad.abs().mean()):.5f}")


LOSS_ARGS = {  # opv2v/GenComm_yamls/gencomm/stage1/m1_att.yaml:168-189
    "pos_cls_weight": 2.0, "cls": {"type": "SigmoidFocalLoss", "alpha": 0.25, "gamma": 2.0, "weight": 2.0},
    "reg": {"type": "WeightedSmoothL1Loss", "sigma": 3.0, "codewise": True, "weight": 2.0},
    "dir": {"type": "WeightedSoftmaxClassificationLoss", "weight": 0.2,
            "args": {"dir_offset": 0.7853, "num_bins": 2, "anchor_yaw": [0, 90]}},
    "depth": {"weight": 1.0}, "generate_weight": 1, "gmatch_weight": 1}


def run_loss_case() -> None:
    """The reference's own PointPillarGencommLoss (opencood/loss/point_pillar_gencomm_loss.py) on seeded synthetic head
    maps and labels: the total, its components and the gradients with respect to every input that carries one."""
    import build_ref
    import json
    assert build_ref.build(), "oracle/_ref/box_overlaps could not be built"   # the loss module imports the post-processor (unused by it)
    import opencood.utils as ou
    sys.modules["opencood.utils.box_overlaps"] = ou.box_overlaps = build_ref.load_box_overlaps()
    from opencood.loss.point_pillar_gencomm_loss import PointPillarGencommLoss
    B, H, W, A, C = 3, 12, 20, 2, 16
    inp = synth.make_loss_inputs(DATA_SEED + 90, B, H, W, A, C)
    t = {k: torch.from_numpy(v) for k, v in inp.items()}
    for k in ("cls_preds", "reg_preds", "dir_preds", "pred_feature"):
        t[k].requires_grad_(True)
    crit = PointPillarGencommLoss(LOSS_ARGS)
    out = {k: t[k] for k in ("cls_preds", "reg_preds", "dir_preds", "gt_feature", "pred_feature")}
    total = crit(out, {k: t[k] for k in ("pos_equal_one", "neg_equal_one", "targets")})
    total.backward()
    rec = {"args": json.dumps(LOSS_ARGS), "data_seed": DATA_SEED + 90, "dims": np.array([B, H, W, A, C]), "total": total.detach().numpy(),
           **{k: np.float64(v) for k, v in crit.loss_dict.items()},
           **{"grad_" + k: t[k].grad.numpy() for k in ("cls_preds", "reg_preds", "dir_preds", "pred_feature")}}
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **rec)
    print("loss:", {k: float(v) for k, v in crit.loss_dict.items()})


APCHAIN_PP = {  # the post-processor block of the stage-1 yamls at the reduced shell geometry (128 x 64 pillars, feature stride 4 -> 32 x 16 anchors x 2)
    "core_method": "VoxelPostprocessor", "gt_range": [-25.6, -12.8, -3, 25.6, 12.8, 1],
    "anchor_args": {"cav_lidar_range": [-25.6, -12.8, -3, 25.6, 12.8, 1], "l": 3.9, "w": 1.6, "h": 1.56, "r": [0, 90],
                    "feature_stride": 4, "num": 2, "vw": 0.4, "vh": 0.4, "vd": 4, "W": 128, "H": 64, "D": 1},
    "target_args": {"pos_threshold": 0.6, "neg_threshold": 0.45, "score_threshold": 0.2},
    "order": "hwl", "max_num": 150, "nms_thresh": 0.15,
    "dir_args": {"dir_offset": 0.7853, "num_bins": 2, "anchor_yaw": [0, 90]},
}
APCHAIN_FRAMES = [[2], [3], [1], [2], [4], [2], [3], [2], [5], [2]]   # agents per frame (one scene per frame: inference.py runs batch size 1)


def run_apchain_case() -> None:
    """north_star's end-to-end criterion on synthetic frames: the reference's own stage-1 shell (CPU) -> the
    reference's VoxelPostprocessor.post_process -> the reference's eval_utils AP@0.3/0.5/0.7, chained on 10 frames, plus a variant with
    the reference's pose noise (pose_utils.generate_noise: std 0.2 m / 0.2 deg) on the agents' poses.  Ground truth = a jittered subset of
    the reference's own detections (so that AP lands strictly inside (0, 1) and matches straddle the IoU thresholds).  Third-party
    arithmetic absent here, as in the shell / postproc / eval cases: DeformConv2d and the shapely IoU run the oracle's restatements."""
    import build_ref
    import copy
    import json
    import detect_port as D
    assert build_ref.build(), "oracle/_ref/box_overlaps could not be built"
    import opencood.utils as ou
    sys.modules["opencood.utils.box_overlaps"] = ou.box_overlaps = build_ref.load_box_overlaps()
    from opencood.data_utils.post_processor.voxel_postprocessor import VoxelPostprocessor
    from opencood.models.heter_model_baseline_w_gencomm_stage1 import HeterModelBaselineWGenComm
    from opencood.utils import common_utils, eval_utils, pose_utils
    from opencood.utils.transformation_utils import x_to_world
    common_utils.convert_format = lambda boxes: np.asarray(boxes, dtype=np.float64)[:, :4, :2]
    common_utils.compute_iou = lambda box, boxes: D.quad_iou_one_to_many(box, np.asarray(boxes).reshape(-1, 4, 2))
    args = shell_args()
    model = HeterModelBaselineWGenComm(copy.deepcopy(args)).eval()
    synth.trained_looking_heads_(model, WEIGHT_SEED + 100)
    pp = VoxelPostprocessor(json.loads(json.dumps(APCHAIN_PP)), train=False)
    anchors = pp.generate_anchor_box()
    nx, ny, L = 128, 64, 5
    rec = dict(weight_seed=WEIGHT_SEED + 100, params=json.dumps(APCHAIN_PP), frames=json.dumps(APCHAIN_FRAMES), nx=nx, ny=ny, M=1500,
               data_seed=DATA_SEED + 100, noise_seed=NOISE_SEED + 100, pos_std=0.2, rot_std=0.2)
    r = np.random.RandomState(DATA_SEED + 101)

    def corners_of(center, size, yaw, z0=-1.0, z1=0.6):
        l, w = size
        loc = np.array([[l / 2, w / 2], [l / 2, -w / 2], [-l / 2, -w / 2], [-l / 2, w / 2]])
        R = np.array([[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]])
        xy = loc @ R.T + center
        return np.concatenate([np.concatenate([xy, np.full((4, 1), z0)], 1), np.concatenate([xy, np.full((4, 1), z1)], 1)], 0).astype(np.float32)

    gts = []
    for variant in ("clean", "posenoise"):
        stat = {t: {"tp": [], "fp": [], "gt": 0, "score": []} for t in (0.3, 0.5, 0.7)}
        nboxes = []
        for f, rl in enumerate(APCHAIN_FRAMES):
            n = sum(rl)
            pil = synth.make_pillars(1500 * n, n, nx, ny, DATA_SEED + 110 + f, voxel_size=[0.4, 0.4, 4.0], pc_range=args["lidar_range"])
            # 6-dof lidar poses [x, y, z, roll, yaw, pitch] (degrees) of the frame's agents; the pairwise matrices come from the reference's
            # own x_to_world exactly as get_pairwise_transformation builds them (transformation_utils.py:21-66)
            rp = np.random.RandomState(DATA_SEED + 130 + f)
            poses = np.zeros((n, 6))
            poses[:, 0], poses[:, 1], poses[:, 4] = rp.uniform(-6, 6, n), rp.uniform(-3, 3, n), rp.uniform(-25, 25, n)
            if variant == "posenoise":
                np.random.seed(DATA_SEED + 150 + f)
                for i in range(n):
                    poses[i] = poses[i] + pose_utils.generate_noise(0.2, 0.2)      # pose_utils.py:41-74 (numpy global RNG, seeded above)
            tl = [x_to_world(list(pz)) for pz in poses]
            ptm = np.tile(np.eye(4), (1, L, L, 1, 1))
            for i in range(n):
                for j in range(n):
                    if i != j:
                        ptm[0, i, j] = np.linalg.solve(tl[j], tl[i])
            rec[f"ptm_{variant}_{f}"] = ptm
            data = {"agent_modality_list": ["m1"] * n, "record_len": torch.tensor(rl), "pairwise_t_matrix": torch.from_numpy(ptm),
                    "inputs_m1": {k: torch.from_numpy(pil[k]) for k in ("voxel_features", "voxel_coords", "voxel_num_points")}}
            with torch.no_grad(), PatchedNoise(NOISE_SEED + 100 + f):
                out = model(data)
            boxes, scores = pp.post_process({"ego": {"transformation_matrix": torch.eye(4), "anchor_box": torch.from_numpy(anchors)}}, {"ego": out})
            nb = 0 if boxes is None else int(boxes.shape[0])
            nboxes.append(nb)
            if variant == "clean":
                # ground truth of the frame (shared by both variants): two thirds of the clean detections, jittered by 5 / 25 / 60 cm
                g = []
                if nb:
                    b = boxes.numpy()
                    for k in range(nb):
                        if k % 3 == 2:
                            continue
                        xy = b[k, :4, :2]
                        c = xy.mean(0)
                        e0, e1 = xy[1] - xy[0], xy[2] - xy[1]
                        size = (float(np.hypot(*e1)), float(np.hypot(*e0)))
                        yaw = float(np.arctan2(e1[1], e1[0]))
                        g.append(corners_of(c + r.normal(0, [0.05, 0.25, 0.6][k % 3 if k % 3 < 2 else 0] if k % 5 else 0.6, 2), size, yaw + r.normal(0, 0.03)))
                for _ in range(2):   # objects nobody detects
                    g.append(corners_of(r.uniform([-22, -10], [22, 10]), (4.2, 1.8), r.uniform(-np.pi, np.pi)))
                gts.append(np.stack(g))
                rec[f"gt_{f}"] = gts[f]
            rec[f"boxes_{variant}_{f}"] = np.zeros((0, 8, 3), np.float32) if boxes is None else boxes.numpy()
            rec[f"scores_{variant}_{f}"] = np.zeros((0,), np.float32) if boxes is None else scores.numpy()
            for t in (0.3, 0.5, 0.7):
                eval_utils.caluclate_tp_fp(boxes, scores, torch.from_numpy(gts[f]), stat, t)
        for t in (0.3, 0.5, 0.7):
            ap, _, _ = eval_utils.calculate_ap(copy.deepcopy(stat), t, True)
            rec[f"ap_{variant}_{t}"] = ap
        rec[f"nboxes_{variant}"] = np.array(nboxes)
        print(f"apchain [{variant}]: boxes per frame {nboxes}, AP@0.3/0.5/0.7 = " + " / ".join(f"{rec[f'ap_{variant}_{t}']:.4f}" for t in (0.3, 0.5, 0.7)))
    np.savez_compressed(os.path.join(OUT, "apchain.npz"), **rec)
    print(f"apchain: wrote apchain.npz ({os.path.getsize(os.path.join(OUT, 'apchain.npz')) / 1024:.0f} KiB)")


ROBUST = dict(C=64, H=32, W=64, T=5, n=3, px_m=0.8, stride=29, pos_rot_std=[0.2, 0.4, 0.8], async_overhead=[100, 300, 500])


def robust_frame_inputs(frame_of_agent):
    """feat / cond of the robustness case: agent i's rows come from the synthetic frame `frame_of_agent[i]` (a stale collaborator = rows
    of an earlier frame).  Shared with tests/test_gpu_robustness.py so that the fixture stores seeds, not tensors."""
    c = ROBUST
    per = {f: synth.make_inputs([c["n"]], c["C"], c["H"], c["W"], DATA_SEED + 200 + f) for f in set(frame_of_agent)}
    feat = np.stack([per[f]["feat"][i] for i, f in enumerate(frame_of_agent)])
    cond = np.stack([per[f]["cond"][i] for i, f in enumerate(frame_of_agent)])
    return feat, cond


def run_robust_case() -> None:
    """BASELINE configs[4] (pose noise + communication delay) on synthetic inputs: both perturbations reach the hot path only through
    `pairwise_t_matrix` and through WHICH frame a collaborator's feature / message rows come from.  This case builds them with the
    reference's own code -- pose_utils.generate_noise under np.random.seed(303) for the std sweep of inference_w_noise.py:66-88,
    the 'sim' / 'random' frame delay of opv2v_basedataset.py:706-744 for the overheads of inference_w_delay.py:66, poses -> pairwise
    matrices through x_to_world / get_pairwise_transformation (transformation_utils.py:21-66, :264-307) -- and runs the reference's
    GenComm -> Enhancer -> AttFusion on them."""
    from collections import OrderedDict
    from opencood.models.gencomm_modules.cond_diff import GenComm
    from opencood.models.gencomm_modules.enhancer import Enhancer
    from opencood.models.fuse_modules.fusion_in_one import AttFusion
    from opencood.utils import pose_utils
    from opencood.utils.transformation_utils import get_pairwise_transformation, normalize_pairwise_tfm
    c = ROBUST
    C, H, W, T, n, st = c["C"], c["H"], c["W"], c["T"], c["n"], c["stride"]
    cfg = synth.default_gencomm_cfg(C, T)
    gen, enh, fus = GenComm(cfg).eval(), Enhancer(C, [8, 8], 4).eval(), AttFusion(C).eval()
    synth.fill_params_(gen, WEIGHT_SEED + 200)
    synth.fill_params_(enh, WEIGHT_SEED + 201)
    bev_h_m, bev_w_m = H * c["px_m"], W * c["px_m"]
    # world: agent i at pose0[i] in frame 0, moving with velocity vel[i] per frame (10 Hz); the current frame is frame 6
    rp = np.random.RandomState(DATA_SEED + 210)
    pose0, vel = np.zeros((n, 6)), np.zeros((n, 6))
    pose0[:, 0], pose0[:, 1], pose0[:, 4] = rp.uniform(-8, 8, n), rp.uniform(-4, 4, n), rp.uniform(-30, 30, n)
    vel[:, 0], vel[:, 1], vel[:, 4] = rp.uniform(-0.8, 0.8, n), rp.uniform(-0.3, 0.3, n), rp.uniform(-1.0, 1.0, n)
    CUR = 6
    pose_at = lambda i, f: pose0[i] + vel[i] * f

    def ptm_of(poses):
        base = OrderedDict((i, {"params": {"lidar_pose": list(poses[i])}}) for i in range(n))
        return get_pairwise_transformation(base, 5, False)[None]

    variants = OrderedDict()
    variants["clean"] = ([CUR] * n, np.stack([pose_at(i, CUR) for i in range(n)]))
    for s_ in c["pos_rot_std"]:
        np.random.seed(303)                                                   # inference_w_noise.py:88
        poses = np.stack([pose_at(i, CUR) + pose_utils.generate_noise(s_, s_) for i in range(n)])   # pose_utils.py:14-33: every cav, ego included
        variants[f"pose{s_}"] = ([CUR] * n, poses)
    for ov in c["async_overhead"]:
        torch.manual_seed(303)
        frames = [CUR]                                                        # opv2v_basedataset.py:720-721: no delay for the ego
        for i in range(1, n):
            delay = (torch.randint(0, ov, (1,)).item() + 100) // 100          # :733-743, async_mode 'sim', async_method 'random'
            frames.append(CUR - delay)
        variants[f"delay{ov}"] = (frames, np.stack([pose_at(i, f) for i, f in enumerate(frames)]))   # stale rows AND the stale pose (:661)
    rec = dict(C=C, H=H, W=W, T=T, n=n, px_m=c["px_m"], stride=st, weight_seed=WEIGHT_SEED + 200, noise_seed=NOISE_SEED + 200,
               variants=np.asarray(list(variants)))
    for name, (frames, poses) in variants.items():
        feat, cond = robust_frame_inputs(frames)
        ptm = ptm_of(poses)
        with torch.no_grad(), PatchedNoise(NOISE_SEED + 200):
            pred = gen(torch.from_numpy(feat), torch.from_numpy(cond), torch.tensor([n]))["pred_feature"]
            affine = normalize_pairwise_tfm(torch.from_numpy(ptm).clone(), bev_h_m, bev_w_m, 1)
            enhd = enh(pred, affine, torch.tensor([n]))
            fused = fus(enhd, torch.tensor([n]), affine)
        rec[f"{name}/frames"], rec[f"{name}/poses"], rec[f"{name}/ptm"] = np.asarray(frames), poses, ptm
        rec[f"{name}/pred_feature"], rec[f"{name}/enhanced"], rec[f"{name}/fused"] = sub(pred.numpy(), st), sub(enhd.numpy(), st), sub(fused.numpy(), 7)
        d = float((fused - (0 if name == "clean" else clean_fused)).abs().mean()) if name != "clean" else 0.0
        if name == "clean":
            clean_fused = fused
        print(f"robust [{name}]: frames {frames}, |fused| {float(fused.abs().mean()):.4f}, mean |fused - clean| {d:.4f}")
    np.savez_compressed(os.path.join(OUT, "robust.npz"), **rec)
    print(f"robust: wrote robust.npz ({os.path.getsize(os.path.join(OUT, 'robust.npz')) / 1024:.0f} KiB)")


def run_wide_case() -> None:
    """General DiffusionUNet widths: the reference's own GenComm with ch = 16, ch_mult [1, 2], num_res_blocks 1 --
    every block of the up path then has a 1x1 nin_shortcut with unequal sides, the levels differ in width, GroupNorm groups hold 4 / 8 /
    12 channels.  Single UNet calls for every t and the whole eval forward with injected noise."""
    from opencood.models.gencomm_modules.cond_diff import GenComm
    C, H, W, T, rl = 24, 12, 20, 3, [2, 1]
    cfg = synth.default_gencomm_cfg(C, T)
    cfg["model"].update({"ch": 16, "ch_mult": [1, 2], "num_res_blocks": 1})
    gen = GenComm(cfg).eval()
    synth.fill_params_(gen, WEIGHT_SEED + 70)
    n = sum(rl)
    inp = synth.make_inputs(rl, C, H, W, DATA_SEED + 70)
    feat, cond = torch.from_numpy(inp["feat"]), torch.from_numpy(inp["cond"])
    import json
    rec = dict(cfg=json.dumps(cfg), C=C, H=H, W=W, T=T, record_len=np.asarray(rl), weight_seed=WEIGHT_SEED + 70, data_seed=DATA_SEED + 70,
               noise_seed=NOISE_SEED + 70, keys=np.asarray(sorted(gen.state_dict().keys())),
               shapes=json.dumps({k: list(v.shape) for k, v in gen.state_dict().items()}))
    with torch.no_grad():
        for t in range(T):
            tt = torch.full((n,), t, dtype=torch.long)
            rec[f"unet_out_t{t}"] = gen.denoiser(torch.cat([cond, feat], dim=1), tt.float()).numpy()
        with PatchedNoise(NOISE_SEED + 70):
            rec["pred_feature"] = gen(feat, cond, torch.tensor(rl))["pred_feature"].numpy()
    np.savez_compressed(os.path.join(OUT, "unet_wide.npz"), **rec)
    print(f"wide: wrote unet_wide.npz ({os.path.getsize(os.path.join(OUT, 'unet_wide.npz')) / 1024:.0f} KiB), {len(rec['keys'])} state_dict keys")


def stage2_args(T: int = 3) -> dict:
    """A reduced stage-2 `model.args` block (structure of opv2v/GenComm_yamls/gencomm/stage2/m1m2_att.yaml): two lidar modalities with
    their own PointPillars encoder / backbone / shrink header / message extractor (m2: the new agent type, a lighter backbone), the
    reference's `diffcomm:` key spelling (stage2.py:36), `trick` (predicted features masked by the occupied cells of the originals)."""
    a = shell_args(T)
    a["diffcomm"] = a.pop("gencomm")
    m2 = json_roundtrip(a["m1"])
    m2["backbone_args"]["layer_nums"] = [1, 2, 1]
    a["m2"] = m2
    a["trick"] = True
    return a


def json_roundtrip(o):
    import json
    return json.loads(json.dumps(o))


def run_shell2_case() -> None:
    """The reference's own stage-2 shell (heter_model_baseline_w_gencomm_stage2.py:31-328) end to end on CPU: ego of modality m1 with
    collaborators of m1 and m2 in two scenes (agents m1 m2 m1 | m1 m2), eval mode.  As in the stage-1 case everything is reference code
    except torchvision's DeformConv2d (the oracle's DCNv1 restatement: PARITY UNPINNED for that op)."""
    import copy
    import json
    from opencood.models.heter_model_baseline_w_gencomm_stage2 import HeterModelBaselineWDiffCommStage2
    args = stage2_args()
    model = HeterModelBaselineWDiffCommStage2(copy.deepcopy(args)).eval()
    synth.fill_params_(model, WEIGHT_SEED + 120)
    synth.fill_bn_stats_(model, WEIGHT_SEED + 121)
    rl, mods = [3, 2], ["m1", "m2", "m1", "m1", "m2"]
    nx, ny = 128, 64
    pil = {m: synth.make_pillars(4000, mods.count(m), nx, ny, DATA_SEED + 120 + i, voxel_size=[0.4, 0.4, 4.0], pc_range=args["lidar_range"])
           for i, m in enumerate(("m1", "m2"))}
    ptm = synth.make_pairwise_t_matrix(rl, 5, DATA_SEED + 123, max_shift=6.0)
    data = {"agent_modality_list": mods, "record_len": torch.tensor(rl), "pairwise_t_matrix": torch.from_numpy(ptm)}
    for m in ("m1", "m2"):
        data[f"inputs_{m}"] = {k: torch.from_numpy(pil[m][k]) for k in ("voxel_features", "voxel_coords", "voxel_num_points")}
    import contextlib, io
    with torch.no_grad(), PatchedNoise(NOISE_SEED + 120), contextlib.redirect_stdout(io.StringIO()):   # the reference prints every forward
        out = model(data)
    rec = dict(args=json.dumps(args), weight_seed=WEIGHT_SEED + 120, bn_seed=WEIGHT_SEED + 121, data_seed=DATA_SEED + 120, pose_seed=DATA_SEED + 123,
               noise_seed=NOISE_SEED + 120, record_len=np.asarray(rl), mods=np.asarray(mods), M=4000, nx=nx, ny=ny, max_shift=6.0,
               out_keys=np.asarray(sorted(out.keys())), keys=np.asarray(sorted(model.state_dict().keys())),
               frozen=np.asarray(sorted(n for n, p in model.named_parameters() if not p.requires_grad)),
               message=out["message"].numpy(), gt_feature=sub(out["gt_feature"].numpy(), 5), pred_feature=sub(out["pred_feature"].numpy(), 5),
               cls_preds=out["cls_preds"].numpy(), reg_preds=out["reg_preds"].numpy(), dir_preds=out["dir_preds"].numpy())
    np.savez_compressed(os.path.join(OUT, "shell2.npz"), **rec)
    print(f"shell2: wrote shell2.npz ({os.path.getsize(os.path.join(OUT, 'shell2.npz')) / 1024:.0f} KiB), {len(rec['keys'])} state_dict keys, "
          f"{len(rec['frozen'])} frozen parameters, |cls| mean {out['cls_preds'].abs().mean().item():.4f}")


def dump_state_dict_keys() -> None:
    """Key names + shapes of the reference modules: the checkpoint contract."""
    from opencood.models.gencomm_modules.cond_diff import GenComm
    from opencood.models.gencomm_modules.enhancer import Enhancer
    import json
    gen = GenComm(synth.default_gencomm_cfg(128, 3))
    enh = Enhancer(128, [8, 8], 4)
    d = {"gencomm": {k: list(v.shape) for k, v in gen.state_dict().items()},
         "enhancer": {k: list(v.shape) for k, v in enh.state_dict().items()},
         "gencomm_param_count": sum(p.numel() for p in gen.parameters()),
         "enhancer_param_count": sum(p.numel() for p in enh.parameters())}
    with open(os.path.join(OUT, "state_dict_keys_C128_T3.json"), "w") as f:
        json.dump(d, f, indent=0)
    print("state_dict keys:", len(d["gencomm"]), len(d["enhancer"]), d["gencomm_param_count"], d["enhancer_param_count"])


def main() -> None:
    if not os.path.isdir(REF):
        raise SystemExit("reference checkout not mounted at /root/reference; fixtures can only be regenerated in the build container")
    _install_stubs()
    sys.path.insert(0, REF)
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    only = set(sys.argv[1:])  # e.g. `python oracle/make_golden.py backbone` regenerates one fixture
    for case in CASES:
        if not only or case["name"] in only:
            run_case(case)
    extra = {"attn": run_attn_case, "pillars": run_pillar_case, "backbone": run_backbone_case, "shell": run_shell_case, "postproc": run_postproc_case, "eval": run_eval_case, "v2xvit": run_v2xvit_case, "late": run_late_case, "where2comm": run_where2comm_case, "loss": run_loss_case, "apchain": run_apchain_case, "wide": run_wide_case, "shell2": run_shell2_case, "robust": run_robust_case, "keys": dump_state_dict_keys}
    for name, fn in extra.items():
        if not only or name in only:
            fn()


if __name__ == "__main__":
    main()
