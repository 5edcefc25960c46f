"""Deterministic synthetic weights / inputs for the GenComm hot path.

Everything here is generated with ``numpy.random.RandomState`` (MT19937, bit-reproducible across
machines and numpy versions), never with torch's RNG, so that

* the golden-vector generator (``oracle/make_golden.py``, runs where the reference is mounted),
* the parity tests (``tests/``, run on the GPU box where the reference is absent) and
* ``bench.py``

all see byte-identical weights, inputs and diffusion noise from nothing but a seed and a shape.

Workload definition follows SURVEY.md section 8(d): ``feat = relu(N(0,1))`` (post-ReLU backbone
statistics, reference ``opencood/models/sub_modules/downsample_conv.py:23``), ``cond = N(0,1)``,
ego pose identity, collaborators random SE(2) with yaw ~ U(-pi, pi), translation ~ U(-40, 40) m.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np

__all__ = [
    "default_gencomm_cfg",
    "fill_params_",
    "make_inputs",
    "make_pairwise_t_matrix",
    "noise_stream",
    "make_eval_noise",
    "make_train_noise",
]


def default_gencomm_cfg(C: int, T: int) -> dict:
    """The ``model.args.gencomm`` block of every shipped yaml (e.g. reference
    ``opencood/hypes_yaml/opv2v/GenComm_yamls/gencomm/stage1/m1_att.yaml:150-165``) with the
    feature width and step count substituted."""
    return {
        "model": {
            "embed_dim": C + 2,
            "in_channels": C,
            "out_ch": C,
            "ch": 8,
            "ch_mult": [1, 1],
            "num_res_blocks": 2,
            "attn_resolutions": [16],
            "dropout": 0.0,
            "resamp_with_conv": True,
        },
        "diffusion": {
            "beta_schedule": "linear",
            "beta_start": 0.0005,
            "beta_end": 0.02,
            "num_diffusion_timesteps": T,
        },
    }


def _param_values(name: str, shape: Tuple[int, ...], rng: np.random.RandomState) -> np.ndarray:
    """One tensor of synthetic weights. Norm scales are drawn around 1 and all biases are
    non-zero so that a kernel which drops an affine term cannot pass parity."""
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = name.rsplit(".", 1)[-1]
    if len(shape) <= 1:
        if leaf == "weight":  # GroupNorm / LayerNorm scale
            v = 1.0 + 0.2 * rng.standard_normal(n)
        else:  # any bias
            v = 0.1 * rng.standard_normal(n)
    else:
        fan_in = int(np.prod(shape[1:]))
        bound = 1.0 / math.sqrt(max(fan_in, 1))
        v = rng.uniform(-bound, bound, size=n)
    return v.astype(np.float32).reshape(shape)


def fill_params_(module, seed: int) -> None:
    """Overwrite every parameter of ``module`` (a ``torch.nn.Module``) in sorted-name order
    from one RandomState(seed) stream. Buffers (the diffusion schedule) are left alone."""
    import torch

    rng = np.random.RandomState(seed)
    with torch.no_grad():
        for name, p in sorted(module.named_parameters(), key=lambda kv: kv[0]):
            v = _param_values(name, tuple(p.shape), rng)
            p.copy_(torch.from_numpy(v).to(p.device, p.dtype))


def make_pairwise_t_matrix(record_len: Sequence[int], max_cav: int, seed: int,
                           max_shift: float = 40.0) -> np.ndarray:
    """[B, L, L, 4, 4] float64; entry [b, i, j] maps agent i's frame into agent j's frame
    (reference ``opencood/utils/transformation_utils.py:20-64``: T_ji = T_jw . T_wi).
    Agent 0 (ego) sits at the world origin."""
    rng = np.random.RandomState(seed)
    B = len(record_len)
    out = np.tile(np.eye(4), (B, max_cav, max_cav, 1, 1))
    for b, n in enumerate(record_len):
        world = []
        for a in range(n):
            T = np.eye(4)
            if a > 0:
                yaw = rng.uniform(-math.pi, math.pi)
                tx, ty = rng.uniform(-max_shift, max_shift, size=2)
                c, s = math.cos(yaw), math.sin(yaw)
                T[0, 0], T[0, 1], T[1, 0], T[1, 1] = c, -s, s, c
                T[0, 3], T[1, 3] = tx, ty
            world.append(T)
        for i in range(n):
            for j in range(n):
                if i != j:
                    out[b, i, j] = np.linalg.solve(world[j], world[i])
    return out


def make_inputs(record_len: Sequence[int], C: int, H: int, W: int, seed: int,
                max_cav: int = 5, max_shift: float = 40.0) -> Dict[str, np.ndarray]:
    """Synthetic batch for the path: ``feat`` [sumN,C,H,W] f32, ``cond`` [sumN,2,H,W] f32,
    ``record_len`` [B] int64, ``pairwise_t_matrix`` [B,L,L,4,4] f64."""
    n = int(sum(record_len))
    rng = np.random.RandomState(seed)
    feat = np.maximum(rng.standard_normal((n, C, H, W)), 0.0).astype(np.float32)
    cond = rng.standard_normal((n, 2, H, W)).astype(np.float32)
    return {
        "feat": feat,
        "cond": cond,
        "record_len": np.asarray(record_len, dtype=np.int64),
        "pairwise_t_matrix": make_pairwise_t_matrix(record_len, max_cav, seed + 7, max_shift),
    }


def noise_stream(seed: int, k: int, shape: Tuple[int, ...]) -> np.ndarray:
    """The k-th N(0,1) draw of a run, as float32. Each draw has its own RandomState so that
    a consumer may skip draws it does not need (the reference draws and discards several)."""
    return np.random.RandomState((seed * 1000003 + k) % (2 ** 31 - 1)).standard_normal(shape).astype(np.float32)


def make_eval_noise(seed: int, n: int, C: int, H: int, W: int, T: int) -> Tuple[np.ndarray, np.ndarray]:
    """Noise in the order the reference's eval branch consumes it
    (``opencood/models/gencomm_modules/cond_diff.py:367-375``, ``:307``):
    draw 0 = x_start noise [n,C,H,W]; draws 1,2 = dead q_samples on the ego [1,C,H,W];
    draws 3 .. 3+T-1 = one per denoise step, in loop order t = T-1 .. 0 (the last is drawn but
    discarded). Returns (noise0 [n,C,H,W], step_noise [T,n,C,H,W]) where step_noise[i] is the draw
    of loop iteration i (timestep T-1-i)."""
    shape = (n, C, H, W)
    noise0 = noise_stream(seed, 0, shape)
    steps = np.stack([noise_stream(seed, 3 + i, shape) for i in range(T)], axis=0)
    return noise0, steps


def make_train_noise(seed: int, n: int, C: int, H: int, W: int, T: int) -> Tuple[np.ndarray, np.ndarray]:
    """Noise in the order of the training branch (``cond_diff.py:342-360``): per agent a:
    one x_start draw [1,C,H,W], then T step draws [1,C,H,W] (t = T-1 .. 0; the t=0 one is discarded).
    Draw index = a*(T+1) + {0, 1+i}. Returns the same layout as :func:`make_eval_noise`."""
    shape1 = (1, C, H, W)
    noise0 = np.concatenate([noise_stream(seed, a * (T + 1), shape1) for a in range(n)], axis=0)
    steps = np.stack([
        np.concatenate([noise_stream(seed, a * (T + 1) + 1 + i, shape1) for a in range(n)], axis=0)
        for i in range(T)], axis=0)
    return noise0, steps


def make_pillars(M: int, B: int, nx: int, ny: int, seed: int, voxel_size=(0.4, 0.4, 4.0),
                 pc_range=(-140.8, -40.0, -3.0, 140.8, 40.0, 1.0), max_points: int = 32) -> Dict[str, np.ndarray]:
    """Synthetic output of the voxeliser for the PointPillars front half:
    ``voxel_features`` [M,32,4] (x,y,z,intensity; unused slots zero), ``voxel_num_points`` [M],
    ``voxel_coords`` [M,4] = (batch, z, y, x) with UNIQUE (batch, y, x) so that the scatter has no
    write conflicts (a voxeliser never emits duplicates)."""
    rng = np.random.RandomState(seed)
    cells = rng.choice(B * ny * nx, size=M, replace=False)
    b, rem = cells // (ny * nx), cells % (ny * nx)
    yy, xx = rem // nx, rem % nx
    coords = np.stack([b, np.zeros_like(b), yy, xx], axis=1).astype(np.int32)
    npts = rng.randint(1, max_points + 1, size=M).astype(np.int32)
    feats = np.zeros((M, max_points, 4), dtype=np.float32)
    u = rng.uniform(0.0, 1.0, size=(M, max_points, 3))
    feats[:, :, 0] = pc_range[0] + (xx[:, None] + u[:, :, 0]) * voxel_size[0]
    feats[:, :, 1] = pc_range[1] + (yy[:, None] + u[:, :, 1]) * voxel_size[1]
    feats[:, :, 2] = pc_range[2] + u[:, :, 2] * voxel_size[2]
    feats[:, :, 3] = rng.uniform(0.0, 1.0, size=(M, max_points))
    mask = np.arange(max_points)[None, :] < npts[:, None]
    feats *= mask[:, :, None]
    return {"voxel_features": feats, "voxel_num_points": npts, "voxel_coords": coords}


def fill_bn_stats_(module, seed: int) -> None:
    """Deterministic non-trivial BatchNorm running statistics (sorted buffer order): mean ~ N(0, 0.5), var ~ U(0.5, 2)."""
    import torch
    r = np.random.RandomState(seed)
    with torch.no_grad():
        for name, b in sorted(module.named_buffers(), key=lambda kv: kv[0]):
            if name.endswith("running_mean"):
                b.copy_(torch.from_numpy(r.normal(0.0, 0.5, tuple(b.shape)).astype(np.float32)))
            elif name.endswith("running_var"):
                b.copy_(torch.from_numpy(r.uniform(0.5, 2.0, tuple(b.shape)).astype(np.float32)))


def make_detection_maps(H: int, W: int, A: int, seed: int, n_obj: int = 60, num_bins: int = 2):
    """Synthetic head outputs for the detection tail: cls logits [1, A, H, W] (background -5, `n_obj` clusters of
    overlapping positives), reg deltas [1, 7A, H, W], dir logits [1, A*num_bins, H, W]. float32, numpy-deterministic."""
    r = np.random.RandomState(seed)
    cls = np.full((1, A, H, W), -5.0, dtype=np.float32) + r.normal(0, 0.3, (1, A, H, W)).astype(np.float32)
    for _ in range(n_obj):
        a, y, x = r.randint(A), r.randint(1, H - 1), r.randint(1, W - 1)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                if r.rand() < 0.7:
                    cls[0, a, y + dy, x + dx] = r.uniform(-1.0, 3.0)
    reg = r.normal(0, 0.15, (1, 7 * A, H, W)).astype(np.float32)
    dirp = r.normal(0, 1.0, (1, A * num_bins, H, W)).astype(np.float32)
    return cls, reg, dirp


def make_loss_inputs(seed: int, B: int, H: int, W: int, A: int, C: int, pos_frac: float = 0.02):
    """Synthetic head maps and anchor labels for the training criterion (PointPillarGencommLoss): cls / reg / dir predictions
    [B, A | 7A | 2A, H, W], labels `pos_equal_one` / `neg_equal_one` [B, H, W, A] (a few percent positives, a "don't care" band
    that is neither), regression `targets` [B, H, W, 7A] (yaw residuals across the whole circle, some inside the smooth-L1
    quadratic zone), and a `gt_feature` / `pred_feature` pair [B, C, H, W]. float32, numpy-deterministic."""
    r = np.random.RandomState(seed)
    f = np.float32
    u = r.rand(B, H, W, A)
    pos = (u < pos_frac).astype(f)
    neg = (u > 3 * pos_frac).astype(f)
    targets = r.normal(0, 0.4, (B, H, W, 7 * A)).astype(f)
    targets[..., 6::7] = r.uniform(-np.pi, np.pi, (B, H, W, A)).astype(f)
    reg = (targets.transpose(0, 3, 1, 2) + r.normal(0, 0.2, (B, 7 * A, H, W)) * (r.rand(B, 7 * A, H, W) < 0.7)).astype(f)
    gt = np.maximum(r.normal(0, 1, (B, C, H, W)), 0).astype(f)
    return {"cls_preds": r.normal(-2, 1.5, (B, A, H, W)).astype(f), "reg_preds": reg, "dir_preds": r.normal(0, 1, (B, 2 * A, H, W)).astype(f),
            "pos_equal_one": pos, "neg_equal_one": neg, "targets": targets, "gt_feature": gt,
            "pred_feature": (gt + r.normal(0, 0.3, gt.shape)).astype(f)}


def stage1_model_args(T: int = 3, lidar_range=(-102.4, -51.2, -3, 102.4, 51.2, 1), layer_nums=(3, 5, 8), C: int = 128) -> dict:
    """The `model.args` block of opv2v/GenComm_yamls/gencomm/stage1/m1_att.yaml (:93-166) as a dict: one lidar modality
    (PointPillars, 0.4 m pillars over `lidar_range` -> 512 x 256 grid by default), BaseBEVBackbone `layer_nums`, shrink to
    C = 128 channels at 1/4 resolution (64 x 128), message extractor, GenComm (T steps), Enhancer, AttFusion, heads with two
    anchors. Used by the training leg of bench.py and tools/shell_bench.py; tests use the reduced geometry of
    tests/golden/shell_state_dict_keys.json."""
    rng_ = [float(v) for v in lidar_range]
    return {
        "ego_modality": "m1", "lidar_range": rng_,
        "m1": {"core_method": "point_pillar", "sensor_type": "lidar",
               "encoder_args": {"voxel_size": [0.4, 0.4, 4], "lidar_range": rng_,
                                "pillar_vfe": {"use_norm": True, "with_distance": False, "use_absolute_xyz": True, "num_filters": [64]},
                                "point_pillar_scatter": {"num_features": 64}},
               "backbone_args": {"layer_nums": list(layer_nums), "layer_strides": [2, 2, 2], "num_filters": [64, 128, 256],
                                 "upsample_strides": [1, 2, 4], "num_upsample_filter": [128, 128, 128]},
               "shrink_header": {"kernal_size": [3], "stride": [2], "padding": [1], "dim": [C], "input_dim": 384}},
        "enhancer": {"in_ch": C}, "message_extractor": {"in_ch": C, "out_ch": 2},
        "fusion_method": "att", "att": {"feat_dim": C}, "in_head": C, "anchor_number": 2,
        "dir_args": {"dir_offset": 0.7853, "num_bins": 2, "anchor_yaw": [0, 90]}, "gmatch": True,
        "gencomm": default_gencomm_cfg(C, T),
    }


STAGE1_LOSS_ARGS = {  # opv2v/GenComm_yamls/gencomm/stage1/m1_att.yaml:168-189
    "pos_cls_weight": 2.0, "cls": {"type": "SigmoidFocalLoss", "alpha": 0.25, "gamma": 2.0, "weight": 2.0},
    "reg": {"type": "WeightedSmoothL1Loss", "sigma": 3.0, "codewise": True, "weight": 2.0},
    "dir": {"type": "WeightedSoftmaxClassificationLoss", "weight": 0.2, "args": {"dir_offset": 0.7853, "num_bins": 2, "anchor_yaw": [0, 90]}},
    "depth": {"weight": 1.0}, "generate_weight": 1, "gmatch_weight": 1}


def trained_looking_heads_(model, weight_seed: int) -> None:
    """Deterministic 'trained-looking' detection heads for the synthetic AP chain (tests/golden/apchain.npz): synthetic weights
    everywhere, then the classification head's bias is lowered so that a few percent of the anchors pass the 0.2 score threshold and
    the regression head is damped so that decoded boxes stay car-sized -- applied identically to the reference's shell by
    oracle/make_golden.py and to this package's shell by tests/test_ap_chain.py."""
    import torch
    fill_params_(model, weight_seed)
    fill_bn_stats_(model, weight_seed + 1)
    with torch.no_grad():
        model.cls_head.weight.mul_(4.0)
        model.cls_head.bias.fill_(-2.2)
        model.reg_head.weight.mul_(0.5)
        model.reg_head.bias.zero_()
