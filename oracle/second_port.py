"""ORACLE (test infrastructure, never imported by the product): the SECOND encoder of the reference on DENSE volumes.

  SECOND.forward            opencood/models/heter_encoders.py:66-81
  MeanVFE.forward           opencood/models/sub_modules/mean_vfe.py:14-33      (sum over all point slots / clamp_min(num_points, 1))
  VoxelBackBone8x           opencood/models/sub_modules/sparse_backbone_3d.py:33-152
  post_act_block            :12-31    (conv without bias, BatchNorm1d(eps 1e-3), ReLU)
  HeightCompression         opencood/models/sub_modules/height_compression.py:10-30   (dense(), [N, C, D, H, W] -> [N, C D, H, W])

PARITY UNPINNED: the sparse convolutions themselves live in spconv (`pip install spconv-cu116`, reference README.md:116; call
sites sparse_backbone_3d.py:7-9, :17-23), which is neither under /root/reference nor installed. Their published semantics are
restated here as dense torch convolutions with an explicit active-site mask:
  SubMConv3d(k, padding ignored)    active set unchanged; out = conv3d(x, w, padding = k // 2) at the active sites, 0 elsewhere
                                    (inactive inputs are zeros in the dense volume, so they contribute nothing to the sum)
  SparseConv3d(k, stride, padding)  active set = max_pool3d(mask, k, stride, padding) > 0 (a site is active as soon as one active
                                    input lies in its receptive field); out = conv3d(x, w, stride, padding) there, 0 elsewhere
  BatchNorm1d / ReLU                act on the active rows only: inactive sites stay exactly zero (BatchNorm's shift never
                                    reaches them), which is what SparseConvTensor.dense() returns
Weights: spconv 2.x layout [Cout, kD, kH, kW, Cin] (a 5-D tensor whose first three trailing sizes are not the kernel size is
read as the spconv 1.x layout [kD, kH, kW, Cin, Cout]).  Anchored by the known-answer cases in tests/test_second.py."""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
BN_EPS = 1e-3  # sparse_backbone_3d.py:37


def mean_vfe(voxel_features: torch.Tensor, voxel_num_points: torch.Tensor) -> torch.Tensor:
    s = voxel_features.sum(dim=1)
    return s / torch.clamp_min(voxel_num_points.view(-1, 1).to(s.dtype), 1.0)


def to_dense(features: torch.Tensor, coords: torch.Tensor, batch: int, shape: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor]:
    """features [M, C], coords [M, 4] (b, z, y, x) -> dense [B, C, D, H, W], mask [B, 1, D, H, W]."""
    D, H, W = shape
    x = features.new_zeros(batch, features.shape[1], D, H, W)
    m = features.new_zeros(batch, 1, D, H, W)
    b, z, y, xx = (coords[:, i].long() for i in range(4))
    x[b, :, z, y, xx] = features
    m[b, 0, z, y, xx] = 1.0
    return x, m


def torch_weight(w: torch.Tensor, kernel: Sequence[int]) -> torch.Tensor:
    """spconv weight -> conv3d weight [Cout, Cin, kD, kH, kW]."""
    if tuple(w.shape[1:4]) == tuple(kernel):
        return w.permute(0, 4, 1, 2, 3).contiguous()      # 2.x: [Cout, kD, kH, kW, Cin]
    assert tuple(w.shape[0:3]) == tuple(kernel), w.shape
    return w.permute(4, 3, 0, 1, 2).contiguous()           # 1.x: [kD, kH, kW, Cin, Cout]


TRAIN_BN = False   # tests flip this to restate BatchNorm1d in TRAINING mode: statistics over the active rows of the batch (biased variance)


def _bn_relu(sd: SD, p: str, y: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    g, b, mu, var = (sd[f"{p}.{k}"].view(1, -1, 1, 1, 1) for k in ("weight", "bias", "running_mean", "running_var"))
    if TRAIN_BN:
        cnt = mask.sum()
        mu = (y * mask).sum((0, 2, 3, 4), keepdim=True) / cnt
        var = (((y - mu) ** 2) * mask).sum((0, 2, 3, 4), keepdim=True) / cnt
    return torch.relu((y - mu) / torch.sqrt(var + BN_EPS) * g + b) * mask


def subm_block(sd: SD, conv: str, bn: str, x: torch.Tensor, mask: torch.Tensor, kernel=(3, 3, 3)) -> torch.Tensor:
    w = torch_weight(sd[conv + ".weight"], kernel)
    y = F.conv3d(x, w, padding=tuple(k // 2 for k in kernel))
    return _bn_relu(sd, bn, y, mask)


def spconv_block(sd: SD, conv: str, bn: str, x: torch.Tensor, mask: torch.Tensor, kernel, stride, padding):
    w = torch_weight(sd[conv + ".weight"], kernel)
    y = F.conv3d(x, w, stride=tuple(stride), padding=tuple(padding))
    m = (F.max_pool3d(mask, tuple(kernel), tuple(stride), tuple(padding)) > 0).to(x.dtype)
    return _bn_relu(sd, bn, y, m), m


def voxel_backbone_8x(sd: SD, p: str, features: torch.Tensor, coords: torch.Tensor, batch: int, sparse_shape: Sequence[int]):
    """sparse_backbone_3d.py:96-150 -> dense encoded tensor [B, C, D', H', W'] (+ the per-stage dense maps and masks)."""
    x, m = to_dense(features, coords, batch, sparse_shape)
    x = subm_block(sd, f"{p}.conv_input.0", f"{p}.conv_input.1", x, m)
    x1 = subm_block(sd, f"{p}.conv1.0.0", f"{p}.conv1.0.1", x, m)
    stages = {"x_conv1": (x1, m)}
    x, pads = x1, {2: (1, 1, 1), 3: (1, 1, 1), 4: (0, 1, 1)}
    for lvl in (2, 3, 4):
        x, m = spconv_block(sd, f"{p}.conv{lvl}.0.0", f"{p}.conv{lvl}.0.1", x, m, (3, 3, 3), (2, 2, 2), pads[lvl])
        for j in (1, 2):
            x = subm_block(sd, f"{p}.conv{lvl}.{j}.0", f"{p}.conv{lvl}.{j}.1", x, m)
        stages[f"x_conv{lvl}"] = (x, m)
    out, mo = spconv_block(sd, f"{p}.conv_out.0", f"{p}.conv_out.1", x, m, (3, 1, 1), (2, 1, 1), (0, 0, 0))
    return out, mo, stages


def second_forward(sd: SD, p: str, voxel_features: torch.Tensor, voxel_coords: torch.Tensor, voxel_num_points: torch.Tensor,
                   grid_size_xyz: Sequence[int]) -> torch.Tensor:
    """SECOND.forward (heter_encoders.py:66-81): grid_size = round((range[3:6] - range[:3]) / voxel_size) (x, y, z);
    sparse_shape = grid_size[::-1] + [1, 0, 0] (sparse_backbone_3d.py:39)."""
    batch = int(voxel_coords[:, 0].max()) + 1
    shape = [int(grid_size_xyz[2]) + 1, int(grid_size_xyz[1]), int(grid_size_xyz[0])]
    feats = mean_vfe(voxel_features, voxel_num_points)
    out, _, _ = voxel_backbone_8x(sd, p + ".spconv_block" if p else "spconv_block", feats, voxel_coords, batch, shape)
    N, C, D, H, W = out.shape
    return out.reshape(N, C * D, H, W)
