"""``PointPillar`` encoder front half -- host-side mirror of ``opencood/models/heter_encoders.py:22-50``
(``PillarVFE`` ``sub_modules/pillar_vfe.py:57-155`` + ``PointPillarScatter``
``sub_modules/point_pillar_scatter.py:9-76``), SURVEY.md 8f-2. Same constructor arguments, attribute
names (``pillar_vfe.pfn_layers.0.{linear,norm}``, ``scatter``) and ``state_dict`` keys; eval mode runs
one fused HIP kernel (augment -> Linear -> folded BatchNorm -> ReLU -> max -> scatter)."""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .runtime import f32c, ptr, require_gpu, stream_ptr


class PFNLayer(nn.Module):  # pillar_vfe.py:10-29
    def __init__(self, in_channels, out_channels, use_norm=True, last_layer=False):
        super().__init__()
        self.last_vfe, self.use_norm = last_layer, use_norm
        if not last_layer:
            out_channels = out_channels // 2
        if use_norm:
            self.linear = nn.Linear(in_channels, out_channels, bias=False)
            self.norm = nn.BatchNorm1d(out_channels, eps=1e-3, momentum=0.01)
        else:
            self.linear = nn.Linear(in_channels, out_channels, bias=True)


class PillarVFE(nn.Module):  # pillar_vfe.py:57-90
    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range):
        super().__init__()
        self.use_norm = model_cfg["use_norm"]
        self.with_distance = model_cfg["with_distance"]
        self.use_absolute_xyz = model_cfg["use_absolute_xyz"]
        num_point_features += 6 if self.use_absolute_xyz else 3
        if self.with_distance:
            num_point_features += 1
        self.num_filters = list(model_cfg["num_filters"])
        nf = [num_point_features] + self.num_filters
        self.pfn_layers = nn.ModuleList([PFNLayer(nf[i], nf[i + 1], self.use_norm, last_layer=(i >= len(nf) - 2))
                                         for i in range(len(nf) - 1)])
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]

    def get_output_feature_dim(self):
        return self.num_filters[-1]


class PointPillarScatter(nn.Module):  # point_pillar_scatter.py:9-17
    def __init__(self, model_cfg):
        super().__init__()
        self.num_bev_features = model_cfg["num_features"]
        self.nx, self.ny, self.nz = (int(v) for v in model_cfg["grid_size"])
        assert self.nz == 1


class PointPillar(nn.Module):
    def __init__(self, args):
        super().__init__()
        grid = (np.array(args["lidar_range"][3:6]) - np.array(args["lidar_range"][0:3])) / np.array(args["voxel_size"])
        args["point_pillar_scatter"]["grid_size"] = np.round(grid).astype(np.int64)
        self.pillar_vfe = PillarVFE(args["pillar_vfe"], num_point_features=4, voxel_size=args["voxel_size"],
                                    point_cloud_range=args["lidar_range"])
        self.scatter = PointPillarScatter(args["point_pillar_scatter"])

    def _check_supported(self):
        v = self.pillar_vfe
        if not (v.use_norm and v.use_absolute_xyz and not v.with_distance and v.num_filters == [64]
                and self.scatter.num_bev_features == 64):
            raise NotImplementedError("gencomm_amd.PointPillar: the HIP kernel covers the shipped configuration "
                                      "(use_norm, use_absolute_xyz, no distance feature, num_filters [64])")
        pfn = v.pfn_layers[0]
        if pfn.norm.training:
            # a frozen encoder inside a model in train mode (stage 2: fix_bn keeps its BatchNorm in eval mode) is fine
            raise NotImplementedError("gencomm_amd.PointPillar: training-mode BatchNorm (batch statistics) is not implemented; use .eval() "
                                      "(or freeze the encoder as stage 2 does)")

    def forward(self, data_dict, modality_name):
        inp = data_dict[f"inputs_{modality_name}"]
        return self.encode(inp["voxel_features"], inp["voxel_coords"], inp["voxel_num_points"])

    def encode(self, voxel_features, voxel_coords, voxel_num_points, batch_size=None):
        """[M,P,4], [M,4] (b,z,y,x), [M] -> [B,64,ny,nx]."""
        self._check_supported()
        require_gpu(voxel_features, "PointPillar.forward")
        vf = f32c(voxel_features)
        M, P = vf.shape[0], vf.shape[1]
        coords = voxel_coords.to(torch.int32).contiguous()
        npts = voxel_num_points.to(torch.int32).contiguous()
        if batch_size is None:  # the reference derives it the same way, with the same sync (point_pillar_scatter.py:45)
            batch_size = int(coords[:, 0].max().item()) + 1 if M > 0 else 1
        pfn = self.pillar_vfe.pfn_layers[0]
        out = torch.empty((batch_size, 64, self.scatter.ny, self.scatter.nx), dtype=torch.float32, device=vf.device)
        scratch = torch.empty(128, dtype=torch.float32, device=vf.device)
        vs = (ctypes.c_float * 3)(*self.pillar_vfe.voxel_size)
        rg = (ctypes.c_float * 6)(*self.pillar_vfe.point_cloud_range)
        _lib.check(_lib.lib().gencomm_pillar_encode_fwd(
            ptr(vf), ptr(npts), ptr(coords), ptr(f32c(pfn.linear.weight.detach())), ptr(f32c(pfn.norm.weight.detach())),
            ptr(f32c(pfn.norm.bias.detach())), ptr(f32c(pfn.norm.running_mean)), ptr(f32c(pfn.norm.running_var)),
            ptr(out), ptr(scratch), M, P, batch_size, self.scatter.nx, self.scatter.ny, vs, rg, stream_ptr(vf.device)),
            "gencomm_pillar_encode_fwd")
        return out
