"""``PointPillar`` encoder front half -- host-side mirror of ``opencood/models/heter_encoders.py:22-50``
(``PillarVFE`` ``sub_modules/pillar_vfe.py:57-155`` + ``PointPillarScatter``
``sub_modules/point_pillar_scatter.py:9-76``), SURVEY.md 8f-2. Same constructor arguments, attribute
names (``pillar_vfe.pfn_layers.0.{linear,norm}``, ``scatter``) and ``state_dict`` keys; inference (and a frozen
encoder inside a training model) runs one fused HIP kernel (augment -> Linear -> folded BatchNorm -> ReLU -> max -> scatter);
training the encoder itself (stage 1) runs the per-pillar network layer by layer on HIP kernels with a HIP backward (`_PillarNetFn`)."""
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


class _PillarNetFn(torch.autograd.Function):
    """PFNLayer (pillar_vfe.py:31-54) with gradients: Linear (no bias) -> BatchNorm1d -> ReLU -> max over the P slots, on HIP kernels:
    1x1 convolution + its weight gradient, BatchNorm with batch statistics (train mode) or the running statistics (eval mode with
    gradients), slot max with arg-max routing."""

    @staticmethod
    def forward(ctx, x4, pfn, M, P, *params):
        from . import train_ops as T
        l, dev = _lib.lib(), x4.device
        w = pfn.linear.weight.detach()
        Cout = w.shape[0]
        pre = T.conv2d(x4, w[:, :, None, None], None, 0)                                       # [1, 64, 1, M P]
        bn = pfn.norm
        if bn.training:
            y, save = T.bn2d_train_fwd(pre, bn, True)
        else:
            save = torch.stack([bn.running_mean.float(), torch.rsqrt(bn.running_var.float() + bn.eps)], 1).contiguous()
            y = torch.relu((pre - save[:, 0].view(1, -1, 1, 1)) * (save[:, 1] * bn.weight.detach().float()).view(1, -1, 1, 1) + bn.bias.detach().float().view(1, -1, 1, 1))
        out = torch.empty(M, Cout, dtype=torch.float32, device=dev)
        arg = torch.empty(M, Cout, dtype=torch.uint8, device=dev)
        _lib.check(l.gencomm_slot_max_fwd(ptr(y), ptr(out), ptr(arg), Cout, M, P, stream_ptr(dev)), "gencomm_slot_max_fwd")
        ctx.pfn, ctx.M, ctx.P, ctx.train = pfn, M, P, bn.training
        ctx.save_for_backward(x4, pre, y, save, arg)
        return out

    @staticmethod
    def backward(ctx, gout):
        from . import train_ops as T
        x4, pre, y, save, arg = ctx.saved_tensors
        pfn, M, P = ctx.pfn, ctx.M, ctx.P
        bn = pfn.norm
        l, dev = _lib.lib(), x4.device
        Cout = pre.shape[1]
        dy = torch.empty_like(y)
        _lib.check(l.gencomm_slot_max_bwd(ptr(f32c(gout)), ptr(arg), ptr(dy), Cout, M, P, stream_ptr(dev)), "gencomm_slot_max_bwd")
        if ctx.train:
            dpre, dg, db = T.bn2d_train_bwd(pre, y, dy, save, bn.weight, True)
        else:
            g = dy * (y > 0)
            dg = (g * (pre - save[:, 0].view(1, -1, 1, 1)) * save[:, 1].view(1, -1, 1, 1)).sum((0, 2, 3))
            db = g.sum((0, 2, 3))
            dpre = g * (save[:, 1] * bn.weight.detach().float()).view(1, -1, 1, 1)
        dw, _ = T.conv2d_wgrad(dpre, x4, 1, 0, False)
        return (None, None, None, None, dw[:, :, 0, 0] if ctx.needs_input_grad[4] else None,
                dg if ctx.needs_input_grad[5] else None, db if ctx.needs_input_grad[6] else None)


class _PillarNetFusedFn(torch.autograd.Function):
    """The same layer in TRAINING mode (batch statistics) as two launches forward and two backward, without the [M P, 64] intermediates
    (csrc/pfn_kernels.h: BatchNorm's statistics and the dense part of its backward follow exactly from the inputs' first and second
    moments; ReLU(BN(.)) is monotone in the Linear output, so the slot max is taken on it).  feats [M, P, F] with masked slots zeroed."""

    @staticmethod
    def forward(ctx, feats, pfn, *params):
        from .runtime import zeros as pool_zeros
        l, dev = _lib.lib(), feats.device
        M, P, F = feats.shape
        w = f32c(pfn.linear.weight.detach())
        bn = pfn.norm
        C = w.shape[0]
        gamma, beta = f32c(bn.weight.detach()), f32c(bn.bias.detach())
        out = torch.empty(M, C, dtype=torch.float32, device=dev)
        arg = torch.empty(M, C, dtype=torch.uint8, device=dev)
        save = torch.empty(C, 2, dtype=torch.float32, device=dev)
        moments = torch.empty(_lib.check_size(l.gencomm_pfn_moment_doubles(F), "gencomm_pfn_moment_doubles"), dtype=torch.float64, device=dev)
        track = bn.track_running_stats and bn.running_mean is not None
        momentum = 0.0 if bn.momentum is None else float(bn.momentum)
        _lib.check(l.gencomm_pfn_train_fwd(ptr(feats), ptr(w), ptr(gamma), ptr(beta), ptr(bn.running_mean) if track else 0, ptr(bn.running_var) if track else 0,
                                           ptr(bn.num_batches_tracked) if track and bn.num_batches_tracked is not None else 0, momentum, float(bn.eps),
                                           ptr(out), ptr(arg), ptr(save), ptr(moments), M, P, F, C, stream_ptr(dev)), "gencomm_pfn_train_fwd")
        ctx.pfn = pfn
        ctx.save_for_backward(feats, w, gamma, beta, save, moments, arg)
        return out

    @staticmethod
    def backward(ctx, gout):
        feats, w, gamma, beta, save, moments, arg = ctx.saved_tensors
        l, dev = _lib.lib(), feats.device
        M, P, F = feats.shape
        C = w.shape[0]
        dw = torch.empty(C, F, dtype=torch.float32, device=dev)
        dg = torch.empty(C, dtype=torch.float32, device=dev)
        db = torch.empty(C, dtype=torch.float32, device=dev)
        scratch = torch.empty(_lib.check_size(l.gencomm_pfn_bwd_scratch_doubles(F, C), "gencomm_pfn_bwd_scratch_doubles"), dtype=torch.float64, device=dev)
        _lib.check(l.gencomm_pfn_train_bwd(ptr(feats), ptr(w), ptr(gamma), ptr(beta), ptr(save), ptr(moments), ptr(f32c(gout)), ptr(arg),
                                           ptr(dw), ptr(dg), ptr(db), ptr(scratch), M, P, F, C, stream_ptr(dev)), "gencomm_pfn_train_bwd")
        return (None, None, dw if ctx.needs_input_grad[2] else None, dg if ctx.needs_input_grad[3] else None, db if ctx.needs_input_grad[4] else None)


class PointPillar(nn.Module):
    def __init__(self, args):
        super().__init__()
        grid = (np.array(args["lidar_range"][3:6]) - np.array(args["lidar_range"][0:3])) / np.array(args["voxel_size"])
        args["point_pillar_scatter"]["grid_size"] = np.round(grid).astype(np.int64)
        self.pillar_vfe = PillarVFE(args["pillar_vfe"], num_point_features=4, voxel_size=args["voxel_size"],
                                    point_cloud_range=args["lidar_range"])
        self.scatter = PointPillarScatter(args["point_pillar_scatter"])
        self.fused_train_pfn = True     # False: the composed 1x1 convolution + BatchNorm + slot-max path (what the fused kernels are tested against)

    def _check_supported(self):
        v = self.pillar_vfe
        if not (v.use_norm and v.use_absolute_xyz and not v.with_distance and v.num_filters == [64]
                and self.scatter.num_bev_features == 64):
            raise NotImplementedError("gencomm_amd.PointPillar: the HIP kernel covers the shipped configuration "
                                      "(use_norm, use_absolute_xyz, no distance feature, num_filters [64])")

    def forward(self, data_dict, modality_name):
        inp = data_dict[f"inputs_{modality_name}"]
        # the number of agents of this modality is known on the host when the shell calls (heter_model.py): no `coords[:, 0].max().item()`
        # synchronisation at the very start of the model (the reference derives it that way, point_pillar_scatter.py:45)
        mods = data_dict.get("agent_modality_list")
        batch = mods.count(modality_name) if isinstance(mods, (list, tuple)) and modality_name in mods else None
        return self.encode(inp["voxel_features"], inp["voxel_coords"], inp["voxel_num_points"], batch_size=batch)

    def _encode_train(self, vf, coords, npts, batch_size):
        """Training / gradient path (stage 1 trains the encoder; BatchNorm1d with batch statistics when in train mode): the 10-feature
        augmentation of the raw points is elementwise tensor arithmetic (no parameters), then `_PillarNetFn` -- Linear 10 -> 64 as a
        1x1 convolution over the [1, 10, 1, M P] point-slot layout, BatchNorm + ReLU, max over the slots, all on HIP kernels with a HIP
        backward -- and the scatter by indexed assignment (point_pillar_scatter.py:42-76). The fused one-launch kernel stays the
        inference / frozen-encoder path."""
        v, pfn = self.pillar_vfe, self.pillar_vfe.pfn_layers[0]
        vx, vy, vz = v.voxel_size
        xo, yo, zo = vx / 2 + v.point_cloud_range[0], vy / 2 + v.point_cloud_range[1], vz / 2 + v.point_cloud_range[2]
        with torch.no_grad():
            c = coords.to(vf.dtype)
            mean = vf[:, :, :3].sum(dim=1, keepdim=True) / npts.to(vf.dtype).view(-1, 1, 1)
            f_cluster = vf[:, :, :3] - mean
            f_center = torch.stack([vf[:, :, 0] - (c[:, 3].unsqueeze(1) * vx + xo), vf[:, :, 1] - (c[:, 2].unsqueeze(1) * vy + yo),
                                    vf[:, :, 2] - (c[:, 1].unsqueeze(1) * vz + zo)], dim=-1)
            mask = (npts.view(-1, 1) > torch.arange(vf.shape[1], device=vf.device).view(1, -1)).unsqueeze(-1).to(vf.dtype)
            feats = torch.cat([vf, f_cluster, f_center], dim=-1) * mask                     # [M, P, 10]
            M, P, F = feats.shape
        cout = pfn.linear.weight.shape[0]
        if pfn.norm.training and F in (9, 10, 11) and cout in (32, 64, 128, 256) and P <= 255 and M > 0 and self.fused_train_pfn:
            pillar = _PillarNetFusedFn.apply(feats.contiguous(), pfn, pfn.linear.weight, pfn.norm.weight, pfn.norm.bias)   # [M, 64]
        else:
            x4 = feats.permute(2, 0, 1).reshape(1, F, 1, M * P).contiguous()               # channel-major point slots
            pillar = _PillarNetFn.apply(x4, pfn, M, P, pfn.linear.weight, pfn.norm.weight, pfn.norm.bias)    # [M, 64]
        out = torch.zeros(batch_size, 64, self.scatter.ny * self.scatter.nx, dtype=vf.dtype, device=vf.device)
        idx = (coords[:, 1] + coords[:, 2] * self.scatter.nx + coords[:, 3]).long()
        out[coords[:, 0].long(), :, idx] = pillar
        return out.view(batch_size, 64, self.scatter.ny, self.scatter.nx)

    def encode(self, voxel_features, voxel_coords, voxel_num_points, batch_size=None):
        """[M,P,4], [M,4] (b,z,y,x), [M] -> [B,64,ny,nx]."""
        self._check_supported()
        require_gpu(voxel_features, "PointPillar.forward")
        pfn0 = self.pillar_vfe.pfn_layers[0]
        if pfn0.norm.training or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            if batch_size is None:
                batch_size = int(voxel_coords[:, 0].max().item()) + 1 if voxel_features.shape[0] > 0 else 1
            return self._encode_train(voxel_features.float(), voxel_coords, voxel_num_points, batch_size)
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
