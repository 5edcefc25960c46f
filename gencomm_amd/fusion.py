"""``AttFusion`` / ``regroup`` / ``normalize_pairwise_tfm`` -- host-side mirrors of
``opencood/models/fuse_modules/fusion_in_one.py:126-151, :48-51`` and
``opencood/utils/transformation_utils.py:68-92``; the warp + per-pixel attention runs as one HIP
gather kernel (``gencomm_warp_attfuse_fwd``)."""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from . import _lib
from .runtime import dev_ints, f32c, ptr, record_len_list, require_gpu, stream_ptr

MAX_AGENTS_PER_SCENE = 8


def regroup(x: torch.Tensor, record_len) -> tuple:
    """Split the sumN axis by scene (fusion_in_one.py:48-51)."""
    lens = record_len_list(record_len)
    return torch.split(x, lens, dim=0)


def normalize_pairwise_tfm(pairwise_t_matrix: torch.Tensor, H, W, discrete_ratio, downsample_rate=1) -> torch.Tensor:
    """[B,L,L,4,4] -> [B,L,L,2,3] normalised affine for ``affine_grid`` (transformation_utils.py:68-92).
    A few dozen float64 scalars per batch: done with torch indexing on whatever device the poses
    are on. Unlike the reference this does not modify a view of its argument."""
    # rows 0, 1 and columns 0, 1, 3 by slicing: indexing with Python lists builds index tensors on the host and copies them to the
    # device -- a host <-> device synchronisation in the middle of every forward (torch.cuda.set_sync_debug_mode found it)
    r = pairwise_t_matrix[:, :, :, 0:2, :]
    a = torch.cat([r[..., 0:2], r[..., 3:4]], dim=-1)
    a[..., 0, 1] = a[..., 0, 1] * H / W
    a[..., 1, 0] = a[..., 1, 0] * W / H
    a[..., 0, 2] = a[..., 0, 2] / (downsample_rate * discrete_ratio * W) * 2
    a[..., 1, 2] = a[..., 1, 2] / (downsample_rate * discrete_ratio * H) * 2
    return a


def gather_ego_thetas(affine_matrix: torch.Tensor, lens: List[int]) -> torch.Tensor:
    """[sumN, 2, 3] float64: for scene b the rows affine_matrix[b, 0, :N_b] (= ``t_matrix[0, :, :, :]``
    in AttFusion.forward, fusion_in_one.py:140-144)."""
    rows = [affine_matrix[b, 0, :n] for b, n in enumerate(lens)]
    return torch.cat(rows, dim=0).to(torch.float64).contiguous()


class ScaledDotProductAttention(nn.Module):  # fusion_in_one.py:14-45 (no parameters)
    def __init__(self, dim):
        super().__init__()
        self.sqrt_dim = float(dim) ** 0.5


class AttFusion(nn.Module):
    def __init__(self, feature_dims):
        super().__init__()
        self.att = ScaledDotProductAttention(feature_dims)

    def forward(self, xx, record_len, affine_matrix):
        """xx [sumN,C,H,W], record_len [B], affine_matrix [B,L,L,2,3] -> [B,C,H,W]."""
        require_gpu(xx, "AttFusion.forward")
        lens = record_len_list(record_len)
        n, C, H, W = xx.shape
        B = affine_matrix.shape[0]
        if len(lens) != B or sum(lens) != n:
            raise ValueError(f"record_len {lens} inconsistent with {n} agents / {B} scenes")
        if min(lens) < 1 or max(lens) > MAX_AGENTS_PER_SCENE:
            raise ValueError(f"each scene needs 1..{MAX_AGENTS_PER_SCENE} agents, got {lens}")
        xx = f32c(xx)
        if torch.is_grad_enabled() and xx.requires_grad:
            from .autograd import AttFusionFunction  # HIP forward and HIP backward
            return AttFusionFunction.apply(self, lens, affine_matrix, xx)
        return self._forward_hip(xx, lens, affine_matrix)

    _entry = "gencomm_warp_attfuse_fwd"

    def _forward_hip(self, xx, lens, affine_matrix):
        n, C, H, W = xx.shape
        B = affine_matrix.shape[0]
        theta = gather_ego_thetas(affine_matrix, lens).to(xx.device)
        off = [0]
        for k in lens:
            off.append(off[-1] + k)
        scene_off = dev_ints(off, xx.device)
        out = torch.empty((B, C, H, W), dtype=torch.float32, device=xx.device)
        _lib.check(getattr(_lib.lib(), self._entry)(ptr(xx), ptr(theta), ptr(scene_off), ptr(out), B, n, C, H, W,
                                                    stream_ptr(xx.device)), self._entry)
        return out


    def _backward_hip(self, xx, lens, affine_matrix, grad_out):
        n, C, H, W = xx.shape
        B = affine_matrix.shape[0]
        theta = gather_ego_thetas(affine_matrix, lens).to(xx.device)
        off = [0]
        for k in lens:
            off.append(off[-1] + k)
        scene_off = dev_ints(off, xx.device)
        gx = torch.empty_like(xx)
        l = _lib.lib()
        scratch = torch.empty(_lib.check_size(l.gencomm_warp_attfuse_bwd_scratch_floats(n, H, W), "gencomm_warp_attfuse_bwd_scratch_floats"),
                              dtype=torch.float32, device=xx.device)
        _lib.check(l.gencomm_warp_attfuse_bwd(ptr(xx), ptr(theta), ptr(scene_off), ptr(grad_out), ptr(gx), ptr(scratch), B, n, C, H, W,
                                              stream_ptr(xx.device)), "gencomm_warp_attfuse_bwd")
        return gx


class MaxFusion(AttFusion):
    """fusion_in_one.py:87-124: warp every agent to the ego frame, element-wise max over the agents (inference path;
    no autograd)."""
    _entry = "gencomm_warp_maxfuse_fwd"

    def __init__(self):
        nn.Module.__init__(self)

    def forward(self, x, record_len, affine_matrix):
        require_gpu(x, "MaxFusion.forward")
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError("MaxFusion: backward is not implemented on the HIP path")
        lens = record_len_list(record_len)
        if len(lens) != affine_matrix.shape[0] or sum(lens) != x.shape[0] or min(lens) < 1 or max(lens) > MAX_AGENTS_PER_SCENE:
            raise ValueError(f"record_len {lens} inconsistent with input / 1..{MAX_AGENTS_PER_SCENE} agents per scene")
        return self._forward_hip(f32c(x), lens, affine_matrix)
