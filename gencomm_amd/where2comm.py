"""``Where2commFusion`` -- host-side mirror of ``opencood/models/fuse_modules/fusion_in_one.py:466-519`` and of the ``EncodeLayer``
it wraps (``fuse_modules/where2comm_attn.py:64-102``): warp every agent to the ego frame, then per pixel a multi-head attention
(``nn.MultiheadAttention``, 8 heads) of the ego token over the agents' tokens, residual + LayerNorm, a two-layer ReLU feed-forward,
residual + LayerNorm. Same constructor argument (``feature_dims``), attribute names and ``state_dict`` keys
(``mha_fusion.attn.in_proj_weight``, ``mha_fusion.attn.out_proj.bias``, ``mha_fusion.linear1.weight``, ``mha_fusion.norm2.bias`` ...).

Everything runs on kernels the library already has, forward and backward (``_Where2commFn``):
  warp to ego           gencomm_warp_affine_fwd / _bwd
  in_proj (q | k | v)   1x1 convolution [3C, C] over the NCHW maps; out_proj, linear1 (+ ReLU), linear2 likewise
  agent attention       gencomm_hgt_attn_fwd / _bwd (per pixel and head, softmax(q k / sqrt(C / heads)) across the agents of a scene: the
                        layout nn.MultiheadAttention sees with seq = agents, batch = pixels); only the ego row is used downstream
  LayerNorm             gencomm_ln_nchw_fwd / _bwd
The reference's dropouts have p = 0 (EncodeLayer's default, the only value fusion_in_one.py constructs)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib, train_ops as T
from .fusion import MAX_AGENTS_PER_SCENE, gather_ego_thetas, record_len_list
from .runtime import dev_ints, f32c, ptr, require_gpu, stream_ptr
from .v2xvit_bwd import _lin, _lin_bwd


class EncodeLayer(nn.Module):  # where2comm_attn.py:64-79 (parameters only; the arithmetic is in _run below)
    def __init__(self, channels, n_head=8, dropout=0):
        super().__init__()
        if dropout != 0:
            raise NotImplementedError("EncodeLayer: dropout > 0 is not used by the reference's fusion and not implemented")
        self.attn = nn.MultiheadAttention(channels, n_head, dropout)
        self.linear1 = nn.Linear(channels, channels)
        self.linear2 = nn.Linear(channels, channels)
        self.norm1 = nn.LayerNorm(channels)
        self.norm2 = nn.LayerNorm(channels)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.relu = nn.ReLU()


def _run(layer: EncodeLayer, x, theta, scene_off, B):
    """The fusion step by step on the HIP kernels; returns (out [B, C, H, W], tape of what the backward needs)."""
    l, dev = _lib.lib(), x.device
    st = stream_ptr(dev)
    n, C, H, W = x.shape
    heads = layer.attn.num_heads
    xw = torch.empty_like(x)
    _lib.check(l.gencomm_warp_affine_fwd(ptr(x), ptr(theta), ptr(xw), n, C, H, W, st), "gencomm_warp_affine_fwd")
    qkv = _lin(xw, layer.attn.in_proj_weight, layer.attn.in_proj_bias.detach())
    o = torch.empty(n, C, H, W, dtype=torch.float32, device=dev)
    _lib.check(l.gencomm_hgt_attn_fwd(ptr(qkv), ptr(scene_off), ptr(o), B, heads, C // heads, H * W, st), "gencomm_hgt_attn_fwd")
    ego = scene_off[:-1].long()
    o_ego, res = o[ego].contiguous(), xw[ego].contiguous()
    s1 = res + _lin(o_ego, layer.attn.out_proj.weight, layer.attn.out_proj.bias.detach())
    y1 = T.ln_fwd(s1, layer.norm1.weight.detach(), layer.norm1.bias.detach(), layer.norm1.eps, False)
    h1 = torch.relu_(_lin(y1, layer.linear1.weight, layer.linear1.bias.detach()))
    s2 = y1 + _lin(h1, layer.linear2.weight, layer.linear2.bias.detach())
    y2 = T.ln_fwd(s2, layer.norm2.weight.detach(), layer.norm2.bias.detach(), layer.norm2.eps, False)
    return y2, (xw, qkv, o_ego, s1, y1, h1, s2)


class _Where2commFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, layer, x, theta, scene_off, B, *params):
        ctx.layer, ctx.B = layer, B
        with torch.no_grad():
            y, tape = _run(layer, x, theta, scene_off, B)
        ctx.save_for_backward(x, theta, scene_off, *tape)
        return y

    @staticmethod
    def backward(ctx, gout):
        x, theta, scene_off, xw, qkv, o_ego, s1, y1, h1, s2 = ctx.saved_tensors
        layer, B = ctx.layer, ctx.B
        l, dev = _lib.lib(), x.device
        st = stream_ptr(dev)
        n, C, H, W = x.shape
        heads = layer.attn.num_heads
        ego = scene_off[:-1].long()
        g = {}
        with torch.no_grad():
            ds2, g[layer.norm2.weight], g[layer.norm2.bias] = T.ln_bwd(s2, layer.norm2.weight.detach(), f32c(gout), layer.norm2.eps)
            dh1, g[layer.linear2.weight], g[layer.linear2.bias] = _lin_bwd(ds2, h1, layer.linear2.weight, True)
            dh1 = dh1 * (h1 > 0)
            dy1, g[layer.linear1.weight], g[layer.linear1.bias] = _lin_bwd(dh1, y1, layer.linear1.weight, True)
            dy1 = dy1 + ds2
            ds1, g[layer.norm1.weight], g[layer.norm1.bias] = T.ln_bwd(s1, layer.norm1.weight.detach(), dy1, layer.norm1.eps)
            do_ego, g[layer.attn.out_proj.weight], g[layer.attn.out_proj.bias] = _lin_bwd(ds1, o_ego, layer.attn.out_proj.weight, True)
            do = torch.zeros(n, C, H, W, dtype=torch.float32, device=dev)
            do[ego] = do_ego
            dqkv = torch.empty_like(qkv)
            _lib.check(l.gencomm_hgt_attn_bwd(ptr(qkv), ptr(scene_off), ptr(do), ptr(dqkv), B, heads, C // heads, H * W, st), "gencomm_hgt_attn_bwd")
            dxw, g[layer.attn.in_proj_weight], g[layer.attn.in_proj_bias] = _lin_bwd(dqkv, xw, layer.attn.in_proj_weight, True)
            dxw[ego] += ds1                                  # the residual of the first LayerNorm is the warped ego map
            dx = None
            if ctx.needs_input_grad[1]:
                dx = torch.empty_like(x)
                _lib.check(l.gencomm_warp_affine_bwd(ptr(theta), ptr(f32c(dxw)), ptr(dx), n, C, H, W, st), "gencomm_warp_affine_bwd")
        return (None, dx, None, None, None, *[g.get(p) if p.requires_grad else None for p in layer.parameters()])


class Where2commFusion(nn.Module):
    def __init__(self, feature_dims):
        super().__init__()
        self.mha_fusion = EncodeLayer(feature_dims)

    def forward(self, x, record_len, affine_matrix):
        """x [sumN, C, H, W], record_len [B], affine_matrix [B, L, L, 2, 3] -> [B, C, H, W]."""
        require_gpu(x, "Where2commFusion.forward")
        lens = record_len_list(record_len)
        n, C, H, W = x.shape
        if len(lens) != affine_matrix.shape[0] or sum(lens) != n or min(lens) < 1 or max(lens) > MAX_AGENTS_PER_SCENE:
            raise ValueError(f"record_len {lens} inconsistent with input / 1..{MAX_AGENTS_PER_SCENE} agents per scene")
        heads = self.mha_fusion.attn.num_heads
        if C != self.mha_fusion.attn.embed_dim or C // heads not in (8, 16, 32, 64):
            raise NotImplementedError(f"Where2commFusion: {C} channels / {heads} heads: the attention kernel has head widths 8, 16, 32, 64")
        theta = gather_ego_thetas(affine_matrix, lens).to(x.device)
        off = [0]
        for v in lens:
            off.append(off[-1] + v)
        scene_off = dev_ints(off, x.device)
        x = f32c(x)
        params = list(self.mha_fusion.parameters())
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            return _Where2commFn.apply(self.mha_fusion, x, theta, scene_off, len(lens), *params)
        with torch.no_grad():
            return _run(self.mha_fusion, x, theta, scene_off, len(lens))[0]
