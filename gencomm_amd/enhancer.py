"""``Enhancer`` -- host-side mirror of ``opencood/models/gencomm_modules/enhancer.py:359-383``.

The full parameter tree of the reference is kept (58 ``state_dict`` keys for C=128: ``block_{1,2,3}``
each with ``attn`` / ``mlp`` / ``norm1`` / ``norm2``, and ``split_attn``) so that checkpoints load
with identical keys, but like the reference's live code only ``block_1.{norm1,norm2,mlp.*}`` and
``split_attn.*`` are ever read (``enhancer.py:352`` comments the attention out, ``:377-380`` only
calls ``block_1``). ``forward`` runs HIP kernels via ``gencomm_enhancer_fwd``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .runtime import PackedParams, f32c, ptr, record_len_list, require_gpu, stream_ptr, workspaces


class LinearProjection(nn.Module):  # enhancer.py:42-60 (parameters only)
    def __init__(self, dim, heads=8, dim_head=64, dropout=0.0, bias=True):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.to_q = nn.Linear(dim, inner_dim, bias=bias)
        self.to_kv = nn.Linear(dim, inner_dim * 2, bias=bias)
        self.dim, self.inner_dim = dim, inner_dim


class Attention(nn.Module):  # enhancer.py:88-107 (parameters only; never executed by the reference)
    def __init__(self, dim, num_heads, token_projection="linear", qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.angle_bins = 5
        self.angle_bias_table = nn.Parameter(torch.ones(self.angle_bins, num_heads))
        self.qkv = LinearProjection(dim, num_heads, dim // num_heads, bias=qkv_bias)
        self.token_projection = token_projection
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.softmax = nn.Softmax(dim=-1)


class FRFN(nn.Module):  # enhancer.py:207-220
    def __init__(self, dim=32, hidden_dim=128, act_layer=nn.GELU, drop=0.0, use_eca=False):
        super().__init__()
        self.linear1 = nn.Sequential(nn.Linear(dim, hidden_dim * 2), act_layer())
        self.dwconv = nn.Sequential(nn.Conv2d(hidden_dim, hidden_dim, groups=hidden_dim, kernel_size=3, stride=1, padding=1), act_layer())
        self.linear2 = nn.Sequential(nn.Linear(hidden_dim, dim))
        self.dim, self.hidden_dim = dim, hidden_dim
        self.dim_conv = self.dim // 4
        self.dim_untouched = self.dim - self.dim_conv
        self.partial_conv3 = nn.Conv2d(self.dim_conv, self.dim_conv, 3, 1, 1, bias=False)


class SplitAttn(nn.Module):  # enhancer.py:302-313
    def __init__(self, input_dim):
        super().__init__()
        self.input_dim = input_dim
        self.fc1 = nn.Linear(input_dim, input_dim, bias=False)
        self.bn1 = nn.LayerNorm(input_dim)
        self.act1 = nn.ReLU()
        self.fc2 = nn.Linear(input_dim, input_dim, bias=False)


class Enhancer_block(nn.Module):  # enhancer.py:335-344
    def __init__(self, C, win_size, num_heads):
        super().__init__()
        self.window_size = win_size
        self.attn = Attention(dim=C, num_heads=num_heads, qkv_bias=True, qk_scale=None, attn_drop=0.1, proj_drop=0.1, token_projection="linear")
        self.mlp = FRFN(dim=C, hidden_dim=C * 2, act_layer=nn.GELU, drop=0.0)
        self.norm1 = nn.LayerNorm(C)
        self.norm2 = nn.LayerNorm(C)
        self.drop_path = nn.Identity()


class Enhancer(nn.Module):
    def __init__(self, C, win_size, num_heads):
        super().__init__()
        self.C = C
        self.block_1 = Enhancer_block(C, [4, 4], num_heads)
        self.block_2 = Enhancer_block(C, win_size, num_heads)
        self.block_3 = Enhancer_block(C, [16, 16], num_heads)
        self.split_attn = SplitAttn(C)
        self._packed = None

    def _raw_params(self, device: torch.device) -> torch.Tensor:
        if self._packed is None:
            table = _lib.enhancer_param_table(self.C)
            self._packed = PackedParams(table, _lib.check_size(_lib.lib().gencomm_enhancer_raw_floats(self.C), "gencomm_enhancer_raw_floats"))
        self._packed.update(dict(self.named_parameters()))
        require_gpu(self._packed.flat, "Enhancer parameters")
        return self._packed.flat

    def forward(self, x, affine_matrix=None, record_len=None):
        """x [sumN,C,H,W] -> [sumN,C,H,W]. ``affine_matrix`` is accepted and ignored exactly like
        the reference's live code (enhancer.py:375 slices it, nothing reads the slice); agents are
        processed independently so ``record_len`` only serves as a consistency check."""
        require_gpu(x, "Enhancer.forward")
        lens = record_len_list(record_len)
        n, C, H, W = x.shape
        if C != self.C:
            raise ValueError(f"Enhancer built for C={self.C}, got {C}")
        if lens is not None and sum(lens) != n:
            raise ValueError(f"record_len sums to {sum(lens)} but x has {n} agents")
        x = f32c(x)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            from .autograd import EnhancerFunction  # HIP forward, recompute-based backward
            return EnhancerFunction.apply(self, x, *list(self.parameters()))
        return self._forward_hip(x)

    def _forward_hip(self, x: torch.Tensor) -> torch.Tensor:
        n, C, H, W = x.shape
        l = _lib.lib()
        raw = self._raw_params(x.device)
        ws = workspaces.get(x.device, _lib.check_size(l.gencomm_enhancer_workspace_bytes(n, C, H, W), "gencomm_enhancer_workspace_bytes"), "enhancer")
        out = torch.empty_like(x)
        _lib.check(l.gencomm_enhancer_fwd(ptr(raw), ptr(x), ptr(out), n, C, H, W, ptr(ws), ws.numel(), stream_ptr(x.device)),
                   "gencomm_enhancer_fwd")
        return out
