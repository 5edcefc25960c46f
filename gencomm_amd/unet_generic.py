"""General-width ``DiffusionUNet`` -- the reference's denoiser for ANY ``ch`` / ``ch_mult`` / ``num_res_blocks``
(``opencood/models/gencomm_modules/unet.py:198-344``), correct first.

The accelerated path (``gencomm_amd/unet.py``: hand-fused 8-channel kernels, ``ch = 8`` with ``ch_mult`` all ones -- every shipped yaml)
covers one point of the family the reference's constructor spans.  Everything else runs here: the same module tree under the same
``state_dict`` keys (``temb.dense.{0,1}``, ``conv_in``, ``down.{i}.block.{j}.{norm1,conv1,temb_proj,norm2,conv2[,nin_shortcut]}``,
``down.{i}.downsample.conv``, ``mid.block_{1,2}``, ``up.{i}.block.{j}``, ``up.{i}.upsample.conv``, ``norm_out``, ``conv_out``), and a
forward composed layer by layer from the library's general primitives -- exact-fp32 implicit-GEMM convolutions
(``gencomm_conv2d_fwd`` / ``_act_res_fwd``), ``gencomm_gn_nchw_fwd`` (GroupNorm + SiLU for any width), channel-slice copies -- with no
torch convolution or normalisation call.  What stays in the framework: the sinusoidal embedding and the timestep MLP on [1, 4 ch]
vectors, the zero pad of the stride-2 Downsample and the nearest-neighbour doubling of the Upsample (data movement).

Inference only, no AttnBlocks (``attn_resolutions`` that would instantiate one raise), one timestep per call (every caller of the
reference passes equal entries, cond_diff.py:327).  It is an order of magnitude slower than the fused path per unit of work and says so;
its purpose is that a config outside the shipped family loads, runs and matches the reference (tests/golden/unet_wide.npz).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import train_ops as T
from .runtime import f32c, ptr, require_gpu, stream_ptr


def _norm(c: int) -> nn.GroupNorm:
    return nn.GroupNorm(num_groups=4, num_channels=c, eps=1e-6, affine=True)   # unet.py:36-37


class _Res(nn.Module):
    """Parameter container of one ResnetBlock (unet.py:81-118); the arithmetic lives in GenericDiffusionUNet._res."""

    def __init__(self, cin: int, cout: int, temb_ch: int):
        super().__init__()
        self.norm1 = _norm(cin)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.temb_proj = nn.Linear(temb_ch, cout)
        self.norm2 = _norm(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1, 1, 0)


class _Resample(nn.Module):
    def __init__(self, c: int, stride: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride, 1 if stride == 1 else 0)


def _level(blocks) -> nn.Module:
    m = nn.Module()
    m.block = nn.ModuleList(blocks)
    m.attn = nn.ModuleList()
    return m


class GenericDiffusionUNet(nn.Module):
    def __init__(self, ch: int, out_ch: int, ch_mult, num_res_blocks: int, in_channels: int, resamp_with_conv: bool, attn_resolutions):
        super().__init__()
        if not resamp_with_conv:
            raise NotImplementedError("gencomm_amd.DiffusionUNet: resamp_with_conv=False is not supported")
        if ch % 4 or any((ch * m) % 4 for m in ch_mult):
            raise ValueError("GroupNorm(4 groups) needs channel counts that are multiples of 4 (unet.py:36-37)")
        self.ch, self.temb_ch, self.ch_mult, self.out_ch = ch, 4 * ch, tuple(ch_mult), out_ch
        if out_ch != in_channels - 2:
            raise NotImplementedError("gencomm_amd.DiffusionUNet: out_ch must equal in_channels")
        self.num_resolutions, self.num_res_blocks = len(self.ch_mult), num_res_blocks
        self.in_channels = in_channels
        res, L = 128, self.num_resolutions
        if any((res >> l) in list(attn_resolutions) for l in range(L)):
            raise NotImplementedError("gencomm_amd.DiffusionUNet: AttnBlocks are implemented for the accelerated family only (ch 8, ch_mult all ones)")
        self.temb = nn.Module()
        self.temb.dense = nn.ModuleList([nn.Linear(ch, self.temb_ch), nn.Linear(self.temb_ch, self.temb_ch)])
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        in_mult = (1,) + self.ch_mult
        self.down = nn.ModuleList()
        cur = ch
        for l in range(L):
            cur, cout = ch * in_mult[l], ch * self.ch_mult[l]
            blocks = []
            for _ in range(num_res_blocks):
                blocks.append(_Res(cur, cout, self.temb_ch))
                cur = cout
            lvl = _level(blocks)
            if l != L - 1:
                lvl.downsample = _Resample(cur, 2)
            self.down.append(lvl)
        self.mid = nn.Module()
        self.mid.block_1 = _Res(cur, cur, self.temb_ch)
        self.mid.block_2 = _Res(cur, cur, self.temb_ch)
        ups = []
        for l in reversed(range(L)):
            cout, skip = ch * self.ch_mult[l], ch * self.ch_mult[l]
            blocks = []
            for b in range(num_res_blocks + 1):
                if b == num_res_blocks:
                    skip = ch * in_mult[l]
                blocks.append(_Res(cur + skip, cout, self.temb_ch))
                cur = cout
            lvl = _level(blocks)
            if l != 0:
                lvl.upsample = _Resample(cur, 1)
            ups.insert(0, lvl)
        self.up = nn.ModuleList(ups)
        self.norm_out = _norm(cur)
        self.conv_out = nn.Conv2d(cur, out_ch, 3, 1, 1)

    @property
    def feature_channels(self) -> int:
        return self.in_channels - 2

    # ------------------------------------------------------------------ primitives
    @staticmethod
    def _gn_silu(x: torch.Tensor, norm: nn.GroupNorm) -> torch.Tensor:
        n, C, H, W = x.shape
        y = torch.empty_like(x)
        stat = torch.empty(n * norm.num_groups * 2, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().gencomm_gn_nchw_fwd(ptr(x), ptr(f32c(norm.weight.detach())), ptr(f32c(norm.bias.detach())), ptr(y), ptr(stat),
                                                  float(norm.eps), 1, n, C, norm.num_groups, H * W, stream_ptr(x.device)), "gencomm_gn_nchw_fwd")
        return y

    def _res(self, blk: _Res, x: torch.Tensor, temb_act: torch.Tensor) -> torch.Tensor:
        # h = conv1(SiLU(GN1(x))) + temb_proj(SiLU(temb)): one timestep for the whole batch, so the projection is a per-channel bias
        bias1 = blk.conv1.bias.detach().float() + F.linear(temb_act, blk.temb_proj.weight.detach().float(), blk.temb_proj.bias.detach().float())[0]
        h = T.conv2d(self._gn_silu(x, blk.norm1), blk.conv1.weight, bias1, 1)
        skip = T.conv2d(x, blk.nin_shortcut.weight, blk.nin_shortcut.bias, 0) if hasattr(blk, "nin_shortcut") else x
        return T.conv2d(self._gn_silu(h, blk.norm2), blk.conv2.weight, blk.conv2.bias, 1, residual=skip)   # x + h (unet.py:138)

    @staticmethod
    def _cat(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        n, ca, H, W = a.shape
        out = torch.empty(n, ca + b.shape[1], H, W, dtype=torch.float32, device=a.device)
        T.copy_slice(a, 0, ca, out, 0)
        T.copy_slice(b, 0, b.shape[1], out, ca)
        return out

    # ------------------------------------------------------------------ forward (unet.py:307-344)
    def forward(self, x: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        require_gpu(x, "DiffusionUNet.forward")
        tv = t.detach().reshape(-1).float()
        if tv.numel() > 1 and not bool((tv == tv[0]).all().item()):
            raise NotImplementedError("per-sample timesteps are not supported (the reference never uses them)")
        x = f32c(x)
        with torch.no_grad():
            half = self.ch // 2                      # get_timestep_embedding, unet.py:10-28
            freq = torch.exp(torch.arange(half, dtype=torch.float32, device=x.device) * -(math.log(10000) / (half - 1)))
            e = tv[:1, None] * freq[None, :]
            emb = torch.cat([torch.sin(e), torch.cos(e)], dim=1)
            if self.ch % 2 == 1:
                emb = F.pad(emb, (0, 1, 0, 0))
            d0, d1 = self.temb.dense
            temb = F.linear(F.silu(F.linear(emb, d0.weight.float(), d0.bias.float())), d1.weight.float(), d1.bias.float())
            temb_act = F.silu(temb)                  # every block applies the nonlinearity before its projection (unet.py:124)
            hs = [T.conv2d(x, self.conv_in.weight, self.conv_in.bias, 1)]
            L = self.num_resolutions
            for l in range(L):
                for b in range(self.num_res_blocks):
                    hs.append(self._res(self.down[l].block[b], hs[-1], temb_act))
                if l != L - 1:                       # Downsample: zero pad right / bottom by one, 3x3 stride 2 pad 0 (unet.py:71-75)
                    ds = self.down[l].downsample.conv
                    hs.append(T.conv2d(F.pad(hs[-1], (0, 1, 0, 1)), ds.weight, ds.bias, 0, stride=2))
            h = self._res(self.mid.block_1, hs[-1], temb_act)
            h = self._res(self.mid.block_2, h, temb_act)
            for l in reversed(range(L)):
                for b in range(self.num_res_blocks + 1):
                    h = self._res(self.up[l].block[b], self._cat(h, hs.pop()), temb_act)
                if l != 0:                           # Upsample: nearest x2, 3x3 (unet.py:51-56)
                    us = self.up[l].upsample.conv
                    h = T.conv2d(h.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3), us.weight, us.bias, 1)
            return T.conv2d(self._gn_silu(h, self.norm_out), self.conv_out.weight, self.conv_out.bias, 1)
