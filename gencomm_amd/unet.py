"""``DiffusionUNet`` -- host-side mirror of the reference denoiser
(``opencood/models/gencomm_modules/unet.py:198-344``).

The module owns the parameters under the reference's ``state_dict`` key names (``temb.dense.{0,1}``,
``conv_in``, ``down.{i}.block.{j}.{norm1,conv1,temb_proj,norm2,conv2}``, ``down.{i}.downsample.conv``,
``mid.block_{1,2}``, ``up.{i}.block.{j}.{...,nin_shortcut}``, ``up.{i}.upsample.conv``, ``norm_out``,
``conv_out``); ``torch.nn`` layer classes are used purely as parameter containers with the same
default initialisation. ``forward`` never calls them: it packs the parameters once per weight
version and runs the hand-written HIP kernels through the C ABI.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .runtime import PackedParams, f32c, ptr, require_gpu, stream_ptr, workspaces


def _cfg_get(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


def Normalize(in_channels: int) -> nn.GroupNorm:  # unet.py:36-37
    return nn.GroupNorm(num_groups=4, num_channels=in_channels, eps=1e-6, affine=True)


class Upsample(nn.Module):  # unet.py:40-56
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)


class Downsample(nn.Module):  # unet.py:59-78
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)


class ResnetBlock(nn.Module):  # unet.py:81-138
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.temb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if in_channels != out_channels:
            if conv_shortcut:
                self.conv_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
            else:
                self.nin_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)


class AttnBlock(nn.Module):  # unet.py:141-193 (no shipped config instantiates it; runs as flash-style HIP attention)
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1)


class DiffusionUNet(nn.Module):
    """Constructor mirrors unet.py:198-305 (``config.model.*`` attribute access)."""

    def __init__(self, config):
        super().__init__()
        m = _cfg_get(config, "model")
        ch, out_ch, ch_mult = _cfg_get(m, "ch"), _cfg_get(m, "out_ch"), tuple(_cfg_get(m, "ch_mult"))
        num_res_blocks = _cfg_get(m, "num_res_blocks")
        attn_resolutions = _cfg_get(m, "attn_resolutions")
        dropout = _cfg_get(m, "dropout")
        in_channels = _cfg_get(m, "in_channels") + 2  # two message channels, unet.py:210
        resolution = 128  # nominal, hard-coded in the reference (unet.py:211)
        resamp_with_conv = _cfg_get(m, "resamp_with_conv")

        self.config = config
        self.ch, self.temb_ch = ch, ch * 4
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.out_ch = out_ch
        self.ch_mult = ch_mult
        self.dropout_p = dropout
        self.resamp_with_conv = resamp_with_conv

        self.temb = nn.Module()
        self.temb.dense = nn.ModuleList([nn.Linear(self.ch, self.temb_ch), nn.Linear(self.temb_ch, self.temb_ch)])
        self.conv_in = nn.Conv2d(in_channels, self.ch, kernel_size=3, stride=1, padding=1)

        curr_res = resolution
        in_ch_mult = (1,) + ch_mult
        self.down = nn.ModuleList()
        block_in = None
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in = ch * in_ch_mult[i_level]
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = curr_res // 2
            self.down.append(down)

        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)

        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            skip_in = ch * ch_mult[i_level]
            for i_block in range(num_res_blocks + 1):
                if i_block == num_res_blocks:
                    skip_in = ch * in_ch_mult[i_level]
                block.append(ResnetBlock(in_channels=block_in + skip_in, out_channels=block_out, temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            self.up.insert(0, up)

        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)

        self._packed = None
        self._prepared = None
        self._prepared_key = None

    # ------------------------------------------------------------------ HIP path
    def _check_supported(self) -> None:
        why = None
        if self.ch != 8 or any(m != 1 for m in self.ch_mult):
            why = f"ch={self.ch}, ch_mult={self.ch_mult}: the HIP kernels cover ch=8 with ch_mult all ones"
        elif not self.resamp_with_conv:
            why = "resamp_with_conv=False is not supported"
        elif self.training and self.dropout_p != 0.0:
            why = "dropout > 0 in training mode is not supported"
        elif self.out_ch != self.in_channels - 2:
            why = "out_ch must equal in_channels"
        if why:
            raise NotImplementedError("gencomm_amd.DiffusionUNet: " + why + " (every shipped GenComm yaml uses ch 8, ch_mult [1,1], attn_resolutions [16])")

    @property
    def feature_channels(self) -> int:
        return self.in_channels - 2

    @property
    def attn_mask(self) -> int:
        """bit l = level l carries AttnBlocks (down and up paths agree by construction, unet.py:252,:286)."""
        return sum(1 << l for l in range(self.num_resolutions) if len(self.down[l].attn) > 0)

    def prepared_params(self, T: int, device: torch.device) -> torch.Tensor:
        """Device blob in kernel layout + the [block][t][8] timestep-bias tables for t < T."""
        self._check_supported()
        l = _lib.lib()
        C, L, R, A = self.feature_channels, self.num_resolutions, self.num_res_blocks, self.attn_mask
        if self._packed is None:
            table = _lib.unet_param_table(C, L, R, A)
            self._packed = PackedParams(table, _lib.check_size(l.gencomm_unet_raw_floats(C, L, R, A), "gencomm_unet_raw_floats"))
        named = dict(self.named_parameters())
        for p in named.values():
            require_gpu(p, "DiffusionUNet parameters")
            break
        changed = self._packed.update(named)
        key = (self._packed.generation, T, str(device))
        if changed or self._prepared is None or self._prepared_key != key:
            nflt = _lib.check_size(l.gencomm_unet_prepared_floats(C, L, R, A, T), "gencomm_unet_prepared_floats")
            prepared = torch.zeros(nflt, dtype=torch.float32, device=device)
            _lib.check(l.gencomm_unet_prepare(ptr(self._packed.flat), ptr(prepared), C, L, R, A, T, stream_ptr(device)),
                       "gencomm_unet_prepare")
            self._prepared, self._prepared_key = prepared, key
        return self._prepared

    def denoise_workspace(self, n: int, H: int, W: int, device: torch.device) -> torch.Tensor:
        nbytes = _lib.check_size(_lib.lib().gencomm_denoise_workspace_bytes(
            n, self.feature_channels, H, W, self.num_resolutions, self.num_res_blocks, self.attn_mask), "gencomm_denoise_workspace_bytes")
        return workspaces.get(device, nbytes, "unet")

    def forward(self, x: torch.Tensor, t: torch.Tensor, T: int = None) -> torch.Tensor:
        """x = cat[cond (2 ch), x_t (C ch)] [n, C+2, H, W]; t [n] (all entries equal, as every
        caller in the reference passes, cond_diff.py:327). Returns x0_hat [n, C, H, W]."""
        require_gpu(x, "DiffusionUNet.forward")
        n, cin, H, W = x.shape
        C = self.feature_channels
        if cin != C + 2:
            raise ValueError(f"expected {C + 2} input channels, got {cin}")
        tv = t.detach().reshape(-1)
        t_int = int(round(float(tv[0].item())))
        if tv.numel() > 1 and not bool((tv == tv[0]).all().item()):
            raise NotImplementedError("per-sample timesteps are not supported (the reference never uses them)")
        T = max(T or 0, t_int + 1)
        x = f32c(x)
        cond, x_t = x[:, :2].contiguous(), x[:, 2:].contiguous()
        prepared = self.prepared_params(T, x.device)
        ws = self.denoise_workspace(n, H, W, x.device)
        out = torch.empty((n, C, H, W), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().gencomm_unet_fwd(ptr(prepared), ptr(x_t), ptr(cond), ptr(out), t_int, n, C, H, W,
                                               self.num_resolutions, self.num_res_blocks, self.attn_mask, T,
                                               ptr(ws), ws.numel(), stream_ptr(x.device)), "gencomm_unet_fwd")
        return out
