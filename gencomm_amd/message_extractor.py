"""``MessageExtractorv2`` -- host-side mirror of
``opencood/models/gencomm_modules/message_extractor_v2.py:70-120``: the module that
turns an agent's BEV feature into the 2-channel spatial message conditioning GenComm. The only part of
the model the reference re-trains per new agent type (stage 2).

Same constructor and ``state_dict`` keys (``bev_extractor.{offset1,dcn1,fuse.0,fuse.2,attn.1,attn.3}``);
``DeformConv2d`` below is a parameter container with torchvision's parameter names and default
initialisation -- the deformable convolution itself runs in HIP (``gencomm_msgext_fwd``), so the
torchvision dependency of the reference (``message_extractor_v2.py:67``) is gone.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib
from .runtime import PackedParams, f32c, ptr, require_gpu, stream_ptr, workspaces


class DeformConv2d(nn.Module):
    """Parameters of torchvision.ops.DeformConv2d(in, out, kernel_size=3, padding=1): ``weight``
    [out, in, 3, 3] and ``bias`` [out], kaiming-uniform(a=sqrt(5)) / uniform(+-1/sqrt(fan_in))."""

    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1):
        super().__init__()
        if kernel_size != 3 or padding != 1:
            raise NotImplementedError("only the 3x3 / padding 1 deformable convolution of MessageExtractorv2 is supported")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_channels * 9)
        nn.init.uniform_(self.bias, -bound, bound)


class BEVDeformableExtractor(nn.Module):
    def __init__(self, in_channels=128, out_channels=2):
        super().__init__()
        if out_channels != 2:
            raise NotImplementedError("the HIP message extractor produces the reference's 2-channel message")
        self.in_channels = in_channels
        self.offset1 = nn.Conv2d(in_channels, 18, kernel_size=3, padding=1)
        self.dcn1 = DeformConv2d(in_channels, 64, kernel_size=3, padding=1)
        self.fuse = nn.Sequential(nn.Conv2d(64, 64, kernel_size=1), nn.ReLU(), nn.Conv2d(64, out_channels, kernel_size=1))
        self.attn = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(64, 32, kernel_size=1), nn.ReLU(),
                                  nn.Conv2d(32, 64, kernel_size=1), nn.Sigmoid())


class MessageExtractorv2(nn.Module):
    def __init__(self, in_channels=128, out_channels=2):
        super().__init__()
        self.bev_extractor = BEVDeformableExtractor(in_channels, out_channels)
        self._packed = None

    def forward(self, bev_feature: torch.Tensor) -> torch.Tensor:
        """[n, C, H, W] -> [n, 2, H, W]."""
        require_gpu(bev_feature, "MessageExtractorv2.forward")
        if torch.is_grad_enabled() and (bev_feature.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("gencomm_amd.MessageExtractorv2: backward is not implemented yet; call under torch.no_grad()")
        x = f32c(bev_feature)
        n, C, H, W = x.shape
        if C != self.bev_extractor.in_channels:
            raise ValueError(f"built for {self.bev_extractor.in_channels} input channels, got {C}")
        l = _lib.lib()
        if self._packed is None:
            self._packed = PackedParams(_lib.msgext_param_table(C), _lib.check_size(l.gencomm_msgext_raw_floats(C), "gencomm_msgext_raw_floats"))
        self._packed.update(dict(self.named_parameters()))
        ws = workspaces.get(x.device, _lib.check_size(l.gencomm_msgext_workspace_bytes(n, C, H, W), "gencomm_msgext_workspace_bytes"), "msgext")
        out = torch.empty((n, 2, H, W), dtype=torch.float32, device=x.device)
        _lib.check(l.gencomm_msgext_fwd(ptr(self._packed.flat), ptr(x), ptr(out), n, C, H, W, ptr(ws), ws.numel(), stream_ptr(x.device)),
                   "gencomm_msgext_fwd")
        return out
