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


def _deform_conv3x3_torch(x, offset, weight, bias):
    """Differentiable restatement of the 3x3 / padding-1 deformable convolution (DCNv1 as torchvision defines it: offset
    channel 2k / 2k+1 = vertical / horizontal displacement of tap k, bilinear sampling, zero outside the map) with torch
    gathers -- used ONLY to differentiate: the forward value always comes from the HIP kernel."""
    n, C, H, W = x.shape
    O = weight.shape[0]
    ys = torch.arange(H, dtype=x.dtype, device=x.device).view(1, H, 1)
    xs = torch.arange(W, dtype=x.dtype, device=x.device).view(1, 1, W)
    xf = x.reshape(n, C, H * W)
    out = x.new_zeros(n, O, H, W)
    wk = weight.reshape(O, C, 9)
    for k in range(9):
        py = ys - 1 + (k // 3) + offset[:, 2 * k]
        px = xs - 1 + (k % 3) + offset[:, 2 * k + 1]
        inside = (py > -1) & (py < H) & (px > -1) & (px < W)
        y0, x0 = torch.floor(py), torch.floor(px)
        ly, lx = py - y0, px - x0
        val = 0
        for yy, xx, wgt in ((y0, x0, (1 - ly) * (1 - lx)), (y0, x0 + 1, (1 - ly) * lx), (y0 + 1, x0, ly * (1 - lx)), (y0 + 1, x0 + 1, ly * lx)):
            ok = (inside & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)).to(x.dtype)
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long().view(n, 1, H * W).expand(n, C, H * W)
            val = val + torch.gather(xf, 2, idx).view(n, C, H, W) * (wgt * ok).unsqueeze(1)
        out = out + torch.einsum("oc,nchw->nohw", wk[:, :, k], val)
    return out + bias.view(1, O, 1, 1)


def _extractor_torch(x, ow, ob, dw, db, f0w, f0b, f2w, f2b, a1w, a1b, a3w, a3b):
    """BEVDeformableExtractor.forward (message_extractor_v2.py:103-118) in differentiable torch ops (backward only)."""
    import torch.nn.functional as F
    off = F.conv2d(x, ow, ob, padding=1)
    b1 = _deform_conv3x3_torch(x, off, dw, db)
    g = b1.mean((2, 3), keepdim=True)
    g = torch.sigmoid(F.conv2d(F.relu(F.conv2d(g, a1w, a1b)), a3w, a3b))
    h = F.relu(F.conv2d(b1 * g, f0w, f0b))
    return F.conv2d(h, f2w, f2b)


class _MsgExtFn(torch.autograd.Function):
    """HIP forward (gencomm_msgext_fwd); the backward re-evaluates the extractor with differentiable torch ops on the GPU from
    the saved input (activation-checkpoint style) -- stage 2 of the reference trains exactly this module (stage2.py:99-101),
    with every other module frozen, so its gradients must exist; dedicated HIP backward kernels are the next step."""

    @staticmethod
    def forward(ctx, mod, x, *params):
        ctx.mod = mod
        ctx.save_for_backward(x, *params)
        with torch.no_grad():
            return mod._forward_hip(x)

    @staticmethod
    def backward(ctx, gy):
        x, *params = ctx.saved_tensors
        with torch.enable_grad():
            xd = x.detach().float().requires_grad_(ctx.needs_input_grad[1])
            pd = [p.detach().float().requires_grad_(need) for p, need in zip(params, ctx.needs_input_grad[2:])]
            y = _extractor_torch(xd, *pd)
            wanted = ([xd] if ctx.needs_input_grad[1] else []) + [p for p in pd if p.requires_grad]
            grads = list(torch.autograd.grad(y, wanted, gy.contiguous().float(), allow_unused=True))
        gx = grads.pop(0) if ctx.needs_input_grad[1] else None
        gp = [grads.pop(0) if need else None for need in ctx.needs_input_grad[2:]]
        return (None, gx, *gp)


class MessageExtractorv2(nn.Module):
    def __init__(self, in_channels=128, out_channels=2):
        super().__init__()
        self.bev_extractor = BEVDeformableExtractor(in_channels, out_channels)
        self._packed = None

    def _param_list(self):
        e = self.bev_extractor
        return [e.offset1.weight, e.offset1.bias, e.dcn1.weight, e.dcn1.bias, e.fuse[0].weight, e.fuse[0].bias, e.fuse[2].weight, e.fuse[2].bias,
                e.attn[1].weight, e.attn[1].bias, e.attn[3].weight, e.attn[3].bias]

    def forward(self, bev_feature: torch.Tensor) -> torch.Tensor:
        """[n, C, H, W] -> [n, 2, H, W]."""
        require_gpu(bev_feature, "MessageExtractorv2.forward")
        if torch.is_grad_enabled() and (bev_feature.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _MsgExtFn.apply(self, bev_feature, *self._param_list())
        return self._forward_hip(bev_feature)

    def _forward_hip(self, bev_feature: torch.Tensor) -> torch.Tensor:
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
