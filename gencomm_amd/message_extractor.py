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


class _MsgExtFn(torch.autograd.Function):
    """HIP forward (gencomm_msgext_fwd, the fused inference kernels) and a backward composed of HIP primitives
    (gencomm_amd/train_ops.py) -- stage 2 of the reference trains exactly this module (stage2.py:99-101) with every other
    module frozen. The deformable convolution is split into its sampling half (gencomm_dcn_sample_fwd) and its GEMM half, so
    its backward is two GEMMs on the general kernels (d weight on the split-K MFMA kernel, d columns) plus one scatter kernel
    (gencomm_dcn_scatter_bwd: d input by atomics, d offsets); the offset convolution and the two 1x1 layers use the general
    convolution's dgrad / wgrad. Elementwise products / sums and the squeeze-excitation gate's MLP on [n, 64] vectors are plain
    torch tensor arithmetic, as in EnhancerFunction."""

    @staticmethod
    def forward(ctx, mod, x, *params):
        ctx.mod = mod
        ctx.save_for_backward(x)
        with torch.no_grad():
            return mod._forward_hip(x)

    @staticmethod
    def backward(ctx, gy):
        import torch.nn.functional as F
        from . import train_ops as T
        (x,) = ctx.saved_tensors
        e_ = ctx.mod.bev_extractor
        ow, ob, dw, db = e_.offset1.weight, e_.offset1.bias, e_.dcn1.weight, e_.dcn1.bias
        f0, f2, a1, a3 = e_.fuse[0], e_.fuse[2], e_.attn[1], e_.attn[3]
        n, C, H, W = x.shape
        HW = H * W
        with torch.no_grad():
            x = x.detach().float().contiguous()
            # ---- forward recompute with every intermediate kept (message_extractor_v2.py:103-118)
            off = T.conv2d(x, ow, ob, 1)
            col = T.dcn_sample(x, off)                                               # [n, 9 C, H, W]
            wd2 = dw.detach().reshape(64, C * 9)[:, :, None, None]
            b1 = T.conv2d(col, wd2, db, 0)                                           # deformable conv = 1x1 GEMM over the samples
            w0 = f0.weight.detach()
            w2 = f2.weight.detach()
        # ---- squeeze-excitation gate on [n, 64] vectors: torch autograd on a few hundred numbers
        gate_params = [a1.weight, a1.bias, a3.weight, a3.bias]
        with torch.enable_grad():
            mean = b1.mean((2, 3), keepdim=True).requires_grad_(True)
            local = [p.detach().requires_grad_(True) for p in gate_params]
            # the two 1x1 convolutions on a 1x1 map are Linear layers on [n, 64] vectors: plain matrix products (a framework
            # convolution here would be the only MIOpen call of the package)
            hid = F.relu(F.linear(mean.flatten(1), local[0].flatten(1), local[1]))
            gate = torch.sigmoid(F.linear(hid, local[2].flatten(1), local[3]))[:, :, None, None]
        with torch.no_grad():
            g = gate.detach()
            e = b1 * g
            hpre = T.conv2d(e, w0, f0.bias, 0)
            h = torch.relu(hpre)
            go = gy.float().contiguous()
            # ---- fuse: 1x1 64 -> 2, ReLU, 1x1 64 -> 64
            dW2, db2 = T.conv2d_wgrad(go, h, 1, 0, True)
            dh = T.conv2d(go, w2.transpose(0, 1).contiguous(), None, 0) * (hpre > 0)
            dW0, db0 = T.conv2d_wgrad(dh, e, 1, 0, True)
            de = T.conv2d(dh, w0.transpose(0, 1).contiguous(), None, 0)
            dgate = (de * b1).sum((2, 3), keepdim=True)
        dmean, *dgp = torch.autograd.grad(gate, [mean] + local, dgate)
        with torch.no_grad():
            db1 = de * g + dmean / HW
            # ---- deformable convolution: GEMM half, then the sampling half
            dWd, dbd = T.conv2d_wgrad(db1, col, 1, 0, True)                          # [64, 9 C, 1, 1]
            dcol = T.conv2d(db1, wd2.transpose(0, 1).contiguous(), None, 0)          # [n, 9 C, H, W]
            dx, doff = T.dcn_scatter_bwd(x, off, dcol)
            # ---- offset convolution
            dWo, dbo = T.conv2d_wgrad(doff, x, 3, 1, True)
            dx = dx + T.conv2d_dgrad(doff, ow, 1)
        grads = [dWo, dbo, dWd.reshape(64, C, 3, 3), dbd, dW0, db0, dW2, db2] + list(dgp)
        gp = [gr if need else None for gr, need in zip(grads, ctx.needs_input_grad[2:])]
        return (None, dx if ctx.needs_input_grad[1] else None, *gp)


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
