"""Training support for the hot path: HIP forward, gradients by differentiable recomputation.

Forward always runs the hand-written HIP kernels. For backward, this round re-evaluates the same
maths with differentiable torch ops ON THE GPU (activation-checkpoint style: nothing but the
inputs and the explicit noise is saved) and lets autograd produce the gradients -- the first
pass SURVEY.md section 7 step 6 plans before dedicated HIP backward kernels. Nothing here touches
the CPU or the oracle. Gradients reach what the reference's do (training-branch row):
the UNet weights, every row of ``conditions`` and -- through the ego repeat -- the ego rows of
``spatial_features``; for the Enhancer and AttFusion the usual dense gradients.

The functional forms below follow the reference line by line:
  UNet            opencood/models/gencomm_modules/unet.py:307-344, :119-138, :71-75, :51-56
  sampler         opencood/models/gencomm_modules/cond_diff.py:262-264, :272-315, :342-360
  Enhancer        opencood/models/gencomm_modules/enhancer.py:346-357, :222-250, :315-333
  warp + fusion   opencood/models/fuse_modules/fusion_in_one.py:131-151, torch_transformation_utils.py:323-332
"""
from __future__ import annotations

import math
from typing import List, Sequence

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------- UNet
def _silu(x):
    return x * torch.sigmoid(x)


def _resblock(blk, x, temb):
    h = F.conv2d(_silu(F.group_norm(x, 4, blk.norm1.weight, blk.norm1.bias, 1e-6)), blk.conv1.weight, blk.conv1.bias, padding=1)
    h = h + F.linear(_silu(temb), blk.temb_proj.weight, blk.temb_proj.bias)[:, :, None, None]
    h = F.conv2d(_silu(F.group_norm(h, 4, blk.norm2.weight, blk.norm2.bias, 1e-6)), blk.conv2.weight, blk.conv2.bias, padding=1)
    if hasattr(blk, "nin_shortcut"):
        x = F.conv2d(x, blk.nin_shortcut.weight, blk.nin_shortcut.bias)
    return x + h


def _attnblock(ab, x):
    h = F.group_norm(x, 4, ab.norm.weight, ab.norm.bias, 1e-6)
    q, k, v = (F.conv2d(h, m.weight, m.bias) for m in (ab.q, ab.k, ab.v))
    b, c, hh, ww = q.shape
    w_ = torch.bmm(q.reshape(b, c, hh * ww).permute(0, 2, 1), k.reshape(b, c, hh * ww)) * (int(c) ** (-0.5))
    w_ = F.softmax(w_, dim=2)
    o = torch.bmm(v.reshape(b, c, hh * ww), w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + F.conv2d(o, ab.proj_out.weight, ab.proj_out.bias)


def unet_forward(unet, x, t_int: int):
    half = unet.ch // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32, device=x.device) * -(math.log(10000) / (half - 1)))
    ang = float(t_int) * freq
    temb = torch.cat([torch.sin(ang), torch.cos(ang)])[None, :].expand(x.shape[0], -1)
    temb = F.linear(temb, unet.temb.dense[0].weight, unet.temb.dense[0].bias)
    temb = F.linear(_silu(temb), unet.temb.dense[1].weight, unet.temb.dense[1].bias)
    hs = [F.conv2d(x, unet.conv_in.weight, unet.conv_in.bias, padding=1)]
    L = unet.num_resolutions
    for lvl in range(L):
        for i, blk in enumerate(unet.down[lvl].block):
            h = _resblock(blk, hs[-1], temb)
            if len(unet.down[lvl].attn) > 0:
                h = _attnblock(unet.down[lvl].attn[i], h)
            hs.append(h)
        if lvl != L - 1:
            c = unet.down[lvl].downsample.conv
            hs.append(F.conv2d(F.pad(hs[-1], (0, 1, 0, 1)), c.weight, c.bias, stride=2))
    h = _resblock(unet.mid.block_2, _resblock(unet.mid.block_1, hs[-1], temb), temb)
    for lvl in reversed(range(L)):
        for i, blk in enumerate(unet.up[lvl].block):
            h = _resblock(blk, torch.cat([h, hs.pop()], dim=1), temb)
            if len(unet.up[lvl].attn) > 0:
                h = _attnblock(unet.up[lvl].attn[i], h)
        if lvl != 0:
            c = unet.up[lvl].upsample.conv
            h = F.conv2d(F.interpolate(h, scale_factor=2.0, mode="nearest"), c.weight, c.bias, padding=1)
    return F.conv2d(_silu(F.group_norm(h, 4, unet.norm_out.weight, unet.norm_out.bias, 1e-6)),
                    unet.conv_out.weight, unet.conv_out.bias, padding=1)


def sampler_forward(gen, feat, cond, src_rows: Sequence[int], noise0, step_noise):
    T = gen.num_timesteps
    idx = torch.as_tensor(list(src_rows), dtype=torch.long, device=feat.device)
    x = gen.sqrt_alphas_cumprod[T - 1] * feat.index_select(0, idx) + gen.sqrt_one_minus_alphas_cumprod[T - 1] * noise0
    for i, t in enumerate(reversed(range(T))):
        x0 = unet_forward(gen.denoiser, torch.cat([cond, x], dim=1), t)
        if t == 0:
            return x0
        x = gen.posterior_mean_coef1[t] * x0 + gen.posterior_mean_coef2[t] * x \
            + (0.5 * gen.posterior_log_variance_clipped[t]).exp() * step_noise[i]
    return x


class DenoiseFunction(torch.autograd.Function):
    """pred = HIP denoise loop; backward = autograd through `sampler_forward` recomputed on the GPU."""

    @staticmethod
    def forward(ctx, gen, src_rows, feat, cond, noise0, step_noise, *params):
        with torch.no_grad():
            pred = gen._denoise(feat, cond, src_rows, (noise0, step_noise), None)
        ctx.gen, ctx.src_rows = gen, list(src_rows)
        ctx.save_for_backward(feat, cond, noise0, step_noise)
        return pred

    @staticmethod
    def backward(ctx, grad_out):
        feat, cond, noise0, step_noise = ctx.saved_tensors
        gen = ctx.gen
        params = [p for p in gen.denoiser.parameters()]
        with torch.enable_grad():
            f = feat.detach().float().requires_grad_(ctx.needs_input_grad[2])
            c = cond.detach().float().requires_grad_(ctx.needs_input_grad[3])
            out = sampler_forward(gen, f, c, ctx.src_rows, noise0, step_noise)
            wanted = [t for t in [f, c] if t.requires_grad] + [p for p in params if p.requires_grad]
            grads = list(torch.autograd.grad(out, wanted, grad_out.float().contiguous(), allow_unused=True)) if wanted else []
        gf = grads.pop(0) if f.requires_grad else None
        gc = grads.pop(0) if c.requires_grad else None
        gp = [grads.pop(0) if p.requires_grad else None for p in params]
        return (None, None, gf, gc, None, None, *gp)


# ----------------------------------------------------------------------------------------- Enhancer
def enhancer_forward(enh, x):
    b1, sa = enh.block_1, enh.split_attn
    B, C, H, W = x.shape
    tok = x.permute(0, 2, 3, 1).reshape(B, H * W, C)
    tok = tok + F.layer_norm(tok, (C,), b1.norm1.weight, b1.norm1.bias, 1e-5)
    z = F.layer_norm(tok, (C,), b1.norm2.weight, b1.norm2.bias, 1e-5)
    m = b1.mlp
    dc = C // 4
    zi = z.transpose(1, 2).reshape(B, C, H, W)
    zi = torch.cat([F.conv2d(zi[:, :dc], m.partial_conv3.weight, None, padding=1), zi[:, dc:]], dim=1)
    hdn = F.gelu(F.linear(zi.reshape(B, C, H * W).transpose(1, 2), m.linear1[0].weight, m.linear1[0].bias))
    h1, h2 = hdn.chunk(2, dim=-1)
    hid = h1.shape[-1]
    h1 = F.gelu(F.conv2d(h1.transpose(1, 2).reshape(B, hid, H, W), m.dwconv[0].weight, m.dwconv[0].bias, padding=1, groups=hid))
    tok = tok + F.linear(h1.reshape(B, hid, H * W).transpose(1, 2) * h2, m.linear2[0].weight, m.linear2[0].bias)
    s = tok.view(B, H, W, C)
    g = F.linear(s.mean((1, 2), keepdim=True), sa.fc1.weight)
    g = F.relu(F.layer_norm(g, (C,), sa.bn1.weight, sa.bn1.bias, 1e-5))
    a = torch.sigmoid(F.linear(g, sa.fc2.weight))
    return (s * a).permute(0, 3, 1, 2).contiguous()


class EnhancerFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enh, x, *params):
        with torch.no_grad():
            out = enh._forward_hip(x)
        ctx.enh = enh
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (x,) = ctx.saved_tensors
        enh = ctx.enh
        params = [p for p in enh.parameters()]
        with torch.enable_grad():
            xi = x.detach().float().requires_grad_(ctx.needs_input_grad[1])
            out = enhancer_forward(enh, xi)
            wanted = ([xi] if xi.requires_grad else []) + [p for p in params if p.requires_grad]
            grads = list(torch.autograd.grad(out, wanted, grad_out.float().contiguous(), allow_unused=True)) if wanted else []
        gx = grads.pop(0) if xi.requires_grad else None
        gp = [grads.pop(0) if p.requires_grad else None for p in params]
        return (None, gx, *gp)


# ----------------------------------------------------------------------------------------- fusion
def att_fusion_forward(xx, lens: List[int], affine_matrix):
    _, C, H, W = xx.shape
    out, o = [], 0
    for b, n in enumerate(lens):
        M = affine_matrix[b][0, :n].to(xx.device)
        grid = F.affine_grid(M, [n, C, H, W], align_corners=False).to(xx)
        x = F.grid_sample(xx[o:o + n], grid, align_corners=False)
        x = x.view(n, C, -1).permute(2, 0, 1)
        score = torch.bmm(x[:, :1], x.transpose(1, 2)) / math.sqrt(C)   # ego row only
        out.append(torch.bmm(F.softmax(score, -1), x)[:, 0].permute(1, 0).view(C, H, W))
        o += n
    return torch.stack(out)


class AttFusionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fus, lens, affine_matrix, xx):
        with torch.no_grad():
            out = fus._forward_hip(xx, lens, affine_matrix)
        ctx.lens, ctx.affine = list(lens), affine_matrix
        ctx.save_for_backward(xx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (xx,) = ctx.saved_tensors
        with torch.enable_grad():
            xi = xx.detach().float().requires_grad_(True)
            out = att_fusion_forward(xi, ctx.lens, ctx.affine)
            (gx,) = torch.autograd.grad(out, [xi], grad_out.float().contiguous())
        return None, None, None, gx
