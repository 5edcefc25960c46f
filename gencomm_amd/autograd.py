"""Training support for the hot path.

GenComm (the T-step chain of UNet calls the reference's training branch back-propagates through, cond_diff.py:342-360):
HIP forward AND HIP backward -- ``UNetFunction`` wraps one UNet call (``gencomm_unet_fwd`` / ``gencomm_unet_bwd``: conv dgrad
and wgrad, GroupNorm+SiLU backward, nin / Downsample / Upsample / timestep-MLP backward; the forward keeps the call's
intermediates in a workspace of its own, nothing is recomputed), the sampler's affine updates between the calls are elementwise torch
ops that autograd composes.
Gradients reach what the reference's do (training-branch row): the UNet weights, every row of ``conditions``
and -- through the ego repeat -- the ego rows of ``spatial_features``.

Enhancer and AttFusion: HIP forward; backward still re-evaluates the stage with differentiable torch ops on the GPU from
the saved input (activation-checkpoint style). Nothing here touches the CPU or the oracle.

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


# ----------------------------------------------------------------------------------------- UNet + sampler
def _silu(x):
    return x * torch.sigmoid(x)


def _resblocks_in_execution_order(unet):
    """(key prefix, module) of every ResnetBlock in the order the library enumerates them (down, mid, up from the deepest level)."""
    out, L = [], unet.num_resolutions
    for l in range(L):
        out += [(f"down.{l}.block.{i}", blk) for i, blk in enumerate(unet.down[l].block)]
    out += [("mid.block_1", unet.mid.block_1), ("mid.block_2", unet.mid.block_2)]
    for l in reversed(range(L)):
        out += [(f"up.{l}.block.{i}", blk) for i, blk in enumerate(unet.up[l].block)]
    return out


class UNetFunction(torch.autograd.Function):
    """x0_hat = DiffusionUNet(cat[cond, x_t], t): HIP forward (gencomm_unet_fwd) and HIP backward (gencomm_unet_bwd: conv
    dgrad / wgrad, GroupNorm+SiLU backward, nin / Downsample / Upsample backward, timestep MLP backward; csrc/unet_bwd_kernels.h).
    The forward keeps every intermediate of the call in a workspace of its own (288 GB of HBM: 0.5 GB per call at 4 x 64 x 200 x 704);
    the backward reads them -- nothing is recomputed.
    `flat` is the UNet's parameters as ONE differentiable vector in the library's blob order (`DiffusionUNet.flat_params()`):
    the backward returns one gradient blob per call, autograd sums T blobs and splits the sum once (instead of 146 small
    accumulations per call)."""

    @staticmethod
    def forward(ctx, unet, t_int, T, x_t, cond, flat):
        with torch.no_grad():
            xt, cd = x_t.detach().float().contiguous(), cond.detach().float().contiguous()
            out, ws = unet.forward_train(xt, cd, int(t_int), int(T))
        ctx.unet, ctx.t_int, ctx.T, ctx.ws = unet, int(t_int), int(T), ws   # the call's intermediates: no recomputation in backward
        ctx.save_for_backward(xt, cd)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x_t, cond = ctx.saved_tensors
        gx, gc, graw = ctx.unet.backward_call(x_t, cond, ctx.t_int, grad_out.float().contiguous(), ctx.T, ws=ctx.ws)
        ctx.ws = None
        return (None, None, None, gx if ctx.needs_input_grad[3] else None, gc if ctx.needs_input_grad[4] else None,
                graw if ctx.needs_input_grad[5] else None)


def sampler_forward(gen, feat, cond, src_rows: Sequence[int], noise0, step_noise):
    """The training branch's maths (cond_diff.py:342-360, :262-264, :272-315) as a chain of `UNetFunction` calls and
    elementwise torch ops; autograd composes the T steps. `step_noise[i]` is the noise of the i-th loop iteration."""
    T = gen.num_timesteps
    idx = torch.as_tensor(list(src_rows), dtype=torch.long, device=feat.device)
    flat = gen.denoiser.flat_params()
    x = gen.sqrt_alphas_cumprod[T - 1] * feat.index_select(0, idx) + gen.sqrt_one_minus_alphas_cumprod[T - 1] * noise0
    for i, t in enumerate(reversed(range(T))):
        x0 = UNetFunction.apply(gen.denoiser, t, T, x, cond, flat)
        if t == 0:
            return x0
        x = gen.posterior_mean_coef1[t] * x0 + gen.posterior_mean_coef2[t] * x \
            + (0.5 * gen.posterior_log_variance_clipped[t]).exp() * step_noise[i]
    return x


# ----------------------------------------------------------------------------------------- Enhancer
class EnhancerFunction(torch.autograd.Function):
    """HIP forward (gencomm_enhancer_fwd, the fused inference kernels) and a backward composed of HIP primitives
    (gencomm_amd/train_ops.py): the stage is recomputed layer by layer in NCHW with the exact-fp32 general convolution
    (1x1 = Linear, 3x3 partial conv), LayerNorm and depthwise kernels, then walked backwards with their gradient kernels.
    Elementwise products / sums and the channel gate's MLP on [n, C] vectors are plain torch tensor arithmetic."""

    @staticmethod
    def forward(ctx, enh, x, *params):
        with torch.no_grad():
            out = enh._forward_hip(x)
        ctx.enh = enh
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from . import train_ops as T
        (x,) = ctx.saved_tensors
        enh = ctx.enh
        b1, sa, m = enh.block_1, enh.split_attn, enh.block_1.mlp
        n, C, H, W = x.shape
        dc, hid, HW = C // 4, m.dwconv[0].weight.shape[0], H * W
        with torch.no_grad():
            # ---- forward recompute, every intermediate kept
            y = T.ln_fwd(x, b1.norm1.weight, b1.norm1.bias, 1e-5, True)            # x + LN1(x)       enhancer.py:351-352
            z = T.ln_fwd(y, b1.norm2.weight, b1.norm2.bias, 1e-5, False)           # LN2              :354
            z1 = z[:, :dc].contiguous()
            zi = torch.cat([T.conv2d(z1, m.partial_conv3.weight, None, 1), z[:, dc:]], dim=1)                    # :229-232
            w1 = m.linear1[0].weight.detach()[:, :, None, None]
            v = T.conv2d(zi, w1, m.linear1[0].bias, 0)                                # Linear1          :235
            hdn = F.gelu(v)
            h1, h2 = hdn[:, :hid].contiguous(), hdn[:, hid:].contiguous()
            u = T.dwconv3x3(h1, m.dwconv[0].weight, m.dwconv[0].bias)                  # depthwise        :241-243
            h1p = F.gelu(u)
            g = h1p * h2
            w2 = m.linear2[0].weight.detach()[:, :, None, None]
            y2 = y + T.conv2d(g, w2, m.linear2[0].bias, 0)                             # Linear2 + residual :247, :354
        # ---- channel gate on [n, C] vectors (split_attn, :315-333): torch autograd on a few hundred numbers
        gate_params = [sa.fc1.weight, sa.bn1.weight, sa.bn1.bias, sa.fc2.weight]
        with torch.enable_grad():
            gap = y2.mean((2, 3)).requires_grad_(True)
            local = [p.detach().requires_grad_(True) for p in gate_params]
            a = torch.sigmoid(F.linear(F.relu(F.layer_norm(F.linear(gap, local[0]), (local[0].shape[0],), local[1], local[2], 1e-5)), local[3]))
            go = grad_out.float()
            da = (go * y2).sum((2, 3))
            dgap, *dgate = torch.autograd.grad(a, [gap] + local, da)
        with torch.no_grad():
            a = a.detach()
            dy2 = go * a[:, :, None, None] + dgap[:, :, None, None] / HW
            # ---- Linear2
            dg = T.conv2d(dy2, w2.transpose(0, 1).contiguous(), None, 0)
            dW2, db2 = T.conv2d_wgrad(dy2, g, 1, 0, True)
            dh1p, dh2 = dg * h2, dg * h1p
            # ---- depthwise + GELU
            du = T.gelu_bwd(u, dh1p)
            dh1 = T.dwconv3x3(du, m.dwconv[0].weight, None, flip=True)
            dWd, dbd = T.dwconv3x3_wgrad(h1, du)
            # ---- Linear1 + GELU
            dv = T.gelu_bwd(v, torch.cat([dh1, dh2], dim=1))
            dzi = T.conv2d(dv, w1.transpose(0, 1).contiguous(), None, 0)
            dW1, db1l = T.conv2d_wgrad(dv, zi, 1, 0, True)
            # ---- partial conv
            dzi1 = dzi[:, :dc].contiguous()
            dz = torch.cat([T.conv2d_dgrad(dzi1, m.partial_conv3.weight, 1), dzi[:, dc:]], dim=1)
            dWp, _ = T.conv2d_wgrad(dzi1, z1, 3, 1, False)
            # ---- LayerNorms and residuals
            dy_ln, dg2, db2n = T.ln_bwd(y, b1.norm2.weight, dz, 1e-5)
            dy = dy2 + dy_ln
            dx_ln, dg1, db1n = T.ln_bwd(x, b1.norm1.weight, dy, 1e-5)
            dx = dy + dx_ln
        grads = {id(b1.norm1.weight): dg1, id(b1.norm1.bias): db1n, id(b1.norm2.weight): dg2, id(b1.norm2.bias): db2n,
                 id(m.partial_conv3.weight): dWp, id(m.linear1[0].weight): dW1[:, :, 0, 0], id(m.linear1[0].bias): db1l,
                 id(m.dwconv[0].weight): dWd, id(m.dwconv[0].bias): dbd, id(m.linear2[0].weight): dW2[:, :, 0, 0], id(m.linear2[0].bias): db2}
        grads.update({id(p): gr for p, gr in zip(gate_params, dgate)})
        gp = [grads.get(id(p)) if p.requires_grad else None for p in enh.parameters()]
        return (None, dx if ctx.needs_input_grad[1] else None, *gp)


# ----------------------------------------------------------------------------------------- fusion
class AttFusionFunction(torch.autograd.Function):
    """HIP forward (gencomm_warp_attfuse_fwd) and HIP backward (gencomm_warp_attfuse_bwd: softmax / dot-product backward
    per pixel + the bilinear gather's adjoint with float atomics)."""

    @staticmethod
    def forward(ctx, fus, lens, affine_matrix, xx):
        with torch.no_grad():
            out = fus._forward_hip(xx, lens, affine_matrix)
        ctx.fus, ctx.lens, ctx.affine = fus, list(lens), affine_matrix
        ctx.save_for_backward(xx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (xx,) = ctx.saved_tensors
        return None, None, None, ctx.fus._backward_hip(xx, ctx.lens, ctx.affine, grad_out.float().contiguous())
