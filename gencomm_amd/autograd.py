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

from . import _lib
from .runtime import dev_ints, ptr, stream_ptr


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


def _lincomb(out, x, a, y=None, b=0.0, z=None, c=0.0):
    """out = a x + b y + c z on the HIP elementwise kernel (out may alias an input)."""
    _lib.check(_lib.lib().gencomm_lincomb_fwd(ptr(out), ptr(x), ptr(y), ptr(z), float(a), float(b), float(c), out.numel(), stream_ptr(out.device)),
               "gencomm_lincomb_fwd")
    return out


# Diagnostic switch (tools/train_bench.py --separate-update): False = the sampler's update between two UNet calls as separate noise /
# linear-combination launches (the first two thirds of round 4) instead of conv_out's fused epilogue
FUSED_SAMPLER_UPDATE = True


class SamplerChainFunction(torch.autograd.Function):
    """The training branch's whole T-step chain (cond_diff.py:342-360, :262-264, :272-315) as ONE autograd node: q_sample, T HIP
    UNet calls with every intermediate kept (`gencomm_unet_fwd_train`), the posterior-mean update between them, and in backward
    the same walk in reverse (`gencomm_unet_bwd` per step) -- the elementwise glue and the accumulation of the T gradient blobs
    run on HIP kernels (`gencomm_lincomb_fwd`), the noise is either the explicit tensors of the caller (tests) or the sampler's
    own in-kernel Philox field (`gencomm_q_sample_fwd` / `gencomm_step_noise_fwd`: the field inference adds for the same seed).
    Round 2 composed the chain from `UNetFunction` and torch tensor arithmetic: 60 framework kernels per training step."""

    @staticmethod
    def forward(ctx, gen, src_rows, noise0, step_noise, seed, feat, cond, flat):
        T = gen.num_timesteps
        unet = gen.denoiser
        dev = feat.device
        n, (C, H, W) = cond.shape[0], feat.shape[1:]
        # raw pointers from here on: GenComm._checked has validated feat / cond / src_rows; the noise tensors once more, because this
        # node is also reachable through autograd.sampler_forward directly
        if tuple(cond.shape) != (n, 2, H, W) or len(src_rows) != n or C != unet.feature_channels:
            raise ValueError(f"sampler chain: cond {tuple(cond.shape)} / feat {tuple(feat.shape)} / {len(src_rows)} source rows do not fit a {unet.feature_channels}-channel denoiser")
        if (noise0 is None) != (step_noise is None):
            raise ValueError("sampler chain: give both noise0 and step_noise, or neither")
        if noise0 is not None and (tuple(noise0.shape) != (n, C, H, W) or tuple(step_noise.shape) != (T, n, C, H, W) or not (noise0.is_cuda and step_noise.is_cuda)):
            raise ValueError(f"sampler chain: noise must be device tensors (noise0 [{n},{C},{H},{W}], step_noise [{T},{n},{C},{H},{W}]), "
                             f"got {tuple(noise0.shape)} and {tuple(step_noise.shape)}")
        l = _lib.lib()
        st = stream_ptr(dev)
        with torch.no_grad():
            f, cd = feat.detach().float().contiguous(), cond.detach().float().contiguous()
            sched = gen._sched_table(dev)
            rows = dev_ints(list(src_rows), dev)
            x = torch.empty((n, C, H, W), dtype=torch.float32, device=dev)
            n0 = None if noise0 is None else noise0.detach().float().contiguous()
            _lib.check(l.gencomm_q_sample_fwd(ptr(sched[T - 1]), ptr(f), f.shape[0], ptr(rows), ptr(n0), int(seed), T, ptr(x), n, C, H, W, st),
                       "gencomm_q_sample_fwd")
            xs, wss = [], []
            coef = gen._sched_host(dev)            # [T][5]: sqrt_ac, sqrt_1m_ac, coef1, coef2, sigma (cached host copy: no sync)
            out = None
            for i, t in enumerate(reversed(range(T))):
                xs.append(x)
                if t == 0:
                    out, ws = unet.forward_train(x, cd, t, T)
                    wss.append(ws)
                    break
                if FUSED_SAMPLER_UPDATE:
                    # x_{t-1} = coef1 x0_hat + coef2 x_t + sigma eps in conv_out's epilogue, as the inference loop runs it (eps: the explicit
                    # tensor, or the sampler's Philox field of (seed, t)); x0_hat is never written
                    nz = None if step_noise is None else step_noise[i].detach().float().contiguous()
                    x, ws = unet.forward_train_step(x, cd, t, T, sched[t], nz, seed)
                    wss.append(ws)
                    continue
                x0, ws = unet.forward_train(x, cd, t, T)
                wss.append(ws)
                if step_noise is None:   # nu_t = fp16(sigma_t z) of the sampler's Philox field, already scaled
                    nu = torch.empty_like(x)
                    _lib.check(l.gencomm_step_noise_fwd(ptr(sched[t]), int(seed), t, ptr(nu), n, C, H, W, 0, st), "gencomm_step_noise_fwd")
                    x = _lincomb(nu, x0, coef[t][2], x, coef[t][3], nu, 1.0)
                else:
                    x = _lincomb(torch.empty_like(x), x0, coef[t][2], x, coef[t][3], step_noise[i].detach().float().contiguous(), coef[t][4])
        ctx.gen, ctx.src_rows, ctx.coef = gen, list(src_rows), coef
        ctx.xs, ctx.wss, ctx.cond = xs, wss, cd
        ctx.feat_shape = tuple(feat.shape)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        gen, coef = ctx.gen, ctx.coef
        unet, T = gen.denoiser, gen.num_timesteps
        if ctx.xs is None:   # the T saved x_t maps and UNet workspaces were released by the first backward
            raise RuntimeError("GenComm's sampler chain was already differentiated once: its saved activations (T UNet workspaces) are "
                               "freed after the first backward; run the forward again (retain_graph=True is not supported here)")
        need_feat, need_cond, need_flat = ctx.needs_input_grad[5], ctx.needs_input_grad[6], ctx.needs_input_grad[7]
        with torch.no_grad():
            g_x0 = grad_out.float().contiguous()
            d_prev = None            # gradient of x_{t-1}, the input of the step walked before this one
            g_cond = g_flat = None
            for t in range(T):       # the loop ran t = T-1 .. 0; xs[T-1-t] is the x_t the call at t saw
                i = T - 1 - t
                if t > 0 and FUSED_SAMPLER_UPDATE:
                    # d x0_hat = coef1_t d x_{t-1}: the call is linear in it, so d x_{t-1} goes in unscaled, d x_t = coef1 (input gradient)
                    # + coef2 d x_{t-1} is formed in the last layer's epilogue, and the parameter / message gradients take coef1 below
                    alpha = coef[t][2]
                    gx, gc, graw = unet.backward_call(ctx.xs[i], ctx.cond, t, d_prev, T, ws=ctx.wss[i], chain=(alpha, coef[t][3], d_prev))
                else:
                    alpha = 1.0
                    if t > 0:
                        g_x0 = _lincomb(torch.empty_like(d_prev), d_prev, coef[t][2])
                    gx, gc, graw = unet.backward_call(ctx.xs[i], ctx.cond, t, g_x0, T, ws=ctx.wss[i])
                    if t > 0:
                        _lincomb(gx, gx, 1.0, d_prev, coef[t][3])      # d x_t = UNet input gradient + c2_t d x_{t-1}
                ctx.wss[i] = None
                d_prev = gx
                g_cond = gc if g_cond is None else _lincomb(g_cond, g_cond, 1.0, gc, alpha)
                g_flat = graw if g_flat is None else _lincomb(g_flat, g_flat, 1.0, graw, alpha)
            g_feat = None
            if need_feat:            # x_{T-1} = sqrt_ac feat[src_rows] + ...: rows of one scene all point at its ego row
                g_feat = torch.zeros(ctx.feat_shape, dtype=torch.float32, device=d_prev.device)
                for k, r in enumerate(ctx.src_rows):
                    _lincomb(g_feat[r], g_feat[r], 1.0, d_prev[k], coef[T - 1][0])
            ctx.xs = ctx.wss = None
        return None, None, None, None, None, g_feat, g_cond if need_cond else None, g_flat if need_flat else None


def sampler_forward(gen, feat, cond, src_rows: Sequence[int], noise0, step_noise, seed: int = 0):
    """The training branch's maths (cond_diff.py:342-360, :262-264, :272-315): one `SamplerChainFunction` node.  `noise0` /
    `step_noise[i]` (the noise of the i-th loop iteration) may be None: the chain then draws the sampler's in-kernel Philox field
    of `seed`."""
    return SamplerChainFunction.apply(gen, list(src_rows), noise0, step_noise, int(seed), feat, cond, gen.denoiser.flat_params())


# ----------------------------------------------------------------------------------------- Enhancer
def _gate_mlp(sa, gap, params):
    """split_attn's channel gate on [n, C] vectors (enhancer.py:315-333): fc1 -> LayerNorm -> ReLU -> fc2 -> sigmoid."""
    return torch.sigmoid(F.linear(F.relu(F.layer_norm(F.linear(gap, params[0]), (params[0].shape[0],), params[1], params[2], 1e-5)), params[3]))


# Diagnostic switch (tools/train_bench.py --enh-split): True = the first half of round 4's flow, which wrote x1, x2 = GELU(Linear1 output).chunk(2)
# in a pass of their own and kept them for the backward; False (default) = their consumers evaluate GELU on the half of v they read.
ENH_MATERIALIZE_GELU = False
# Diagnostic switch (tools/train_bench.py --pconv-general): False = FRFN.partial_conv3 and its input gradient through the general convolution at every size
ENH_PCONV_C16 = True


class EnhancerFunction(torch.autograd.Function):
    """The Enhancer for a call that will be differentiated.  The forward runs the stage layer by layer in NCHW on HIP primitives
    (gencomm_amd/train_ops.py: LayerNorm, the partial 3x3 and the Linear layers as convolutions, depthwise 3x3, fused GELU / gate
    kernels) and KEEPS every intermediate; the backward walks them with the matching gradient kernels -- nothing is recomputed
    (round 2 ran the fused inference kernels forward and recomputed this chain in backward: 0.75 ms per 4-agent scene more).
    The channel gate's MLP on [n, C] vectors is plain torch tensor arithmetic."""

    @staticmethod
    def forward(ctx, enh, x, *params):
        from . import train_ops as T
        b1, sa, m = enh.block_1, enh.split_attn, enh.block_1.mlp
        x = x.detach().float().contiguous()
        n, C, H, W = x.shape
        dc, hid, HW = C // 4, m.dwconv[0].weight.shape[0], H * W
        with torch.no_grad():
            y = T.ln_fwd(x, b1.norm1.weight, b1.norm1.bias, 1e-5, True)            # x + LN1(x)       enhancer.py:351-352
            # LN2 (:354) writes straight into Linear1's input zi = cat[pconv(z[:, :dc]), z[:, dc:]] (:229-232): its first dc channels are copied
            # out (the partial convolution's input, kept for its weight gradient) and then overwritten by the convolution -- no z, no cat
            zi = T.ln_fwd(y, b1.norm2.weight, b1.norm2.bias, 1e-5, False)
            z1 = T.copy_slice(zi, 0, dc)
            if ENH_PCONV_C16 and dc == 16 and n * HW >= (1 << 16):   # C = 64 on large maps: the UNet's 8-channel kernel (the general one re-stages 16-pixel segments)
                T.conv3x3_c16(z1, m.partial_conv3.weight, zi)
            else:
                T.conv2d(z1, m.partial_conv3.weight, None, 1, out=zi, out_coff=0)
            w1 = m.linear1[0].weight.detach()[:, :, None, None]
            v = T.conv2d(zi, w1, m.linear1[0].bias, 0)                                # Linear1          :235
            h1 = h2 = None
            if ENH_MATERIALIZE_GELU:
                h1, h2 = torch.empty(n, hid, H, W, dtype=torch.float32, device=x.device), torch.empty(n, hid, H, W, dtype=torch.float32, device=x.device)
                T.ew_slice(T.EW_GELU_SPLIT, v, o0=h1, o1=h2, n=n, nch=hid, HW=HW)         # GELU, chunk      :236-240
                u = T.dwconv3x3(h1, m.dwconv[0].weight, m.dwconv[0].bias)
                g = torch.empty_like(u)
                T.ew_slice(T.EW_GELU_GATE, u, h2, o0=g, n=n, nch=hid, HW=HW)
            else:
                # x1, x2 = GELU(v).chunk(2) (:236-240) are never written: the depthwise layer and the gate evaluate GELU on the half of v they read
                u = T.dwconv3x3(v, m.dwconv[0].weight, m.dwconv[0].bias, gelu_in=True)     # depthwise(x1)    :241-243
                g = torch.empty_like(u)
                T.ew_slice(T.EW_GELU2_GATE, u, d=v, o0=g, n=n, nch=hid, HW=HW, o1_ct=2 * hid, o1_c0=hid)   # GELU(u) * x2     :244-246
            w2 = m.linear2[0].weight.detach()[:, :, None, None]
            y2 = T.conv2d(g, w2, m.linear2[0].bias, 0, residual=y)                     # Linear2 + residual :247, :354
            gap = T.nc_dot(y2, None) / HW                                              # global average pool   :325
            a = _gate_mlp(sa, gap, [sa.fc1.weight, sa.bn1.weight, sa.bn1.bias, sa.fc2.weight])
            out = T.nc_scale(y2, a, None)                                              # x * gate         :333
        ctx.enh = enh
        ctx.h12 = (h1, h2)
        ctx.save_for_backward(x, y, z1, zi, v, u, g, y2, gap)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from . import train_ops as T
        x, y, z1, zi, v, u, g, y2, gap0 = ctx.saved_tensors
        enh = ctx.enh
        b1, sa, m = enh.block_1, enh.split_attn, enh.block_1.mlp
        n, C, H, W = x.shape
        dc, hid, HW = C // 4, m.dwconv[0].weight.shape[0], H * W
        w1 = m.linear1[0].weight.detach()[:, :, None, None]
        w2 = m.linear2[0].weight.detach()[:, :, None, None]
        with torch.no_grad():
            go = grad_out.float().contiguous()
            da = T.nc_dot(go, y2)
        # ---- channel gate on [n, C] vectors (split_attn, :315-333): torch autograd on a few hundred numbers
        gate_params = [sa.fc1.weight, sa.bn1.weight, sa.bn1.bias, sa.fc2.weight]
        with torch.enable_grad():
            gap = gap0.detach().clone().requires_grad_(True)
            local = [p.detach().requires_grad_(True) for p in gate_params]
            a = _gate_mlp(sa, gap, local)
            dgap, *dgate = torch.autograd.grad(a, [gap] + local, da)
        # Weight gradients feed nothing in this walk: on large maps they run on a side stream beside the input-gradient chain (the same
        # overlap as GENCOMM_MODE_BWD_STREAMS inside gencomm_unet_bwd) and are joined before d y2 is overwritten by LayerNorm2's backward.
        from .runtime import overlap
        # (a `with` block: an exception between fork and join -- a shape error, a failed library call -- still joins the side stream and
        # hands its outputs to the current stream before it propagates)
        with overlap(x.device, n * HW) as ov, torch.no_grad():
            on_side = ov.run
            dy2 = T.nc_scale(go, a.detach(), dgap / HW)
            # ---- Linear2
            dg = T.conv2d(dy2, w2.transpose(0, 1).contiguous(), None, 0)
            dW2, db2 = on_side(lambda: T.conv2d_wgrad(dy2, g, 1, 0, True))
            # ---- gate, depthwise + GELU, Linear1's GELU: d u and the x2 half of d v in one pass, the x1 half behind the depthwise dgrad
            du, dv = torch.empty_like(u), torch.empty_like(v)
            h1, h2 = ctx.h12
            if h2 is not None:
                T.ew_slice(T.EW_GATE_BWD, u, h2, dg, v, o0=du, o1=dv, n=n, nch=hid, HW=HW, o1_ct=2 * hid, o1_c0=hid)
            else:
                T.ew_slice(T.EW_GATE_BWD2, u, None, dg, v, o0=du, o1=dv, n=n, nch=hid, HW=HW, o1_ct=2 * hid, o1_c0=hid)
            dh1 = T.dwconv3x3(du, m.dwconv[0].weight, None, flip=True)
            dWd, dbd = on_side((lambda: T.dwconv3x3_wgrad(h1, du)) if h1 is not None else (lambda: T.dwconv3x3_wgrad(v, du, gelu_in=True)))
            T.ew_slice(T.EW_GELU_BWD, v, dh1, o0=dv, n=n, nch=hid, HW=HW, o0_ct=2 * hid, o0_c0=0)
            # ---- Linear1
            dzi = T.conv2d(dv, w1.transpose(0, 1).contiguous(), None, 0)
            dW1, db1l = on_side(lambda: T.conv2d_wgrad(dv, zi, 1, 0, True))
            # ---- partial conv: its input gradient overwrites the first dc channels of d zi (= d z)
            dzi1 = T.copy_slice(dzi, 0, dc)
            if ENH_PCONV_C16 and dc == 16 and n * HW >= (1 << 16):
                T.conv3x3_c16(dzi1, m.partial_conv3.weight, dzi, transposed=True)
            else:
                T.conv2d(dzi1, m.partial_conv3.weight.detach().flip(2, 3).transpose(0, 1).contiguous(), None, 1, out=dzi, out_coff=0)
            dz = dzi
            dWp, _ = on_side(lambda: T.conv2d_wgrad(dzi1, z1, 3, 1, False))
            ov.join()                                                                      # idempotent; __exit__ joins again on every path
            # ---- LayerNorms and residuals
            dy, dg2, db2n = T.ln_bwd(y, b1.norm2.weight, dz, 1e-5, accumulate_into=dy2)       # d y = d y2 + LN2 backward
            dx, dg1, db1n = T.ln_bwd(x, b1.norm1.weight, dy, 1e-5)
            _lincomb(dx, dx, 1.0, dy, 1.0)                                                 # d x = d y + LN1 backward
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
