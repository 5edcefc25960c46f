"""Backward of ``V2XViTFusion`` on the HIP kernels (training with ``fusion_method: v2xvit``: stage 1, and stage 2 where the frozen
fusion net must still pass gradients back to the new agent's message extractor).

``V2XViTFunction`` wraps the whole fusion: HIP forward (``V2XViTFusion._forward_hip``); the backward re-runs the forward layer by
layer with the same HIP kernels, keeping what the gradient kernels need, then walks the layers in reverse:
  Linear layers         input gradient = the 1x1 convolution with the transposed weight, weight gradient on the split-K MFMA kernel
  LayerNorm             gencomm_ln_nchw_bwd
  agent-wise attention  gencomm_hgt_attn_bwd (relation matrices folded into k / v: their gradients are unfolded by torch autograd
                        through the 3 small weight products of ``_hgt_weights``)
  window attention      gencomm_win_attn_bwd (two launches: lane = query for dQ / d pos, lane = key for dK / dV)
  split attention       elementwise products / sums and the gate MLP on [n, C] vectors in torch tensor arithmetic (as EnhancerFunction)
  GELU                  gencomm_gelu_bwd;   warp to ego: gencomm_warp_affine_bwd (bilinear scatter)
Reference: fusion_in_one.py:355-407, sub_modules/{v2xvit_basic, hmsa, mswin, split_attn, base_transformer}.py.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib, train_ops as T
from .runtime import f32c, ptr, stream_ptr


def _lin(x, w, b, act=0):
    return T.conv2d(x, w.detach()[:, :, None, None], b, 0) if act == 0 else F.gelu(T.conv2d(x, w.detach()[:, :, None, None], b, 0))


def _lin_bwd(dy, x, w, has_bias):
    """(dx, dW [out, in], db) of y = W x + b over NCHW pixels."""
    dx = T.conv2d(dy, w.detach().t().contiguous()[:, :, None, None], None, 0)
    dw, db = T.conv2d_wgrad(dy, x, 1, 0, has_bias)
    return dx, dw[:, :, 0, 0], db


class _Masks:
    """Dropout masks (already scaled by 1 / (1 - p)) in the order the forward draws them; replayed in the same order by the
    backward's recomputation. The reference's dropouts sit after the attention output projections, after the FeedForward's GELU
    and after its last Linear (hmsa.py:148, mswin.py:41-44, base_transformer.py:31-36); they are active whenever the module is in
    train mode -- also for a frozen fusion net in stage 2 (fix_bn only touches BatchNorm)."""

    def __init__(self, active: bool, recorded=None):
        self.active, self.recorded, self.pos = active, ([] if recorded is None else recorded), 0
        self.replay = recorded is not None

    def apply(self, y: torch.Tensor, p: float):
        """(dropout(y), mask or None)."""
        if not self.active or p <= 0.0:
            return y, None
        if self.replay:
            m = self.recorded[self.pos]
            self.pos += 1
        else:
            m = (torch.rand_like(y) >= p).to(y.dtype) / (1.0 - p)
            self.recorded.append(m)
        return y * m, m


def _drop_p(seq) -> float:
    for m in seq.modules() if hasattr(seq, "modules") else []:
        if isinstance(m, torch.nn.Dropout):
            return float(m.p)
    return 0.0


def run_layers(mod, x, theta, scene_off, B, masks: _Masks):
    """The fusion layer by layer on the HIP kernels (unfused epilogues), every intermediate kept: (h_final, tape)."""
    from .v2xvit import HGTCavAttention, _hgt_weights
    enc = mod.fusion_net.encoder
    l, dev = _lib.lib(), x.device
    st = stream_ptr(dev)
    n, C, H, W = x.shape
    HW = H * W
    x = f32c(x)
    h = torch.empty_like(x)
    _lib.check(l.gencomm_warp_affine_fwd(ptr(x), ptr(theta), ptr(h), n, C, H, W, st), "gencomm_warp_affine_fwd")
    tape = []
    for block, ff in enc.layers:
        for cav, pwin in block.layers:
            att = cav.fn
            a = T.ln_fwd(h, cav.norm.weight, cav.norm.bias, 1e-5, False)
            if isinstance(att, HGTCavAttention):
                wq, bq = _hgt_weights(att)
                out_lin, p_att = att.a_linears[0], float(att.drop_out.p)
            else:
                wq, bq = att.to_qkv.weight.detach(), None
                out_lin, p_att = att.to_out[0], _drop_p(att.to_out)
            qkv = _lin(a, wq, bq)
            inner = att.heads * att.dim_head
            o = torch.empty(n, inner, H, W, dtype=torch.float32, device=dev)
            _lib.check(l.gencomm_hgt_attn_fwd(ptr(qkv), ptr(scene_off), ptr(o), B, att.heads, att.dim_head, HW, st), "gencomm_hgt_attn_fwd")
            y, m_att = masks.apply(_lin(o, out_lin.weight, out_lin.bias), p_att)
            tape.append(("cav", cav, h, a, wq, qkv, o, out_lin, m_att))
            h = y + h
            b_ = T.ln_fwd(h, pwin.norm.weight, pwin.norm.bias, 1e-5, False)
            branches = []
            for wa in pwin.fn.pwmsa:
                qkv_w = _lin(b_, wa.to_qkv.weight, None)
                inner_w = wa.heads * wa.dim_head
                o_w = torch.empty(n, inner_w, H, W, dtype=torch.float32, device=dev)
                pos = f32c(wa.pos_embedding.detach())
                _lib.check(l.gencomm_win_attn_fwd(ptr(qkv_w), ptr(pos), ptr(o_w), n, wa.heads, wa.dim_head, wa.window_size, H, W, st), "gencomm_win_attn_fwd")
                p_w = _drop_p(wa.to_out)
                y_w, m_w = masks.apply(_lin(o_w, wa.to_out[0].weight, wa.to_out[0].bias), p_w)
                branches.append((wa, qkv_w, o_w, y_w, pos, m_w))
            sa = pwin.fn.split_attn
            ys = torch.stack([br[3] for br in branches], 1)                                  # [n, 3, C, H, W]
            gap = ys.sum(1).mean((2, 3))                                                     # split_attn.py:50-53
            gate = F.linear(F.relu(F.layer_norm(F.linear(gap, sa.fc1.weight), (C,), sa.bn1.weight, sa.bn1.bias, 1e-5)),
                            sa.fc2.weight).view(n, 3, C).softmax(dim=1)
            tape.append(("pwin", pwin, h, b_, branches, ys, gap))
            h = (ys * gate[:, :, :, None, None]).sum(1) + h
        c_ = T.ln_fwd(h, ff.norm.weight, ff.norm.bias, 1e-5, False)
        l0, l3 = ff.fn.net[0], ff.fn.net[3]
        p_ff = _drop_p(ff.fn.net)
        pre = _lin(c_, l0.weight, l0.bias)
        mid, m_mid = masks.apply(F.gelu(pre), p_ff)
        y, m_out = masks.apply(_lin(mid, l3.weight, l3.bias), p_ff)
        tape.append(("ff", ff, h, c_, pre, mid, m_mid, m_out))
        h = y + h
    return h, tape


class V2XViTFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, x, theta, scene_off, B, *params):
        ctx.mod, ctx.B = mod, B
        ctx.save_for_backward(x, theta, scene_off)
        dropout = mod.training and any(isinstance(m, torch.nn.Dropout) and m.p > 0 for m in mod.modules())
        with torch.no_grad():
            if not dropout:
                ctx.masks = None
                return mod._forward_hip(x, theta, scene_off, B)
            masks = _Masks(True)
            h, _ = run_layers(mod, x, theta, scene_off, B, masks)
            ctx.masks = masks.recorded
            return h[scene_off[:-1].long()].contiguous()

    @staticmethod
    def backward(ctx, gout):
        from .v2xvit import HGTCavAttention, _hgt_weights
        x, theta, scene_off = ctx.saved_tensors
        mod, B = ctx.mod, ctx.B
        l, dev = _lib.lib(), x.device
        st = stream_ptr(dev)
        n, C, H, W = x.shape
        HW = H * W
        grads = {}

        def acc(p, g):
            if p.requires_grad:
                grads[p] = g if p not in grads else grads[p] + g

        def unmask(g_, m_):
            return g_ if m_ is None else g_ * m_

        with torch.no_grad():
            h, tape = run_layers(mod, x, theta, scene_off, B, _Masks(ctx.masks is not None, ctx.masks))
            g = torch.zeros_like(h)
            g[scene_off[:-1].long()] = gout.float()
            for entry in reversed(tape):
                kind = entry[0]
                if kind == "ff":
                    _, ff, h_in, c_, pre, mid, m_mid, m_out = entry
                    l0, l3 = ff.fn.net[0], ff.fn.net[3]
                    dmid, dw3, db3 = _lin_bwd(unmask(g, m_out), mid, l3.weight, True)
                    acc(l3.weight, dw3); acc(l3.bias, db3)
                    dpre = T.gelu_bwd(pre, unmask(dmid, m_mid))
                    dc, dw0, db0 = _lin_bwd(dpre, c_, l0.weight, True)
                    acc(l0.weight, dw0); acc(l0.bias, db0)
                    dh, dgm, dbt = T.ln_bwd(h_in, ff.norm.weight, dc, 1e-5)
                    acc(ff.norm.weight, dgm); acc(ff.norm.bias, dbt)
                    g = g + dh
                elif kind == "pwin":
                    _, pwin, h_in, b_, branches, ys, gap = entry
                    sa = pwin.fn.split_attn
                    # radix-3 split attention: gate MLP on [n, C] vectors through torch autograd
                    gate_params = [sa.fc1.weight, sa.bn1.weight, sa.bn1.bias, sa.fc2.weight]
                    with torch.enable_grad():
                        gp = gap.detach().requires_grad_(True)
                        local = [p.detach().requires_grad_(True) for p in gate_params]
                        gate = F.linear(F.relu(F.layer_norm(F.linear(gp, local[0]), (C,), local[1], local[2], 1e-5)), local[3]).view(n, 3, C).softmax(dim=1)
                        dgate = (g[:, None] * ys).sum((3, 4))                                        # [n, 3, C]
                        dgap, *dgp = torch.autograd.grad(gate, [gp] + local, dgate)
                    for p_, gr in zip(gate_params, dgp):
                        acc(p_, gr)
                    gate = gate.detach()
                    db_total = torch.zeros_like(b_)
                    for r, (wa, qkv_w, o_w, y_w, pos, m_w) in enumerate(branches):
                        dy = unmask(g * gate[:, r, :, None, None] + dgap[:, :, None, None] / HW, m_w)
                        do, dwo, dbo = _lin_bwd(dy, o_w, wa.to_out[0].weight, True)
                        acc(wa.to_out[0].weight, dwo); acc(wa.to_out[0].bias, dbo)
                        dqkv = torch.empty_like(qkv_w)
                        dpos = torch.zeros_like(pos)
                        scratch = torch.empty(_lib.check_size(l.gencomm_win_attn_bwd_scratch_floats(n, wa.heads, wa.window_size, H, W),
                                                              "gencomm_win_attn_bwd_scratch_floats"), dtype=torch.float32, device=dev)
                        _lib.check(l.gencomm_win_attn_bwd(ptr(qkv_w), ptr(pos), ptr(o_w), ptr(f32c(do)), ptr(dqkv), ptr(dpos), ptr(scratch), n, wa.heads,
                                                          wa.dim_head, wa.window_size, H, W, st), "gencomm_win_attn_bwd")
                        acc(wa.pos_embedding, dpos.view_as(wa.pos_embedding))
                        dbb, dwq, _ = _lin_bwd(dqkv, b_, wa.to_qkv.weight, False)
                        acc(wa.to_qkv.weight, dwq)
                        db_total += dbb
                    dh, dgm, dbt = T.ln_bwd(h_in, pwin.norm.weight, db_total, 1e-5)
                    acc(pwin.norm.weight, dgm); acc(pwin.norm.bias, dbt)
                    g = g + dh
                else:
                    _, cav, h_in, a, wq, qkv, o, out_lin, m_att = entry
                    att = cav.fn
                    do, dwo, dbo = _lin_bwd(unmask(g, m_att), o, out_lin.weight, True)
                    acc(out_lin.weight, dwo); acc(out_lin.bias, dbo)
                    dqkv = torch.empty_like(qkv)
                    _lib.check(l.gencomm_hgt_attn_bwd(ptr(qkv), ptr(scene_off), ptr(f32c(do)), ptr(dqkv), B, att.heads, att.dim_head, HW, st),
                               "gencomm_hgt_attn_bwd")
                    hetero = isinstance(att, HGTCavAttention)
                    da, dwq, dbq = _lin_bwd(dqkv, a, wq, hetero)
                    if hetero:   # unfold the relation matrices: autograd through the small weight products of _hgt_weights
                        srcs = [att.q_linears[0].weight, att.q_linears[0].bias, att.k_linears[0].weight, att.k_linears[0].bias,
                                att.v_linears[0].weight, att.v_linears[0].bias, att.relation_att, att.relation_msg]
                        need = [p for p in srcs if p.requires_grad]
                        if need:
                            with torch.enable_grad():
                                wf, bf = _hgt_weights(att, detach=False)
                                gs = torch.autograd.grad([wf, bf], need, [dwq, dbq], allow_unused=True)
                            for p_, gr in zip(need, gs):
                                if gr is not None:
                                    acc(p_, gr)
                    else:
                        acc(att.to_qkv.weight, dwq)
                    dh, dgm, dbt = T.ln_bwd(h_in, cav.norm.weight, da, 1e-5)
                    acc(cav.norm.weight, dgm); acc(cav.norm.bias, dbt)
                    g = g + dh
            dx = None
            if ctx.needs_input_grad[1]:
                dx = torch.empty_like(x)
                _lib.check(l.gencomm_warp_affine_bwd(ptr(theta), ptr(f32c(g)), ptr(dx), n, C, H, W, st), "gencomm_warp_affine_bwd")
        gp_out = [grads.get(p) if p.requires_grad else None for p in mod.parameters()]
        return (None, dx, None, None, None, *gp_out)
