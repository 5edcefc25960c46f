"""ORACLE (test infrastructure, never imported by the product): functional restatement of the reference's V2XViTFusion in
eval mode as GenComm's shells call it (prior encoding all zero => agent type 0 everywhere, no relative temporal encoding,
identity spatial correction) -- pinned by tests/golden/v2xvit.npz, produced by the reference's own modules
(oracle/make_golden.py run_v2xvit_case).

  V2XViTFusion.forward            opencood/models/fuse_modules/fusion_in_one.py:361-407   (regroup + zero padding, warp to ego)
  V2XTEncoder / V2XFusionBlock    opencood/models/sub_modules/v2xvit_basic.py:82-178      (depth x [HGT attn, window attn, MLP], pre-norm residuals)
  HGTCavAttention                 opencood/models/sub_modules/hmsa.py:7-150               (per-pixel attention across agents, relation matrices)
  PyramidWindowAttention          opencood/models/sub_modules/mswin.py:19-122             (3 window sizes, relative position bias)
  SplitAttn (radix 3)             opencood/models/sub_modules/split_attn.py:6-62
  FeedForward / PreNorm           opencood/models/sub_modules/base_transformer.py:7-40

Padded agents are not materialised: their attention columns are masked to -inf in the reference (com_mask, (B,H,W,1,L)) and
their rows never reach a real agent, so attention over the scene's real agents is the same function. STTF with the identity
correction (v2xvit_basic.py:13-34, what fusion_in_one.py:401 passes) is a bilinear resampling at the pixel centres themselves:
the identity up to float rounding of the grid (checked against the fixture at 1e-5)."""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

from torch_port import warp_affine_simple

SD = Dict[str, torch.Tensor]


def _ln(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def hgt_attention(sd: SD, p: str, x: torch.Tensor, heads: int, dim_head: int) -> torch.Tensor:
    """x [N, H, W, C] (the N real agents of one scene) -> [N, H, W, C]; every agent has type 0 (hmsa.py:117-150)."""
    N, H, W, _ = x.shape
    q = F.linear(x, sd[p + ".q_linears.0.weight"], sd[p + ".q_linears.0.bias"]).view(N, H, W, heads, dim_head)
    k = F.linear(x, sd[p + ".k_linears.0.weight"], sd[p + ".k_linears.0.bias"]).view(N, H, W, heads, dim_head)
    v = F.linear(x, sd[p + ".v_linears.0.weight"], sd[p + ".v_linears.0.bias"]).view(N, H, W, heads, dim_head)
    w_att, w_msg = sd[p + ".relation_att"][0], sd[p + ".relation_msg"][0]           # relation type 0*2+0 (hmsa.py:69-70)
    att = torch.einsum("ihwmp,mpq,jhwmq->mhwij", q, w_att, k) * dim_head ** -0.5       # hmsa.py:131-133
    att = att.softmax(dim=-1)
    v_msg = torch.einsum("mpc,jhwmp->mhwjc", w_msg, v)                                # hmsa.py:140-141 (same for every i)
    out = torch.einsum("mhwij,mhwjc->ihwmc", att, v_msg).reshape(N, H, W, heads * dim_head)
    return F.linear(out, sd[p + ".a_linears.0.weight"], sd[p + ".a_linears.0.bias"])


def window_attention(sd: SD, p: str, x: torch.Tensor, heads: int, dim_head: int, ws: int) -> torch.Tensor:
    """x [N, H, W, C] -> [N, H, W, C] (mswin.py:47-83), relative position embedding."""
    N, H, W, _ = x.shape
    nh, nw = H // ws, W // ws
    qkv = F.linear(x, sd[p + ".to_qkv.weight"]).chunk(3, dim=-1)

    def part(t):  # n (nh wh) (nw ww) (m c) -> n m (nh nw) (wh ww) c
        return t.view(N, nh, ws, nw, ws, heads, dim_head).permute(0, 5, 1, 3, 2, 4, 6).reshape(N, heads, nh * nw, ws * ws, dim_head)

    q, k, v = (part(t) for t in qkv)
    dots = torch.einsum("nmhic,nmhjc->nmhij", q, k) * dim_head ** -0.5
    idx = torch.tensor([[a, b] for a in range(ws) for b in range(ws)])
    rel = idx[None, :, :] - idx[:, None, :] + ws - 1                                   # mswin.py:12-16, :34-35
    dots = dots + sd[p + ".pos_embedding"][rel[:, :, 0], rel[:, :, 1]]
    out = torch.einsum("nmhij,nmhjc->nmhic", dots.softmax(dim=-1), v)
    out = out.view(N, heads, nh, nw, ws, ws, dim_head).permute(0, 2, 4, 3, 5, 1, 6).reshape(N, H, W, heads * dim_head)
    return F.linear(out, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])


def split_attn(sd: SD, p: str, wl: List[torch.Tensor]) -> torch.Tensor:
    """split_attn.py:43-62, radix 3: per-agent channel-wise softmax over the three window branches."""
    sw, mw, bw = wl
    C = sw.shape[-1]
    gap = (sw + mw + bw).mean((1, 2))                                                  # [N, C]
    g = F.relu(F.layer_norm(F.linear(gap, sd[p + ".fc1.weight"]), (C,), sd[p + ".bn1.weight"], sd[p + ".bn1.bias"], 1e-5))
    a = F.linear(g, sd[p + ".fc2.weight"]).view(-1, 3, C).softmax(dim=1)               # [N, 3, C]
    return sw * a[:, None, None, 0] + mw * a[:, None, None, 1] + bw * a[:, None, None, 2]


def encoder_scene(sd: SD, p: str, x: torch.Tensor, enc: dict) -> torch.Tensor:
    """x [N, H, W, C] of one scene's real agents -> ego row [H, W, C] (v2xvit_basic.py:150-178, :186-192)."""
    cav, pw = enc["cav_att_config"], enc["pwindow_att_config"]
    assert not enc.get("use_RTE", False) and not cav.get("use_RTE", False), "relative temporal encoding: not used by any GenComm yaml"
    for d in range(enc["depth"]):
        for blk in range(enc["num_blocks"]):
            q = f"{p}.layers.{d}.0.layers.{blk}"
            x = hgt_attention(sd, q + ".0.fn", _ln(sd, q + ".0.norm", x), cav["heads"], cav["dim_head"]) + x
            xn = _ln(sd, q + ".1.norm", x)
            wl = [window_attention(sd, f"{q}.1.fn.pwmsa.{i}", xn, h, dh, ws)
                  for i, (h, dh, ws) in enumerate(zip(pw["heads"], pw["dim_head"], pw["window_size"]))]
            x = split_attn(sd, q + ".1.fn.split_attn", wl) + x
        q = f"{p}.layers.{d}.1"
        xn = _ln(sd, q + ".norm", x)
        x = F.linear(F.gelu(F.linear(xn, sd[q + ".fn.net.0.weight"], sd[q + ".fn.net.0.bias"])), sd[q + ".fn.net.3.weight"], sd[q + ".fn.net.3.bias"]) + x
    return x[0]


def v2xvit_fusion(sd: SD, args: dict, x: torch.Tensor, record_len, affine_matrix: torch.Tensor) -> torch.Tensor:
    """x [sumN, C, H, W], record_len [B], affine_matrix [B, L, L, 2, 3] -> [B, C, H, W] (fusion_in_one.py:361-407)."""
    enc = args["transformer"]["encoder"]
    lens = [int(v) for v in (record_len.tolist() if hasattr(record_len, "tolist") else record_len)]
    _, C, H, W = x.shape
    out, o = [], 0
    for b, n in enumerate(lens):   # differentiable: the backward tests take float64 autograd through this function
        xb = warp_affine_simple(x[o:o + n], affine_matrix[b, 0, :n], (H, W))          # fusion_in_one.py:393-395
        out.append(encoder_scene(sd, "fusion_net.encoder", xb.permute(0, 2, 3, 1), enc).permute(2, 0, 1))
        o += n
    return torch.stack(out)
