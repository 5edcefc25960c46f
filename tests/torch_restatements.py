"""Differentiable torch restatements used ONLY by tests and diagnostics to check the HIP backward kernels through torch
autograd (moved out of the product package gencomm_amd/ in round 3: nothing under gencomm_amd/ calls them)."""
import math
from typing import List

import torch
import torch.nn.functional as F


def enhancer_forward(enh, x):
    """Differentiable torch restatement of Enhancer.forward (enhancer.py:367-383, :346-357, :222-250, :315-333)."""
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


def att_fusion_forward(xx, lens: List[int], affine_matrix):
    """Differentiable torch restatement of warp + ego-row attention (fusion_in_one.py:131-151)."""
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


def deform_conv3x3_torch(x, offset, weight, bias):
    """Differentiable restatement of the 3x3 / padding-1 deformable convolution (DCNv1 as torchvision defines it: offset
    channel 2k / 2k+1 = vertical / horizontal displacement of tap k, bilinear sampling, zero outside the map) with torch
    gathers. Diagnostic only (tools/diag_msgext_bwd.py)."""
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


def extractor_torch(x, ow, ob, dw, db, f0w, f0b, f2w, f2b, a1w, a1b, a3w, a3b):
    """BEVDeformableExtractor.forward (message_extractor_v2.py:103-118) in differentiable torch ops (diagnostic only)."""
    off = F.conv2d(x, ow, ob, padding=1)
    b1 = deform_conv3x3_torch(x, off, dw, db)
    g = b1.mean((2, 3), keepdim=True)
    g = torch.sigmoid(F.conv2d(F.relu(F.conv2d(g, a1w, a1b)), a3w, a3b))
    h = F.relu(F.conv2d(b1 * g, f0w, f0b))
    return F.conv2d(h, f2w, f2b)
