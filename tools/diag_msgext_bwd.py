#!/usr/bin/env python3
"""Debug: message extractor backward pieces vs float64 autograd of the torch restatement, on the GPU."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch, torch.nn.functional as F
from gencomm_amd import MessageExtractorv2, synth, train_ops as T
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from torch_restatements import deform_conv3x3_torch as _deform_conv3x3_torch, extractor_torch as _extractor_torch
C, H, W, n = 128, 32, 48, 1
me = MessageExtractorv2(C, 2).train(); synth.fill_params_(me, 41)
with torch.no_grad():
    me.bev_extractor.offset1.weight.mul_(5.0); me.bev_extractor.offset1.bias.mul_(5.0)
me = me.cuda()
x = torch.randn(n, C, H, W, generator=torch.Generator().manual_seed(C)).cuda()
ps = [p.detach().double() for p in me._param_list()]
ow, ob, dw, db = ps[:4]
xd = x.double().requires_grad_(True)
off = F.conv2d(xd, ow, ob, padding=1)
offh0 = T.conv2d(x, ow.float(), ob.float(), 1)
offd = (offh0.double() if "--same-offsets" in sys.argv else off.detach()).requires_grad_(True)
xs = x.double().requires_grad_(True)
b1 = _deform_conv3x3_torch(xs, offd, dw, db)
gb1 = torch.randn(b1.shape, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
gx_ref, goff_ref = torch.autograd.grad(b1, [xs, offd], gb1)
# HIP pieces
offh = T.conv2d(x, ow.float(), ob.float(), 1)
print("off err", float((offh.double() - off.detach()).abs().max()), float(off.abs().max()))
col = T.dcn_sample(x, offh)
wd2 = dw.float().reshape(64, C * 9)[:, :, None, None]
b1h = T.conv2d(col, wd2, db.float(), 0)
print("b1 err", float((b1h.double() - b1.detach()).abs().max()), float(b1.abs().max()))
dcol = T.conv2d(gb1.float(), wd2.transpose(0, 1).contiguous(), None, 0)
dcol_ref = torch.einsum("ok,nohw->nkhw", dw.reshape(64, C * 9), gb1)
print("dcol err", float((dcol.double() - dcol_ref).abs().max()), float(dcol_ref.abs().max()))
dx, doff = T.dcn_scatter_bwd(x, offh, dcol)
print("dx(scatter) err", float((dx.double() - gx_ref).abs().max()), float(gx_ref.abs().max()))
print("doff err", float((doff.double() - goff_ref).abs().max()), float(goff_ref.abs().max()))
e = (doff.double() - goff_ref).abs()
i = int(e.argmax()); print("doff worst idx", i, float(doff.flatten()[i]), float(goff_ref.flatten()[i]))
# fractional parts of sample positions near integers?
k = (i // (H * W)) % 18
print("channel", k)
