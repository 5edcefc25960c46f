#!/usr/bin/env python3
"""CPU validation of the 'latent' sampler algebra (development tool, uses the oracle).

Standard step (cond_diff.py:272-315 + unet.py conv_in):   x' = c1*(W_out (*) A + b_out) + c2*x + s*eps ;  hs0' = W_c (*) cond + W_x (*) x' + b_in
Latent step:   hs0' = k + c2*(hs0 - k) + c1*[ Wc5 (*) A + bsum + fix ] + s*(W_x (*) eps),   k = W_c (*) cond + b_in
  Wc5  = 5x5 composite of W_x and W_out; `fix` removes, on the 1-pixel image border, the contributions that
  would pass through x0_hat positions OUTSIDE the image (a true two-stage zero-padded conv never sees them).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

torch.manual_seed(0)
C, H, W, n = 16, 9, 11, 2
Wx = torch.randn(8, C, 3, 3, dtype=torch.float64) * 0.1      # conv_in weights of the x channels
Wc = torch.randn(8, 2, 3, 3, dtype=torch.float64) * 0.1
b_in = torch.randn(8, dtype=torch.float64)
Wo = torch.randn(C, 8, 3, 3, dtype=torch.float64) * 0.1
b_out = torch.randn(C, dtype=torch.float64)
A = torch.randn(n, 8, H, W, dtype=torch.float64)             # act(GN(h)) of the last block
x = torch.randn(n, C, H, W, dtype=torch.float64)
eps = torch.randn(n, C, H, W, dtype=torch.float64)
cond = torch.randn(n, 2, H, W, dtype=torch.float64)
c1, c2, s = 0.3, 0.6, 0.2

# ---- standard
x0 = F.conv2d(A, Wo, b_out, padding=1)
xn = c1 * x0 + c2 * x + s * eps
hs0_std = F.conv2d(torch.cat([cond, xn], 1), torch.cat([Wc, Wx], 1), b_in, padding=1)

# ---- latent
k = F.conv2d(cond, Wc, b_in, padding=1)
hs0 = k + F.conv2d(x, Wx, None, padding=1)
# composite 5x5: Wc5[o,i,d] = sum_c sum_{t1+t2=d} Wx[o,c,t1] Wo[c,i,t2]  (full 2-D convolution of the kernels)
Wc5 = torch.zeros(8, 8, 5, 5, dtype=torch.float64)
for t1y in range(3):
    for t1x in range(3):
        for t2y in range(3):
            for t2x in range(3):
                Wc5[:, :, t1y + t2y, t1x + t2x] += torch.einsum("oc,ci->oi", Wx[:, :, t1y, t1x], Wo[:, :, t2y, t2x])
Wc1 = torch.einsum("ocab,cide->abodie", Wx, Wo)              # [t1y,t1x,o,t2y,t2x... ] -> index as Wc1[t1y,t1x][o, t2y(d), i, t2x(e)]
bring = torch.einsum("ocab,c->abo", Wx, b_out)               # [t1y,t1x,o]
bsum = bring.sum((0, 1))
comp = F.conv2d(A, Wc5, None, padding=2) + bsum.view(1, 8, 1, 1)
# border fix
fix = torch.zeros_like(comp)
Ap = F.pad(A, (2, 2, 2, 2))
for y in range(H):
    for xx in range(W):
        if 0 < y < H - 1 and 0 < xx < W - 1:
            continue
        for t1y in range(3):
            for t1x in range(3):
                qy, qx = y + t1y - 1, xx + t1x - 1
                if 0 <= qy < H and 0 <= qx < W:
                    continue
                # x0_virtual(q) = sum_{i,t2} Wo[c,i,t2] A(q+t2-1) + b_out ; passes through Wx[o,c,t1]
                patch = Ap[:, :, qy + 2 - 1: qy + 2 + 2, qx + 2 - 1: qx + 2 + 2]            # [n,8,3,3] zero outside
                w = torch.einsum("oc,cide->oide", Wx[:, :, t1y, t1x], Wo)                    # = Wc1[t1]
                fix[:, :, y, xx] -= torch.einsum("oide,nide->no", w, patch) + bring[t1y, t1x]
lat = k + c2 * (hs0 - k) + c1 * (comp + fix) + s * F.conv2d(eps, Wx, None, padding=1)
print("max |latent - standard| =", (lat - hs0_std).abs().max().item(), " (|hs0| max", hs0_std.abs().max().item(), ")")
print("without border fix       =", (k + c2 * (hs0 - k) + c1 * comp + s * F.conv2d(eps, Wx, None, padding=1) - hs0_std).abs().max().item())
