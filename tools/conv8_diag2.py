#!/usr/bin/env python3
"""Diagnostic: where (inside the 64x16 tile, which agent) do sporadic split-vs-f32 differences sit? (UP variant only)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gencomm_amd import GenComm, synth
DEV = "cuda:0"
C, T = 64, 20
gen = GenComm(synth.default_gencomm_cfg(C, T)).eval()
synth.fill_params_(gen, 0)
gen = gen.to(DEV)
os.environ["GENCOMM_CONV8H_MASK"] = sys.argv[1] if len(sys.argv) > 1 else "16"
N, H, W = 4, 200, 704
g = torch.Generator(device=DEV).manual_seed(3)
x = torch.randn(N, C + 2, H, W, generator=g, device=DEV)
t = torch.full((N,), 7.0, device=DEV)
os.environ["GENCOMM_CONV8"] = "f32"
with torch.no_grad():
    ref = gen.denoiser(x, t, T=T).clone()
os.environ["GENCOMM_CONV8"] = "split"
for rep in range(3):
    with torch.no_grad():
        y = gen.denoiser(x, t, T=T).clone()
    d = (y - ref).abs().amax(dim=1)  # [N, H, W]
    bad = (d > 1e-4)
    print(f"rep {rep}: bad pixels {int(bad.sum())} of {bad.numel()}; per agent {[int(b.sum()) for b in bad]}")
    idx = bad.nonzero()
    if len(idx):
        ys, xs = idx[:, 1], idx[:, 2]
        print("  tile rows (y//16) hit:", sorted(set((ys // 16).tolist())))
        print("  tile cols (x//64) hit:", sorted(set((xs // 64).tolist())))
        # connected blobs: print the bounding boxes of the first few clusters by tile
        tiles = {}
        for n_, y_, x_ in idx.tolist():
            k = (n_, y_ // 16, x_ // 64)
            b = tiles.setdefault(k, [y_, y_, x_, x_, 0.0])
            b[0] = min(b[0], y_); b[1] = max(b[1], y_); b[2] = min(b[2], x_); b[3] = max(b[3], x_)
            b[4] = max(b[4], d[n_, y_, x_].item())
        items = sorted(tiles.items(), key=lambda kv: -kv[1][4])[:12]
        for k, b in items:
            print(f"  agent {k[0]} tile ({k[1]},{k[2]}): rows {b[0]}..{b[1]} (mod16 {b[0]%16}..{b[1]%16}) cols {b[2]}..{b[3]} (mod64 {b[2]%64}..{b[3]%64}) max {b[4]:.3e}")
