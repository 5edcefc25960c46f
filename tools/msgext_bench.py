#!/usr/bin/env python3
"""Measurement for MessageExtractorv2: ms per call and achieved TFLOP/s / GB/s.
    python tools/msgext_bench.py [--n 4 --C 64 --H 200 --W 704]
Algorithmic work per pixel: offset conv 9*C*18 + deformable conv 9*C*64 + fuse 64*64 + 64*2 MACs;
bytes: read x (C), write message (2) -- the 18-ch offsets and the 64-ch intermediate are extra traffic."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gencomm_amd import MessageExtractorv2

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4)
ap.add_argument("--C", type=int, default=64)
ap.add_argument("--H", type=int, default=200)
ap.add_argument("--W", type=int, default=704)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = MessageExtractorv2(a.C, 2).eval().to(dev)
x = torch.randn(a.n, a.C, a.H, a.W, device=dev).clamp_(min=0)
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        y = m(x)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
px = a.n * a.H * a.W
flops = 2.0 * px * (9 * a.C * 18 + 9 * a.C * 64 + 64 * 64 + 128)
byts = 4.0 * px * (a.C + 2)
print(json.dumps({"op": "MessageExtractorv2", "n": a.n, "C": a.C, "H": a.H, "W": a.W, "ms": ms,
                  "achieved_tflops": flops / ms / 1e9, "frac_fp32_peak": flops / ms / 1e9 / 157.3,
                  "algorithmic_gbs": byts / ms / 1e6, "finite": bool(torch.isfinite(y).all())}))
