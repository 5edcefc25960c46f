#!/usr/bin/env python3
"""Per-shape timing of gencomm_conv2d_fwd (prepared weights cached) on the three-term f16-pipe kernels (default) against the exact-fp32
kernel (GENCOMM_MODE_ARITH = 1): the stage-1 training leg's convolution shapes.   python tools/conv_h3_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gencomm_amd import _lib
from gencomm_amd.runtime import conv2d_prepare, ptr, stream_ptr

dev = torch.device("cuda:0")
l = _lib.lib()
SHAPES = [  # N, Cin, Cout, H, W, K, stride
    (4, 64, 64, 128, 256, 3, 1), (4, 128, 128, 64, 128, 3, 1), (4, 256, 256, 32, 64, 3, 1),   # the stage-1 leg's backbone levels (9.7 GFLOP each)
    (4, 64, 64, 128, 64, 3, 1), (4, 128, 128, 64, 32, 3, 1), (4, 256, 256, 32, 16, 3, 1), (4, 384, 256, 128, 64, 3, 1),
    (4, 64, 64, 256, 128, 3, 2), (4, 128, 128, 128, 64, 1, 1), (4, 256, 128, 64, 128, 1, 1), (4, 128, 512, 64, 128, 1, 1), (4, 64, 64, 200, 704, 3, 1),
]
for (N, Cin, Cout, H, W, K, S) in SHAPES:
    x = torch.randn(N, Cin, H, W, device=dev)
    w = torch.randn(Cout, Cin, K, K, device=dev) / (K * Cin ** 0.5)
    pad = K // 2
    Ho, Wo = (H + 2 * pad - K) // S + 1, (W + 2 * pad - K) // S + 1
    y = torch.empty(N, Cout, Ho, Wo, device=dev)
    ss = torch.stack([torch.ones(Cout), torch.zeros(Cout)]).to(dev)
    prepared = conv2d_prepare(w, Cin, Cout, K, K, 0, dev)
    st = stream_ptr(dev)
    flops = 2.0 * N * Ho * Wo * Cout * Cin * K * K
    out = []
    for arith in (0, 1):
        _lib.check(l.gencomm_set_mode(_lib.MODE_ARITH, arith), "set_mode")
        def run():
            _lib.check(l.gencomm_conv2d_fwd(ptr(x), ptr(prepared), ptr(ss[0]), ptr(ss[1]), ptr(y), N, Cin, H, W, Cout, K, K, S, pad, 0, 1, Cout, 0, st), "conv2d_fwd")
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        out.append((us, flops / us * 1e-6))
    _lib.check(l.gencomm_set_mode(_lib.MODE_ARITH, 0), "set_mode")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): conv2d_prepare(w, Cin, Cout, K, K, 0, dev)
    e1.record(); torch.cuda.synchronize()
    print("N %d %3d->%3d %3dx%3d k%d s%d (%.1f GFLOP): three-term %7.1f us = %6.1f TFLOP/s | exact fp32 %7.1f us = %5.1f TFLOP/s | prepare %.1f us" % (
        N, Cin, Cout, H, W, K, S, flops * 1e-9, out[0][0], out[0][1], out[1][0], out[1][1], e0.elapsed_time(e1) * 1e3 / 20), flush=True)
