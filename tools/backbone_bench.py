#!/usr/bin/env python3
"""Time the HIP BaseBEVBackbone + DownsampleConv + heads at the shipped shape (m1_att.yaml: 64 x 256 x 512
pillar map, layer_nums [3,5,8], filters [64,128,256], deblocks to 3 x 128) and, for orientation only, the same
layers through torch's own conv (MIOpen) on the same GPU.   python tools/backbone_bench.py [--n 2]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gencomm_amd import synth
from gencomm_amd.bev_backbone import BaseBEVBackbone, DownsampleConv, HipConv2d
import _mode
_mode.apply_env_modes()   # GENCOMM_TOOL_ARITH=3: the opt-in two-term general convolutions

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2)
ap.add_argument("--H", type=int, default=256)
ap.add_argument("--W", type=int, default=512)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = {"layer_nums": [3, 5, 8], "layer_strides": [2, 2, 2], "num_filters": [64, 128, 256],
       "upsample_strides": [1, 2, 4], "num_upsample_filter": [128, 128, 128]}
bb = BaseBEVBackbone(cfg, 64).eval().to(dev)
sh = DownsampleConv({"kernal_size": [3], "stride": [2], "padding": [1], "dim": [128], "input_dim": 384}).eval().to(dev)
heads = [HipConv2d(128, 2, 1).to(dev), HipConv2d(128, 14, 1).to(dev), HipConv2d(128, 4, 1).to(dev)]
synth.fill_bn_stats_(bb, 1)
x = torch.randn(a.n, 64, a.H, a.W, device=dev).relu_()


def hip():
    z = sh(bb({"spatial_features": x})["spatial_features_2d"])
    return [h(z) for h in heads]


def torch_ref():
    y = x
    ups = []
    for i, blk in enumerate(bb.blocks):
        y = blk(y)
        ups.append(bb.deblocks[i](y))
    z = torch.cat(ups, 1)
    for l in sh.layers:
        z = l.double_conv(z)
    return [torch.nn.functional.conv2d(z, h.weight, h.bias) for h in heads]


def macs():
    H, W, cin, tot = a.H, a.W, 64, 0
    for ln, f, uf, us in zip(cfg["layer_nums"], cfg["num_filters"], cfg["num_upsample_filter"], cfg["upsample_strides"]):
        H, W = H // 2, W // 2
        tot += H * W * (cin * f * 9 + ln * f * f * 9) + H * W * f * uf * us * us
        cin = f
    Ho, Wo = a.H // 4, a.W // 4
    tot += Ho * Wo * (384 * 128 * 9 + 128 * 128 * 9 + 128 * 20)
    return tot * a.n


with torch.no_grad():
    for name, fn in (("hip", hip), ("torch/MIOpen (orientation)", torch_ref)):
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            out = fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.iters * 1e3
        print(f"{name:28s} {ms:8.3f} ms / call  ({2 * macs() / ms / 1e9:.1f} TFLOP/s fp32, n={a.n}, {a.H}x{a.W})")
    r, g = torch_ref(), hip()
    print("max |hip - torch| over heads:", max((p - q).abs().max().item() for p, q in zip(r, g)))
