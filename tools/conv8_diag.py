#!/usr/bin/env python3
"""Diagnostic: per layer kind (GENCOMM_CONV8H_MASK), run-to-run repeatability of a full-size UNet call with the f16-pipe
kernels and its distance to the exact-fp32 kernels, at launch sizes that put several workgroups on every CU.

    python tools/conv8_diag.py
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gencomm_amd import _lib as _L
def _set(key, v): _L.check(_L.lib().gencomm_set_mode(key, int(v)), 'gencomm_set_mode')

from gencomm_amd import GenComm, synth

DEV = "cuda:0"
C, T = 64, 20
gen = GenComm(synth.default_gencomm_cfg(C, T)).eval()
synth.fill_params_(gen, 0)
gen = gen.to(DEV)
for mask in ("1", "2", "4", "8", "16", "31"):
    _set(_L.MODE_CONV8H_MASK, mask)
    for (N, H, W) in [(64, 64, 128), (4, 200, 704), (16, 200, 704)]:
        g = torch.Generator(device=DEV).manual_seed(3)
        x = torch.randn(N, C + 2, H, W, generator=g, device=DEV)
        t = torch.full((N,), 7.0, device=DEV)
        ys = {}
        for mode in ("f32", "split", "split2", "split3"):
            _set(_L.MODE_ARITH, mode == "f32")
            with torch.no_grad():
                ys[mode] = gen.denoiser(x, t, T=T).clone()
        torch.cuda.synchronize()
        rr = max((ys["split"] - ys["split2"]).abs().max().item(), (ys["split"] - ys["split3"]).abs().max().item())
        d = (ys["split"] - ys["f32"]).abs()
        print(f"mask {mask:>2} N {N:2} {H}x{W}: split run-to-run {rr:.3e}   split vs f32 max {d.max().item():.3e} "
              f"mean {d.mean().item():.3e}", flush=True)
