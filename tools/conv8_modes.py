#!/usr/bin/env python3
"""A/B of the two 8-channel convolution kernels (GENCOMM_CONV8=f32 | split): error against the reference golden
vectors with the 64x16 tile forced onto the small fixtures, and the time of one full-size UNet call.

    python tools/conv8_modes.py
"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch
from gencomm_amd import _lib as _L
def _set(key, v): _L.check(_L.lib().gencomm_set_mode(key, int(v)), 'gencomm_set_mode')

from gencomm_amd import GenComm, synth
from helpers import build_inputs, build_modules, eval_noise, load_case, sub

DEV = "cuda:0"


def golden_err(name):
    g = load_case(name)
    _, gen, _ = build_modules(g, DEV)
    inp = build_inputs(g, DEV)
    noise = eval_noise(g, DEV)
    with torch.no_grad():
        pred = gen(inp["feat"], inp["cond"], inp["record_len"], noise=noise)["pred_feature"]
    want = g["pred_feature"].astype(np.float64)
    got = sub(pred, int(g["stride"])).astype(np.float64)
    err = np.abs(got - want)
    return err.max(), (err / (1e-5 + 1e-4 * np.abs(want))).max()


def main():
    for tile in ("512", "1"):
        _set(_L.MODE_TILE_WANT, tile)
        for mode in ("f32", "split"):
            _set(_L.MODE_ARITH, mode == "f32")
            for name in ("tiny", "ragged", "mid", "shipped"):
                e, r = golden_err(name)
                print(f"tile_want {tile:>3} conv8 {mode:5} {name:8}: max abs err {e:.3e}  worst err/tol {r:.3f}", flush=True)
    _set(_L.MODE_TILE_WANT, 512)
    n, C, H, W, T = 16, 64, 200, 704, 20
    torch.manual_seed(0)
    gen = GenComm(synth.default_gencomm_cfg(C, T)).eval().to(DEV)
    x = torch.randn(n, C + 2, H, W, device=DEV)
    t = torch.full((n,), 3.0, device=DEV)
    outs = {}
    for rep in range(2):
        for mode in ("f32", "split"):
            _set(_L.MODE_ARITH, mode == "f32")
            with torch.no_grad():
                y = gen.denoiser(x, t, T=T)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    y = gen.denoiser(x, t, T=T)
                torch.cuda.synchronize()
            outs[mode] = y
            print(f"full-size UNet call, {n} agents, conv8 {mode}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms", flush=True)
    d = (outs["f32"] - outs["split"]).abs()
    print(f"f32 vs split on the full-size call: max abs diff {d.max().item():.3e}, max |y| {outs['f32'].abs().max().item():.3e}")


if __name__ == "__main__":
    main()
