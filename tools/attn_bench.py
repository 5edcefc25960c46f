#!/usr/bin/env python3
"""Cost of the UNet's BEV self-attention (AttnBlock, unet.py:141-193: single head, 8 channels, softmax over ALL pixels of the level) -- off in
every shipped yaml (attn_resolutions [16] never matches), measured here so that the path has a number: one UNet call with and without the five
half-resolution AttnBlocks (attn_resolutions [64] with ch_mult [1, 1]) at small maps.   python tools/attn_bench.py"""
import copy, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gencomm_amd import GenComm, synth

dev = "cuda:0"
for (n, C, H, W) in ((2, 128, 64, 128), (2, 64, 96, 176)):
    res = {}
    for label, attn in (("without", [16]), ("with", [64])):
        cfg = copy.deepcopy(synth.default_gencomm_cfg(C, 3))
        cfg["model"]["attn_resolutions"] = attn
        gen = GenComm(cfg).eval().to(dev)
        synth.fill_params_(gen, 3)
        x = torch.randn(n, C + 2, H, W, device=dev)
        t = torch.full((n,), 1, device=dev)
        with torch.no_grad():
            for _ in range(3):
                gen.denoiser(x, t, T=3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            K = 10
            for _ in range(K):
                gen.denoiser(x, t, T=3)
            torch.cuda.synchronize()
        res[label] = 1e3 * (time.perf_counter() - t0) / K
    keys = (H // 2) * (W // 2)
    extra = res["with"] - res["without"]
    flops = 5 * n * 2 * 2 * keys * keys * 8      # 5 blocks, QK^T and PV, 8 channels
    print(f"UNet call, {n} agents, C={C}, {H}x{W}: {res['without']:.3f} ms without / {res['with']:.3f} ms with 5 AttnBlocks over {keys} keys "
          f"(+{extra:.3f} ms = {flops / extra / 1e9:.1f} TFLOP/s on the two contractions)", flush=True)
