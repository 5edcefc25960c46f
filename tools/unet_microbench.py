#!/usr/bin/env python3
"""Run a few bare UNet calls (gencomm_unet_fwd) at a given geometry -- the subject for
`rocprofv3 --kernel-trace --stats` / `--pmc` passes on single kernels.

    python tools/unet_microbench.py [--n 4 --C 64 --H 200 --W 704 --calls 5]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gencomm_amd import GenComm, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4)
    ap.add_argument("--C", type=int, default=64)
    ap.add_argument("--H", type=int, default=200)
    ap.add_argument("--W", type=int, default=704)
    ap.add_argument("--T", type=int, default=20)
    ap.add_argument("--calls", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    gen = GenComm(synth.default_gencomm_cfg(a.C, a.T)).eval().to(dev)
    x = torch.randn(a.n, a.C + 2, a.H, a.W, device=dev)
    t = torch.full((a.n,), 3.0, device=dev)
    with torch.no_grad():
        gen.denoiser(x, t, T=a.T)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.calls):
            gen.denoiser(x, t, T=a.T)
        torch.cuda.synchronize()
    print(f"unet call: {(time.perf_counter() - t0) / a.calls * 1e3:.3f} ms (incl. host-side slicing)")


if __name__ == "__main__":
    main()
