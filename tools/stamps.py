#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase timeline of conv8_kernel from in-kernel real-time stamps.

Build the diagnostic library (never shipped) and run on the GPU box:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DGC_STAMPS gencomm_amd/csrc/gencomm_abi.hip -o tools/probes/libgencomm_hip_diag.so
    GENCOMM_HIP_LIB=tools/probes/libgencomm_hip_diag.so python tools/stamps.py
The LAST conv8 launch of a UNet call overwrites the buffer, i.e. up.0.block.2 conv2 (full-res,
RES=2) at the default geometry; use --levels 1 to look at other shapes.
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gencomm_amd import GenComm, _lib, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4)
ap.add_argument("--C", type=int, default=64)
ap.add_argument("--H", type=int, default=200)
ap.add_argument("--W", type=int, default=704)
ap.add_argument("--blocks", type=int, default=0, help="workgroups of the stamped launch (persistent grid)")
a = ap.parse_args()
dev = torch.device("cuda:0")
gen = GenComm(synth.default_gencomm_cfg(a.C, 20)).eval().to(dev)
x = torch.randn(a.n, a.C + 2, a.H, a.W, device=dev)
t = torch.full((a.n,), 3.0, device=dev)
with torch.no_grad():
    for _ in range(3):
        gen.denoiser(x, t, T=20)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
nb = a.blocks or -(-a.W // 64) * -(-a.H // 16) * a.n
buf = np.zeros((nb, 8), dtype=np.uint64)
rc = lib.gencomm_diag_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), nb)
assert rc == 0
st = buf[:, :7].astype(np.float64) * 0.01  # us (100 MHz)
t0 = st[:, 0].min()
names = ["iter start", "wait tile+coeffs", "gn+store+prefetch+sync", "mfma", "(src2)", "epilogue", "stats"]
print(f"{nb} workgroups; kernel span {st[:, 6].max() - t0:.2f} us (first start -> last end)")
print(f"start spread: {st[:, 0].max() - t0:.2f} us; per-phase mean / p95 duration (us):")
for i in range(1, 7):
    d = st[:, i] - st[:, i - 1]
    print(f"  {names[i]:24s} mean {d.mean():6.2f}  p95 {np.percentile(d, 95):6.2f}  max {d.max():6.2f}")
life = st[:, 6] - st[:, 0]
print(f"workgroup lifetime mean {life.mean():.2f} us, max {life.max():.2f} us; last end at {st[:, 6].max() - t0:.2f}, median end {np.median(st[:, 6]) - t0:.2f}")
