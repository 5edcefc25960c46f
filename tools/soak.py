#!/usr/bin/env python3
"""Soak: 150 scene batches of the benchmark geometry on 3 overlapped HIP streams; every (stream, Philox key) pair recurs ten times
and must reproduce its first result (up to the order of the GroupNorm statistics atomics) -- the check that caught nothing
after the two co-residency faults of DESIGN.md section 4 were fixed.   python tools/soak.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gencomm_amd import Enhancer, GenComm, normalize_pairwise_tfm, synth
from gencomm_amd.pipeline import ScenePipeline
DEV = torch.device("cuda:0")
N, C, H, W, T, B = 4, 64, 200, 704, 20, 4
torch.manual_seed(0)
gen = GenComm(synth.default_gencomm_cfg(C, T)).eval().to(DEV)
enh = Enhancer(C, [8, 8], 4).eval().to(DEV)
streams = [torch.cuda.Stream() for _ in range(3)]
pipes, data = [], []
for si in range(3):
    g = torch.Generator(device=DEV).manual_seed(si + 1)
    feat = torch.randn(B * N, C, H, W, generator=g, device=DEV).clamp_(min=0)
    cond = torch.randn(B * N, 2, H, W, generator=g, device=DEV)
    ptm = torch.from_numpy(synth.make_pairwise_t_matrix([N] * B, 5, si + 8, 40.0))
    p = ScenePipeline(gen, enh, [N] * B, C, H, W, DEV)
    p.set_affine(normalize_pairwise_tfm(ptm, H * 0.4, W * 0.4, 1))
    pipes.append(p); data.append((feat, cond))
outs = []
t0 = time.time()
with torch.no_grad():
    for it in range(150):
        si = it % 3
        seed = (it // 3) % 5
        with torch.cuda.stream(streams[si]):
            outs.append(((si, seed), pipes[si].run(data[si][0], data[si][1], seed=seed).clone()))
torch.cuda.synchronize()
el = time.time() - t0
ref, bad = {}, 0
for key, out in outs:
    assert torch.isfinite(out).all(), key
    if key in ref:
        d = (out - ref[key]).abs().max().item()
        if d > 2e-5:
            bad += 1
            print(key, "differs from its first run by", d, flush=True)
    else:
        ref[key] = out
print(f"soak (3 streams overlapped): {len(outs)} scene batches ({len(outs) * B} scenes) in {el:.1f} s = {len(outs) * B / el:.1f} scenes/s, repeats out of tolerance: {bad}")
