#!/usr/bin/env python3
"""The shader clock while the benchmark's launch pattern runs: 3 scene pipelines on 3 streams (metric workload) keep the GPU busy, a fourth
stream runs gencomm_clock_probe (one wave spinning 200 us of the constant 100 MHz counter, counting shader-clock ticks) every few steps.
Also the idle clock right after a synchronise and the clock under a pure-matrix and a pure-streaming load of this library's kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from gencomm_amd import _lib, normalize_pairwise_tfm
from gencomm_amd.pipeline import ScenePipeline
from gencomm_amd.runtime import ptr
dev = torch.device("cuda:0")
l = _lib.lib()
N, C, H, W, T = bench.WORKLOADS["metric"]
gen, enh = bench.build_modules(C, T, dev)
S, B = 3, 4
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
probe_stream = torch.cuda.Stream(device=dev)
scenes, pipes = [], []
for si in range(S):
    feat, cond, ptm = bench.make_scene(N, C, H, W, 1 + si, dev, B)
    p = ScenePipeline(gen, enh, [N] * B, C, H, W, dev)
    p.set_affine(normalize_pairwise_tfm(ptm, H * bench.PX_M, W * bench.PX_M, 1))
    scenes.append((feat, cond)); pipes.append(p)
out = torch.zeros(64, 2, dtype=torch.int64, device=dev)
def probe(i):
    _lib.check(l.gencomm_clock_probe(ptr(out[i]), 200, probe_stream.cuda_stream), "gencomm_clock_probe")
torch.cuda.synchronize()
for i in range(4): probe(i)
torch.cuda.synchronize()
idle = out[:4].cpu().double()
print("idle (nothing else in flight): %s MHz" % ", ".join("%.0f" % (100 * a / b) for a, b in idle.tolist()))
out.zero_()
with torch.no_grad():
    for i in range(6):
        with torch.cuda.stream(streams[i % S]): pipes[i % S].run(*scenes[i % S], seed=i)
    t0 = time.perf_counter()
    k = 0
    for i in range(90):
        with torch.cuda.stream(streams[i % S]): pipes[i % S].run(*scenes[i % S], seed=100 + i)
        if i % 3 == 2 and k < 30:
            probe(k); k += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
mhz = [100 * a / b for a, b in out[:k].cpu().double().tolist() if b > 0]
print("under the benchmark's load (%.1f scenes/s over 90 steps): min %.0f / median %.0f / max %.0f MHz over %d probes" % (90 * B / dt, min(mhz), sorted(mhz)[len(mhz) // 2], max(mhz), len(mhz)))
print("device property clock_rate: %s" % getattr(torch.cuda.get_device_properties(0), "clock_rate", "n/a"))
