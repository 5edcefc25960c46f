#!/usr/bin/env python3
"""Benchmark of the GenComm generative-communication hot path on MI355X.

One "step" = one scene through GenComm (q_sample + T x0-parameterised ancestral denoise steps of
the diffusion UNet) -> Enhancer -> warp + AttFusion, on synthetic tensors already resident in HBM.
Workload = BASELINE.json's metric configuration: 4 agents, C=64, 200x704 BEV, T=20.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Scenes are independent, so N GPUs = N replicas each running its own scenes (no data-path
collective; weak scaling). Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np
import torch

WORKLOADS = {
    # name: (agents, C, H, W, T)
    "metric": (4, 64, 200, 704, 20),   # BASELINE.json metric: 4 agents, 64x200x704 BEV, 20 steps
    "cfg2": (2, 64, 200, 704, 10),     # BASELINE.json configs[1]
    "shipped": (2, 128, 64, 128, 3),   # shape of every shipped OPV2V / DAIR-V2X yaml (control)
    "v2xreal": (2, 256, 64, 128, 3),   # V2X-Real yamls: 256 feature channels
}
PX_M = 0.4           # metres per BEV pixel at 200x704 (OPV2V range +-140.8 x +-40 m)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32 vector == f32-input MFMA peak

# dominant kernel (profiles/r1_bench_default_kernel_stats.csv: 19 % of kernel time, the largest single kernel): family id
# in the library's timer.  The 8-channel convolutions together are 59 % (five conv8h_kernel variants): the largest of
# them, the 16 -> 8 layers (family 2), is reported beside it against the HBM roof in a separate pass.
DOMINANT = {"family": 15, "name": "latent_step_kernel"}
SECONDARY_FAMILY = 2


def algorithmic_work(N, C, HW, T):
    """FLOPs and bytes (fp32) per scene."""
    flops = 2.0 * HW * (N * T * (144 * C + 11280) + N * (6.5625 * C * C + 18 * C) + 2 * N * C)
    byts = 4.0 * HW * (N * T * (2 * C + 2) + 2 * N * C + 2 * N * C + (N + 1) * C)
    return flops, byts


def make_scene(N, C, H, W, seed, device, B=1):
    """Synthetic inputs for B scenes of N agents, generated on the device (seeded torch generator;
    the parity tests use the numpy generator of gencomm_amd.synth -- here only shape/statistics matter)."""
    from gencomm_amd import synth
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    feat = torch.randn(B * N, C, H, W, generator=g, device=device).clamp_(min=0)
    cond = torch.randn(B * N, 2, H, W, generator=g, device=device)
    ptm = torch.from_numpy(synth.make_pairwise_t_matrix([N] * B, 5, seed + 7, 40.0))
    return feat, cond, ptm


def build_modules(C, T, device):
    from gencomm_amd import Enhancer, GenComm, synth
    torch.manual_seed(0)  # "default PyTorch init, seed 0" 
    gen = GenComm(synth.default_gencomm_cfg(C, T)).eval()
    enh = Enhancer(C, [8, 8], 4).eval()
    return gen.to(device), enh.to(device)


def cpu_baseline(name, N, C, H, W, T, gen, enh, ptm, sample_steps=2):
    """The CPU oracle (plain PyTorch restatement of the reference, oracle/torch_port.py) on this
    host's cores, on a bounded sample: q_sample + `sample_steps` of the T denoise steps + Enhancer +
    fusion at full tensor size; the per-step time is extrapolated to T steps (every step has
    identical cost)."""
    from gencomm_amd import synth
    from oracle import torch_port as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cfg = synth.default_gencomm_cfg(C, T)
    sd_g = {k: v.detach().cpu() for k, v in gen.state_dict().items()}
    sd_e = {k: v.detach().cpu() for k, v in enh.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    feat = torch.randn(N, C, H, W, generator=g).clamp_(min=0)
    cond = torch.randn(N, 2, H, W, generator=g)
    noise = torch.randn(N, C, H, W, generator=g)
    sched = O.make_schedule(T)
    # thread count: ATen's CPU convolutions on these 8-channel maps stop scaling (and then slow down
    # badly) well below the core count of a 2-socket host, so calibrate on one denoise step and
    # keep the fastest of a few candidates; `cores` reports the threads actually used
    best = None
    with torch.no_grad():
        for th in [c for c in (8, 16, 32, 64) if c <= avail] or [avail]:
            torch.set_num_threads(th)
            tc = time.perf_counter()
            O.p_sample(sd_g, sched, cfg["model"], cond, feat, T - 1, noise)
            tc = time.perf_counter() - tc
            if best is None or tc < best[1]:
                best = (th, tc)
            if tc > 20.0:
                break
    cores = best[0]
    torch.set_num_threads(cores)
    with torch.no_grad():
        t0 = time.perf_counter()
        x = O.q_sample(sched, O.ego_repeat(feat, [N]), T - 1, noise)
        t1 = time.perf_counter()
        for i in range(sample_steps):
            x = O.p_sample(sd_g, sched, cfg["model"], cond, x, T - 1 - i, noise)
        t2 = time.perf_counter()
        e = O.enhancer_forward(sd_e, x, [N])
        t3 = time.perf_counter()
        affine = O.normalize_pairwise_tfm(ptm, H * PX_M, W * PX_M, 1.0)
        O.att_fusion(e, [N], affine)
        t4 = time.perf_counter()
    per_step = (t2 - t1) / sample_steps
    scene_s = (t1 - t0) + per_step * T + (t3 - t2) + (t4 - t3)
    return {"value": 1.0 / scene_s, "unit": "scenes/sec", "cores": cores, "kind": "port",
            "sample": f"workload '{name}': q_sample + {sample_steps} of {T} denoise steps ({per_step:.2f} s each, extrapolated x{T}) "
                      f"+ enhancer {t3 - t2:.2f} s + fusion {t4 - t3:.2f} s, torch {torch.__version__} CPU, {cores} threads "
                      f"(fastest of 8/16/32/64 on one step; host exposes {avail} logical CPUs)",
            "scene_seconds": scene_s}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="metric", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-enhancer", action="store_true")
    ap.add_argument("--timer-family", type=int, default=DOMINANT["family"])
    ap.add_argument("--batch", type=int, default=4, help="scenes per step (batched in one launch sequence, record_len=[N]*B)")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent scenes in flight per GPU, each on its own HIP stream with its own buffers")
    ap.add_argument("--graph", type=int, default=0, help="1: replay the scene's launch sequence as a captured HIP graph")
    ap.add_argument("--mode", action="append", default=[], metavar="KEY=VALUE",
                    help="library mode for this run (gencomm_set_mode; keys: arith sampler tile_want enh_fuse conv8h_mask xcd), e.g. --mode xcd=0")
    args = ap.parse_args()

    from gencomm_amd import dist as gdist
    rank, world, local_rank = gdist.env_rank_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = gdist.init_process_group("nccl", device)  # RCCL; only the timing barrier/reduction use it

    from gencomm_amd import _lib, normalize_pairwise_tfm
    from gencomm_amd.pipeline import ScenePipeline
    lib = _lib.lib()
    mode_keys = {"arith": _lib.MODE_ARITH, "sampler": _lib.MODE_SAMPLER, "tile_want": _lib.MODE_TILE_WANT,
                 "enh_fuse": _lib.MODE_ENH_FUSE, "conv8h_mask": _lib.MODE_CONV8H_MASK, "xcd": _lib.MODE_XCD_REMAP}
    for kv in args.mode:
        k, v = kv.split("=")
        _lib.check(lib.gencomm_set_mode(mode_keys[k], int(v)), "gencomm_set_mode")

    N, C, H, W, T = WORKLOADS[args.workload]
    gen, enh = build_modules(C, T, device)
    # S scenes in flight: the path is a chain of ~600 short dependent launches per scene, so two
    # independent scenes on two streams fill each other's latency gaps (memory phases of one overlap
    # compute phases of the other). Every stream has its own inputs, workspace and outputs.
    S, B = max(1, args.streams), max(1, args.batch)
    streams = [torch.cuda.Stream(device=device) for _ in range(S)]
    scenes, pipes = [], []
    for si in range(S):
        feat, cond, ptm = make_scene(N, C, H, W, 1 + rank * S + si, device, B)
        pipe = ScenePipeline(gen, None if args.no_enhancer else enh, [N] * B, C, H, W, device, graph=bool(args.graph))
        pipe.set_affine(normalize_pairwise_tfm(ptm, H * PX_M, W * PX_M, 1))
        scenes.append((feat, cond))
        pipes.append(pipe)
    torch.cuda.synchronize(device)

    def run_scene(i, seed):
        si = i % S
        with torch.cuda.stream(streams[si]):
            pipes[si].run(scenes[si][0], scenes[si][1], seed=seed)

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    with torch.no_grad():
        for i in range(args.warmup):
            run_scene(i, 1000 + i)
        barrier()
        # arm the kernel timer for the dominant kernel on rank 0 (HIP events on the launch stream)
        timed = rank == 0 and args.timer_family >= 0 and not args.graph  # event pairs cannot be recorded into a replayed graph
        if timed:
            _lib.check(lib.gencomm_timer_start(args.timer_family, args.steps * (T + 4) * 16), "gencomm_timer_start")
        t0 = time.perf_counter()
        for i in range(args.steps):
            run_scene(i, 2000 + i)
        barrier()
        elapsed = time.perf_counter() - t0
    k_ms, k_n = ctypes.c_double(0.0), ctypes.c_int(0)
    iso_ms, iso_n = ctypes.c_double(0.0), ctypes.c_int(0)
    if timed:
        _lib.check(lib.gencomm_timer_stop(ctypes.byref(k_ms), ctypes.byref(k_n)), "gencomm_timer_stop")
        # the same kernel with nothing else in flight (one stream, outside the timed region): what
        # the kernel itself achieves when it does not share the chip with another scene
        with torch.no_grad():
            _lib.check(lib.gencomm_timer_start(args.timer_family, 4 * (T + 4) * 16), "gencomm_timer_start")
            for i in range(2):
                pipes[0].run(scenes[0][0], scenes[0][1], seed=3000 + i)
            torch.cuda.synchronize(device)
            _lib.check(lib.gencomm_timer_stop(ctypes.byref(iso_ms), ctypes.byref(iso_n)), "gencomm_timer_stop")
    sec_ms, sec_n = ctypes.c_double(0.0), ctypes.c_int(0)
    if timed:
        with torch.no_grad():  # same single-stream pass for the 16 -> 8 channel layers (HBM-bound)
            _lib.check(lib.gencomm_timer_start(SECONDARY_FAMILY, 4 * (T + 4) * 16), "gencomm_timer_start")
            for i in range(2):
                pipes[0].run(scenes[0][0], scenes[0][1], seed=3100 + i)
            torch.cuda.synchronize(device)
            _lib.check(lib.gencomm_timer_stop(ctypes.byref(sec_ms), ctypes.byref(sec_n)), "gencomm_timer_stop")
    # the same workload with the exact-fp32 MFMA kernels everywhere (GENCOMM_CONV8=f32, read per call by the library): a short
    # untimed-region pass so that the JSON line carries both arithmetic modes
    exact = None
    if rank == 0 and lib.gencomm_get_mode(_lib.MODE_ARITH) == 0:
        with _lib.mode(_lib.MODE_ARITH, 1):
            with torch.no_grad():
                for i in range(S):
                    run_scene(i, 4000 + i)
                torch.cuda.synchronize(device)
                te = time.perf_counter()
                ne = 2 * S
                for i in range(ne):
                    run_scene(i, 4100 + i)
                torch.cuda.synchronize(device)
                te = time.perf_counter() - te
            exact = {"value": ne * B / te, "unit": "scenes/sec", "steps": ne, "n_gpus": 1,
                     "note": "rank 0 only, same pipelines and streams, exact-fp32 v_mfma_f32_4x4x1 / 32x32x2 kernels"}
    for pipe in pipes:
        assert torch.isfinite(pipe.fused).all(), "non-finite output"

    # every rank ran `steps` scenes of its own; whole-job rate = all scenes / slowest rank
    value, elapsed, total_scenes = gdist.aggregate_throughput(args.steps * B, elapsed, dist, device)

    if rank == 0:
        HW = H * W
        flops, byts = algorithmic_work(N, C, HW, T)
        out = {
            "metric": "scenes/sec", "value": value, "unit": "scenes/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "total_scenes": total_scenes,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: GenComm->Enhancer->AttFusion, {N} agents, C={C}, {H}x{W} BEV, "
                                   f"T={T} x0-param ancestral steps, {B} scene(s)/step/GPU",
                       "agents": N, "C": C, "H": H, "W": W, "T": T, "enhancer": not args.no_enhancer,
                       "noise": "in-kernel Philox4x32-7 + Box-Muller (16-bit uniforms), step noise rounded to fp16",
                       "arithmetic": "fp32 tensors in HBM, fp32 accumulation; 3x3 / 5x5 / Linear products formed on the f16 matrix pipe from "
                                     "exact two-term fp16 splits of both operands (22-bit products, same parity tolerance as the exact-fp32 "
                                     "kernels; gencomm_set_mode(GENCOMM_MODE_ARITH, 1) selects those: see exact_fp32_mode)" if lib.gencomm_get_mode(_lib.MODE_ARITH) == 0
                                     else "exact fp32 MFMA kernels (GENCOMM_MODE_ARITH = 1)",
                       "streams_per_gpu": S, "scenes_per_step": B, "hip_graph": bool(args.graph), "parallelism": f"replicas x{world} (scene-sharded, no collective)"},
            "scene_algorithmic": {"gflop": flops / 1e9, "gbyte": byts / 1e9,
                                  "achieved_tflops": flops * args.steps * B / elapsed / 1e12,
                                  "achieved_gbs": byts * args.steps * B / elapsed / 1e9},
        }
        # roofline of the dominant kernel: algorithmic FLOPs per launch / measured launch time
        roof = None
        if timed and k_n.value > 0:
            fam = args.timer_family
            name = lib.gencomm_timer_kernel_name(fam).decode()
            per_launch_ms = k_ms.value / k_n.value
            # MACs per agent-pixel the kernel executes: conv_in 72(C+2); conv_out 72C; latent step = 5x5 composite
            # (8*8*25) + noise conv (72C) -- DESIGN.md section 4
            macs_px = {0: 72.0 * (C + 2), 5: 72.0 * C, 15: 1600.0 + 72.0 * C}.get(fam)
            traffic = None
            pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get(name, {}).get("hbm_bytes_per_launch")  # measured at 1 scene/launch
                    traffic = traffic * B if traffic is not None else None
                except Exception:
                    traffic = None
            if macs_px is not None:
                fl = 2.0 * macs_px * HW * N * B
                # primary figure = the kernel's own duration: HIP events around every launch of two extra pipeline runs
                # with ONE scene batch in flight. Inside the timed region two streams share the chip, so an event pair
                # there also counts the time a launch waits for the other stream's kernels to drain; rocprofv3 (which
                # serialises dispatches) reports the same duration as the single-stream pass, see profiles/.
                own_ms = iso_ms.value / iso_n.value if iso_n.value else per_launch_ms
                roof = {"kernel": name, "bound": "mfma", "achieved": fl / (own_ms * 1e-3) / 1e12,
                        "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / (own_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                        "traffic": traffic, "launches": iso_n.value if iso_n.value else k_n.value, "avg_launch_ms": own_ms,
                        "flops_per_launch": fl,
                        "in_timed_region": {"launches": k_n.value, "avg_launch_ms": per_launch_ms,
                                            "achieved": fl / (per_launch_ms * 1e-3) / 1e12,
                                            "frac": fl / (per_launch_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                            "note": f"event pairs with {S} streams sharing the chip: includes queueing behind the other stream"},
                        "note": "latent_step_h_kernel = conv_out + sampler update + conv_in of one step fused by linearity; "
                                "algorithmic FLOPs = 2*(1600 + 72*C) per agent-pixel, priced against the 157.3 TFLOP/s fp32 "
                                "matrix/vector peak (the dtype of the path). The products run on the f16 matrix pipe from exact "
                                "fp16 hi/lo operand splits (3 v_mfma_f32_16x16x32_f16 per fp32 product block, fp32 accumulate, "
                                "22-bit products: same parity tolerance as the exact-fp32 kernel, GENCOMM_CONV8=f32); the kernel "
                                "is now bound by the in-kernel Philox4x32-10 + Box-Muller VALU work, not by the matrix pipe. "
                                "duration = HIP events on the launch stream, one scene batch in flight (agrees with the "
                                "rocprofv3 --kernel-trace --stats average)"}
                if sec_n.value > 0:
                    # 16 -> 8 channel layers (conv1 of the up blocks): per UNet call 3 launches per level, level l at 1/4^l of the
                    # pixels; algorithmic bytes = read 16 channels + write 8 channels, fp32
                    L = len(gen.denoiser.ch_mult) if hasattr(gen.denoiser, "ch_mult") else 2
                    lvl = sum(0.25 ** l for l in range(L)) / L
                    bytes_launch = 4.0 * 24 * HW * N * B * lvl
                    s_ms = sec_ms.value / sec_n.value
                    out["roofline_conv16"] = {
                        "kernel": "conv8h_kernel<NSRC=2> (16 -> 8 ch 3x3 + GroupNorm + SiLU, all levels)", "bound": "hbm",
                        "achieved": bytes_launch / (s_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": bytes_launch / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "launches": sec_n.value, "avg_launch_ms": s_ms, "bytes_per_launch_avg": bytes_launch,
                        "note": "average over the launches of one scene batch (full- and half-resolution levels); algorithmic "
                                "bytes = 4 B * (16 read + 8 written channels) * pixels * agents"}
            else:
                roof = {"kernel": name, "launches": k_n.value, "avg_launch_ms": per_launch_ms}
        out["roofline"] = roof
        out["exact_fp32_mode"] = exact
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, N, C, H, W, T, gen, enh, ptm)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
