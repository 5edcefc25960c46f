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

`python bench.py --gpus N` without torch.distributed.run starts its own N ranks (gencomm_amd/launch.py) before the parent
makes any GPU call; under torch.distributed.run the environment it set is used as is. `--share-device` puts every rank on
cuda:0 with a gloo process group (rehearsal on a 1-GPU box; RCCL refuses two ranks on one device).

`--workload train` is BASELINE.json configs[3]: one optimiser step of the stage-1 recipe (PointPillars -> backbone -> message
extractor -> GenComm training branch -> Enhancer -> AttFusion -> heads, PointPillarGencommLoss, Adam) per step and rank,
DistributedDataParallel(find_unused_parameters=True) over RCCL (train_ddp.py:121-125), scenes sharded over the ranks.
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
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16/f16 MFMA peak (the 5 PF headline includes 2:1 sparsity)
MFMA_TERMS = 6.0          # matrix instructions per fp32-grade product block (conv8h_kernels.h header)

# Kernel families of the library's timer (include/gencomm_hip.h, csrc/common.h KernelFamily).  The dominant kernel is the
# conv8h_kernel template -- the 8-channel 3x3 layers of the UNet (ResnetBlock conv1 / conv2, Upsample), ~55 % of kernel time in
# profiles/r2_*_kernel_stats*.csv, HBM-bound: the headline `roofline`.  The latent sampler step (the largest single
# instantiation, ~14 %, VALU/MFMA-bound) is reported beside it as `roofline_latent_step`.
CONV8_FAMILIES = {1: "conv1 8->8 (GN+SiLU)", 16: "conv2 + identity residual", 17: "conv2 + 1x1 nin_shortcut", 2: "conv1 16->8 (skip concat)",
                  4: "Upsample conv (nearest x2)"}
LATENT_FAMILY = 15
ENH_FRONT_FAMILY = 9
N_FAMILIES = 20
CONV2D_FAMILY = 19      # the general convolution around the path (conv2d_h3l / conv2d_h3 / conv2d_igemm kernels): its timer slot carries algorithmic FLOPs
RESULT_OUT = sys.stdout   # main() replaces it by a private copy of the original stdout


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


def cpu_baseline(name, N, C, H, W, T, gen, enh, ptm, timed_steps=5):
    """The CPU oracle (plain PyTorch restatement of the reference, oracle/torch_port.py) on this host's cores, on a bounded
    sample of the same workload: after one untimed warm-up step, `timed_steps` of the T denoise steps are timed one by one
    at full tensor size (every step has identical cost), q_sample once, Enhancer and fusion twice each; the scene time is
    q_sample + T * mean(step) + enhancer + fusion, reported with the 95 % interval that the step-to-step spread implies."""
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
    # thread count: ATen's CPU convolutions on these 8-channel maps stop scaling (and then slow down badly) well below the
    # core count of a 2-socket host, so calibrate on one denoise step and keep the fastest of a few candidates (this doubles
    # as the warm-up); `cores` reports the threads actually used
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
    steps = []
    with torch.no_grad():
        t0 = time.perf_counter()
        x = O.q_sample(sched, O.ego_repeat(feat, [N]), T - 1, noise)
        t_q = time.perf_counter() - t0
        x = O.p_sample(sd_g, sched, cfg["model"], cond, x, T - 1, noise)  # warm-up at the chosen thread count
        for i in range(timed_steps):
            t0 = time.perf_counter()
            x = O.p_sample(sd_g, sched, cfg["model"], cond, x, T - 2 - i, noise)
            steps.append(time.perf_counter() - t0)
        t_e, t_f = [], []
        affine = O.normalize_pairwise_tfm(ptm, H * PX_M, W * PX_M, 1.0)
        for _ in range(2):
            t0 = time.perf_counter()
            e = O.enhancer_forward(sd_e, x, [N])
            t_e.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            O.att_fusion(e, [N], affine)
            t_f.append(time.perf_counter() - t0)
    mean = float(np.mean(steps))
    sem = float(np.std(steps, ddof=1) / np.sqrt(len(steps))) if len(steps) > 1 else 0.0
    tcrit = {2: 12.71, 3: 4.303, 4: 3.182, 5: 2.776, 6: 2.571, 7: 2.447, 8: 2.365}.get(len(steps), 2.0)  # Student t, 95 %
    rest = t_q + min(t_e) + min(t_f)
    scene_s = rest + T * mean
    lo, hi = rest + T * (mean - tcrit * sem), rest + T * (mean + tcrit * sem)
    return {"value": 1.0 / scene_s, "unit": "scenes/sec", "cores": cores, "kind": "port",
            "ci95": [1.0 / hi, 1.0 / max(lo, 1e-9)],
            "sample": f"workload '{name}': q_sample {t_q:.2f} s + 1 warm-up and {len(steps)} individually timed denoise steps of {T} "
                      f"(mean {mean:.3f} s, standard error {sem:.3f} s, extrapolated x{T}) + enhancer {min(t_e):.2f} s + fusion {min(t_f):.2f} s "
                      f"(best of 2), torch {torch.__version__} CPU, {cores} threads (fastest of 8/16/32/64 on one step; host exposes "
                      f"{avail} logical CPUs); value = 1 / scene time, ci95 from the step-to-step spread",
            "scene_seconds": scene_s}


def cpu_train_baseline(B, N, C, H, W, T, model):
    """CPU baseline of the TRAINING leg: the hot-path part of the stage-1 step -- GenComm's training branch (per-agent chains, T
    steps, cond_diff.py:342-360) -> Enhancer -> AttFusion -> the generation loss MSE(pred, gt) (point_pillar_gencomm_loss.py:46-52),
    forward + backward through torch autograd -- on the oracle (oracle/torch_port.py) with this model's weights, one warm-up and
    `reps` timed steps at the leg's shapes.  The encoder / backbone / heads AROUND the path have no training-mode restatement in the
    oracle (its BatchNorm is the eval form), so they are not part of this number: it is the cost of the path the HIP kernels replace."""
    from gencomm_amd import synth
    from oracle import torch_port as O
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(16, avail)
    torch.set_num_threads(cores)
    cfg = synth.default_gencomm_cfg(C, T)
    sd_g = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in model.gencomm.state_dict().items()}
    sd_e = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in model.enhancer.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    n = B * N
    feat = torch.randn(n, C, H, W, generator=g).clamp_(min=0).requires_grad_(True)
    cond = torch.randn(n, 2, H, W, generator=g).requires_grad_(True)
    n0, sn = torch.randn(n, C, H, W, generator=g), torch.randn(T, n, C, H, W, generator=g)
    ptm = torch.from_numpy(synth.make_pairwise_t_matrix([N] * B, 5, 10, max_shift=20.0))
    affine = O.normalize_pairwise_tfm(ptm, H * 1.6, W * 1.6, 1.0)
    times, reps = [], 3
    for i in range(reps + 1):
        t0 = time.perf_counter()
        pred = O.gencomm_forward(sd_g, cfg, feat, cond, [N] * B, n0, sn, per_agent=True)
        fused = O.att_fusion(O.enhancer_forward(sd_e, pred, [N] * B), [N] * B, affine)
        loss = torch.nn.functional.mse_loss(pred, feat.detach()) + fused.pow(2).mean()     # the generation loss + a stand-in for the heads' gradient
        loss.backward()
        dt = time.perf_counter() - t0
        if i > 0:
            times.append(dt)
    step_s = float(np.mean(times))
    return {"value": B / step_s, "unit": "train scenes/sec", "cores": cores, "kind": "port", "step_seconds": step_s,
            "sample": f"hot-path part of the step only (GenComm training branch T={T} -> Enhancer -> AttFusion -> MSE, forward + backward by torch "
                      f"autograd on oracle/torch_port.py), {B} scene(s) x {N} agents, C={C}, {H}x{W}; 1 warm-up + {reps} timed steps, torch {torch.__version__} "
                      f"CPU, {cores} threads (host exposes {avail}); encoder / backbone / heads / optimiser are NOT in it (no training-mode restatement)"}


def selftest_main(args, rank, world):
    """`--workload launch_selftest`: the launcher, the rank environment, the process group (gloo, CPU only -- no GPU call anywhere),
    the barrier-bracketed timed region and the aggregation of bench.py with a stand-in step (rank r "processes" `--batch` scenes
    per step in 10 ms * (1 + r)). tests/test_launch.py runs this through `python bench.py --gpus 2 --workload launch_selftest`."""
    from gencomm_amd import dist as gdist
    if os.environ.get("GENCOMM_SELFTEST_FAIL_RANK") == str(rank):
        print(f"rank {rank}: failing on request", file=sys.stderr)
        raise SystemExit(3)
    dist = gdist.init_process_group("gloo")
    cpu = torch.device("cpu")
    B = max(1, args.batch)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.01 * (1 + rank))
    busy = time.perf_counter() - t0          # this rank's own work; the closing barrier makes every rank wait for the slowest
    if dist is not None:
        dist.barrier()
    elapsed_here = time.perf_counter() - t0
    per_rank = [None] * world
    if dist is not None:
        dist.all_gather_object(per_rank, {"rank": rank, "scenes": args.steps * B, "elapsed": elapsed_here, "busy": busy, "pid": os.getpid(),
                                          "env": {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "GENCOMM_LAUNCHED")}})
    else:
        per_rank = [{"rank": 0, "scenes": args.steps * B, "elapsed": elapsed_here, "busy": busy, "pid": os.getpid(), "env": {}}]
    value, elapsed, total = gdist.aggregate_throughput(args.steps * B, elapsed_here, dist, cpu)
    if rank == 0:
        print(json.dumps({"metric": "scenes/sec", "value": value, "unit": "scenes/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * elapsed / args.steps, "total_scenes": total, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "synthetic",
                          "config": {"workload": "launch_selftest: stand-in steps on the CPU (gloo); exercises launcher + aggregation only"},
                          "ranks": per_rank}), file=RESULT_OUT, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def train_main(args, rank, world, device, backend):
    """`--workload train` (BASELINE.json configs[3]): the stage-1 recipe's optimiser step under DistributedDataParallel.

    Model = HeterModelBaselineWGenCommStage1 built from the model block of opv2v/GenComm_yamls/gencomm/stage1/m1_att.yaml
    (512 x 256 pillars of 0.4 m -> BaseBEVBackbone [3,5,8] -> 128 x 64 x 128 -> message extractor -> GenComm T = 3 training branch
    -> Enhancer -> AttFusion -> heads), criterion PointPillarGencommLoss with the yaml's weights, Adam(lr 2e-3, eps 1e-10,
    weight_decay 1e-4). Every rank runs `--train-batch` scenes of `--train-agents` agents per step on its own synthetic shard
    (train_ddp.py:62-66 DistributedSampler); gradients are averaged by DDP's bucketed all-reduce over RCCL (train_ddp.py:121-125,
    find_unused_parameters=True: Enhancer blocks 2/3 and the attention tables never receive gradients). The process group is
    created at world size 1 too, so the N = 1 line exercises the same RCCL communicator + DDP reducer code."""
    from gencomm_amd import dist as gdist
    from gencomm_amd import synth
    from gencomm_amd.heter_model_baseline_w_gencomm_stage1 import HeterModelBaselineWGenCommStage1
    from gencomm_amd.point_pillar_gencomm_loss import PointPillarGencommLoss
    dist = gdist.init_process_group(backend, device, force=True)
    red_dev = device if backend == "nccl" else torch.device("cpu")
    B, N, T = max(1, args.train_batch), max(1, args.train_agents), 3
    margs = synth.stage1_model_args(T=T)
    torch.manual_seed(0)                       # identical initial weights on every rank (DDP also broadcasts rank 0's)
    model = HeterModelBaselineWGenCommStage1(margs)
    synth.fill_params_(model, 3)
    synth.fill_bn_stats_(model, 4)
    model = model.to(device).train()
    # gradient_as_bucket_view: the gradients live in the all-reduce buckets (no per-parameter copy in and out of them: ~770 framework
    # launches per step less in profiles/r4_train_leg_kernel_stats.csv); same arithmetic as the reference's plain DDP wrapper
    sync = None
    if args.grad_sync == "ddp":
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[device.index], find_unused_parameters=True, gradient_as_bucket_view=True)
    else:          # one flat bucket per step (gencomm_amd/dist.py FlatGradSync): the same average, 4 launches instead of 2 per parameter
        ddp = model
        sync = gdist.FlatGradSync(model.parameters(), dist, module=model)   # broadcasts rank 0's parameters and buffers
    crit = PointPillarGencommLoss(synth.STAGE1_LOSS_ARGS)
    params = [p for p in model.parameters() if p.requires_grad]
    # m1_att.yaml:191-196; capturable: the step counter lives on the device, so that the whole step can be recorded into a HIP graph
    opt = torch.optim.Adam(params, lr=2e-3, eps=1e-10, weight_decay=1e-4, fused=True, capturable=bool(args.graph))
    rng = margs["lidar_range"]
    n = B * N
    pil = synth.make_pillars(12000 * n, n, 512, 256, 9 + 31 * rank, voxel_size=[0.4, 0.4, 4.0], pc_range=rng)
    ptm = synth.make_pairwise_t_matrix([N] * B, 5, 10 + rank, max_shift=20.0)
    data = {"agent_modality_list": ["m1"] * n, "record_len": torch.tensor([N] * B), "pairwise_t_matrix": torch.from_numpy(ptm).to(device),
            "inputs_m1": {k: torch.from_numpy(pil[k]).to(device) for k in ("voxel_features", "voxel_coords", "voxel_num_points")}}
    labels = None

    def step():
        nonlocal labels
        opt.zero_grad(set_to_none=True)
        out = ddp(data)
        if labels is None:                     # anchor labels of the head geometry, once
            _, A, Hh, Wh = out["cls_preds"].shape
            li = synth.make_loss_inputs(50 + rank, B, Hh, Wh, A, 1)
            labels = {k: torch.from_numpy(li[k]).to(device) for k in ("pos_equal_one", "neg_equal_one", "targets")}
        loss = crit(out, labels)
        loss.backward()
        if sync is not None:
            sync.sync()
        opt.step()
        return loss

    def barrier():
        torch.cuda.synchronize(device)
        dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(max(1, args.warmup)):
        step()
    barrier()
    eager_step, graph_note = step, None
    if args.graph:
        # --graph 1: forward + backward + gradient averaging + Adam of ONE step recorded into a HIP graph on a side stream (after the
        # eager warm-up: the gradient bucket's layout is agreed and every cache of the library is warm) and replayed per step -- the
        # ~1 600 launches of a step cost one hipGraphLaunch on the host.  Same kernels, same arithmetic; inputs are the static synthetic
        # shard.  The reference's loop is eager PyTorch (train_ddp.py:173-197); this is the MI355X-side answer to a host-bound step.
        g = torch.cuda.CUDAGraph()
        cs = torch.cuda.Stream(device=device)
        cs.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(cs):
            for _ in range(2):     # on the capture stream first: allocator pools and autograd's streams settle
                eager_step()
        torch.cuda.current_stream(device).wait_stream(cs)
        barrier()
        with torch.cuda.graph(g, stream=cs):
            static_loss = eager_step()

        def step():
            g.replay()
            return static_loss
        graph_note = "one HIP graph per step (torch.cuda.CUDAGraph: forward + backward + flat gradient bucket + fused Adam), replayed"
        barrier()
    if args.sync_debug:      # diagnostic: one step with torch's synchronisation detector (every host <-> device sync point warns with a stack)
        torch.cuda.set_sync_debug_mode("warn")
        step()
        torch.cuda.set_sync_debug_mode("default")
        barrier()
    if args.op_census:       # diagnostic: which Python lines issue the step's small framework operators (wrappers around the torch entry points
        import collections   # that launch fill / copy / elementwise kernels; the caller's file:line is the key), to stderr
        torch.autograd.set_multithreading_enabled(False)
        sites, saved = collections.Counter(), []

        def wrap(owner, name):
            orig = getattr(owner, name)

            def w(*a, **k):
                f = sys._getframe(1)
                while f is not None and ("bench.py" in f.f_code.co_filename and f.f_code.co_name == "w"):
                    f = f.f_back
                sites[(name, f"{os.path.basename(f.f_code.co_filename)}:{f.f_lineno} {f.f_code.co_name}")] += 1
                return orig(*a, **k)
            saved.append((owner, name, orig))
            setattr(owner, name, w)
        for nm in ("fill_", "zero_", "copy_", "contiguous", "clone", "float", "sum", "__mul__", "__add__", "__iadd__", "__sub__", "__gt__", "mul", "add", "view_as", "t", "reshape"):
            wrap(torch.Tensor, nm)
        for nm in ("zeros", "ones", "full", "zeros_like", "ones_like", "empty_like", "cat", "stack", "rsqrt"):
            wrap(torch, nm)
        step()
        for owner, name, orig in saved:
            setattr(owner, name, orig)
        torch.autograd.set_multithreading_enabled(True)
        barrier()
        for (name, where), c in sites.most_common(80):
            print(f"{c:5d}  {name:12s} {where}", file=sys.stderr)
    if args.dispatch_census:   # diagnostic: every ATen operator of one step as the dispatcher sees it (autograd-engine nodes included), with the
        import collections, traceback   # innermost frame of this repository on the Python stack at that moment, to stderr
        from torch.utils._python_dispatch import TorchDispatchMode
        seen, allops = collections.Counter(), collections.Counter()

        class Census(TorchDispatchMode):
            def __torch_dispatch__(self, func, types, a=(), kw=None):
                name = str(func).replace("aten.", "")
                allops[name] += 1
                if any(k in name for k in ("fill", "zero", "copy", "clone", "zeros", "add", "mul", "cat", "contiguous", "sum", "index", "select_backward", "slice_backward")):
                    fr = [f for f in traceback.extract_stack(limit=40) if "/gencomm_amd/" in f.filename or f.filename.endswith("bench.py")]
                    fr = [f for f in fr if f.name not in ("__torch_dispatch__",)]
                    where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in fr[-2:][::-1]) or "?"
                    shp = next((tuple(t.shape) for t in a if isinstance(t, torch.Tensor)), ())
                    seen[(name, str(shp), where)] += 1
                return func(*a, **(kw or {}))
        torch.autograd.set_multithreading_enabled(False)
        with Census():
            step()
        torch.autograd.set_multithreading_enabled(True)
        barrier()
        print("all operators of the step: " + ", ".join(f"{n} x{c}" for n, c in allops.most_common(60)), file=sys.stderr)
        for (name, shp, where), c in seen.most_common(90):
            print(f"{c:5d}  {name:28s} {shp:26s} {where}"[:250], file=sys.stderr)
    if args.op_stacks:       # diagnostic: torch.profiler over one step -- the framework operators behind the fill / copy launches, with Python stacks
        from torch.profiler import ProfilerActivity, profile
        torch.autograd.set_multithreading_enabled(False)
        with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
            step()
        torch.autograd.set_multithreading_enabled(True)
        barrier()
        import collections
        agg = collections.Counter()
        for e in prof.events():
            if e.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::clone", "aten::zeros", "aten::zeros_like", "aten::contiguous", "aten::add_", "aten::add", "aten::mul"):
                st = [f for f in (e.stack or []) if "site-packages" not in f and "dist-packages" not in f and "<built-in" not in f][:3]
                shp = str(e.input_shapes[:1]) if e.input_shapes else ""
                agg[(e.name, " <- ".join(x.split("/")[-1] for x in st) or "(autograd engine / C++)", shp)] += 1
        totals = collections.Counter()
        for (name, _, _), c in agg.items():
            totals[name] += c
        print("operators of the step by name: " + ", ".join(f"{n} x{c}" for n, c in totals.most_common()), file=sys.stderr)
        for (name, where, shp), c in agg.most_common(160):
            print(f"{c:5d}  {name:18s} {shp:28s} {where}"[:230], file=sys.stderr)
    if args.host_profile:    # diagnostic: cProfile of the host side of 10 steps (autograd on this thread), written to stderr
        import cProfile, io, pstats
        torch.autograd.set_multithreading_enabled(False)
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(10):
            step()
        pr.disable()
        torch.autograd.set_multithreading_enabled(True)
        barrier()
        for key in ("tottime", "cumulative"):
            sio = io.StringIO()
            pstats.Stats(pr, stream=sio).sort_stats(key).print_stats(60)
            print("\n".join(l[:180] for l in sio.getvalue().split("\n")), file=sys.stderr)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    enqueue_here = time.perf_counter() - t0      # the host is done issuing; close to the elapsed time = the step is host-bound
    barrier()
    elapsed_here = time.perf_counter() - t0
    assert torch.isfinite(loss.detach()).all(), "non-finite training loss"
    value, elapsed, total_scenes = gdist.aggregate_throughput(args.steps * B, elapsed_here, dist, red_dev)
    grad_bytes = sum(p.numel() * p.element_size() for p in params) if sync is None else sync.bucket_bytes
    # roofline of the step's dominant kernel family -- the general convolution (BEV backbone, shrink conv, heads, the Enhancer's Linear
    # layers; forward and input gradients): HIP-event time and algorithmic FLOPs of every launch of ONE extra eager step (the library's
    # kernel timer, family CONV2D_FAMILY), priced against the f16 pipe's peak over the six matrix instructions of a product block
    roofline = None
    if rank == 0:
        from gencomm_amd import _lib
        lib = _lib.lib()
        ms = (ctypes.c_double * N_FAMILIES)()
        cnt = (ctypes.c_int * N_FAMILIES)()
        wk = (ctypes.c_double * N_FAMILIES)()
        _lib.check(lib.gencomm_timer_start_mask(1 << CONV2D_FAMILY, 4096), "gencomm_timer_start_mask")
    eager_step()        # EVERY rank runs the extra step (it contains the gradient collective); only rank 0's launches are timed
    barrier()
    if rank == 0:
        _lib.check(lib.gencomm_timer_stop_families(ms, cnt, wk, N_FAMILIES), "gencomm_timer_stop_families")
        if cnt[CONV2D_FAMILY] > 0:
            f_ms, f_fl, f_n = ms[CONV2D_FAMILY], wk[CONV2D_FAMILY], cnt[CONV2D_FAMILY]
            ach = f_fl / (f_ms * 1e-3) / 1e12
            roofline = {"kernel": "conv2d_h3l_kernel / conv2d_h3_kernel<KH, KW, ...> (general 3x3 / 2x2 convolutions of the step on the f16 matrix "
                                  "pipe, six matrix instructions per product block from exact three-term operand splits: csrc/conv_h3_kernels.h; the few "
                                  "shapes it does not take -- 1x1, fewer than 32 GEMM rows or 16 input channels -- run conv2d_igemm_kernel on v_mfma_f32_32x32x2_f32)",
                        "bound": "mfma", "achieved": ach, "peak": F16_MFMA_PEAK_TFLOPS / MFMA_TERMS, "unit": "TFLOP/s (fp32-equivalent)",
                        "frac": ach * MFMA_TERMS / F16_MFMA_PEAK_TFLOPS, "traffic": None, "launches": f_n, "avg_launch_ms": f_ms / f_n,
                        "flops_per_launch": f_fl / f_n, "share_of_step": f_ms / (1e3 * elapsed / args.steps),
                        "frac_of_fp32_mfma_peak": ach / FP32_PEAK_TFLOPS,
                        "note": "algorithmic FLOPs = 2 N Ho Wo Cout Cin KH KW per launch (host-computed from the launch shape), HIP events around "
                                "every launch of one untimed eager step (launches of the side streams run beside them: durations are not "
                                "isolated-kernel times); peak = the f16 pipe's 2 500 TFLOP/s over the six matrix instructions of a product block; "
                                "share_of_step = the family's summed device time over the step's wall time"}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_train_baseline(B, N, 128, 64, 128, T, model)
    if rank == 0:
        from gencomm_amd import _lib as _lib_modes
        d = crit.logging(0, args.steps - 1, args.steps)
        print(json.dumps({
            "metric": "train scenes/sec", "value": value, "unit": "scenes/sec", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_scene_per_gpu": 1e3 * elapsed / args.steps / B, "total_scenes": total_scenes,
            "host_enqueue_ms_per_step": 1e3 * enqueue_here / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"train: stage-1 recipe (opv2v/GenComm_yamls/gencomm/stage1/m1_att.yaml), {N} agents/scene, {B} scene(s)/step/GPU, "
                                   f"512x256 pillars -> 128x64x128 BEV, GenComm T={T}, forward + backward + Adam",
                       "modes": {k: int(_lib_modes.lib().gencomm_get_mode(v)) for k, v in (("arith", _lib_modes.MODE_ARITH), ("bwd_streams", _lib_modes.MODE_BWD_STREAMS))},
                       "parallelism": (f"dp{world} (DistributedDataParallel, find_unused_parameters=True)" if sync is None else
                                       f"dp{world} (one flat gradient bucket per step: cat -> all_reduce -> scale -> multi-tensor copy)"),
                       "grad_sync": args.grad_sync, "process_group": backend, "hip_graph": graph_note,
                       "rccl_ranks": dist.get_world_size() if backend == "nccl" else 0, "share_device": backend != "nccl",
                       "grad_bytes_allreduced_per_step": grad_bytes, "trainable_parameters": sum(p.numel() for p in params),
                       "ddp_bucket_cap_mb": 25 if sync is None else None, "pillars_per_agent": 12000},
            "loss": d, "roofline": roofline, "cpu_baseline": cpu}), file=RESULT_OUT, flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="metric", choices=sorted(WORKLOADS) + ["train", "launch_selftest"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-enhancer", action="store_true")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-fp32 pass (profiling runs)")
    ap.add_argument("--no-timer", action="store_true",
                    help="profiling runs: only warm-up + the K-step region (no per-kernel HIP-event passes, no >= 1 s repeat, no single-scene latency leg)")
    ap.add_argument("--batch", type=int, default=4, help="scenes per step (batched in one launch sequence, record_len=[N]*B)")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent scenes in flight per GPU, each on its own HIP stream with its own buffers")
    ap.add_argument("--sustain", type=float, default=12.0,
                    help="seconds of the sustained repeat of the timed loop (reported beside `value`; long enough for a 5 s utilisation sampler)")
    ap.add_argument("--graph", type=int, default=0, help="1: replay the scene's launch sequence as a captured HIP graph")
    ap.add_argument("--mode", action="append", default=[], metavar="KEY=VALUE",
                    help="library mode for this run (gencomm_set_mode; keys: arith sampler tile_want enh_fuse conv8h_mask xcd dataflow resfuse_emu tile8), e.g. --mode xcd=0")
    ap.add_argument("--share-device", action="store_true",
                    help="with --gpus N > 1 on a 1-GPU box: every rank on cuda:0, gloo process group (launcher / DDP rehearsal, not a scaling number)")
    ap.add_argument("--op-census", action="store_true", help="--workload train: call sites of the fill / zero / copy operators of one step (torch.profiler), to stderr")
    ap.add_argument("--dispatch-census", action="store_true", help="--workload train: ATen operators of one step at the dispatcher (fill / copy / add / ...), by shape and repository frame, to stderr")
    ap.add_argument("--op-stacks", action="store_true", help="--workload train: torch.profiler over one step, fill / copy / add operators grouped by Python stack, to stderr")
    ap.add_argument("--host-profile", action="store_true", help="--workload train: cProfile of the host side of 10 untimed steps, to stderr")
    ap.add_argument("--sync-debug", action="store_true", help="run one untimed step under torch.cuda.set_sync_debug_mode('warn'): every host <-> device synchronisation warns with its stack")
    ap.add_argument("--grad-sync", choices=["flat", "ddp"], default="flat",
                    help="--workload train: gradient averaging -- one flat bucket per step (gencomm_amd.dist.FlatGradSync) or torch DistributedDataParallel")
    ap.add_argument("--train-batch", type=int, default=2, help="--workload train: scenes per rank and step (m1_att.yaml batch_size: 2)")
    ap.add_argument("--train-agents", type=int, default=2, help="--workload train: agents per scene")
    ap.add_argument("--launch-timeout", type=float, default=None, help="seconds before the self-launcher stops its ranks")
    args = ap.parse_args()

    from gencomm_amd import dist as gdist
    from gencomm_amd import launch
    if launch.needs_launch(args.gpus):
        # N fresh ranks of this very command line, started before this process has made any GPU call; it never makes one
        argv = [a for a in sys.argv[1:] if a != "--share-device"]
        raise SystemExit(launch.self_launch(os.path.abspath(__file__), argv, args.gpus, share_device=args.share_device, timeout=args.launch_timeout))
    rank, world, local_rank = gdist.env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    # stdout carries exactly ONE line, the result: native libraries (RCCL's version banner, gloo's connection notes) write to file
    # descriptor 1 behind Python's back, so fd 1 is pointed at stderr for the rest of the run and the result goes to a saved copy
    global RESULT_OUT
    sys.stdout.flush()
    RESULT_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.workload == "launch_selftest":
        return selftest_main(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU"
    dev_index = launch.device_index(local_rank)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = launch.backend()
    from gencomm_amd import _lib, normalize_pairwise_tfm
    lib = _lib.lib()
    mode_keys = {"arith": _lib.MODE_ARITH, "sampler": _lib.MODE_SAMPLER, "tile_want": _lib.MODE_TILE_WANT,
                 "enh_fuse": _lib.MODE_ENH_FUSE, "conv8h_mask": _lib.MODE_CONV8H_MASK, "xcd": _lib.MODE_XCD_REMAP,
                 "dataflow": _lib.MODE_DATAFLOW, "resfuse_emu": _lib.MODE_RESFUSE_EMU, "tile8": _lib.MODE_TILE8, "bwd_streams": _lib.MODE_BWD_STREAMS, "persist": _lib.MODE_PERSIST}
    for kv in args.mode:      # (before the training leg branches off: until the end of round 5 `--mode` was applied behind it and the leg ignored it)
        k, v = kv.split("=")
        _lib.check(lib.gencomm_set_mode(mode_keys[k], int(v)), "gencomm_set_mode")
    if args.workload == "train":
        return train_main(args, rank, world, device, backend)
    dist = gdist.init_process_group(backend, device)  # RCCL; only the timing barrier/reduction use it
    red_dev = device if backend == "nccl" else torch.device("cpu")   # gloo reduces host scalars

    from gencomm_amd.pipeline import ScenePipeline

    N, C, H, W, T = WORKLOADS[args.workload]
    HW = H * W
    gen, enh = build_modules(C, T, device)
    # S scene batches in flight: the path is a chain of ~600 short dependent launches per scene batch, so independent
    # batches on separate streams fill each other's latency gaps. Every stream has its own inputs, workspace and outputs.
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

    def timed_region(steps, seed0, use_barrier=True):
        """`steps` steps bracketed by barrier + synchronize on both sides; returns elapsed seconds."""
        (barrier if use_barrier else (lambda: torch.cuda.synchronize(device)))()
        t0 = time.perf_counter()
        for i in range(steps):
            run_scene(i, seed0 + i)
        (barrier if use_barrier else (lambda: torch.cuda.synchronize(device)))()
        return time.perf_counter() - t0

    def family_pass(runs=2):
        """Per-kernel-family device time, launches and algorithmic bytes (HIP events on the launch stream, one scene
        batch in flight so that an event pair brackets the kernel alone)."""
        ms = (ctypes.c_double * N_FAMILIES)()
        cnt = (ctypes.c_int * N_FAMILIES)()
        byt = (ctypes.c_double * N_FAMILIES)()
        _lib.check(lib.gencomm_timer_start_mask((1 << N_FAMILIES) - 1, runs * (T + 4) * 40), "gencomm_timer_start_mask")
        with torch.no_grad(), _lib.kernel_log() as kl:
            for i in range(runs):
                pipes[0].run(scenes[0][0], scenes[0][1], seed=3000 + i)
        torch.cuda.synchronize(device)
        _lib.check(lib.gencomm_timer_stop_families(ms, cnt, byt, N_FAMILIES), "gencomm_timer_stop_families")
        family_pass.instantiations = {k: v // runs for k, v in kl.counts.items()}   # launches per scene batch
        return {f: {"ms": ms[f], "launches": cnt[f], "bytes": byt[f], "name": lib.gencomm_timer_kernel_name(f).decode()}
                for f in range(N_FAMILIES) if cnt[f] > 0}

    def rooflines(fam, arith_note):
        """Headline roofline of the conv8h_kernel template (HBM) and the latent step beside it, from one family pass."""
        out = {}
        tot_ms = sum(v["ms"] for v in fam.values())
        conv = {f: fam[f] for f in CONV8_FAMILIES if f in fam}
        if conv:
            c_ms, c_b, c_n = sum(v["ms"] for v in conv.values()), sum(v["bytes"] for v in conv.values()), sum(v["launches"] for v in conv.values())
            # counter traffic is a committed measurement, valid only for the kernels it was taken on: the JSON carries the source
            # stamp of the library it ran (gencomm_build_info() " src=..."); any other library -> null
            traffic, traffic_src = None, None
            for pmc in ("r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json"):
                pmc = os.path.join(REPO, "profiles", pmc)
                if not os.path.exists(pmc):
                    continue
                try:
                    tj = json.load(open(pmc))
                    if (tj.get("workload") == args.workload and tj.get("scenes_per_launch") == B
                            and tj.get("library_src") and tj.get("library_src") == _lib.library_src_hash()):
                        traffic, traffic_src = tj["conv8h_family"]["hbm_bytes_per_launch"], os.path.basename(pmc)
                        break
                except Exception:
                    traffic = None
            # the second resource: vector-instruction issue, from the committed SQ counter pass of the same command (stamped like the
            # traffic): fraction of a wave's resident cycles with an instruction in flight x the three waves a SIMD holds
            issue = None
            for sq_name in ("r5_pmc_sq.json", "r4_pmc_sq.json"):
                sq = os.path.join(REPO, "profiles", sq_name)
                if issue is not None or not os.path.exists(sq):
                    continue
                try:
                    sj = json.load(open(sq))
                    # an unstamped library (hash "") must never match an unstamped JSON: both sides have to carry a stamp
                    if (sj.get("workload") == args.workload and sj.get("library_src")
                            and sj.get("library_src") == _lib.library_src_hash()):
                        ks = {k: v for k, v in sj["kernels"].items() if "conv8h_kernel<" in k}
                        w = sum(v["launches"] * v["raw_means"]["SQ_WAVE_CYCLES"] for v in ks.values())
                        act = sum(v["launches"] * v["raw_means"]["SQ_ACTIVE_INST_ANY"] for v in ks.values()) / w
                        wait = sum(v["launches"] * v["raw_means"]["SQ_WAIT_ANY"] for v in ks.values()) / w
                        issue = {"active_inst_frac_per_wave": act, "waves_per_simd": 3, "simd_issue_busy": min(1.0, 3 * act),
                                 "parked_on_waitcnt_or_barrier_frac_per_wave": wait, "source": f"profiles/{sq_name} (tools/pmc_sq_pass.sh)"}
                except Exception:
                    issue = None
            out["roofline"] = {
                "kernel": "conv8h_kernel<NSRC, GN, UP, RES> (8-channel 3x3 layers of the UNet: ResnetBlock conv1 / conv2, Upsample; all levels)",
                "bound": "hbm", "issue": issue, "achieved": c_b / (c_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": c_b / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "launches": c_n, "avg_launch_ms": c_ms / c_n, "algorithmic_bytes_per_launch": c_b / c_n,
                "share_of_kernel_time": c_ms / tot_ms,
                "variants": [{"variant": CONV8_FAMILIES[f], "launches": v["launches"], "avg_launch_ms": v["ms"] / v["launches"],
                              "achieved_gbs": v["bytes"] / (v["ms"] * 1e-3) / 1e9} for f, v in conv.items()],
                "note": "bound = the roof the number is priced against; `issue` = the co-limiting resource measured by the SQ counters (null unless "
                        "profiles/r4_pmc_sq.json was taken on this library). achieved = ALGORITHMIC bytes (source maps + residual sources + destination of each launch, fp32, computed by the "
                        "host from the launch shape) / HIP-event time of the launches, one scene batch in flight; averages over the full- "
                        "and half-resolution levels. traffic = (2 x FETCH_SIZE + WRITE_SIZE) per launch from the committed rocprofv3 --pmc "
                        "passes of this command (profiles/r4_pmc_traffic.json, tools/pmc_pass.sh), null unless that file was measured on this workload "
                        "AND on a library with this library's source stamp (library_src). " + arith_note}
        if LATENT_FAMILY in fam:
            v = fam[LATENT_FAMILY]
            fl = 2.0 * (1600.0 + 72.0 * C) * HW * N * B
            ms1 = v["ms"] / v["launches"]
            fp32_eq = fl / (ms1 * 1e-3) / 1e12
            # north_star: "MFMA utilisation on the contractions against MI355X peak".  The contraction runs on the f16 matrix pipe,
            # six instructions per fp32-grade product block, so the pipe executes 6 x the algorithmic FLOPs; frac is THAT against the
            # dense f16 peak (2.5 PFLOP/s).  The fp32-equivalent rate (what the reference's arithmetic would need) is kept beside it.
            out["roofline_latent_step"] = {
                "kernel": "latent_step_h_kernel (conv_out + sampler update + conv_in of one step fused by linearity, in-kernel noise)",
                "bound": "mfma", "achieved": MFMA_TERMS * fp32_eq, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": MFMA_TERMS * fp32_eq / F16_MFMA_PEAK_TFLOPS, "traffic": None, "launches": v["launches"], "avg_launch_ms": ms1,
                "flops_per_launch": fl, "matrix_pipe_flops_per_launch": MFMA_TERMS * fl, "share_of_kernel_time": v["ms"] / tot_ms,
                "fp32_equivalent": {"achieved": fp32_eq, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fp32_eq / FP32_PEAK_TFLOPS},
                "achieved_algorithmic_gbs": v["bytes"] / (v["ms"] * 1e-3) / 1e9,
                "note": "algorithmic FLOPs = 2*(1600 + 72*C) per agent-pixel (conv_out 8 -> C, posterior update, conv_in C + 2 -> 8 and the noise's "
                        "conv_in, fused by linearity). achieved / frac = 6 x those FLOPs (a product block is six f16-pipe matrix instructions on "
                        "three-term operands; the hi-only noise convolution issues three, so this is an upper estimate of the pipe's load) against the "
                        "2.5 PFLOP/s dense f16 MFMA peak; fp32_equivalent = the algorithmic FLOPs against the 157.3 TFLOP/s fp32 peak. The in-kernel "
                        "Philox4x32-7 + Box-Muller generator is 13.5 % of the kernel's time (measured by subtraction with diagnostic builds: "
                        "profiles/r4_latent_noise_budget.txt), not its bound"}
        if ENH_FRONT_FAMILY in fam and C == 64:
            # Enhancer front: Linear C -> 4C, GELU, depthwise 3x3 + gate, Linear 2C -> C in one launch (enhancer.py:222-250); the two Linears
            # are the contractions: 2 * (4C^2 + 2C^2) FLOPs per token
            v = fam[ENH_FRONT_FAMILY]
            fl = 2.0 * 6.0 * C * C * HW * N * B * v["launches"]
            fp32_eq = fl / (v["ms"] * 1e-3) / 1e12
            out["roofline_enh_front"] = {
                "kernel": "enh_front_h_kernel (FRFN: Linear1 + GELU + depthwise 3x3 + gate + Linear2 + residual, fused)",
                "bound": "mfma", "achieved": MFMA_TERMS * fp32_eq, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": MFMA_TERMS * fp32_eq / F16_MFMA_PEAK_TFLOPS, "traffic": None, "launches": v["launches"], "avg_launch_ms": v["ms"] / v["launches"],
                "fp32_equivalent": {"achieved": fp32_eq, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fp32_eq / FP32_PEAK_TFLOPS},
                "share_of_kernel_time": v["ms"] / tot_ms,
                "note": "contraction FLOPs = 12*C^2 per token (the two Linears), x 6 matrix instructions per product block against the dense f16 peak"}
        out["kernel_time_shares"] = {v["name"]: round(v["ms"] / tot_ms, 4) for f, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}
        return out

    with torch.no_grad():
        for i in range(args.warmup):
            run_scene(i, 1000 + i)
        if args.sync_debug:      # diagnostic: one scene batch per stream under torch's synchronisation detector (warnings carry the stack)
            torch.cuda.synchronize(device)
            torch.cuda.set_sync_debug_mode("warn")
            for i in range(S):
                run_scene(i, 1900 + i)
            torch.cuda.set_sync_debug_mode("default")
            torch.cuda.synchronize(device)
        elapsed = timed_region(args.steps, 2000)
        # the K-step region above is the reported number; 20 steps = 0.4 s is too short for a 5 s-period utilisation sampler to
        # see, so the same loop is repeated to >= --sustain seconds (default 12: at least two sampler periods) in ONE bracket and
        # reported beside it -- `value` stays on the K-step region
        sustained = None
        if elapsed < args.sustain and not args.no_timer:   # profiling runs (--no-timer) execute the K-step region only
            reps = int(np.ceil(args.sustain / max(elapsed, 1e-3)))
            ts = timed_region(args.steps * reps, 2500)
            sustained = {"steps": args.steps * reps, "seconds": ts, "value_this_rank": args.steps * reps * B / ts, "unit": "scenes/sec"}
        # latency of ONE scene alone (one stream, batch 1): throughput above needs S x B scenes in flight
        latency = None
        if rank == 0 and not args.no_timer:
            f1, c1, p1 = make_scene(N, C, H, W, 77, device, 1)
            pipe1 = ScenePipeline(gen, None if args.no_enhancer else enh, [N], C, H, W, device)
            pipe1.set_affine(normalize_pairwise_tfm(p1, H * PX_M, W * PX_M, 1))
            for i in range(3):
                pipe1.run(f1, c1, seed=5000 + i)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for i in range(10):
                pipe1.run(f1, c1, seed=5100 + i)
            torch.cuda.synchronize(device)
            latency = 1e3 * (time.perf_counter() - t0) / 10
            del pipe1, f1, c1
    timed = rank == 0 and not args.graph and not args.no_timer  # event pairs cannot be recorded into a replayed graph
    arith_mode = lib.gencomm_get_mode(_lib.MODE_ARITH)
    split_default = arith_mode in (0, 3)   # 3 = 0 on this path (the opt-in two-term kernels are the general convolutions AROUND it)
    roofs = rooflines(family_pass(), "Arithmetic: see config.arithmetic.") if timed else {}
    instantiations = getattr(family_pass, "instantiations", None)
    # the same workload with the exact-fp32 MFMA kernels everywhere, the arithmetic that is identical to the reference's:
    # the full step count, timed the same way (rank 0), with its own family pass
    exact = None
    if rank == 0 and split_default and not args.no_exact:
        with _lib.mode(_lib.MODE_ARITH, 1), torch.no_grad():
            for i in range(S):
                run_scene(i, 4000 + i)
            te = timed_region(args.steps, 4100, use_barrier=False)
            exact = {"value": args.steps * B / te, "unit": "scenes/sec", "steps": args.steps, "ms_per_step": 1e3 * te / args.steps,
                     "n_gpus": 1, "dtype": "f32",
                     "note": "rank 0, same pipelines and streams, gencomm_set_mode(GENCOMM_MODE_ARITH, 1): exact-fp32 "
                             "v_mfma_f32_4x4x1 / 32x32x2 kernels everywhere"}
            if timed:
                exact.update({k: v for k, v in rooflines(family_pass(), "Exact-fp32 kernels (conv8_kernel).").items() if k.startswith("roofline")})
    for pipe in pipes:
        assert torch.isfinite(pipe.fused).all(), "non-finite output"

    # every rank ran `steps` steps of B scenes of its own; whole-job rate = all scenes / slowest rank
    value, elapsed, total_scenes = gdist.aggregate_throughput(args.steps * B, elapsed, dist, red_dev)

    if rank == 0:
        flops, byts = algorithmic_work(N, C, HW, T)
        out = {
            "metric": "scenes/sec", "value": value, "unit": "scenes/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "total_scenes": total_scenes,
            "latency_ms_one_scene": latency, "sustained": sustained,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if arith_mode == 2 else "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: GenComm->Enhancer->AttFusion, {N} agents, C={C}, {H}x{W} BEV, "
                                   f"T={T} x0-param ancestral steps, {B} scene(s)/step/GPU",
                       "agents": N, "C": C, "H": H, "W": W, "T": T, "enhancer": not args.no_enhancer,
                       "noise": ("in-kernel Philox4x32-7 + Box-Muller: 16-bit angle; radius uniform on a 16-bit midpoint grid, refined by 32 more "
                                 "Philox bits below u = 2^-12 (|z| > 4.08), so radii reach 8.2 sigma (tail mass lost: 2e-16; round 2 stopped at "
                                 "4.85 sigma = 1.2e-6); the step noise added is nu = fp16(sigma_t z): relative rounding <= 2^-11 = 4.9e-4, "
                                 "variance added 8e-8 relative; q_sample's eps is unrounded fp32. The reference draws torch.randn "
                                 "(cond_diff.py:307); KS / tail / correlation tests on 1.4e8 samples of this field: tests/test_gpu_noise_stats.py; "
                                 "the field is exported (gencomm_step_noise_fwd) and replayed through the oracle: tests/test_gpu_philox_replay.py"),
                       "arithmetic": ("fp32 tensors in HBM, fp32 accumulation, fp32-EQUIVALENT products: every 3x3 / 5x5 / Linear operand is split "
                                      "EXACTLY into three terms (activations: fp16 hi + fp16 lo + a bf8 third term holding the last bit or two; weights: three "
                                      "fp16 terms + one bf8 copy) and a product is six matrix instructions (five v_mfma_f32_*_f16, one v_mfma_f32_*_bf8_bf8) "
                                      "accurate to 2^-26 -- all 24 bits of both operands enter it (|x| >= 2^-3; absolute operand accuracy 2^-28 below, "
                                      "the fp16 subnormal range of the second term). Per layer the result is closer to a float64 convolution than the "
                                      "exact-fp32 kernel's (rms 0.60x; tests/test_gpu_conv8.py::test_f16_pipe_layer_is_at_least_as_accurate_as_the_exact_fp32_kernel). "
                                      "Round 2's two-term / three-instruction form (22-bit products) is gone; exact_fp32_mode = the fp32 matrix-core kernels") if split_default else
                                     ("exact fp32 MFMA kernels (GENCOMM_MODE_ARITH = 1)" if arith_mode == 1 else
                                      "bf16 denoise mode (GENCOMM_MODE_ARITH = 2): the UNet's 8-channel maps stored as bf16, single bf16 MFMA products, "
                                      "fp32 accumulation, f64 GroupNorm statistics, fp32 sampler state; Enhancer / fusion as in the fp32 mode. "
                                      "NOT the fp32 headline: accuracy of bf16 storage, see tests/test_gpu_bf16.py"),
                       "streams_per_gpu": S, "scenes_per_step": B, "hip_graph": bool(args.graph),
                       "modes": {k: lib.gencomm_get_mode(v) for k, v in mode_keys.items()},
                       "parallelism": f"replicas x{world} (scene-sharded, no collective)",
                       "process_group": backend if dist is not None else None, "rccl_ranks": (dist.get_world_size() if backend == "nccl" else 0) if dist is not None else 1,
                       "share_device": backend != "nccl",
                       "instantiations_per_scene_batch": instantiations},
            "scene_algorithmic": {"gflop": flops / 1e9, "gbyte": byts / 1e9,
                                  "achieved_tflops": flops * args.steps * B / elapsed / 1e12,
                                  "achieved_gbs": byts * args.steps * B / elapsed / 1e9},
        }
        out["roofline"] = roofs.get("roofline")
        out["roofline_latent_step"] = roofs.get("roofline_latent_step")
        out["roofline_enh_front"] = roofs.get("roofline_enh_front")
        out["kernel_time_shares"] = roofs.get("kernel_time_shares")
        out["exact_fp32_mode"] = exact
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, N, C, H, W, T, gen, enh, ptm)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), file=RESULT_OUT, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
