#!/usr/bin/env python3
"""Training-step timing of the hot path (HIP forward + HIP backward): GenComm (training branch) -> Enhancer -> AttFusion,
loss = mean(fused^2), at the shipped shape (2 agents, C=128, 64x128, T=3) and at the metric geometry's per-scene size with
T=3. Prints scenes/s for forward+backward."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gencomm_amd import AttFusion, Enhancer, GenComm, normalize_pairwise_tfm, synth

DEV = "cuda:0"
OPTIMIZER = "--no-optimizer" not in sys.argv
ONLY = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""   # "shipped" / "large": one shape (profiling runs)
BATCH = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 1   # scenes per step (the shipped yamls train with batch_size 1, 2 or 4)
if "--stream-object" in sys.argv:   # diagnostic: the stream handle through a torch.cuda.Stream object per launch (before the raw accessor)
    from gencomm_amd import runtime as _rt
    _rt._raw_stream = None
if "--pconv-general" in sys.argv:   # diagnostic: the Enhancer's partial 3x3 convolution through the general implicit-GEMM kernel
    from gencomm_amd import autograd as _ag2
    _ag2.ENH_PCONV_C16 = False
if "--separate-update" in sys.argv:   # diagnostic: the sampler's update as separate noise / linear-combination launches
    from gencomm_amd import autograd as _ag3
    _ag3.FUSED_SAMPLER_UPDATE = False
if "--enh-split" in sys.argv:   # diagnostic: the Enhancer's training forward writes GELU(Linear1 output) in a pass of its own again
    from gencomm_amd import autograd as _ag
    _ag.ENH_MATERIALIZE_GELU = True
if "--mode" in sys.argv:   # library modes for this run, e.g. --mode bwd_streams=0 (repeatable)
    from gencomm_amd import _lib
    _keys = {"arith": _lib.MODE_ARITH, "xcd": _lib.MODE_XCD_REMAP, "bwd_streams": _lib.MODE_BWD_STREAMS, "tile_want": _lib.MODE_TILE_WANT}
    for _i, _a in enumerate(sys.argv):
        if _a == "--mode":
            _k, _v = sys.argv[_i + 1].split("=")
            _lib.check(_lib.lib().gencomm_set_mode(_keys[_k], int(_v)), "gencomm_set_mode")
SHAPES = {"shipped (2 agents, C=128, 64x128, T=3)": (2, 128, 64, 128, 3), "large: 4 agents, C=64, 200x704, T=3": (4, 64, 200, 704, 3)}
for name, (N, C, H, W, T) in SHAPES.items():
    if ONLY and not name.startswith(ONLY):
        continue
    gen, enh, fus = GenComm(synth.default_gencomm_cfg(C, T)).train().to(DEV), Enhancer(C, [8, 8], 4).train().to(DEV), AttFusion(C)
    g = torch.Generator(device=DEV).manual_seed(1)
    rl = [N] * BATCH
    feat = torch.randn(N * BATCH, C, H, W, generator=g, device=DEV).clamp_(min=0)
    cond = torch.randn(N * BATCH, 2, H, W, generator=g, device=DEV).requires_grad_(True)
    ptm = torch.from_numpy(synth.make_pairwise_t_matrix(rl, 5, 7, 10.0))
    affine = normalize_pairwise_tfm(ptm, H * 0.4, W * 0.4, 1).to(DEV)   # the training loop moves the batch to the device before the model (train_utils.to_device)
    params = [p for p in list(gen.parameters()) + list(enh.parameters()) if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-5, fused=True) if OPTIMIZER else None   # train.py uses Adam (hypes_yaml optimizer block)

    def step():
        for p in params:
            p.grad = None
        pred = gen(feat, cond, rl, seed=3)["pred_feature"]
        if pred.dim() == 3:
            pred = pred.unsqueeze(0)
        out = fus(enh(pred, affine, rl), rl, affine)
        out.square().mean().backward()
        if opt is not None:
            opt.step()          # the weights change: the next forward re-packs and re-prepares them (gencomm_unet_prepare)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    if "--sync-debug" in sys.argv:   # one step under torch's synchronisation detector (warnings carry the stack)
        torch.cuda.set_sync_debug_mode("warn")
        step()
        torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        step()
    t_enq = (time.perf_counter() - t0) / K / BATCH     # the host is done issuing
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K / BATCH
    print(f"train step [{name}, {BATCH} scene(s) per step]: {1e3 * dt:.1f} ms per scene (forward + backward{' + Adam step' if OPTIMIZER else ''}) = {1.0 / dt:.2f} scenes/s; host enqueue {1e3 * t_enq:.1f} ms", flush=True)
