#!/usr/bin/env python3
"""Where the HOST spends a training step of the hot path at the shipped shape (the step is host-bound there): cProfile over 20 steps of
tools/train_bench.py's loop, top functions by own and by cumulative time.  python tools/diag/host_profile.py [shipped|large]"""
import cProfile, io, os, pstats, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
from gencomm_amd import AttFusion, Enhancer, GenComm, normalize_pairwise_tfm, synth

which = sys.argv[1] if len(sys.argv) > 1 else "shipped"
N, C, H, W, T = (2, 128, 64, 128, 3) if which == "shipped" else (4, 64, 200, 704, 3)
DEV = "cuda:0"
gen, enh, fus = GenComm(synth.default_gencomm_cfg(C, T)).train().to(DEV), Enhancer(C, [8, 8], 4).train().to(DEV), AttFusion(C)
g = torch.Generator(device=DEV).manual_seed(1)
rl = [N]
feat = torch.randn(N, C, H, W, generator=g, device=DEV).clamp_(min=0)
cond = torch.randn(N, 2, H, W, generator=g, device=DEV).requires_grad_(True)
affine = normalize_pairwise_tfm(torch.from_numpy(synth.make_pairwise_t_matrix(rl, 5, 7, 10.0)), H * 0.4, W * 0.4, 1).to(DEV)
params = [p for p in list(gen.parameters()) + list(enh.parameters()) if p.requires_grad]
opt = torch.optim.Adam(params, lr=1e-5, fused=True)


def step():
    for p in params:
        p.grad = None
    pred = gen(feat, cond, rl, seed=3)["pred_feature"]
    if pred.dim() == 3:
        pred = pred.unsqueeze(0)
    out = fus(enh(pred, affine, rl), rl, affine)
    out.square().mean().backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)   # the backward runs on this thread: cProfile sees its Python functions
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print("\n".join(l[:170] for l in s.getvalue().split("\n")))
