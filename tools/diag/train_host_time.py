"""Where the host spends a large training step: enqueue time (before the final synchronise) against total, and a cProfile of one step."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gencomm_amd import AttFusion, Enhancer, GenComm, normalize_pairwise_tfm, synth
DEV = "cuda:0"
N, C, H, W, T = 4, 64, 200, 704, 3
gen, enh, fus = GenComm(synth.default_gencomm_cfg(C, T)).train().to(DEV), Enhancer(C, [8, 8], 4).train().to(DEV), AttFusion(C)
g = torch.Generator(device=DEV).manual_seed(1)
feat = torch.randn(N, C, H, W, generator=g, device=DEV).clamp_(min=0)
cond = torch.randn(N, 2, H, W, generator=g, device=DEV).requires_grad_(True)
affine = normalize_pairwise_tfm(torch.from_numpy(synth.make_pairwise_t_matrix([N], 5, 7, 10.0)), H * 0.4, W * 0.4, 1)
params = [p for p in list(gen.parameters()) + list(enh.parameters()) if p.requires_grad]
opt = torch.optim.Adam(params, lr=1e-5, fused=True)
def step():
    for p in params: p.grad = None
    pred = gen(feat, cond, [N], seed=3)["pred_feature"]
    out = fus(enh(pred, affine, [N]), [N], affine)
    out.square().mean().backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(4):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0):.1f} ms, total {1e3 * (t2 - t0):.1f} ms")
pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
