#!/usr/bin/env python3
"""Round-2 diagnostics (GPU): (1) latent vs direct sampler error statistics per noise kind; (2) HIP vs fp32 oracle vs float64
oracle at the BASELINE configurations: how far is the reference's own fp32 arithmetic from the exact result?"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gencomm_amd import GenComm, Enhancer, AttFusion, normalize_pairwise_tfm, synth, _lib
from oracle import torch_port as O
DEV = "cuda:0"


def setm(k, v):
    _lib.check(_lib.lib().gencomm_set_mode(k, v), "set_mode")


def latent_direct():
    for (C, H, W, n, T) in [(16, 12, 20, 2, 4), (64, 70, 132, 3, 5), (8, 16, 64, 1, 3), (64, 64, 256, 5, 5)]:
        gen = GenComm(synth.default_gencomm_cfg(C, T)).eval()
        synth.fill_params_(gen, 7)
        gen = gen.to(DEV)
        inp = synth.make_inputs([n], C, H, W, 8)
        feat, cond = torch.from_numpy(inp["feat"]).to(DEV), torch.from_numpy(inp["cond"]).to(DEV)
        noise = tuple(torch.from_numpy(a).to(DEV) for a in synth.make_eval_noise(9, n, C, H, W, T))
        outs = {}
        for mode in (0, 1):
            setm(_lib.MODE_SAMPLER, mode)
            with torch.no_grad():
                outs[mode, "explicit"] = gen(feat, cond, [n], noise=noise)["pred_feature"].cpu()
                for seed in (77, 78, 79):
                    outs[mode, f"philox{seed}"] = gen(feat, cond, [n], seed=seed)["pred_feature"].cpu()
        setm(_lib.MODE_SAMPLER, 0)
        for kind in ("explicit", "philox77", "philox78", "philox79"):
            a, b = outs[0, kind], outs[1, kind]
            err = (a - b).abs()
            ratio = err / (2e-5 + 2e-4 * b.abs())
            idx = int(ratio.argmax())
            print(f"latent-vs-direct C{C} {H}x{W} n{n} T{T} {kind:9}: max err {float(err.max()):.3e} worst ratio {float(ratio.max()):.3f} "
                  f"(#>1: {int((ratio > 1).sum())}/{ratio.numel()}) at flat {idx} ref {float(b.flatten()[idx]):.4f}; rms a {float(a.pow(2).mean().sqrt()):.3f}", flush=True)


def three_way(name, N, C, H, W, T, seed, px):
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg = synth.default_gencomm_cfg(C, T)
    gen, enh = GenComm(cfg).eval(), Enhancer(C, [8, 8], 4).eval()
    synth.fill_params_(gen, seed); synth.fill_params_(enh, seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    feat = torch.randn(N, C, H, W, generator=g).clamp_(min=0); cond = torch.randn(N, 2, H, W, generator=g)
    n0 = torch.randn(N, C, H, W, generator=g); sn = torch.randn(T, N, C, H, W, generator=g)
    rl = torch.tensor([N]); ptm = torch.from_numpy(synth.make_pairwise_t_matrix([N], 5, seed + 3, 40.0))
    sd_g = {k: v.detach() for k, v in gen.state_dict().items()}; sd_e = {k: v.detach() for k, v in enh.state_dict().items()}
    t = time.time(); r32 = O.path_forward(sd_g, sd_e, cfg, feat, cond, rl, ptm, H * px, W * px, n0, sn); t32 = time.time() - t
    d = lambda x: x.double()
    t = time.time(); r64 = O.path_forward({k: d(v) for k, v in sd_g.items()}, {k: d(v) for k, v in sd_e.items()}, cfg, d(feat), d(cond), rl, ptm, H * px, W * px, d(n0), d(sn)); t64 = time.time() - t
    gen, enh = gen.to(DEV), enh.to(DEV)
    res = {}
    for arith in (0, 1):
        setm(_lib.MODE_ARITH, arith)
        with torch.no_grad():
            affine = normalize_pairwise_tfm(ptm, H * px, W * px, 1)
            pred = gen(feat.to(DEV), cond.to(DEV), rl, noise=(n0.to(DEV), sn.to(DEV)))["pred_feature"]
            e = enh(pred, affine, rl)
            f = AttFusion(C)(e, rl, affine)
        res[arith] = {"pred_feature": pred.cpu().double(), "enhanced": e.cpu().double(), "fused": f.cpu().double()}
    setm(_lib.MODE_ARITH, 0)
    print(f"== {name}: oracle f32 {t32:.1f} s, f64 {t64:.1f} s")
    for k in ("pred_feature", "enhanced", "fused"):
        truth = r64[k].double()
        tol = 1e-5 + 1e-4 * truth.abs()
        for label, x in (("oracle-f32", r32[k].double()), ("HIP split ", res[0][k]), ("HIP exact ", res[1][k])):
            err = (x - truth).abs(); ratio = err / tol
            print(f"   {k:12} {label} vs f64: max err {float(err.max()):.3e} rms err {float(err.pow(2).mean().sqrt()):.3e} worst ratio {float(ratio.max()):.3f} "
                  f"#>1 {int((ratio > 1).sum())}/{ratio.numel()}  p99.99 ratio {float(ratio.flatten().kthvalue(int(0.9999 * ratio.numel())).values):.3f}", flush=True)
        err = (res[0][k] - r32[k].double()).abs(); ratio = err / (1e-5 + 1e-4 * r32[k].double().abs())
        print(f"   {k:12} HIP split vs oracle-f32: max err {float(err.max()):.3e} worst ratio {float(ratio.max()):.3f} #>1 {int((ratio > 1).sum())}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["ld", "cfg2", "shipped", "metric"]
    if "ld" in what: latent_direct()
    if "cfg2" in what: three_way("config 2 (2 agents, T=10)", 2, 64, 200, 704, 10, 91, 0.4)
    if "shipped" in what: three_way("shipped (2 agents, C=128, 64x128, T=3)", 2, 128, 64, 128, 3, 228, 0.8)
    if "metric" in what: three_way("metric (4 agents, T=20)", 4, 64, 200, 704, 20, 81, 0.4)
