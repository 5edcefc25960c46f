"""BASELINE.json configurations and the shipped shapes through the HIP path with the DEFAULT kernel selection, against the
CPU oracle.

Tolerance policy (SURVEY.md 8c sets rtol 1e-4 / atol 1e-5 for the float path), with the measurements behind it in
DESIGN.md section 2 / docs/history/DESIGN_rounds1-4.md section 2 (tools/diag_r2.py):

* PER STAGE -- every stage fed the oracle's own output of the stage before: single sampler steps (UNet call + update from
  the oracle's x_t), the Enhancer on the oracle's pred_feature, warp + AttFusion on the oracle's enhanced map -- the bar is
  ELEMENTWISE rtol 1e-4 / atol 1e-5, no exceptions.
* THE T-STEP CHAIN end to end cannot be held to that bar by ANY float32 implementation: the reference's own float32
  arithmetic (oracle in float32) deviates from the exact result (the same oracle evaluated in float64) by up to 1.9x that
  tolerance after 20 steps at 200x704 (78 of 36 M elements; 2.3x for the Enhancer output), because rounding differences
  of 1e-7 are amplified through 20 UNet evaluations. So the chain is judged against the FLOAT64 evaluation, relative to
  what the reference's float32 arithmetic achieves on the same inputs: rms error <= 2.5x the float32 oracle's, worst
  elementwise error/tolerance <= 2.5x max(1, the float32 oracle's); and against the float32 oracle itself with at most
  1e-4 of the elements outside the elementwise tolerance and none beyond 6x.
  (The fused map is compared with the float32 oracle only: the reference casts the float64 sampling grid to float32,
  torch_transformation_utils.py:329-331, so a float64 evaluation samples at different positions -- it is not "more exact".)

Cases:
* config 2 (2 agents, C=64, 200x704, T=10): at 2 agents the half-resolution level has fewer than 160 workgroups of 64x16,
  so the f16-pipe kernels (full resolution) and the exact-fp32 32x16 / 32x8-tile kernels (half resolution) run in ONE
  UNet call -- the mixed kernel set no other full-size test exercises;
* the metric configuration (4 agents, T=20);
* the shipped shape (2 agents, C=128, 64x128, T=3) and the V2X-Real shape (C=256) through ScenePipeline with 4 scenes per
  pipeline on 3 concurrent HIP streams (the benchmark's launch pattern).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL, ATOL = 1e-4, 1e-5


def _stats(got, want, rtol=RTOL, atol=ATOL):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (got.shape, want.shape)
    err = (got - want).abs()
    ratio = err / (atol + rtol * want.abs())
    return {"max": float(err.max()), "rms": float(err.pow(2).mean().sqrt()), "worst": float(ratio.max()),
            "over": int((ratio > 1).sum()), "n": ratio.numel(), "ref_max": float(want.abs().max())}


def check_elementwise(name, got, want):
    s = _stats(got, want)
    print(f"{name}: max abs err {s['max']:.3e}, worst err/tol {s['worst']:.3f} (elementwise rtol {RTOL} atol {ATOL}), "
          f"max |ref| {s['ref_max']:.2f}")
    assert np.isfinite(s["worst"]) and s["worst"] <= 1.0, (name, s)


def check_chain(name, hip, ref32, ref64):
    """End-to-end T-step chain: against float64, relative to the float32 reference arithmetic (module docstring)."""
    h64, r64, h32 = _stats(hip, ref64), _stats(ref32, ref64), _stats(hip, ref32)
    print(f"{name}: vs float64 -- HIP rms {h64['rms']:.3e} worst err/tol {h64['worst']:.2f} ({h64['over']}/{h64['n']} over) | "
          f"float32 oracle rms {r64['rms']:.3e} worst {r64['worst']:.2f} ({r64['over']} over); "
          f"HIP vs float32 oracle: worst {h32['worst']:.2f} ({h32['over']} over)")
    assert np.isfinite(h64["worst"])
    assert h64["rms"] <= 2.5 * r64["rms"] + 1e-8, (name, h64, r64)
    assert h64["worst"] <= 2.5 * max(1.0, r64["worst"]), (name, h64, r64)
    assert h32["over"] <= 1e-4 * h32["n"] and h32["worst"] <= 6.0, (name, h32)


def check_chain32(name, hip, ref32):
    s = _stats(hip, ref32)
    print(f"{name}: vs float32 oracle -- max abs err {s['max']:.3e}, worst err/tol {s['worst']:.2f} ({s['over']}/{s['n']} over)")
    assert np.isfinite(s["worst"]) and s["over"] <= 1e-4 * s["n"] and s["worst"] <= 6.0, (name, s)


def _modules(C, T, seed):
    from gencomm_amd import Enhancer, GenComm, synth
    cfg = synth.default_gencomm_cfg(C, T)
    gen, enh = GenComm(cfg).eval(), Enhancer(C, [8, 8], 4).eval()
    synth.fill_params_(gen, seed)
    synth.fill_params_(enh, seed + 1)
    return cfg, gen, enh


def _sd(m, dtype=torch.float32):
    return {k: v.detach().cpu().to(dtype) if v.is_floating_point() else v.detach().cpu() for k, v in m.state_dict().items()}


def _oracle_chain(O, sd, cfg, feat, cond, rl, n0, sn, keep):
    """oracle/torch_port.gencomm_forward with the x_t of the timesteps in `keep` recorded."""
    T = cfg["diffusion"]["num_diffusion_timesteps"]
    sched = O.make_schedule(T)
    x = O.q_sample(sched, O.ego_repeat(feat, rl), T - 1, n0)
    xs = {}
    for i, t in enumerate(reversed(range(T))):
        if t in keep:
            xs[t] = (x.clone(), i)
        x = O.p_sample(sd, sched, cfg["model"], cond, x, t, sn[i] if t > 0 else None)
    return x, xs, sched


@pytest.mark.parametrize("N,T,seed", [(2, 10, 91), (4, 20, 81)], ids=["config2_2agents_T10", "metric_4agents_T20"])
def test_full_size_config_vs_oracle(N, T, seed):
    from gencomm_amd import AttFusion, normalize_pairwise_tfm, synth
    from oracle import torch_port as O
    C, H, W = 64, 200, 704
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg, gen, enh = _modules(C, T, seed)
    g = torch.Generator().manual_seed(seed + 2)
    feat = torch.randn(N, C, H, W, generator=g).clamp_(min=0)
    cond = torch.randn(N, 2, H, W, generator=g)
    n0 = torch.randn(N, C, H, W, generator=g)
    sn = torch.randn(T, N, C, H, W, generator=g)
    rl = torch.tensor([N])
    ptm = torch.from_numpy(synth.make_pairwise_t_matrix([N], 5, seed + 3, 40.0))
    d = torch.float64
    with torch.no_grad():
        keep = {T - 1, T // 2, 0}
        pred32, xs, sched = _oracle_chain(O, _sd(gen), cfg, feat, cond, rl, n0, sn, keep)
        enh32 = O.enhancer_forward(_sd(enh), pred32, rl)
        aff32 = O.normalize_pairwise_tfm(ptm, H * 0.4, W * 0.4, 1.0)
        fus32 = O.att_fusion(enh32, rl, aff32)
        pred64, _, _ = _oracle_chain(O, _sd(gen, d), cfg, feat.to(d), cond.to(d), rl, n0.to(d), sn.to(d), set())
        enh64 = O.enhancer_forward(_sd(enh, d), pred64, rl)
    gen, enh = gen.to(DEV), enh.to(DEV)
    fus = AttFusion(C)
    tag = f"[{N} agents, T={T}]"
    with torch.no_grad():
        affine = normalize_pairwise_tfm(ptm, H * 0.4, W * 0.4, 1)
        # ---- per stage, elementwise: single sampler steps from the oracle's x_t
        for t in sorted(keep, reverse=True):
            x_t, i = xs[t]
            x0 = gen.denoiser(torch.cat([cond, x_t], dim=1).to(DEV), torch.full((N,), float(t), device=DEV), T=T).cpu()
            if t > 0:  # cond_diff.py:272-315
                x0 = sched["posterior_mean_coef1"][t] * x0 + sched["posterior_mean_coef2"][t] * x_t \
                    + (0.5 * sched["posterior_log_variance_clipped"][t]).exp() * sn[i]
            want = O.p_sample(_sd(gen), sched, cfg["model"], cond, x_t, t, sn[i] if t > 0 else None)
            check_elementwise(f"{tag} sampler step t={t} from the oracle's x_t", x0, want)
        check_elementwise(f"{tag} Enhancer on the oracle's pred_feature", enh(pred32.to(DEV), affine, rl), enh32)
        check_elementwise(f"{tag} warp + AttFusion on the oracle's enhanced map", fus(enh32.to(DEV), rl, affine), fus32)
        # ---- the chain end to end
        pred = gen(feat.to(DEV), cond.to(DEV), rl, noise=(n0.to(DEV), sn.to(DEV)))["pred_feature"]
        enh_e2e = enh(pred, affine, rl)
        fus_e2e = fus(enh_e2e, rl, affine)
    torch.cuda.synchronize()
    check_chain(f"{tag} pred_feature after {T} steps", pred, pred32, pred64)
    check_chain(f"{tag} enhanced, end to end", enh_e2e, enh32, enh64)
    check_chain32(f"{tag} fused, end to end", fus_e2e, fus32)


@pytest.mark.parametrize("C", [128, 256], ids=["shipped_C128", "v2xreal_C256"])
def test_shipped_shapes_scene_pipeline_4x3_streams_vs_oracle(C):
    from gencomm_amd import normalize_pairwise_tfm, synth
    from gencomm_amd.pipeline import ScenePipeline
    from oracle import torch_port as O
    N, H, W, T, B, S = 2, 64, 128, 3, 4, 3
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg, gen, enh = _modules(C, T, 100 + C)
    d = torch.float64
    sd_g, sd_e, sd_g64, sd_e64 = _sd(gen), _sd(enh), _sd(gen, d), _sd(enh, d)
    gen, enh = gen.to(DEV), enh.to(DEV)
    dev = torch.device(DEV)
    streams = [torch.cuda.Stream() for _ in range(S)]
    pipes, data, refs = [], [], []
    for si in range(S):
        g = torch.Generator().manual_seed(200 + C + si)
        n = N * B
        feat = torch.randn(n, C, H, W, generator=g).clamp_(min=0)
        cond = torch.randn(n, 2, H, W, generator=g)
        n0 = torch.randn(n, C, H, W, generator=g)
        sn = torch.randn(T, n, C, H, W, generator=g)
        ptm = torch.from_numpy(synth.make_pairwise_t_matrix([N] * B, 5, 300 + si, 20.0))
        rl = torch.tensor([N] * B)
        r32 = O.path_forward(sd_g, sd_e, cfg, feat, cond, rl, ptm, H * 0.8, W * 0.8, n0, sn)
        with torch.no_grad():
            p64 = O.gencomm_forward(sd_g64, cfg, feat.to(d), cond.to(d), rl, n0.to(d), sn.to(d))
        refs.append((r32, p64))
        p = ScenePipeline(gen, enh, [N] * B, C, H, W, dev)
        p.set_affine(normalize_pairwise_tfm(ptm, H * 0.8, W * 0.8, 1))
        pipes.append(p)
        data.append((feat.to(dev), cond.to(dev), (n0.to(dev), sn.to(dev))))
    torch.cuda.synchronize()
    outs = []
    with torch.no_grad():
        for rep in range(2):  # second round: the streams are busy with each other's kernels from the start
            for si in range(S):
                with torch.cuda.stream(streams[si]):
                    fused = pipes[si].run(data[si][0], data[si][1], noise=data[si][2]).clone()
                    outs.append((si, rep, fused, pipes[si].pred.clone()))
    torch.cuda.synchronize()
    for si, rep, fused, pred in outs:
        check_chain(f"C={C} stream {si} round {rep} pred_feature after {T} steps", pred, refs[si][0]["pred_feature"], refs[si][1])
        check_chain32(f"C={C} stream {si} round {rep} fused", fused, refs[si][0]["fused"])
