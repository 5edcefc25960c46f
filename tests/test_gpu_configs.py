"""BASELINE.json configurations and the shipped shapes through the HIP path with the DEFAULT kernel selection, against the
CPU oracle ELEMENTWISE (rtol 1e-4 / atol 1e-5) -- per stage (each stage fed the oracle's output of the stage
before, so that an error cannot hide behind the one upstream) and end to end.

* config 2 (2 agents, C=64, 200x704, T=10): at 2 agents the half-resolution level has fewer than 160 workgroups of 64x16,
  so the f16-pipe kernels (full resolution) and the exact-fp32 32x16 / 32x8-tile kernels (half resolution) run in ONE
  UNet call -- the mixed kernel set no other full-size test exercises;
* the shipped shape (2 agents, C=128, 64x128, T=3) and the V2X-Real shape (C=256) through ScenePipeline with 4 scenes per
  pipeline on 3 concurrent HIP streams (the benchmark's launch pattern).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL, ATOL = 1e-4, 1e-5


def check_elementwise(name, got, want, rtol=RTOL, atol=ATOL):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    err = (got - want).abs()
    ratio = err / (atol + rtol * want.abs())
    worst = float(ratio.max())
    print(f"{name}: max abs err {float(err.max()):.3e}, worst err/tol {worst:.3f} (elementwise rtol {rtol} atol {atol}), "
          f"max |ref| {float(want.abs().max()):.2f}, elements over {int((ratio > 1).sum())}/{ratio.numel()}")
    assert np.isfinite(worst) and worst <= 1.0, (name, worst, float(err.max()))


def _modules(C, T, seed):
    from gencomm_amd import Enhancer, GenComm, synth
    cfg = synth.default_gencomm_cfg(C, T)
    gen, enh = GenComm(cfg).eval(), Enhancer(C, [8, 8], 4).eval()
    synth.fill_params_(gen, seed)
    synth.fill_params_(enh, seed + 1)
    return cfg, gen, enh


def _sd(m):
    return {k: v.detach().cpu() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("N,T,seed", [(2, 10, 91), (4, 20, 81)], ids=["config2_2agents_T10", "metric_4agents_T20"])
def test_full_size_config_vs_oracle_per_stage_elementwise(N, T, seed):
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
    ref = O.path_forward(_sd(gen), _sd(enh), cfg, feat, cond, rl, ptm, H * 0.4, W * 0.4, n0, sn)
    gen, enh = gen.to(DEV), enh.to(DEV)
    fus = AttFusion(C)
    with torch.no_grad():
        affine = normalize_pairwise_tfm(ptm, H * 0.4, W * 0.4, 1)
        pred = gen(feat.to(DEV), cond.to(DEV), rl, noise=(n0.to(DEV), sn.to(DEV)))["pred_feature"]
        # per stage: fed the oracle's result of the stage before
        enh_iso = enh(ref["pred_feature"].to(DEV), affine, rl)
        fus_iso = fus(ref["enhanced"].to(DEV), rl, affine)
        # end to end
        enh_e2e = enh(pred, affine, rl)
        fus_e2e = fus(enh_e2e, rl, affine)
    torch.cuda.synchronize()
    check_elementwise(f"[{N} agents, T={T}] pred_feature (GenComm, {T} steps)", pred, ref["pred_feature"])
    check_elementwise(f"[{N} agents, T={T}] enhanced (Enhancer on the oracle's pred_feature)", enh_iso, ref["enhanced"])
    check_elementwise(f"[{N} agents, T={T}] fused (warp + AttFusion on the oracle's enhanced)", fus_iso, ref["fused"])
    check_elementwise(f"[{N} agents, T={T}] enhanced, end to end", enh_e2e, ref["enhanced"])
    check_elementwise(f"[{N} agents, T={T}] fused, end to end", fus_e2e, ref["fused"])


@pytest.mark.parametrize("C", [128, 256], ids=["shipped_C128", "v2xreal_C256"])
def test_shipped_shapes_scene_pipeline_4x3_streams_vs_oracle(C):
    from gencomm_amd import normalize_pairwise_tfm, synth
    from gencomm_amd.pipeline import ScenePipeline
    from oracle import torch_port as O
    N, H, W, T, B, S = 2, 64, 128, 3, 4, 3
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg, gen, enh = _modules(C, T, 100 + C)
    sd_g, sd_e = _sd(gen), _sd(enh)
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
        refs.append(O.path_forward(sd_g, sd_e, cfg, feat, cond, rl, ptm, H * 0.8, W * 0.8, n0, sn))
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
                    outs.append((si, fused, pipes[si].pred.clone()))
    torch.cuda.synchronize()
    for si, fused, pred in outs:
        check_elementwise(f"C={C} stream {si} pred_feature", pred, refs[si]["pred_feature"])
        check_elementwise(f"C={C} stream {si} fused", fused, refs[si]["fused"])
