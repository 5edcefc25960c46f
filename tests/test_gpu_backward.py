"""HIP backward of the GenComm training branch (gencomm_unet_bwd: conv dgrad / wgrad, GroupNorm+SiLU, nin, Downsample,
Upsample backward kernels) against torch autograd through the CPU oracle's functional restatement (float64): EVERY parameter
gradient, the gradient of the message channels and of the features, for one UNet call and for the T-step sampler chain the
reference trains through (cond_diff.py:342-360); plus two DistributedDataParallel ranks (gloo, both on cuda:0) whose
all-reduced gradients must equal the single-process gradients of the concatenated batch (train_ddp.py:121-125)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _close(name, got, want, rtol=2e-3):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    assert err <= rtol * scale + 1e-9, (name, err, scale)
    return err / (scale + 1e-30)


def _oracle_grads(cfg, gen, feat, cond, rl, n0, sn, single_t=None):
    """float64 autograd through oracle/torch_port on the CPU: grads of mean(out^2) w.r.t. every parameter, feat, cond."""
    from oracle import torch_port as O
    sd = {k: v.detach().cpu().double().requires_grad_(v.is_floating_point() and "denoiser" in k) for k, v in gen.state_dict().items()}
    f = feat.double().requires_grad_(True)
    c = cond.double().requires_grad_(True)
    if single_t is None:
        out = O.gencomm_forward(sd, cfg, f, c, rl, n0.double(), sn.double())
    else:
        tt = torch.full((f.shape[0],), float(single_t), dtype=torch.float64)
        out = O.unet_forward(sd, "denoiser", torch.cat([c, f], dim=1), tt, cfg["model"])
    loss = (out ** 2).mean()
    names = [k for k in sd if sd[k].requires_grad]
    grads = torch.autograd.grad(loss, [sd[k] for k in names] + [f, c])
    return out.detach(), float(loss), dict(zip(names, grads[:-2])), grads[-2], grads[-1]


@pytest.mark.parametrize("C,H,W,n,levels,t", [(16, 16, 24, 2, 2, 1), (8, 24, 40, 1, 3, 0), (32, 20, 68, 3, 2, 2),
                                              (8, 192, 384, 4, 2, 1)])   # > 768 tile-items per weight-gradient launch: several tiles per workgroup + partial sums
def test_unet_call_backward_all_gradients_vs_oracle_autograd(C, H, W, n, levels, t):
    from gencomm_amd import GenComm, synth
    from gencomm_amd.autograd import UNetFunction
    T = 3
    cfg = synth.default_gencomm_cfg(C, T)
    cfg["model"]["ch_mult"] = [1] * levels
    gen = GenComm(cfg).train()
    synth.fill_params_(gen, 300 + C)
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(n, C, H, W, generator=g)
    cond = torch.randn(n, 2, H, W, generator=g)
    ref_out, _, pg, gx, gc = _oracle_grads(cfg, gen, x, cond, None, None, None, single_t=t)
    gen = gen.to(DEV)
    xd, cd = x.to(DEV).requires_grad_(True), cond.to(DEV).requires_grad_(True)
    out = UNetFunction.apply(gen.denoiser, t, T, xd, cd, gen.denoiser.flat_params())
    _close("x0_hat", out, ref_out, 1e-4)
    (out ** 2).mean().backward()
    worst = max(_close("grad x_t", xd.grad, gx), _close("grad cond", cd.grad, gc))
    for name, p in gen.denoiser.named_parameters():
        assert p.grad is not None, name
        worst = max(worst, _close("grad " + name, p.grad, pg["denoiser." + name]))
    print(f"UNet call backward C={C} {H}x{W} n={n} levels={levels} t={t}: {len(pg)} parameter gradients, worst relative error {worst:.2e}")


def test_sampler_chain_backward_vs_oracle_autograd():
    from gencomm_amd import GenComm, synth
    C, H, W, T, rl = 16, 16, 24, 3, [2, 1]
    n = sum(rl)
    cfg = synth.default_gencomm_cfg(C, T)
    gen = GenComm(cfg).train()
    synth.fill_params_(gen, 77)
    inp = {k: torch.from_numpy(v) for k, v in synth.make_inputs(rl, C, H, W, 78).items()}
    n0, sn = (torch.from_numpy(a) for a in synth.make_train_noise(79, n, C, H, W, T))
    ref_out, ref_loss, pg, gf, gc = _oracle_grads(cfg, gen, inp["feat"], inp["cond"], inp["record_len"], n0, sn)
    gen = gen.to(DEV)
    feat, cond = inp["feat"].to(DEV).requires_grad_(True), inp["cond"].to(DEV).requires_grad_(True)
    pred = gen(feat, cond, inp["record_len"], noise=(n0.to(DEV), sn.to(DEV)))["pred_feature"]
    _close("pred_feature (train)", pred, ref_out, 1e-4)
    loss = (pred ** 2).mean()
    assert abs(loss.item() - ref_loss) < 1e-4 * abs(ref_loss)
    loss.backward()
    worst = max(_close("grad feat", feat.grad, gf), _close("grad cond", cond.grad, gc))
    assert float(feat.grad[1].abs().max()) == 0.0 and float(feat.grad[0].abs().max()) > 0  # only the ego rows feed x_start (cond_diff.py:332-337)
    for name, p in gen.denoiser.named_parameters():
        worst = max(worst, _close("grad " + name, p.grad, pg["denoiser." + name]))
    print(f"sampler chain backward (T={T}): worst relative error {worst:.2e}")


def test_training_chain_with_in_kernel_noise_is_the_inference_chain_of_the_same_seed():
    """noise=None in the training branch: q_sample and the T - 1 step noises come from the sampler's Philox field (no framework
    generator, no noise tensors from the caller) -- the same field inference adds for that seed, so the differentiable chain
    (per-step HIP UNet calls) and the fused inference loop must agree, and the gradients must flow."""
    from gencomm_amd import GenComm, synth
    C, H, W, T, rl = 16, 16, 24, 3, [2, 1]
    cfg = synth.default_gencomm_cfg(C, T)
    gen = GenComm(cfg).train().to(DEV)
    synth.fill_params_(gen, 77)
    inp = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_inputs(rl, C, H, W, 78).items()}
    with torch.no_grad():
        ref = gen(inp["feat"], inp["cond"], inp["record_len"], seed=1234)["pred_feature"]
    cond = inp["cond"].clone().requires_grad_(True)
    pred = gen(inp["feat"], cond, inp["record_len"], seed=1234)["pred_feature"]
    assert pred.requires_grad
    err = (pred - ref).abs()
    assert (err <= 2e-5 + 1e-4 * ref.abs()).all(), err.max().item()
    (pred ** 2).mean().backward()
    assert float(cond.grad.abs().max()) > 0 and all(p.grad is not None for p in gen.denoiser.parameters() if p.requires_grad)
    other = gen(inp["feat"], cond, inp["record_len"], seed=1235)["pred_feature"]
    assert float((other - pred).abs().max()) > 1e-3


DDP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)   # before any GPU call
from gencomm_amd import GenComm, synth
C, H, W, T = 16, 16, 24, 3
gen = GenComm(synth.default_gencomm_cfg(C, T)).train()
synth.fill_params_(gen, 5)
gen = gen.to("cuda:0")
ddp = torch.nn.parallel.DistributedDataParallel(gen, find_unused_parameters=True)   # train_ddp.py:121-125 (gloo: CPU-side reduction of cuda grads)
inp = {k: torch.from_numpy(v) for k, v in synth.make_inputs([2], C, H, W, 100 + rank).items()}
n0, sn = (torch.from_numpy(a).cuda() for a in synth.make_train_noise(200 + rank, 2, C, H, W, T))
pred = ddp(inp["feat"].cuda(), inp["cond"].cuda(), inp["record_len"], noise=(n0, sn))["pred_feature"]
(pred ** 2).mean().backward()
torch.cuda.synchronize()
if rank == 0:
    torch.save({k: p.grad.cpu() for k, p in gen.named_parameters()}, out)
dist.barrier()
dist.destroy_process_group()
"""


def test_two_rank_ddp_gradient_sync_matches_single_process(tmp_path):
    from gencomm_amd import GenComm, synth
    script = tmp_path / "ddp_worker.py"
    script.write_text(DDP_WORKER)
    out = tmp_path / "grads.pt"
    port = str(29600 + os.getpid() % 300)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), REPO, str(r), "2", port, str(out)], env=env) for r in range(2)]
    rcs = [p.wait(timeout=600) for p in procs]
    assert rcs == [0, 0], rcs
    ddp_grads = torch.load(out)
    # single process: mean of the two ranks' losses == DDP's averaged gradients
    C, H, W, T = 16, 16, 24, 3
    gen = GenComm(synth.default_gencomm_cfg(C, T)).train()
    synth.fill_params_(gen, 5)
    gen = gen.to(DEV)
    total = 0.0
    for rank in range(2):
        inp = {k: torch.from_numpy(v) for k, v in synth.make_inputs([2], C, H, W, 100 + rank).items()}
        n0, sn = (torch.from_numpy(a).to(DEV) for a in synth.make_train_noise(200 + rank, 2, C, H, W, T))
        pred = gen(inp["feat"].to(DEV), inp["cond"].to(DEV), inp["record_len"], noise=(n0, sn))["pred_feature"]
        total = total + 0.5 * (pred ** 2).mean()
    total.backward()
    worst = 0.0
    for k, p in gen.named_parameters():
        worst = max(worst, _close("ddp grad " + k, ddp_grads[k], p.grad, 1e-4))
    print(f"2-rank DDP (gloo) averaged gradients vs single process: worst relative difference {worst:.2e}")


@pytest.mark.parametrize("C,H,W,rl", [(16, 12, 20, [2, 1]), (64, 24, 40, [3]), (128, 16, 16, [1, 2]),
                                      (16, 192, 384, [2])])   # >= 2^17 pixels: the weight gradients run on the side stream
def test_enhancer_backward_hip_vs_torch_autograd(C, H, W, rl):
    """EnhancerFunction.backward (LayerNorm / conv / depthwise / GELU gradient kernels, gencomm_amd/train_ops.py) against
    torch autograd through the differentiable restatement of the stage: input gradient and every live parameter gradient."""
    from gencomm_amd import Enhancer, normalize_pairwise_tfm, synth
    from torch_restatements import enhancer_forward
    enh = Enhancer(C, [8, 8], 4).to(DEV)
    synth.fill_params_(enh, 3 + C)
    inp = synth.make_inputs(rl, C, H, W, 4)
    x = torch.from_numpy(inp["feat"]).to(DEV).requires_grad_(True)
    g = torch.Generator(device=DEV).manual_seed(C)
    wgt = torch.randn(sum(rl), C, H, W, generator=g, device=DEV)
    y = enh(x, None, rl)
    (y * wgt).sum().backward()
    got = {"x": x.grad.clone(), **{k: p.grad.clone() for k, p in enh.named_parameters() if p.grad is not None}}
    for p in enh.parameters():
        p.grad = None
    x2 = x.detach().clone().requires_grad_(True)
    y2 = enhancer_forward(enh, x2)
    _close("enhancer forward", y, y2, 1e-4)
    (y2 * wgt).sum().backward()
    want = {"x": x2.grad, **{k: p.grad for k, p in enh.named_parameters() if p.grad is not None}}
    assert set(got) == set(want) and len(got) >= 16, (sorted(got), sorted(want))
    worst = max(_close("enhancer grad " + k, got[k], want[k]) for k in want)
    print(f"Enhancer backward C={C} {H}x{W}: {len(want) - 1} parameter gradients + input, worst relative error {worst:.2e}")


@pytest.mark.parametrize("C,H,W,rl,shift", [(16, 12, 20, [2, 1], 3.0), (64, 40, 72, [5], 30.0), (32, 30, 50, [1, 4, 2], 12.0)])
def test_fusion_backward_hip_vs_torch_autograd(C, H, W, rl, shift):
    """gencomm_warp_attfuse_bwd (softmax / dot-product backward + the bilinear gather's adjoint) against torch autograd through
    affine_grid + grid_sample + softmax attention, with agents partly and wholly out of range."""
    from gencomm_amd import AttFusion, normalize_pairwise_tfm, synth
    from torch_restatements import att_fusion_forward
    inp = synth.make_inputs(rl, C, H, W, 9, max_shift=shift)
    affine = normalize_pairwise_tfm(torch.from_numpy(inp["pairwise_t_matrix"]), H * 0.8, W * 0.8, 1)
    x = torch.from_numpy(inp["feat"]).to(DEV).requires_grad_(True)
    g = torch.Generator(device=DEV).manual_seed(C + 1)
    wgt = torch.randn(len(rl), C, H, W, generator=g, device=DEV)
    y = AttFusion(C)(x, rl, affine)
    (y * wgt).sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    y2 = att_fusion_forward(x2, rl, affine)
    _close("fusion forward", y, y2, 1e-4)
    (y2 * wgt).sum().backward()
    worst = _close("fusion grad x", x.grad, x2.grad)
    print(f"fusion backward C={C} {H}x{W} scenes {rl}: worst relative error {worst:.2e}")


@pytest.mark.parametrize("distort", ["zoom", "ego"])
def test_fusion_backward_falls_back_to_the_scatter_for_non_rigid_maps(distort):
    """The gather pass of gencomm_warp_attfuse_bwd is only taken for rigid-like transforms with an identity ego (decided on the device);
    a zooming transform (more than FUSE_KM output pixels may sample one source pixel) or a non-identity ego row must take the
    scatter with atomics and still match torch autograd."""
    from gencomm_amd import AttFusion, normalize_pairwise_tfm, synth
    from torch_restatements import att_fusion_forward
    C, H, W, rl = 16, 20, 28, [3, 2]
    inp = synth.make_inputs(rl, C, H, W, 11, max_shift=4.0)
    affine = normalize_pairwise_tfm(torch.from_numpy(inp["pairwise_t_matrix"]), H * 0.8, W * 0.8, 1).clone()
    if distort == "zoom":
        affine[0, 0, 1, :, :2] *= 2.5          # agent 1 of scene 0: the output samples a 2.5 x larger area of the source
        affine[1, 0, 1, :, :2] *= 0.3          # agent 1 of scene 1: magnified
    else:
        affine[0, 0, 0, 0, 2] = 0.07           # the ego of scene 0 is shifted: no identity warp
    x = torch.from_numpy(inp["feat"]).to(DEV).requires_grad_(True)
    g = torch.Generator(device=DEV).manual_seed(3)
    wgt = torch.randn(len(rl), C, H, W, generator=g, device=DEV)
    y = AttFusion(C)(x, rl, affine)
    (y * wgt).sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    y2 = att_fusion_forward(x2, rl, affine)
    _close("fusion forward", y, y2, 1e-4)
    (y2 * wgt).sum().backward()
    _close("fusion grad x", x.grad, x2.grad)


def test_fusion_backward_is_deterministic_for_rigid_maps():
    """Gather pass: no float atomics -- two runs give bit-identical gradients (the round-2 scatter did not)."""
    from gencomm_amd import AttFusion, normalize_pairwise_tfm, synth
    C, H, W, rl = 32, 40, 72, [4, 2]
    inp = synth.make_inputs(rl, C, H, W, 5, max_shift=20.0)
    affine = normalize_pairwise_tfm(torch.from_numpy(inp["pairwise_t_matrix"]), H * 0.8, W * 0.8, 1)
    g = torch.Generator(device=DEV).manual_seed(2)
    wgt = torch.randn(len(rl), C, H, W, generator=g, device=DEV)
    grads = []
    for _ in range(2):
        x = torch.from_numpy(inp["feat"]).to(DEV).requires_grad_(True)
        (AttFusion(C)(x, rl, affine) * wgt).sum().backward()
        grads.append(x.grad.clone())
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize("C,H,W,n", [(16, 12, 20, 2), (128, 16, 24, 1)])
def test_message_extractor_backward_vs_oracle_autograd(C, H, W, n):
    """MessageExtractorv2 is THE module stage 2 trains (stage2.py:99-101): HIP forward, HIP-composed backward; gradients of all
    12 parameters and of the input against float64 autograd through the oracle (offsets scaled so that the sampling positions
    cross pixels). The gradient of bilinear sampling is DISCONTINUOUS where a sampling position crosses an integer, and the HIP
    path's float32 offsets differ from the float64 ones by ~2e-5: the input is drawn (by seed) so that no position lies within
    1e-4 of an integer -- otherwise a handful of pixels legitimately land in the neighbouring cell (tools/diag_msgext_bwd.py)."""
    import torch.nn.functional as F
    from gencomm_amd import MessageExtractorv2, synth
    from oracle import torch_port as O
    me = MessageExtractorv2(C, 2).train()
    synth.fill_params_(me, 41)
    with torch.no_grad():
        me.bev_extractor.offset1.weight.mul_(5.0)
        me.bev_extractor.offset1.bias.mul_(5.0)
    for seed in range(200):
        x = torch.randn(n, C, H, W, generator=torch.Generator().manual_seed(1000 * C + seed))
        off = F.conv2d(x.double(), me.bev_extractor.offset1.weight.double(), me.bev_extractor.offset1.bias.double(), padding=1)
        if float((off - off.round()).abs().min()) > 1e-4:
            break
    else:
        pytest.skip("no seed keeps every sampling position away from the integers")
    sd = {k: v.detach().double().requires_grad_(True) for k, v in me.state_dict().items()}
    xd = x.double().requires_grad_(True)
    ref = O.message_extractor_forward(sd, xd)
    names = list(sd)
    rg = torch.autograd.grad((ref ** 2).mean(), [sd[k] for k in names] + [xd])
    me = me.to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    out = me(xg)
    _close("message", out, ref, 1e-4)
    (out ** 2).mean().backward()
    worst = _close("grad input", xg.grad, rg[-1])
    got = dict(me.named_parameters())
    for k, r in zip(names, rg[:-1]):
        assert got[k].grad is not None, k
        worst = max(worst, _close("grad " + k, got[k].grad, r))
    print(f"MessageExtractorv2 backward C={C} {H}x{W}: {len(names)} parameter gradients + input, worst relative error {worst:.2e}")


@pytest.mark.parametrize("new_agent,fusion", [("point_pillar", "att"), ("second", "att"), ("second", "v2xvit"), ("point_pillar", "where2comm")])
def test_stage2_training_step_reaches_only_the_new_agents_message_extractor(new_agent, fusion):
    """Stage 2 (heter_model_baseline_w_gencomm_stage2.py:99-101, :180-185): every module is frozen except the message
    extractor of the new (non-ego) modality. One training step through the stage-2 shell on the HIP path: loss.backward()
    must leave finite, non-zero gradients on message_extractor_m2 and on nothing else. The new agent is a second PointPillars
    model (m1m1-style) or a SECOND model (the shipped stage2/m1m3_att.yaml / m1m3_v2xvit.yaml pairings: sparse 3-D encoder, stride-1 first
    backbone block; AttFusion or the frozen V2X-ViT transformer as the fusion net)."""
    import copy, json, os
    from gencomm_amd import synth
    from gencomm_amd.heter_model_baseline_w_gencomm_stage2 import HeterModelBaselineWDiffCommStage2
    with open(os.path.join(REPO, "tests", "golden", "shell_state_dict_keys.json")) as f:
        args = copy.deepcopy(json.load(f)["args"])
    args["m2"] = copy.deepcopy(args["m1"])            # the new agent type: its own encoder / backbone / shrinker / extractor
    if fusion == "v2xvit":                            # stage2/m1m3_v2xvit.yaml: the frozen V2X-ViT (dropouts active) passes the gradient back
        from helpers import load_case
        args["fusion_method"] = "v2xvit"
        args["v2xvit"] = json.loads(str(load_case("v2xvit")["args"]))
    if fusion == "where2comm":                        # Diffcomm/*/m1_diffcomm_where2comm.yaml:130-131
        args["fusion_method"], args["where2comm"] = "where2comm", 128
    if new_agent == "second":
        args["m2"].update({"core_method": "second",
                           "encoder_args": {"voxel_size": [0.1, 0.1, 0.1], "lidar_range": args["lidar_range"], "mean_vfe": {"num_point_features": 4},
                                            "spconv": {"num_features_in": 4, "num_features_out": 64}, "map2bev": {"feature_num": 128}},
                           "backbone_args": {"layer_nums": [3, 5, 8], "layer_strides": [1, 2, 2], "num_filters": [64, 128, 256],
                                             "upsample_strides": [1, 2, 4], "num_upsample_filter": [128, 128, 128], "inplanes": 128}})
    model = HeterModelBaselineWDiffCommStage2(args)
    synth.fill_params_(model, 3)
    synth.fill_bn_stats_(model, 4)
    model = model.to(DEV).train()
    model.model_train_init_stage2()                   # train.py:136-139 calls the model's train-init each epoch
    trainable = sorted({n.split(".")[0] for n, p in model.named_parameters() if p.requires_grad})
    assert trainable == ["message_extractor_m2"], trainable
    rl = [2, 1]
    pil = synth.make_pillars(600, 3, 128, 64, 9, voxel_size=[0.4, 0.4, 4.0], pc_range=args["lidar_range"])
    ptm = synth.make_pairwise_t_matrix(rl, 5, 10, max_shift=4.0)
    coords = torch.from_numpy(pil["voxel_coords"])
    inputs = {k: torch.from_numpy(pil[k]) for k in ("voxel_features", "voxel_coords", "voxel_num_points")}
    # agents 0 and 2 are ego-type (m1), agent 1 is the new type (m2): split the pillars by agent, re-number the batch index
    def pick(agents):
        sel = torch.zeros(len(coords), dtype=torch.bool)
        out_c = coords.clone()
        for new_i, a in enumerate(agents):
            m = coords[:, 0] == a
            sel |= m
            out_c[m, 0] = new_i
        return {"voxel_features": inputs["voxel_features"][sel].to(DEV), "voxel_coords": out_c[sel].to(DEV),
                "voxel_num_points": inputs["voxel_num_points"][sel].to(DEV)}
    data = {"agent_modality_list": ["m1", "m2", "m1"], "record_len": torch.tensor(rl), "pairwise_t_matrix": torch.from_numpy(ptm).to(DEV),
            "inputs_m1": pick([0, 2]), "inputs_m2": pick([1])}
    if new_agent == "second":
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from test_second import _voxels
        vf, vc, vn = _voxels(np.random.RandomState(5), [3000], 512, 256, 40)
        data["inputs_m2"] = {"voxel_features": vf.to(DEV), "voxel_coords": vc.to(DEV), "voxel_num_points": vn.to(DEV)}
    out = model(data)
    loss = out["cls_preds"].square().mean() + out["reg_preds"].square().mean() + (out["pred_feature"] - out["gt_feature"]).square().mean()
    loss.backward()
    with_grad = sorted({n.split(".")[0] for n, p in model.named_parameters() if p.grad is not None})
    assert with_grad == ["message_extractor_m2"], with_grad
    tot = 0.0
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n
            tot += float(p.grad.abs().sum())
    assert tot > 0.0
    print(f"stage-2 training step: loss {float(loss):.4f}, sum |grad| over message_extractor_m2 {tot:.3e}")


@pytest.mark.parametrize("encoder", ["point_pillar", "second"])
def test_stage1_training_step_reaches_every_trained_module(encoder):
    """Stage 1 of the reference trains the whole model (train.py; BatchNorm with batch statistics in the encoder and the backbone,
    GenComm's training branch, Enhancer, fusion, heads). One training step through the stage-1 shell in train mode: finite gradients on
    the encoder, backbone, shrinker, message extractor, GenComm, the live Enhancer block and the heads; running statistics move."""
    import copy, json, os
    from gencomm_amd import synth
    from gencomm_amd.heter_model_baseline_w_gencomm_stage1 import HeterModelBaselineWGenCommStage1
    with open(os.path.join(REPO, "tests", "golden", "shell_state_dict_keys.json")) as f:
        args = copy.deepcopy(json.load(f)["args"])
    if encoder == "second":   # the m3 modality of the shipped stage1/m3_att.yaml as the (only) agent type
        args["m1"].update({"core_method": "second",
                           "encoder_args": {"voxel_size": [0.1, 0.1, 0.1], "lidar_range": args["lidar_range"], "mean_vfe": {"num_point_features": 4},
                                            "spconv": {"num_features_in": 4, "num_features_out": 64}, "map2bev": {"feature_num": 128}},
                           "backbone_args": {"layer_nums": [3, 5, 8], "layer_strides": [1, 2, 2], "num_filters": [64, 128, 256],
                                             "upsample_strides": [1, 2, 4], "num_upsample_filter": [128, 128, 128], "inplanes": 128}})
    model = HeterModelBaselineWGenCommStage1(args)
    synth.fill_params_(model, 3)
    synth.fill_bn_stats_(model, 4)
    model = model.to(DEV).train()
    rm0 = model.backbone_m1.blocks[0][2].running_mean.clone()
    rl = [2, 1]
    ptm = synth.make_pairwise_t_matrix(rl, 5, 10, max_shift=4.0)
    if encoder == "second":
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from test_second import _voxels
        vf, vc, vn = _voxels(np.random.RandomState(5), [3000, 2000, 2500], 512, 256, 40)
        inputs = {"voxel_features": vf.to(DEV), "voxel_coords": vc.to(DEV), "voxel_num_points": vn.to(DEV)}
    else:
        pil = synth.make_pillars(600, 3, 128, 64, 9, voxel_size=[0.4, 0.4, 4.0], pc_range=args["lidar_range"])
        inputs = {k: torch.from_numpy(pil[k]).to(DEV) for k in ("voxel_features", "voxel_coords", "voxel_num_points")}
    data = {"agent_modality_list": ["m1"] * 3, "record_len": torch.tensor(rl), "pairwise_t_matrix": torch.from_numpy(ptm).to(DEV), "inputs_m1": inputs}
    out = model(data)
    loss = (out["cls_preds"].square().mean() + out["reg_preds"].square().mean() + out["dir_preds"].square().mean()
            + (out["pred_feature"] - out["gt_feature"].detach()).square().mean())
    loss.backward()
    groups = {}
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n
            groups[n.split(".")[0]] = groups.get(n.split(".")[0], 0.0) + float(p.grad.abs().sum())
    for g in ("encoder_m1", "backbone_m1", "shrinker_m1", "message_extractor_m1", "gencomm", "enhancer", "cls_head", "reg_head", "dir_head"):
        assert groups.get(g, 0.0) > 0.0, (g, sorted(groups))
    assert not torch.equal(rm0, model.backbone_m1.blocks[0][2].running_mean)      # batch statistics were used and tracked
    print("stage-1 training step: loss %.4f, gradient mass per module: %s" % (float(loss), {k: "%.2e" % v for k, v in sorted(groups.items())}))


def test_training_loop_reduces_the_generation_loss():
    """Twenty Adam steps of GenComm's training branch (HIP forward + backward through T = 3 UNet calls) on one fixed scene batch:
    the generation loss || pred_feature - ego feature ||^2 (the reference's point_pillar_gencomm_loss feature term) must fall."""
    from gencomm_amd import GenComm, synth
    C, H, W, T, rl = 32, 32, 48, 3, [2, 2]
    gen = GenComm(synth.default_gencomm_cfg(C, T)).to(DEV).train()
    inp = {k: torch.from_numpy(v) for k, v in synth.make_inputs(rl, C, H, W, 50).items()}
    feat, cond = inp["feat"].to(DEV), inp["cond"].to(DEV)
    target = torch.cat([feat[0:1].expand(2, -1, -1, -1), feat[2:3].expand(2, -1, -1, -1)])          # every agent regenerates its scene's ego feature
    opt = torch.optim.Adam(gen.parameters(), lr=2e-3)
    losses = []
    for step in range(20):
        opt.zero_grad(set_to_none=True)
        pred = gen(feat, cond, inp["record_len"], seed=100 + step)["pred_feature"]
        loss = (pred - target).square().mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    print("generation loss over 20 Adam steps:", " ".join(f"{v:.4f}" for v in losses[::3]))
    assert all(np.isfinite(losses)) and losses[-1] < 0.7 * losses[0], losses


def test_gradient_path_validates_shapes_before_taking_pointers_and_refuses_a_second_backward():
    """with gradients enabled GenComm runs `SamplerChainFunction`, which hands raw device pointers to the library --
    the same ValueErrors as the no-grad path must come first (a wrong cond / noise shape was an out-of-bounds read)."""
    from gencomm_amd import GenComm, synth
    C, H, W, T, rl = 16, 16, 24, 3, [2]
    gen = GenComm(synth.default_gencomm_cfg(C, T)).train().to(DEV)
    inp = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_inputs(rl, C, H, W, 5).items()}
    n0, sn = (torch.from_numpy(a).to(DEV) for a in synth.make_train_noise(6, 2, C, H, W, T))
    feat, cond = inp["feat"], inp["cond"].clone().requires_grad_(True)
    with pytest.raises(ValueError, match="conditions must be"):
        gen(feat, cond[:, :1], rl, noise=(n0, sn))
    with pytest.raises(ValueError, match="conditions must be"):
        gen(feat, cond[:, :, :-1], rl, noise=(n0, sn))
    with pytest.raises(ValueError, match="channels"):
        gen(feat[:, :8], cond, rl, noise=(n0, sn))
    with pytest.raises(ValueError, match="noise must be"):
        gen(feat, cond, rl, noise=(n0[:1], sn))
    with pytest.raises(ValueError, match="noise must be"):
        gen(feat, cond, rl, noise=(n0, sn[:-1]))
    with pytest.raises(Exception, match="CPU|device|cuda|GPU"):
        gen(feat, cond, rl, noise=(n0.cpu(), sn))
    with pytest.raises(ValueError):
        gen(feat, cond, [3], noise=(n0, sn))
    pred = gen(feat, cond, rl, noise=(n0, sn))["pred_feature"]
    loss = pred.square().mean()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="already differentiated"):
        loss.backward()


def test_side_stream_weight_gradients_equal_the_single_stream_walk(modes):
    """GENCOMM_MODE_BWD_STREAMS: the UNet call's weight gradients on the library's side stream (2 = on every call) against the same
    walk on the caller's stream (0); and the call must leave its results ordered on the CALLER's stream: they are read right after it,
    on that stream, with no device synchronisation in between."""
    from gencomm_amd import GenComm, synth
    from gencomm_amd.autograd import UNetFunction
    C, H, W, n, T, t = 16, 48, 80, 3, 3, 1
    cfg = synth.default_gencomm_cfg(C, T)
    gen = GenComm(cfg).train()
    synth.fill_params_(gen, 77)
    gen = gen.to(DEV)
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(n, C, H, W, generator=g, device=DEV)
    cond = torch.randn(n, 2, H, W, generator=g, device=DEV)
    got = {}
    for mode in (0, 2, 2, 0):
        modes(bwd_streams=mode)
        for p in gen.parameters():
            p.grad = None
        xd, cd = x.clone().requires_grad_(True), cond.clone().requires_grad_(True)
        out = UNetFunction.apply(gen.denoiser, t, T, xd, cd, gen.denoiser.flat_params())
        (out ** 2).mean().backward()
        flat = torch.cat([p.grad.reshape(-1) for p in gen.denoiser.parameters()] + [xd.grad.reshape(-1), cd.grad.reshape(-1)]).clone()
        got.setdefault(mode, []).append(flat)
    ref = got[0][0]
    scale = float(ref.abs().max())
    for mode, runs in got.items():
        for r in runs:
            assert torch.isfinite(r).all()
            assert float((r - ref).abs().max()) <= 2e-6 * scale, (mode, float((r - ref).abs().max()), scale)   # float atomics of the reduce stage: order only
