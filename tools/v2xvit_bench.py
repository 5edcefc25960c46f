#!/usr/bin/env python3
"""Forward latency of V2XViTFusion (HIP path) with the shipped m1_v2xvit.yaml transformer block at the shipped map size
(C=128, 64x128) for 2 and 5 agents, and at 100x352 (the 200x704 grid after the backbone's stride 2) for 4 agents."""
import json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from gencomm_amd import normalize_pairwise_tfm, synth
from gencomm_amd.v2xvit import V2XViTFusion
import _mode
_mode.apply_env_modes()   # GENCOMM_TOOL_ARITH=3: the opt-in two-term general convolutions

args = json.loads(str(np.load(os.path.join(REPO, "tests", "golden", "v2xvit.npz"))["args"]))
net = V2XViTFusion(args).eval()
synth.fill_params_(net, 5)
net = net.cuda()
for (H, W, rl) in ((64, 128, [2]), (64, 128, [5]), (96, 352, [4])):
    inp = synth.make_inputs(rl, 128, H, W, 9, max_shift=10.0)
    aff = normalize_pairwise_tfm(torch.from_numpy(inp["pairwise_t_matrix"]), H * 0.8, W * 0.8, 1)
    x = torch.from_numpy(inp["feat"]).cuda()
    with torch.no_grad():
        for _ in range(3):
            net(x, rl, aff)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 10
        for _ in range(K):
            net(x, rl, aff)
        torch.cuda.synchronize()
    print(f"v2xvit forward: {sum(rl)} agents, C=128, {H}x{W}: {1e3 * (time.perf_counter() - t0) / K:.2f} ms per scene", flush=True)

# Where2commFusion (fusion_method: where2comm) at the same shapes: forward, and forward + backward (stage-2 training passes the gradient
# through the frozen fusion net)
from gencomm_amd.where2comm import Where2commFusion
w2c = Where2commFusion(128).eval()
synth.fill_params_(w2c, 6)
w2c = w2c.cuda()
for (H, W, rl) in ((64, 128, [2]), (64, 128, [5]), (96, 352, [4])):
    inp = synth.make_inputs(rl, 128, H, W, 9, max_shift=10.0)
    aff = normalize_pairwise_tfm(torch.from_numpy(inp["pairwise_t_matrix"]), H * 0.8, W * 0.8, 1).cuda()
    x = torch.from_numpy(inp["feat"]).cuda()
    K = 10
    with torch.no_grad():
        for _ in range(3):
            w2c(x, torch.tensor(rl), aff)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            w2c(x, torch.tensor(rl), aff)
        torch.cuda.synchronize()
        fwd = 1e3 * (time.perf_counter() - t0) / K
    xg = x.clone().requires_grad_(True)
    for _ in range(2):
        w2c(xg, torch.tensor(rl), aff).square().mean().backward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        w2c(xg, torch.tensor(rl), aff).square().mean().backward()
    torch.cuda.synchronize()
    print(f"where2comm: {sum(rl)} agents, C=128, {H}x{W}: forward {fwd:.2f} ms, forward + backward {1e3 * (time.perf_counter() - t0) / K:.2f} ms per scene", flush=True)
