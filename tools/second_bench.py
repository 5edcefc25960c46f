#!/usr/bin/env python3
"""Forward latency of the SECOND encoder (HIP sparse path) at the shipped m3 geometry: range +-102.4 x +-51.2 x [-3, 1] m,
0.1 m voxels = 2048 x 1024 x 40 cells (sparse shape [41, 1024, 2048]), 5 points per voxel, synthetic sweeps: a ground
disc plus vertical clusters, N voxels per agent (max_voxel_test is 70000, m3_att.yaml:40-43)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from gencomm_amd import second
from gencomm_amd.second import SECOND
import _mode
_mode.apply_env_modes()   # GENCOMM_TOOL_ARITH=3: the opt-in two-term general convolutions

NX, NY, NZ = 2048, 1024, 40
args = {"voxel_size": [0.1, 0.1, 0.1], "lidar_range": [-102.4, -51.2, -3, 102.4, 51.2, 1], "mean_vfe": {"num_point_features": 4},
        "spconv": {"num_features_in": 4, "num_features_out": 64}, "map2bev": {"feature_num": 128}}
net = SECOND(args).eval().cuda()


def sweep(rng, n):
    """n unique voxels: 60 % on a noisy ground surface with 1/r density, 40 % in vertical clusters (cars, walls, poles)."""
    r = np.exp(rng.uniform(np.log(20), np.log(900), size=3 * n))
    th = rng.uniform(0, 2 * np.pi, size=3 * n)
    g = np.stack([np.clip(12 + rng.standard_normal(3 * n) * 0.7, 0, NZ), 512 + r * np.sin(th) * 0.5, 1024 + r * np.cos(th)], 1)
    k = max(n // 150, 1)
    cen = np.stack([rng.uniform(12, 24, k), rng.uniform(100, 924, k), rng.uniform(200, 1848, k)], 1)
    c = cen[rng.randint(0, k, 2 * n)] + rng.standard_normal((2 * n, 3)) * [4, 6, 6]
    pts = np.concatenate([g[: int(1.8 * n)], c[: int(1.2 * n)]])
    pts = np.round(pts).astype(np.int64)
    ok = (pts >= 0).all(1) & (pts[:, 0] <= NZ) & (pts[:, 1] < NY) & (pts[:, 2] < NX)
    pts = pts[ok]
    rng.shuffle(pts)
    _, first = np.unique((pts[:, 0] * NY + pts[:, 1]) * NX + pts[:, 2], return_index=True)
    return pts[np.sort(first)][:n]


rng = np.random.RandomState(0)
for agents, n in ((1, 70000), (2, 70000), (5, 70000), (2, 32000)):
    coords = np.concatenate([np.concatenate([np.full((n, 1), b), sweep(rng, n)[:n]], 1) for b in range(agents)])
    m = len(coords)
    vc = torch.from_numpy(coords).int().cuda()
    vf = torch.randn(m, 5, 4, device="cuda")
    vn = torch.randint(1, 6, (m,), dtype=torch.int32, device="cuda")
    data = {"inputs_m3": {"voxel_features": vf, "voxel_coords": vc, "voxel_num_points": vn}}
    with torch.no_grad():
        for _ in range(2):
            out = net(data, "m3")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 10
        for _ in range(K):
            out = net(data, "m3")
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    with torch.no_grad():
        second.TRACE = []
        net(data, "m3")
        pairs = [(ci, co, int(nbr.shape[1]), int((nbr >= 0).sum())) for ci, co, nbr in second.TRACE]
        second.TRACE = None
    flops = sum(2.0 * p * ci * co for ci, co, _, p in pairs)
    print("  sites per layer:", [n for _, _, n, _ in pairs], f"| active (site, offset) pairs {sum(p for *_, p in pairs) / 1e6:.1f} M | "
          f"{flops / 1e9:.1f} GFLOP of sparse MACs -> {flops / dt / 1e12:.1f} TFLOP/s end to end (indexing included)")
    print(f"SECOND forward: {agents} agent(s) x {m // agents} voxels -> {tuple(out.shape)}: {1e3 * dt:.2f} ms "
          f"({float((out != 0).float().mean()) * 100:.1f} % of the BEV map non-zero)", flush=True)
