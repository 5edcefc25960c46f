#!/usr/bin/env python3
"""End-to-end latency of the stage-1 model shell (HeterModelBaselineWGenCommStage1) at the geometry of the shipped
opv2v/GenComm_yamls/gencomm/stage1/m1_att.yaml: lidar range +-102.4 x +-51.2 m, 0.4 m pillars (512 x 256 grid), PointPillars -> BaseBEVBackbone [3, 5, 8]
-> shrink (stride 2) -> 128 x 64 x 128 features -> message extractor -> GenComm (T = 3) -> Enhancer -> AttFusion (or V2X-ViT) -> heads; synthetic
pillars, random weights. Prints milliseconds per scene with one scene in flight and the share of each stage (forward hooks + device synchronisation
in a second, instrumented pass).   python tools/shell_bench.py [--agents 2] [--fusion att|v2xvit|where2comm] [--pillars 12000]"""
import argparse, copy, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from gencomm_amd import synth
from gencomm_amd.heter_model_baseline_w_gencomm_stage1 import HeterModelBaselineWGenCommStage1
import _mode
_mode.apply_env_modes()   # GENCOMM_TOOL_ARITH=3: the opt-in two-term general convolutions

ap = argparse.ArgumentParser()
ap.add_argument("--agents", type=int, default=2)
ap.add_argument("--fusion", default="att")
ap.add_argument("--pillars", type=int, default=12000, help="non-empty pillars per agent")
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
with open(os.path.join(REPO, "tests", "golden", "shell_state_dict_keys.json")) as f:
    args = copy.deepcopy(json.load(f)["args"])
rng = [-102.4, -51.2, -3, 102.4, 51.2, 1]
args["lidar_range"] = rng
args["m1"]["encoder_args"]["lidar_range"] = rng
args["m1"]["backbone_args"]["layer_nums"] = [3, 5, 8]
if a.fusion == "v2xvit":
    args["fusion_method"] = "v2xvit"
    args["v2xvit"] = json.loads(str(np.load(os.path.join(REPO, "tests", "golden", "v2xvit.npz"))["args"]))
elif a.fusion == "where2comm":
    args["fusion_method"], args["where2comm"] = "where2comm", 128
model = HeterModelBaselineWGenCommStage1(args).eval()
synth.fill_params_(model, 3)
synth.fill_bn_stats_(model, 4)
model = model.to(dev)
N = a.agents
pil = synth.make_pillars(a.pillars * N, N, 512, 256, 9, voxel_size=[0.4, 0.4, 4.0], pc_range=rng)
ptm = synth.make_pairwise_t_matrix([N], 5, 10, max_shift=20.0)
data = {"agent_modality_list": ["m1"] * N, "record_len": torch.tensor([N]), "pairwise_t_matrix": torch.from_numpy(ptm).to(dev),
        "inputs_m1": {k: torch.from_numpy(pil[k]).to(dev) for k in ("voxel_features", "voxel_coords", "voxel_num_points")}}
with torch.no_grad():
    for _ in range(3):
        out = model(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        out = model(data)
    torch.cuda.synchronize()
    total = 1e3 * (time.perf_counter() - t0) / a.iters
    # stage shares: synchronise around every top-level child
    acc, marks = {}, {}
    hooks = []
    for name, mod in model.named_children():
        def pre(m, i, name=name):
            torch.cuda.synchronize(); marks[name] = time.perf_counter()
        def post(m, i, o, name=name):
            torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - marks[name]
        hooks += [mod.register_forward_pre_hook(pre), mod.register_forward_hook(post)]
    for _ in range(5):
        model(data)
    for h_ in hooks:
        h_.remove()
print(f"stage-1 shell, {N} agents x {a.pillars} pillars, fusion {a.fusion}: {total:.2f} ms per scene (cls map {tuple(out['cls_preds'].shape)})")
print("  synchronised per-stage times (ms): " + ", ".join(f"{k} {1e3 * v / 5:.2f}" for k, v in sorted(acc.items(), key=lambda kv: -kv[1])))
