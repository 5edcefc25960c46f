"""Where a stage-1 training step (bench.py --workload train, one rank, no DDP) spends its time: host enqueue vs total, a cProfile of one
step, and the framework (aten) operators of one step with their call counts (torch.profiler)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gencomm_amd import synth
from gencomm_amd.heter_model_baseline_w_gencomm_stage1 import HeterModelBaselineWGenCommStage1
from gencomm_amd.point_pillar_gencomm_loss import PointPillarGencommLoss
dev = torch.device("cuda:0")
B, N = 2, 2
margs = synth.stage1_model_args(T=3)
torch.manual_seed(0)
model = HeterModelBaselineWGenCommStage1(margs)
synth.fill_params_(model, 3); synth.fill_bn_stats_(model, 4)
model = model.to(dev).train()
crit = PointPillarGencommLoss(synth.STAGE1_LOSS_ARGS)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=2e-3, eps=1e-10, weight_decay=1e-4, fused=True)
n = B * N
pil = synth.make_pillars(12000 * n, n, 512, 256, 9, voxel_size=[0.4, 0.4, 4.0], pc_range=margs["lidar_range"])
ptm = synth.make_pairwise_t_matrix([N] * B, 5, 10, max_shift=20.0)
data = {"agent_modality_list": ["m1"] * n, "record_len": torch.tensor([N] * B), "pairwise_t_matrix": torch.from_numpy(ptm).to(dev),
        "inputs_m1": {k: torch.from_numpy(pil[k]).to(dev) for k in ("voxel_features", "voxel_coords", "voxel_num_points")}}
labels = None
def step():
    global labels
    opt.zero_grad(set_to_none=True)
    out = model(data)
    if labels is None:
        _, A, Hh, Wh = out["cls_preds"].shape
        li = synth.make_loss_inputs(50, B, Hh, Wh, A, 1)
        labels = {k: torch.from_numpy(li[k]).to(dev) for k in ("pos_equal_one", "neg_equal_one", "targets")}
    loss = crit(out, labels)
    loss.backward()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(4):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0):.1f} ms, total {1e3 * (t2 - t0):.1f} ms")
# forward-only and backward-only split
torch.cuda.synchronize(); t0 = time.perf_counter(); out = model(data); loss = crit(out, labels); torch.cuda.synchronize(); t1 = time.perf_counter()
loss.backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"forward {1e3 * (t1 - t0):.1f} ms, backward {1e3 * (t2 - t1):.1f} ms (synchronised)")
pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=False) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=40))
