"""PointPillarGencommLoss (the caller on the training side of the path) against the reference's own criterion:
tests/golden/loss.npz was written by oracle/make_golden.py from opencood/loss/point_pillar_gencomm_loss.py on the
seeded synthetic maps of gencomm_amd.synth.make_loss_inputs -- total, components and every gradient."""
import json
import os

import numpy as np
import pytest
import torch

from gencomm_amd import synth
from gencomm_amd.point_pillar_gencomm_loss import PointPillarGencommLoss

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss.npz")


def _run(device):
    g = np.load(GOLD)
    B, H, W, A, C = (int(v) for v in g["dims"])
    t = {k: torch.from_numpy(v).to(device) for k, v in synth.make_loss_inputs(int(g["data_seed"]), B, H, W, A, C).items()}
    for k in ("cls_preds", "reg_preds", "dir_preds", "pred_feature"):
        t[k].requires_grad_(True)
    crit = PointPillarGencommLoss(json.loads(str(g["args"])))
    total = crit({k: t[k] for k in ("cls_preds", "reg_preds", "dir_preds", "gt_feature", "pred_feature")},
                 {k: t[k] for k in ("pos_equal_one", "neg_equal_one", "targets")})
    total.backward()
    assert all(isinstance(v, torch.Tensor) for v in crit.loss_dict.values())   # no host synchronisation inside forward
    for k in ("total_loss", "reg_loss", "cls_loss", "dir_loss", "generate_loss"):
        assert float(crit.loss_dict[k]) == pytest.approx(float(g[k]), rel=2e-6, abs=1e-7), k
    assert float(total.detach()) == pytest.approx(float(g["total"]), rel=2e-6)
    for k in ("cls_preds", "reg_preds", "dir_preds", "pred_feature"):
        np.testing.assert_allclose(t[k].grad.cpu().numpy(), g["grad_" + k], rtol=1e-5, atol=1e-8, err_msg=k)
    d = crit.logging(0, 0, 1)
    assert d["total_loss"] == pytest.approx(float(g["total"]), rel=2e-6)


def test_loss_matches_the_reference_criterion_cpu():
    _run("cpu")


@pytest.mark.gpu
def test_loss_matches_the_reference_criterion_gpu():
    _run("cuda:0")


def test_psm_rm_dm_aliases_and_record_len_forms():
    """point_pillar_loss.py:43-44, :59-65: `record_len` in the output dict sets the batch size; models that emit psm / rm / dm are
    renamed to cls_preds / reg_preds / dir_preds"""
    g = np.load(GOLD)
    B, H, W, A, C = (int(v) for v in g["dims"])
    t = {k: torch.from_numpy(v) for k, v in synth.make_loss_inputs(int(g["data_seed"]), B, H, W, A, C).items()}
    crit = PointPillarGencommLoss(json.loads(str(g["args"])))
    tgt = {k: t[k] for k in ("pos_equal_one", "neg_equal_one", "targets")}
    for rl in ([1] * B, torch.ones(B, dtype=torch.int64)):
        out = {"psm": t["cls_preds"], "rm": t["reg_preds"], "dm": t["dir_preds"], "gt_feature": t["gt_feature"], "pred_feature": t["pred_feature"],
               "record_len": rl}
        assert float(crit(out, tgt)) == pytest.approx(float(g["total"]), rel=2e-6)
        assert out["cls_preds"] is t["cls_preds"] and out["reg_preds"] is t["reg_preds"] and out["dir_preds"] is t["dir_preds"]


def test_resolver_name_and_unsupported_keys():
    import gencomm_amd.point_pillar_gencomm_loss as m
    # train_utils.create_loss: module `point_pillar_gencomm_loss`, class whose lower-cased name is the module name without underscores
    assert [n for n in dir(m) if n.lower() == "pointpillargencommloss"] == ["PointPillarGencommLoss"]
    args = json.loads(str(np.load(GOLD)["args"]))
    with pytest.raises(NotImplementedError):
        PointPillarGencommLoss(dict(args, iou={"sigma": 3.0, "weight": 1.0}))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["golden_shape", "wide", "no_positives", "gamma_1p5_no_dir"])
def test_one_launch_head_loss_equals_the_operator_composition(case):
    """gencomm_head_loss (csrc/loss_kernels.h) against the framework-operator composition of the same criterion on the GPU: totals,
    components and the gradients of every head map; a sample without positives (pos_norm clamps to 1), a non-quadratic focal exponent
    and a criterion without the direction term included."""
    g = np.load(GOLD)
    args = json.loads(str(g["args"]))
    B, H, W, A, C = (int(v) for v in g["dims"])
    if case == "wide":
        B, H, W = 3, 50, 88
    t = {k: torch.from_numpy(v).to("cuda:0") for k, v in synth.make_loss_inputs(7 + len(case), B, H, W, A, C).items()}
    if case == "no_positives":
        t["pos_equal_one"][0] = 0
    if case == "gamma_1p5_no_dir":
        args = dict(args, cls=dict(args["cls"], gamma=1.5))
        args.pop("dir")
    res = {}
    for fused in (True, False):
        crit = PointPillarGencommLoss(args)
        crit.fuse_heads = fused
        leaves = {k: t[k].clone().requires_grad_(True) for k in ("cls_preds", "reg_preds", "dir_preds", "pred_feature")}
        out = dict(leaves, gt_feature=t["gt_feature"])
        total = crit(out, {k: t[k] for k in ("pos_equal_one", "neg_equal_one", "targets")})
        total.backward()
        res[fused] = (float(total.detach()), {k: float(v) for k, v in crit.loss_dict.items()}, {k: v.grad for k, v in leaves.items()})
    assert res[True][0] == pytest.approx(res[False][0], rel=3e-6)
    assert set(res[True][1]) == set(res[False][1])
    for k, v in res[False][1].items():
        assert res[True][1][k] == pytest.approx(v, rel=3e-6, abs=1e-7), k
    for k, gr in res[False][2].items():
        if gr is None:                       # dir_preds without a direction term
            assert res[True][2][k] is None
            continue
        scale = float(gr.abs().max()) + 1e-30
        assert float((res[True][2][k] - gr).abs().max()) <= 2e-5 * scale, k
