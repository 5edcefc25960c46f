"""Detection tail. CPU: the oracle restatement against the reference's own VoxelPostprocessor
output (tests/golden/postproc.npz) and against the compiled reference bbox_overlaps. GPU: the HIP kernels through the
C ABI against the same vectors -- kept-box SET and order exact, coordinates within fp32 rounding."""
import json

import numpy as np
import pytest
import torch

from helpers import load_case
from gencomm_amd import synth


def _case(g, tag):
    params = json.loads(str(g["params"]))
    H, W, A = int(g["H"]), int(g["W"]), int(g["A"])
    cls, reg, dirp = synth.make_detection_maps(H, W, A, int(g[f"seed_{tag}"]))
    return params, torch.from_numpy(cls), torch.from_numpy(reg), torch.from_numpy(dirp), torch.from_numpy(g[f"T_{tag}"])


def test_oracle_anchor_boxes_match_reference():
    from oracle import detect_port as D
    g = load_case("postproc")
    np.testing.assert_array_equal(D.generate_anchor_box(json.loads(str(g["params"]))), g["anchors"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_post_process_matches_reference(tag):
    from oracle import detect_port as D
    g = load_case("postproc")
    params, cls, reg, dirp, T = _case(g, tag)
    boxes, scores = D.post_process(cls, reg, dirp, torch.from_numpy(g["anchors"]), T, params)
    np.testing.assert_array_equal(scores.numpy(), g[f"scores_{tag}"])
    np.testing.assert_array_equal(boxes.numpy(), g[f"boxes_{tag}"])


def test_oracle_bbox_overlaps_matches_compiled_reference_bit_exact():
    from oracle import detect_port as D
    g = load_case("postproc")
    np.testing.assert_array_equal(D.bbox_overlaps(g["ov_boxes"], g["ov_query"]), g["ov"])


def test_oracle_quad_iou_known_answers():
    from oracle import detect_port as D
    sq = np.array([[0, 0], [2, 0], [2, 2], [0, 2]], dtype=np.float64)
    assert D.quad_iou(sq, sq) == pytest.approx(1.0, abs=1e-15)
    assert D.quad_iou(sq, sq + [1, 0]) == pytest.approx(2 / 6, abs=1e-15)
    assert D.quad_iou(sq, sq[::-1] + [1, 1]) == pytest.approx(1 / 7, abs=1e-15)      # opposite orientation
    assert D.quad_iou(sq, sq + [5, 5]) == 0.0
    c, s = np.cos(np.pi / 4), np.sin(np.pi / 4)
    diamond = (sq - 1) @ np.array([[c, s], [-s, c]]) + 1                            # same square rotated 45 deg about its centre
    assert D.quad_iou(sq, diamond) == pytest.approx((8 * np.sqrt(2) - 8) / (8 - (8 * np.sqrt(2) - 8)), abs=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_hip_post_process_vs_reference_golden(tag):
    from gencomm_amd.postprocess import VoxelPostprocessor
    g = load_case("postproc")
    params, cls, reg, dirp, T = _case(g, tag)
    pp = VoxelPostprocessor(params, train=False)
    np.testing.assert_array_equal(pp.generate_anchor_box(), g["anchors"])
    dev = "cuda:0"
    data = {"ego": {"transformation_matrix": T.to(dev), "anchor_box": torch.from_numpy(g["anchors"])}}
    out = {"ego": {"cls_preds": cls.to(dev), "reg_preds": reg.to(dev), "dir_preds": dirp.to(dev)}}
    boxes, scores = pp.post_process(data, out)
    ref_b, ref_s = g[f"boxes_{tag}"], g[f"scores_{tag}"]
    assert boxes.shape == ref_b.shape and scores.shape == ref_s.shape          # same number of kept boxes ...
    np.testing.assert_allclose(scores.cpu().numpy(), ref_s, rtol=0, atol=2e-7)  # ... in the same order (sigmoid: 1 ulp)
    np.testing.assert_allclose(boxes.cpu().numpy(), ref_b, rtol=0, atol=3e-5)   # fp32 trig / matmul rounding at |x| <= 40 m


@pytest.mark.gpu
def test_hip_post_process_two_agents_and_empty():
    """Late-fusion style call (two agents appended into one candidate list) against the oracle; and no candidate at all."""
    from gencomm_amd.postprocess import VoxelPostprocessor
    from oracle import detect_port as D
    g = load_case("postproc")
    dev = "cuda:0"
    pa, cls_a, reg_a, dir_a, T_a = _case(g, "a")
    _, cls_b, reg_b, dir_b, T_b = _case(g, "b")
    anchors = torch.from_numpy(g["anchors"])
    pp = VoxelPostprocessor(pa)
    data = {"ego": {"transformation_matrix": T_a.to(dev), "anchor_box": anchors}, "1": {"transformation_matrix": T_b.to(dev), "anchor_box": anchors}}
    out = {"ego": {"cls_preds": cls_a.to(dev), "reg_preds": reg_a.to(dev), "dir_preds": dir_a.to(dev)},
           "1": {"cls_preds": cls_b.to(dev), "reg_preds": reg_b.to(dev), "dir_preds": dir_b.to(dev)}}
    boxes, scores = pp.post_process(data, out)
    # oracle: decode each agent, stack, one NMS
    cs, ss = [], []
    for cls, reg, dirp, T in ((cls_a, reg_a, dir_a, T_a), (cls_b, reg_b, dir_b, T_b)):
        p = dict(pa); p = {**pa, "nms_thresh": 2.0, "gt_range": [-1e9] * 3 + [1e9] * 3}  # no suppression, no range mask
        c, s = D.post_process(cls, reg, dirp, anchors, T, p)
        order = np.argsort(-s.numpy(), kind="stable")  # undo the score sort of the (disabled) NMS: any order works for stacking
        cs.append(c); ss.append(s)
    c_all, s_all = torch.cat(cs), torch.cat(ss)
    k = D.nms_rotated(c_all.numpy(), s_all.numpy(), pa["nms_thresh"])
    c_k, s_k = c_all[k], s_all[k]
    m = D.mask_boxes_outside_range(c_k.numpy(), pa["gt_range"])
    np.testing.assert_allclose(scores.cpu().numpy(), s_k.numpy()[m], rtol=0, atol=2e-7)
    np.testing.assert_allclose(boxes.cpu().numpy(), c_k.numpy()[m], rtol=0, atol=3e-5)
    out_empty = {"ego": {"cls_preds": torch.full_like(cls_a, -9.0).to(dev), "reg_preds": reg_a.to(dev), "dir_preds": dir_a.to(dev)}}
    assert pp.post_process({"ego": data["ego"]}, out_empty) == (None, None)


@pytest.mark.gpu
def test_hip_bbox_overlaps_bit_exact():
    from gencomm_amd.postprocess import bbox_overlaps
    g = load_case("postproc")
    got = bbox_overlaps(torch.from_numpy(g["ov_boxes"]).cuda(), torch.from_numpy(g["ov_query"]).cuda()).cpu().numpy()
    np.testing.assert_array_equal(got, g["ov"])
