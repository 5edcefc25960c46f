"""AP evaluation (the parity metric of SURVEY.md 8d: AP@0.5 / AP@0.7 within +-0.1): the oracle restatement
(oracle/eval_port.py) and the product's host-side mirror (gencomm_amd/eval_utils.py) against tests/golden/eval.npz -- the
reference's own caluclate_tp_fp / calculate_ap / voc_ap run on synthetic frames (oracle/make_golden.py run_eval_case; the
frames are regenerated here from the stored seed by the same numpy generator). Bit-exact tp / fp lists, AP to 1e-12."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))

from helpers import load_case


def _frames(seed):
    import make_golden  # the generator of the fixture's inputs (numpy only; no reference import at module level)
    return make_golden.make_eval_frames(int(seed))


def _run(mod, frames, as_tensor):
    stat = {t: {"tp": [], "fp": [], "gt": 0, "score": []} for t in (0.3, 0.5, 0.7)}
    for det, score, gt in frames:
        conv = (lambda a: None if a is None else torch.from_numpy(a)) if as_tensor else (lambda a: a)
        for thr in (0.3, 0.5, 0.7):
            mod.caluclate_tp_fp(conv(det), conv(score), conv(gt), stat, thr)
    return stat


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_ap_matches_reference_golden(which):
    g = load_case("eval")
    frames = _frames(g["seed"])
    assert len(frames) == int(g["n_frames"])
    if which == "oracle":
        import eval_port as mod
        stat = _run(mod, frames, as_tensor=False)
    else:
        from gencomm_amd import eval_utils as mod
        stat = _run(mod, frames, as_tensor=True)
    for thr in (0.3, 0.5, 0.7):
        k = str(thr)
        assert stat[thr]["tp"] == g["tp_" + k].tolist() and stat[thr]["fp"] == g["fp_" + k].tolist()
        assert stat[thr]["gt"] == int(g["gt_" + k])
        np.testing.assert_array_equal(np.array(stat[thr]["score"], dtype=np.float64), g["score_" + k])
        for gs in (True, False):
            ap, mrec, mpre = mod.calculate_ap(copy.deepcopy(stat), thr, gs)
            assert abs(ap - float(g[f"ap_{thr}_{int(gs)}"])) < 1e-12
            np.testing.assert_allclose(np.array(mrec), g[f"mrec_{thr}_{int(gs)}"], rtol=0, atol=1e-12)
            np.testing.assert_allclose(np.array(mpre), g[f"mpre_{thr}_{int(gs)}"], rtol=0, atol=1e-12)
    if which == "product":
        ap30, ap50, ap70 = mod.eval_final_results(stat, None, True)
        assert abs(ap50 - float(g["ap_0.5_1"])) < 1e-12 and abs(ap70 - float(g["ap_0.7_1"])) < 1e-12 and abs(ap30 - float(g["ap_0.3_1"])) < 1e-12


def test_voc_ap_known_answers():
    from gencomm_amd.eval_utils import voc_ap
    assert voc_ap([], [])[0] == 0.0                                  # no detection at all
    assert abs(voc_ap([0.5, 1.0], [1.0, 1.0])[0] - 1.0) < 1e-15       # every detection right, every object found
    ap, mrec, mpre = voc_ap([0.5, 0.5, 1.0], [1.0, 0.5, 2 / 3])       # tp, fp, tp over two objects
    assert abs(ap - (0.5 * 1.0 + 0.5 * 2 / 3)) < 1e-15 and mpre[2] == 2 / 3  # envelope lifts the dip


def test_product_iou_matrix_vs_oracle_clipping():
    import detect_port as D
    from gencomm_amd.eval_utils import quad_iou_matrix
    frames = _frames(7)
    det, _, gt = frames[0]
    m = quad_iou_matrix(det[:, :4, :2], gt[:, :4, :2])
    for i in range(len(det)):
        ref = D.quad_iou_one_to_many(det[i, :4, :2], gt[:, :4, :2]).astype(np.float64)
        np.testing.assert_allclose(m[i], ref, rtol=0, atol=1e-6)
    sq = np.array([[[0, 0], [2, 0], [2, 2], [0, 2]]], dtype=np.float64)
    assert abs(quad_iou_matrix(sq, sq + [1.0, 0.0])[0, 0] - 1.0 / 3.0) < 1e-15   # half overlap: 2 / (4 + 4 - 2)
    assert quad_iou_matrix(sq, sq + 5.0)[0, 0] == 0.0
    assert abs(quad_iou_matrix(sq, sq[:, ::-1])[0, 0] - 1.0) < 1e-15               # orientation does not matter


def test_frames_without_detections_or_ground_truth():
    from gencomm_amd import eval_utils as E
    stat = {0.5: {"tp": [], "fp": [], "gt": 0, "score": []}}
    gt = np.zeros((3, 8, 3), np.float32)
    E.caluclate_tp_fp(None, None, torch.from_numpy(gt), stat, 0.5)
    assert stat[0.5] == {"tp": [], "fp": [], "gt": 3, "score": []}
    det = np.random.RandomState(0).rand(2, 8, 3).astype(np.float32)
    E.caluclate_tp_fp(torch.from_numpy(det), torch.tensor([0.4, 0.9]), torch.zeros(0, 8, 3), stat, 0.5)
    assert stat[0.5]["fp"] == [1, 1] and stat[0.5]["tp"] == [0, 0] and stat[0.5]["score"] == pytest.approx([0.9, 0.4])
