"""AP evaluation -- host-side mirror of ``opencood/utils/eval_utils.py`` (:181-347) with the reference's function names
(including its spelling ``caluclate_tp_fp``), argument meaning and ``result_stat`` layout, so that the reference's
inference scripts (``opencood/tools/inference.py:171-185, :231-234``) can call it unchanged:

    result_stat = {0.3: {'tp': [], 'fp': [], 'gt': 0, 'score': []}, 0.5: {...}, 0.7: {...}}
    caluclate_tp_fp(pred_box_tensor, pred_score, gt_box_tensor, result_stat, 0.7)      # per frame
    ap30, ap50, ap70 = eval_final_results(result_stat, save_path, global_sort_detections)

The metric is CPU work in the reference as well (numpy + shapely polygons, float64); it is not part of the accelerated path.
The polygon IoU (shapely ``intersection.area / union.area`` in the reference, ``common_utils.py:230-252``) is a float64
convex-quadrilateral clip here -- no shapely / GEOS dependency. Checked against ``tests/golden/eval.npz`` (the reference's
own functions run on synthetic detections) in ``tests/test_eval.py``.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np
import torch


def _to_numpy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


_IOU_MEMO: dict = {}   # last (a, b) -> IoU matrix


def _signed_area(p: np.ndarray) -> np.ndarray:
    """p [..., K, 2] -> signed area [...]."""
    x, y = p[..., 0], p[..., 1]
    return 0.5 * np.sum(x * np.roll(y, -1, axis=-1) - np.roll(x, -1, axis=-1) * y, axis=-1)


def quad_iou_matrix(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """IoU of every convex quadrilateral a[i] (4, 2) with every b[j] -> [len(a), len(b)] float64 (Sutherland-Hodgman
    clipping of a[i] by the four edges of b[j]; union = area_a + area_b - intersection)."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 4, 2)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 4, 2)
    out = np.zeros((len(a), len(b)), dtype=np.float64)
    if len(a) == 0 or len(b) == 0:
        return out
    # one frame is scored at three IoU thresholds (inference.py:171-185): the matrix is computed once per (a, b)
    memo_key = (a.tobytes(), b.tobytes())
    if _IOU_MEMO.get("key") == memo_key:
        return _IOU_MEMO["iou"].copy()
    sa, sb = _signed_area(a), _signed_area(b)
    a = np.where((sa < 0)[:, None, None], a[:, ::-1], a)   # counter-clockwise
    b = np.where((sb < 0)[:, None, None], b[:, ::-1], b)
    area_a, area_b = np.abs(sa), np.abs(sb)
    # only pairs whose axis-aligned bounding boxes overlap can intersect (the clip below is a Python loop per pair)
    amin, amax, bmin, bmax = a.min(1), a.max(1), b.min(1), b.max(1)
    cand = np.all((amin[:, None, :] <= bmax[None, :, :]) & (bmin[None, :, :] <= amax[:, None, :]), axis=-1)
    for j in range(len(b)):
        for i in np.nonzero(cand[:, j])[0]:
            poly = a[i]
            for e in range(4):
                p0, p1 = b[j, e], b[j, (e + 1) % 4]
                if len(poly) == 0:
                    break
                ex, ey = p1[0] - p0[0], p1[1] - p0[1]
                side = ex * (poly[:, 1] - p0[1]) - ey * (poly[:, 0] - p0[0])
                nxt, snx = np.roll(poly, -1, axis=0), np.roll(side, -1)
                pts = []
                for k in range(len(poly)):
                    if side[k] >= 0:
                        pts.append(poly[k])
                    if (side[k] > 0 and snx[k] < 0) or (side[k] < 0 and snx[k] > 0):
                        pts.append(poly[k] + side[k] / (side[k] - snx[k]) * (nxt[k] - poly[k]))
                poly = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
            inter = abs(float(_signed_area(poly))) if len(poly) >= 3 else 0.0
            union = area_a[i] + area_b[j] - inter
            out[i, j] = inter / union if union > 0 else 0.0
    _IOU_MEMO["key"], _IOU_MEMO["iou"] = memo_key, out.copy()
    return out


def voc_ap(rec, prec):
    """VOC-2010 all-point average precision (eval_utils.py:181-204). Returns (ap, mrec, mpre)."""
    mrec = [0.0] + list(rec) + [1.0]
    mpre = [0.0] + list(prec) + [0.0]
    for i in range(len(mpre) - 2, -1, -1):
        mpre[i] = max(mpre[i], mpre[i + 1])
    ap = 0.0
    for i in range(1, len(mrec)):
        if mrec[i] != mrec[i - 1]:
            ap += (mrec[i] - mrec[i - 1]) * mpre[i]
    return ap, mrec, mpre


def caluclate_tp_fp(det_boxes, det_score, gt_boxes, result_stat: Dict[float, dict], iou_thresh: float) -> None:
    """True / false positives of one frame (eval_utils.py:207-261). det_boxes (N, 8, 3) or (N, 4, 2) or None, det_score (N,),
    gt_boxes (M, 8, 3) or (M, 4, 2); the first four corners' x, y form the BEV polygon (common_utils.convert_format)."""
    fp, tp = [], []
    gt = int(gt_boxes.shape[0])
    if det_boxes is not None:
        det, score, gtb = _to_numpy(det_boxes), _to_numpy(det_score), _to_numpy(gt_boxes)
        order = np.argsort(-score)
        score = score[order]
        iou = quad_iou_matrix(det[:, :4, :2], gtb[:, :4, :2]).astype(np.float32)  # compute_iou returns float32
        alive = list(range(gt))
        for i in order:
            if not alive or np.max(iou[i, alive]) < iou_thresh:
                fp.append(1)
                tp.append(0)
                continue
            fp.append(0)
            tp.append(1)
            alive.pop(int(np.argmax(iou[i, alive])))
        result_stat[iou_thresh]["score"] += score.tolist()
    result_stat[iou_thresh]["fp"] += fp
    result_stat[iou_thresh]["tp"] += tp
    result_stat[iou_thresh]["gt"] += gt


def calculate_ap(result_stat: Dict[float, dict], iou: float, global_sort_detections: bool):
    """eval_utils.py:264-318: (ap, mrec, mpre). ``global_sort_detections``: sort all frames' detections by score, else keep
    the per-frame order they were appended in. Works on copies (the reference accumulates in place in the second variant)."""
    s = result_stat[iou]
    if global_sort_detections:
        fp, tp, score = np.array(s["fp"]), np.array(s["tp"]), np.array(s["score"])
        assert len(fp) == len(tp) == len(score)
        idx = np.argsort(-score)
        fp, tp = fp[idx], tp[idx]
    else:
        fp, tp = np.array(s["fp"]), np.array(s["tp"])
        assert len(fp) == len(tp)
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = (tp / float(s["gt"])).tolist() if len(tp) else []
    prec = (tp / (fp + tp).astype(np.float64)).tolist() if len(tp) else []
    return voc_ap(rec, prec)


def eval_final_results(result_stat, save_path: Optional[str], global_sort_detections: bool, infer_info=None):
    """eval_utils.py:321-347: AP@0.3/0.5/0.7, dumped to ``eval[_global_sort][_<infer_info>].yaml`` under ``save_path`` (skipped when None)."""
    ap_30, _, _ = calculate_ap(result_stat, 0.30, global_sort_detections)
    ap_50, mrec_50, mpre_50 = calculate_ap(result_stat, 0.50, global_sort_detections)
    ap_70, mrec_70, mpre_70 = calculate_ap(result_stat, 0.70, global_sort_detections)
    if save_path is not None:
        import yaml
        dump = {"ap30": ap_30, "ap_50": ap_50, "ap_70": ap_70, "mpre_50": mpre_50, "mrec_50": mrec_50, "mpre_70": mpre_70, "mrec_70": mrec_70}
        name = ("eval" if not global_sort_detections else "eval_global_sort") + (f"_{infer_info}" if infer_info is not None else "")
        with open(os.path.join(save_path, name + ".yaml"), "w") as f:
            yaml.dump(dump, f, default_flow_style=False)
    print('The Average Precision at IOU 0.3 is %.4f, The Average Precision at IOU 0.5 is %.4f, '
          'The Average Precision at IOU 0.7 is %.4f' % (ap_30, ap_50, ap_70))
    return ap_30, ap_50, ap_70
