"""ORACLE (test infrastructure, never imported by the product): CPU restatement of the detection tail
 -- anchors, box decoding, direction fix, corners, projection, size / z filters, rotated NMS,
range mask, and the axis-aligned overlap matrix used for target assignment.

Pinned by tests/golden/postproc.npz: the reference's own ``VoxelPostprocessor.post_process`` and helper functions run on
CPU by oracle/make_golden.py (`postproc` case). Two things in that run are NOT reference code and stay
PARITY UNPINNED: the polygon intersection / union areas inside ``nms_rotated`` come from shapely==2.0.0 (GEOS), which is
absent here, so ``common_utils.convert_format`` / ``compute_iou`` were bound to `quad_iou_one_to_many` below (convex
clipping in float64 -- the published Sutherland-Hodgman construction, same result as a polygon overlay for convex
quads up to rounding); everything else (the greedy loop, the sort, the thresholds) is the reference's.
`bbox_overlaps` is pinned against the reference's Cython source compiled into oracle/_ref (oracle/build_ref.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ anchors (voxel_postprocessor.py:68-121)
def generate_anchor_box(params: dict) -> np.ndarray:
    a = params["anchor_args"]
    W, H = a["W"], a["H"]
    r = [math.radians(e) for e in a["r"]]
    num = len(r)
    vh, vw = a["vh"], a["vw"]
    xr = [a["cav_lidar_range"][0], a["cav_lidar_range"][3]]
    yr = [a["cav_lidar_range"][1], a["cav_lidar_range"][4]]
    fs = a.get("feature_stride", 2)
    x = np.linspace(xr[0] + vw, xr[1] - vw, W // fs)
    y = np.linspace(yr[0] + vh, yr[1] - vh, H // fs)
    cx, cy = np.meshgrid(x, y)
    cx = np.tile(cx[..., np.newaxis], num)
    cy = np.tile(cy[..., np.newaxis], num)
    cz = np.ones_like(cx) * -1.0
    w = np.ones_like(cx) * a["w"]
    l = np.ones_like(cx) * a["l"]
    h = np.ones_like(cx) * a["h"]
    r_ = np.ones_like(cx)
    for i in range(num):
        r_[..., i] = r[i]
    if params["order"] == "hwl":
        return np.stack([cx, cy, cz, h, w, l, r_], axis=-1)
    if params["order"] == "lhw":
        return np.stack([cx, cy, cz, l, h, w, r_], axis=-1)
    raise ValueError("Unknown bbx order.")


# ------------------------------------------------------------------ decode (voxel_postprocessor.py:1351-1396)
def delta_to_boxes3d(deltas: torch.Tensor, anchors: torch.Tensor) -> torch.Tensor:
    N = deltas.shape[0]
    deltas = deltas.permute(0, 2, 3, 1).contiguous().view(N, -1, 7)
    boxes = torch.zeros_like(deltas)
    an = anchors.view(-1, 7).float()
    d = torch.sqrt(an[:, 4] ** 2 + an[:, 5] ** 2)
    d = d.repeat(N, 2, 1).transpose(1, 2)
    an = an.repeat(N, 1, 1)
    boxes[..., [0, 1]] = torch.mul(deltas[..., [0, 1]], d) + an[..., [0, 1]]
    boxes[..., [2]] = torch.mul(deltas[..., [2]], an[..., [3]]) + an[..., [2]]
    boxes[..., [3, 4, 5]] = torch.exp(deltas[..., [3, 4, 5]]) * an[..., [3, 4, 5]]
    boxes[..., 6] = deltas[..., 6] + an[..., 6]
    return boxes


def limit_period(val: torch.Tensor, offset=0.5, period=2 * np.pi) -> torch.Tensor:  # common_utils.py:104-113
    return val - torch.floor(val / period + offset) * period


def boxes_to_corners_3d(boxes3d: torch.Tensor, order: str) -> torch.Tensor:  # box_utils.py:152-204
    b = boxes3d[:, [0, 1, 2, 5, 4, 3, 6]] if order == "hwl" else boxes3d
    template = b.new_tensor(([1, -1, -1], [1, 1, -1], [-1, 1, -1], [-1, -1, -1],
                             [1, -1, 1], [1, 1, 1], [-1, 1, 1], [-1, -1, 1])) / 2
    corners = b[:, None, 3:6].repeat(1, 8, 1) * template[None, :, :]
    cosa, sina = torch.cos(b[:, 6]), torch.sin(b[:, 6])  # rotate_points_along_z, common_utils.py:139-161
    zeros, ones = cosa.new_zeros(b.shape[0]), cosa.new_ones(b.shape[0])
    rot = torch.stack((cosa, sina, zeros, -sina, cosa, zeros, zeros, zeros, ones), dim=1).view(-1, 3, 3).float()
    corners = torch.matmul(corners.view(-1, 8, 3).float(), rot)
    return corners + b[:, None, 0:3]


def project_box3d(box3d: torch.Tensor, T: torch.Tensor) -> torch.Tensor:  # box_utils.py:278-316
    c = box3d.transpose(1, 2)
    c = torch.cat((c, torch.ones((c.shape[0], 1, 8))), dim=1)
    return torch.matmul(T, c)[:, :3, :].transpose(1, 2)


def remove_large_pred_bbx(b: torch.Tensor) -> torch.Tensor:  # box_utils.py:1062-1091 (the z extent is taken from y, as there)
    x_len = b[:, :, 0].max(1)[0] - b[:, :, 0].min(1)[0]
    y_len = b[:, :, 1].max(1)[0] - b[:, :, 1].min(1)[0]
    z_len = b[:, :, 1].max(1)[0] - b[:, :, 1].min(1)[0]
    return torch.logical_and(torch.logical_and(x_len <= 6, y_len <= 6), z_len)


def remove_bbx_abnormal_z(b: torch.Tensor) -> torch.Tensor:  # box_utils.py:1094-1112
    return torch.logical_and(b[:, :, 2].min(1)[0] >= -3, b[:, :, 2].max(1)[0] <= 1)


def mask_boxes_outside_range(corners: np.ndarray, limit_range, min_num_corners=8) -> np.ndarray:  # box_utils.py:384-421
    lr = np.asarray(limit_range)
    mask = ((corners >= lr[0:3]) & (corners <= lr[3:6])).all(axis=2)
    return mask.sum(axis=1) >= min_num_corners


# ------------------------------------------------------------------ rotated NMS (box_utils.py:915-960)
def _clip(subject: np.ndarray, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Keep the part of polygon `subject` on the left of (or on) the directed edge a->b."""
    out = []
    n = len(subject)
    if n == 0:
        return subject
    ex, ey = b[0] - a[0], b[1] - a[1]
    side = ex * (subject[:, 1] - a[1]) - ey * (subject[:, 0] - a[0])
    for i in range(n):
        j = (i + 1) % n
        pi, pj, si, sj = subject[i], subject[j], side[i], side[j]
        if si >= 0:
            out.append(pi)
        if (si > 0 and sj < 0) or (si < 0 and sj > 0):
            t = si / (si - sj)
            out.append(pi + t * (pj - pi))
    return np.asarray(out, dtype=np.float64).reshape(-1, 2)


def _area(p: np.ndarray) -> float:
    if len(p) < 3:
        return 0.0
    x, y = p[:, 0], p[:, 1]
    return 0.5 * float(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))


def quad_iou(p: np.ndarray, q: np.ndarray) -> float:
    """IoU of two convex quadrilaterals given as (4, 2) corner arrays (either orientation), float64:
    intersection by Sutherland-Hodgman clipping, union = area(p) + area(q) - intersection
    (common_utils.py:230-252: shapely `intersection(...).area / union(...).area`)."""
    p = np.asarray(p, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64)
    ap, aq = _area(p), _area(q)
    if ap < 0:
        p, ap = p[::-1], -ap
    if aq < 0:
        q, aq = q[::-1], -aq
    poly = p
    for i in range(4):
        poly = _clip(poly, q[i], q[(i + 1) % 4])
        if len(poly) == 0:
            break
    inter = abs(_area(poly))
    union = ap + aq - inter
    return inter / union if union > 0 else 0.0


def quad_iou_one_to_many(box: np.ndarray, boxes: np.ndarray) -> np.ndarray:
    return np.array([quad_iou(box, b) for b in boxes], dtype=np.float32)


def nms_rotated(boxes: np.ndarray, scores: np.ndarray, threshold: float, top: int = 1000) -> np.ndarray:
    """boxes (N, 8, 3) or (N, 4, 2): the first four corners' x, y form the BEV quadrilateral.
    Ties in `scores` are ordered by DESCENDING index (a stable ascending argsort, reversed); the reference's
    `argsort()[::-1]` leaves the order of exact ties to numpy's unstable default sort."""
    if boxes.shape[0] == 0:
        return np.array([], dtype=np.int32)
    quads = np.asarray(boxes, dtype=np.float64)[:, :4, :2]
    ixs = np.argsort(scores, kind="stable")[::-1][:top]
    pick = []
    while len(ixs) > 0:
        i = ixs[0]
        pick.append(i)
        iou = quad_iou_one_to_many(quads[i], quads[ixs[1:]])
        remove = np.where(iou > np.float32(threshold))[0] + 1
        ixs = np.delete(ixs, remove)
        ixs = np.delete(ixs, 0)
    return np.array(pick, dtype=np.int32)


# ------------------------------------------------------------------ whole tail, one agent (voxel_postprocessor.py:1084-1244)
def post_process(cls_preds: torch.Tensor, reg_preds: torch.Tensor, dir_preds: Optional[torch.Tensor], anchor_box: torch.Tensor,
                 transformation_matrix: torch.Tensor, params: dict) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    prob = torch.sigmoid(cls_preds.permute(0, 2, 3, 1)).reshape(1, -1)
    batch_box3d = delta_to_boxes3d(reg_preds, anchor_box)
    mask = torch.gt(prob, params["target_args"]["score_threshold"]).view(1, -1)
    boxes3d = torch.masked_select(batch_box3d[0], mask.unsqueeze(2).repeat(1, 1, 7)[0]).view(-1, 7)
    scores = torch.masked_select(prob[0], mask[0])
    if dir_preds is not None and len(boxes3d) != 0:
        off, nb = params["dir_args"]["dir_offset"], params["dir_args"]["num_bins"]
        dcp = dir_preds.permute(0, 2, 3, 1).contiguous().reshape(1, -1, nb)[mask]
        labels = torch.max(dcp, dim=-1)[1]
        period = 2 * np.pi / nb
        rot = limit_period(boxes3d[..., 6] - off, 0, period)
        boxes3d[..., 6] = rot + off + period * labels.to(dcp.dtype)
        boxes3d[..., 6] = limit_period(boxes3d[..., 6], 0.5, 2 * np.pi)
    if len(boxes3d) == 0:
        return None, None
    corners = project_box3d(boxes_to_corners_3d(boxes3d, params["order"]), transformation_matrix)
    keep = torch.logical_and(remove_large_pred_bbx(corners), remove_bbx_abnormal_z(corners))
    corners, scores = corners[keep], scores[keep]
    k = nms_rotated(corners.numpy(), scores.numpy(), params["nms_thresh"])
    corners, scores = corners[k], scores[k]
    m = mask_boxes_outside_range(corners.numpy(), params["gt_range"])
    return corners[torch.from_numpy(m)], scores[torch.from_numpy(m)]


# ------------------------------------------------------------------ axis-aligned overlaps (box_overlaps.pyx:17-57)
def bbox_overlaps(boxes: np.ndarray, query: np.ndarray) -> np.ndarray:
    """(N, 4), (K, 4) float32 [x1, y1, x2, y2] -> (N, K) float32, the `+ 1` pixel convention of the reference.
    Precision follows the C that Cython emits for the source: coordinate differences are float, the literal `1` is a
    double, so `d + 1` and the area products are double; `box_area`, `iw`, `ih`, `ua` are float variables (rounded on
    assignment); `iw * ih` and the final division are float."""
    f32, f64 = np.float32, np.float64
    b, q = boxes.astype(f32), query.astype(f32)
    qa = ((q[:, 2] - q[:, 0]).astype(f64) + 1.0) * ((q[:, 3] - q[:, 1]).astype(f64) + 1.0)
    qa = qa.astype(f32)
    ba = ((b[:, 2] - b[:, 0]).astype(f64) + 1.0) * ((b[:, 3] - b[:, 1]).astype(f64) + 1.0)  # stays double inside `ua`
    iw = ((np.minimum(b[:, None, 2], q[None, :, 2]) - np.maximum(b[:, None, 0], q[None, :, 0])).astype(f64) + 1.0).astype(f32)
    ih = ((np.minimum(b[:, None, 3], q[None, :, 3]) - np.maximum(b[:, None, 1], q[None, :, 1])).astype(f64) + 1.0).astype(f32)
    inter = iw * ih  # float
    ua = (ba[:, None] + qa[None, :].astype(f64) - inter.astype(f64)).astype(f32)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.where((iw > 0) & (ih > 0), inter / ua, f32(0))
    return out.astype(f32)
