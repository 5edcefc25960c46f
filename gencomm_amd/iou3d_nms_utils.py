"""``iou3d_nms_utils`` -- the reference's Python API for its iou3d_nms CUDA extension
(``opencood/pcdet_utils/iou3d_nms/iou3d_nms_utils.py``: ``boxes_iou_bev`` :32-46, ``boxes_iou3d_gpu`` :147-181, ``nms_gpu``
:255-271, ``nms_normal_gpu`` :274-289) with the same names, argument meaning and return values, on the HIP kernels of
``csrc/iou3d_kernels.h`` through the C ABI. Boxes are ``(N, 7) [x, y, z, dx, dy, dz, heading]`` float32 device tensors.
No CPU fallback: a CPU tensor raises.
"""
from __future__ import annotations

import torch

from . import _lib
from .runtime import f32c, ptr, require_gpu, stream_ptr, workspaces


def _pairwise(boxes_a: torch.Tensor, boxes_b: torch.Tensor, mode: int) -> torch.Tensor:
    require_gpu(boxes_a, "iou3d_nms_utils")
    require_gpu(boxes_b, "iou3d_nms_utils")
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a, b = f32c(boxes_a), f32c(boxes_b)
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    _lib.check(_lib.lib().gencomm_iou3d_pairwise_fwd(ptr(a), a.shape[0], ptr(b), b.shape[0], mode, ptr(out), stream_ptr(a.device)),
               "gencomm_iou3d_pairwise_fwd")
    return out


def boxes_overlap_bev(boxes_a, boxes_b):
    """(N, M) BEV overlap areas (``iou3d_nms_cuda.boxes_overlap_bev_gpu``)."""
    return _pairwise(boxes_a, boxes_b, 0)


def boxes_iou_bev(boxes_a, boxes_b):
    """(N, M) BEV IoU (iou3d_nms_utils.py:32-46)."""
    return _pairwise(boxes_a, boxes_b, 1)


def boxes_iou3d_gpu(boxes_a, boxes_b, return_union=False):
    """(N, M) 3-D IoU = BEV overlap x height overlap / union volume (iou3d_nms_utils.py:147-181)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    a_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    b_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_bev = boxes_overlap_bev(boxes_a, boxes_b)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    union = torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)
    iou3d = overlaps_3d / union
    return (iou3d, union) if return_union else iou3d


def _nms(boxes: torch.Tensor, scores: torch.Tensor, thresh: float, normal: bool, pre_maxsize=None):
    require_gpu(boxes, "iou3d_nms_utils.nms")
    assert boxes.shape[1] == 7
    order = scores.sort(dim=0, descending=True, stable=True)[1]  # ties keep input order (unspecified in the reference)
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = f32c(boxes[order])
    n = b.shape[0]
    l = _lib.lib()
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=b.device)
    count = torch.zeros(1, dtype=torch.int32, device=b.device)
    ws = workspaces.get(b.device, _lib.check_size(l.gencomm_iou3d_nms_workspace_bytes(n), "gencomm_iou3d_nms_workspace_bytes"), "iou3d_nms")
    _lib.check(l.gencomm_iou3d_nms_fwd(ptr(b), n, float(thresh), int(normal), ptr(keep), ptr(count), ptr(ws), ws.numel(), stream_ptr(b.device)),
               "gencomm_iou3d_nms_fwd")
    num_out = int(count.item())  # the reference returns the count on the host as well (iou3d_nms.cpp:135)
    return order[keep[:num_out]].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """Rotated-BEV NMS (iou3d_nms_utils.py:255-271): (indices into `boxes` in keep order, None)."""
    return _nms(boxes, scores, thresh, False, pre_maxsize)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """Axis-aligned BEV NMS ignoring the heading (iou3d_nms_utils.py:274-289)."""
    return _nms(boxes, scores, thresh, True)
