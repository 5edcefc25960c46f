"""``VoxelPostprocessor`` -- host-side mirror of the reference's detection tail for inference
(opencood/data_utils/post_processor/voxel_postprocessor.py: ``generate_anchor_box`` :68-121, ``post_process``
:1084-1244) and ``bbox_overlaps`` (opencood/utils/box_overlaps.pyx:17-57). SURVEY.md 8f rank 3.

``post_process(data_dict, output_dict)`` takes the reference's dictionaries -- per agent id:
``data_dict[cav]['transformation_matrix']`` (4x4), ``['anchor_box']`` ([H, W, A, 7]);
``output_dict[cav]['cls_preds' | 'reg_preds' | 'dir_preds']`` (also the ``psm / rm / dm`` spellings) -- and returns
``(pred_box3d_tensor [M, 8, 3], scores [M])`` or ``(None, None)``. Sigmoid, score filter, box decoding, direction fix,
corners, projection, the size and z filters, the score sort, the rotated IoU (float64), the greedy suppression and the
range mask all run in the HIP library; the only host round trip is the final read of M (the reference goes through numpy
three times on the same path). Anchor generation is constructor-time numpy, as in the reference.
Not mirrored: the training-time target assignment (``generate_label``), the v2xreal multi-class variants, ``iou_preds``.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from .runtime import f32c, ptr, require_gpu, stream_ptr


class VoxelPostprocessor:
    def __init__(self, anchor_params: dict, train: bool = False):
        self.params = anchor_params
        self.train = train
        self.anchor_num = self.params["anchor_args"]["num"]
        self._cache = {}

    # ------------------------------------------------------------------ anchors (numpy, constructor-time)
    def generate_anchor_box(self) -> np.ndarray:
        a = self.params["anchor_args"]
        W, H = a["W"], a["H"]
        r = [math.radians(e) for e in a["r"]]
        assert self.anchor_num == len(r)
        vh, vw = a["vh"], a["vw"]
        xrange = [a["cav_lidar_range"][0], a["cav_lidar_range"][3]]
        yrange = [a["cav_lidar_range"][1], a["cav_lidar_range"][4]]
        fs = a["feature_stride"] if "feature_stride" in a else 2
        x = np.linspace(xrange[0] + vw, xrange[1] - vw, W // fs)
        y = np.linspace(yrange[0] + vh, yrange[1] - vh, H // fs)
        cx, cy = np.meshgrid(x, y)
        cx = np.tile(cx[..., np.newaxis], self.anchor_num)
        cy = np.tile(cy[..., np.newaxis], self.anchor_num)
        cz = np.ones_like(cx) * -1.0
        w, l, h = np.ones_like(cx) * a["w"], np.ones_like(cx) * a["l"], np.ones_like(cx) * a["h"]
        r_ = np.ones_like(cx)
        for i in range(self.anchor_num):
            r_[..., i] = r[i]
        if self.params["order"] == "hwl":
            return np.stack([cx, cy, cz, h, w, l, r_], axis=-1)
        if self.params["order"] == "lhw":
            return np.stack([cx, cy, cz, l, h, w, r_], axis=-1)
        raise ValueError("Unknown bbx order.")

    # ------------------------------------------------------------------ inference tail
    def _buffers(self, device, H, W, A):
        key = (str(device), H, W, A)
        if key not in self._cache:
            l = _lib.lib()
            cap = min(H * W * A, l.gencomm_nms_max_candidates())
            top = 1000
            ws = max(_lib.check_size(l.gencomm_det_workspace_bytes(H, W, A), "gencomm_det_workspace_bytes"),
                     _lib.check_size(l.gencomm_nms_workspace_bytes(), "gencomm_nms_workspace_bytes"))
            self._cache[key] = dict(
                cap=cap, top=top,
                corners=torch.empty(cap, 8, 3, dtype=torch.float32, device=device), scores=torch.empty(cap, dtype=torch.float32, device=device),
                aidx=torch.empty(cap, dtype=torch.int32, device=device), counts=torch.zeros(2, dtype=torch.int32, device=device),
                out_boxes=torch.empty(top, 8, 3, dtype=torch.float32, device=device), out_scores=torch.empty(top, dtype=torch.float32, device=device),
                out_index=torch.empty(top, dtype=torch.int32, device=device), ws=torch.empty(ws, dtype=torch.uint8, device=device),
                range6=torch.tensor([float(v) for v in self.params["gt_range"]], dtype=torch.float32, device=device))
        return self._cache[key]

    def post_process(self, data_dict, output_dict) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        l = _lib.lib()
        buf = None
        for cav_id in output_dict.keys():
            assert cav_id in data_dict
            cav, out = data_dict[cav_id], output_dict[cav_id]
            cls = out["psm"] if "psm" in out else out["cls_preds"]
            reg = out["rm"] if "rm" in out else out["reg_preds"]
            dirp = out["dm"] if "dm" in out else out.get("dir_preds")
            if "iou_preds" in out:
                raise NotImplementedError("iou_preds rescoring is not part of this build")
            require_gpu(cls, "VoxelPostprocessor.post_process")
            if reg.dim() != 4 or cls.shape[0] != 1:
                raise NotImplementedError("anchor-based heads with batch size 1 (as the reference asserts, :1153)")
            dev = cls.device
            anchors = cav["anchor_box"]
            anchors = torch.as_tensor(anchors).to(device=dev, dtype=torch.float32).contiguous()
            H, W, A = anchors.shape[:3]
            if tuple(cls.shape) != (1, A, H, W) or tuple(reg.shape) != (1, 7 * A, H, W):
                raise ValueError(f"head shapes {tuple(cls.shape)} / {tuple(reg.shape)} do not match anchors {tuple(anchors.shape)}")
            nb = int(self.params["dir_args"]["num_bins"]) if dirp is not None else 0
            T = torch.as_tensor(cav["transformation_matrix"]).to(device=dev, dtype=torch.float32).contiguous()
            st = stream_ptr(dev)
            if buf is None:
                buf = self._buffers(dev, H, W, A)
                buf["counts"].zero_()
            elif buf["corners"].device != dev:
                raise ValueError("all agents of one call must live on the same device")
            cls, reg = f32c(cls), f32c(reg)
            dirp = f32c(dirp) if dirp is not None else None
            _lib.check(l.gencomm_det_decode_fwd(
                ptr(cls), ptr(reg), ptr(dirp), ptr(anchors), ptr(T), H, W, A, nb,
                float(self.params["target_args"]["score_threshold"]),
                float(self.params["dir_args"]["dir_offset"]) if dirp is not None else 0.0,
                1 if self.params["order"] == "hwl" else 0,
                ptr(buf["corners"]), ptr(buf["scores"]), ptr(buf["aidx"]), ptr(buf["counts"][0:1]), buf["cap"],
                ptr(buf["ws"]), buf["ws"].numel(), st), "gencomm_det_decode_fwd")
        if buf is None:
            return None, None
        st = stream_ptr(buf["corners"].device)
        _lib.check(l.gencomm_nms_rotated_fwd(
            ptr(buf["corners"]), ptr(buf["scores"]), ptr(buf["counts"][0:1]), float(self.params["nms_thresh"]), buf["top"],
            ptr(buf["range6"]), ptr(buf["out_boxes"]), ptr(buf["out_scores"]), ptr(buf["out_index"]), ptr(buf["counts"][1:2]),
            ptr(buf["ws"]), buf["ws"].numel(), st), "gencomm_nms_rotated_fwd")
        n_cand, m = (int(v) for v in buf["counts"].tolist())  # the one host synchronisation of the tail
        if n_cand > buf["cap"]:
            raise RuntimeError(f"{n_cand} candidates above the score threshold exceed the capacity {buf['cap']} of the device sort")
        if n_cand == 0:
            return None, None
        return buf["out_boxes"][:m].clone(), buf["out_scores"][:m].clone()


def bbox_overlaps(boxes: torch.Tensor, query_boxes: torch.Tensor) -> torch.Tensor:
    """(N, 4), (K, 4) float32 [x1, y1, x2, y2] on the GPU -> (N, K) overlaps, box_overlaps.pyx:17-57."""
    require_gpu(boxes, "bbox_overlaps")
    b, q = f32c(boxes), f32c(query_boxes)
    if b.dim() != 2 or q.dim() != 2 or b.shape[1] != 4 or q.shape[1] != 4:
        raise ValueError("expected (N, 4) and (K, 4)")
    out = torch.zeros(b.shape[0], q.shape[0], dtype=torch.float32, device=b.device)
    _lib.check(_lib.lib().gencomm_bbox_overlaps_fwd(ptr(b), ptr(q), ptr(out), b.shape[0], q.shape[0], stream_ptr(b.device)),
               "gencomm_bbox_overlaps_fwd")
    return out
