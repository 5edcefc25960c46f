"""ORACLE (test infrastructure): ctypes access to oracle/csrc/detect_port.c -- the plain-C restatement of the reference's
iou3d_nms arithmetic and of spconv's point-to-voxel (see the header of that file for the reference lines and the
'parity unpinned' statements). Built by `build()` (gcc, -ffp-contract=off so that float32 operations are not fused) into
oracle/_build/, which `__graft_entry__.build()` calls; the built .so travels to the GPU box with the snapshot."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "detect_port.c")
OUT_DIR = os.path.join(HERE, "_build")
SO = os.path.join(OUT_DIR, "libdetect_port.so")
_lib = None


def build(verbose: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if os.path.exists(SO) and os.path.getmtime(SO) >= os.path.getmtime(SRC):
        return SO
    cmd = ["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", SRC, "-lm", "-o", SO]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        l = C.CDLL(SO)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        l.gc_oracle_boxes_pairwise.argtypes = [fp, C.c_int, fp, C.c_int, C.c_int, fp]
        l.gc_oracle_nms.argtypes = [fp, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_int64)]
        l.gc_oracle_nms.restype = C.c_int
        l.gc_oracle_points_to_voxel.argtypes = [fp, C.c_int, C.c_int, fp, fp, C.c_int, C.c_int, fp, ip, ip]
        l.gc_oracle_points_to_voxel.restype = C.c_int
        _lib = l
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def boxes_overlap_bev(boxes_a, boxes_b):
    a, pa = _f(boxes_a); b, pb = _f(boxes_b)
    out = np.zeros((len(a), len(b)), np.float32)
    lib().gc_oracle_boxes_pairwise(pa, len(a), pb, len(b), 0, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def boxes_iou_bev(boxes_a, boxes_b):
    a, pa = _f(boxes_a); b, pb = _f(boxes_b)
    out = np.zeros((len(a), len(b)), np.float32)
    lib().gc_oracle_boxes_pairwise(pa, len(a), pb, len(b), 1, out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def boxes_iou3d(boxes_a, boxes_b):
    """iou3d_nms_utils.py:147-181 (float32 torch arithmetic restated in numpy float32)."""
    a, b = np.asarray(boxes_a, np.float32), np.asarray(boxes_b, np.float32)
    ov = boxes_overlap_bev(a, b)
    a_max, a_min = (a[:, 2] + a[:, 5] / 2)[:, None], (a[:, 2] - a[:, 5] / 2)[:, None]
    b_max, b_min = (b[:, 2] + b[:, 5] / 2)[None, :], (b[:, 2] - b[:, 5] / 2)[None, :]
    oh = np.clip(np.minimum(a_max, b_max) - np.maximum(a_min, b_min), 0, None).astype(np.float32)
    o3 = ov * oh
    va, vb = (a[:, 3] * a[:, 4] * a[:, 5])[:, None], (b[:, 3] * b[:, 4] * b[:, 5])[None, :]
    return (o3 / np.clip(va + vb - o3, 1e-6, None)).astype(np.float32)


def nms(boxes, scores, thresh, normal=False, pre_maxsize=None):
    """iou3d_nms_utils.py:255-289: indices into the ORIGINAL box array, in keep order. Ties in `scores` keep input order
    (stable), torch's sort on equal keys is not specified by the reference."""
    boxes, scores = np.asarray(boxes, np.float32), np.asarray(scores, np.float32)
    order = np.argsort(-scores, kind="stable")
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    sb, ps = _f(boxes[order])
    keep = np.zeros(len(order), np.int64)
    n = lib().gc_oracle_nms(ps, len(order), float(thresh), int(normal), keep.ctypes.data_as(C.POINTER(C.c_int64)))
    return order[keep[:n]]


def points_to_voxel(points, voxel_size, lidar_range, max_points, max_voxels):
    pts, pp = _f(points)
    vs, pv = _f(voxel_size); rg, pr = _f(lidar_range)
    nf = pts.shape[1]
    voxels = np.zeros((max_voxels, max_points, nf), np.float32)
    coords = np.zeros((max_voxels, 3), np.int32)
    npts = np.zeros((max_voxels,), np.int32)
    m = lib().gc_oracle_points_to_voxel(pp, len(pts), nf, pv, pr, max_points, max_voxels, voxels.ctypes.data_as(C.POINTER(C.c_float)),
                                        coords.ctypes.data_as(C.POINTER(C.c_int32)), npts.ctypes.data_as(C.POINTER(C.c_int32)))
    return voxels[:m], coords[:m], npts[:m]
