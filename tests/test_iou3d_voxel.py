"""iou3d_nms (the reference extension's semantics) and the spconv-style voxeliser.

CPU: the C oracle (oracle/csrc/detect_port.c) against known answers -- both pieces are PARITY UNPINNED (the reference's CPU
twin iou3d_cpu.cpp needs <cuda.h>, spconv is not under /root/reference), so they are anchored by closed forms.
GPU: the HIP kernels through the C ABI / the reference-named Python wrappers against the oracle: voxel indices, counts and
NMS keep lists bit-exact, overlap areas to 1e-5 (device vs glibc cosf / sinf / atan2f differ in the last ulp)."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import native_port as N  # noqa: E402


def random_boxes(n, seed, extent=40.0):
    r = np.random.RandomState(seed)
    b = np.zeros((n, 7), np.float32)
    b[:, 0:2] = r.uniform(-extent, extent, (n, 2))
    b[:, 2] = r.uniform(-1, 1, n)
    b[:, 3] = r.uniform(3.5, 5.5, n)
    b[:, 4] = r.uniform(1.6, 2.2, n)
    b[:, 5] = r.uniform(1.4, 2.0, n)
    b[:, 6] = r.uniform(-np.pi, np.pi, n)
    # clusters of near-duplicates so that NMS has something to suppress
    for i in range(0, n - 3, 4):
        b[i + 1] = b[i] + r.normal(0, [0.3, 0.3, 0.05, 0.1, 0.05, 0.05, 0.05]).astype(np.float32)
        b[i + 2] = b[i] + r.normal(0, [1.0, 0.8, 0.05, 0.1, 0.05, 0.05, 0.3]).astype(np.float32)
    return b


def random_points(n, seed, rng=(-140.8, -40, -3, 140.8, 40, 1)):
    r = np.random.RandomState(seed)
    p = np.zeros((n, 4), np.float32)
    # dense near the origin (many points per pillar), sparse far out, some outside the range, some on cell edges
    rad = np.abs(r.normal(0, 35, n))
    ang = r.uniform(-np.pi, np.pi, n)
    p[:, 0], p[:, 1] = rad * np.cos(ang), rad * np.sin(ang) * 0.5
    p[:, 2] = r.uniform(-3.5, 1.5, n)
    p[:, 3] = r.uniform(0, 1, n)
    edge = r.rand(n) < 0.02
    p[edge, 0] = np.round(p[edge, 0] / 0.4) * 0.4
    return p


# ------------------------------------------------------------------------------------------ CPU: oracle known answers
def test_oracle_iou3d_known_answers():
    sq = np.array([[0, 0, 0, 2, 2, 1, 0.0]], np.float32)
    sh = np.array([[1, 0, 0, 2, 2, 1, 0.0]], np.float32)
    rot = np.array([[0, 0, 0, 2, 2, 1, np.pi / 4]], np.float32)
    far = np.array([[10, 10, 0, 1, 1, 1, 0.3]], np.float32)
    assert abs(N.boxes_overlap_bev(sq, sq)[0, 0] - 4.0) < 1e-5 and abs(N.boxes_iou_bev(sq, sq)[0, 0] - 1.0) < 1e-6
    assert abs(N.boxes_overlap_bev(sq, sh)[0, 0] - 2.0) < 1e-5 and abs(N.boxes_iou_bev(sq, sh)[0, 0] - 1 / 3) < 1e-6
    assert abs(N.boxes_overlap_bev(sq, rot)[0, 0] - 8 * (np.sqrt(2) - 1)) < 1e-4      # regular octagon
    assert N.boxes_overlap_bev(sq, far)[0, 0] == 0.0
    a3, b3 = np.array([[0, 0, 0, 2, 2, 2, 0.0]], np.float32), np.array([[0, 0, 1, 2, 2, 2, 0.0]], np.float32)
    assert abs(N.boxes_iou3d(a3, b3)[0, 0] - 4.0 / 12.0) < 1e-6                        # half the height overlaps
    boxes = np.concatenate([sq, sh, rot, far])
    assert N.nms(boxes, np.array([0.9, 0.8, 0.7, 0.6], np.float32), 0.3).tolist() == [0, 3]
    assert N.nms(boxes, np.array([0.9, 0.8, 0.7, 0.6], np.float32), 0.5).tolist() == [0, 1, 3]   # 1/3 survives, 0.707 does not
    assert N.nms(boxes, np.array([0.1, 0.8, 0.7, 0.6], np.float32), 0.3, normal=True).tolist() == [1, 3]


def test_oracle_points_to_voxel_known_answers():
    pts = np.array([[0.1, 0.1, 0.0, 1], [0.15, 0.1, 0.0, 2], [5.0, 5.0, 0, 3], [-1, 0, 0, 4], [0.1, 0.1, 0, 5], [7.99, 0.01, 1.9, 6]], np.float32)
    v, c, k = N.points_to_voxel(pts, [0.4, 0.4, 4], [0, 0, -2, 8, 8, 2], 2, 10)
    assert c.tolist() == [[0, 0, 0], [0, 12, 12], [0, 0, 19]] and k.tolist() == [2, 1, 1]
    assert v[0, :, 3].tolist() == [1.0, 2.0] and v[1, 0, 3] == 3.0 and (v[1, 1] == 0).all()    # third point of cell 0 dropped, padding zero
    v, c, k = N.points_to_voxel(pts, [0.4, 0.4, 4], [0, 0, -2, 8, 8, 2], 2, 2)                  # max_voxels: the later cell is dropped
    assert c.tolist() == [[0, 0, 0], [0, 12, 12]]


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("na,nb", [(1, 1), (37, 130), (300, 257)])
def test_hip_iou3d_pairwise_vs_oracle(na, nb):
    from gencomm_amd import iou3d_nms_utils as U
    a, b = random_boxes(na, 1), random_boxes(nb, 2)
    b[: min(na, nb)] = a[: min(na, nb)] + np.random.RandomState(3).normal(0, 0.2, (min(na, nb), 7)).astype(np.float32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    ov, iou, iou3 = U.boxes_overlap_bev(ta, tb).cpu().numpy(), U.boxes_iou_bev(ta, tb).cpu().numpy(), U.boxes_iou3d_gpu(ta, tb).cpu().numpy()
    np.testing.assert_allclose(ov, N.boxes_overlap_bev(a, b), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(iou, N.boxes_iou_bev(a, b), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(iou3, N.boxes_iou3d(a, b), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("n,thresh", [(1, 0.5), (63, 0.3), (64, 0.1), (65, 0.7), (1000, 0.15), (4099, 0.5)])
def test_hip_iou3d_nms_keep_lists_vs_oracle(n, thresh):
    from gencomm_amd import iou3d_nms_utils as U
    boxes = random_boxes(n, 10 + n, extent=25.0 if n > 500 else 40.0)
    scores = np.random.RandomState(n).uniform(0, 1, n).astype(np.float32)
    tb, ts = torch.from_numpy(boxes).cuda(), torch.from_numpy(scores).cuda()
    keep, _ = U.nms_gpu(tb, ts, thresh)
    assert keep.cpu().numpy().tolist() == N.nms(boxes, scores, thresh).tolist()
    keep, _ = U.nms_gpu(tb, ts, thresh, pre_maxsize=max(1, n // 2))
    assert keep.cpu().numpy().tolist() == N.nms(boxes, scores, thresh, pre_maxsize=max(1, n // 2)).tolist()
    keep, _ = U.nms_normal_gpu(tb, ts, thresh)
    assert keep.cpu().numpy().tolist() == N.nms(boxes, scores, thresh, normal=True).tolist()


@pytest.mark.gpu
def test_hip_iou3d_empty_and_cpu_tensor():
    from gencomm_amd import _lib, iou3d_nms_utils as U
    e = torch.zeros(0, 7, device="cuda")
    assert U.boxes_iou_bev(e, torch.zeros(3, 7, device="cuda")).shape == (0, 3)
    keep, _ = U.nms_gpu(e, torch.zeros(0, device="cuda"), 0.5)
    assert keep.numel() == 0
    with pytest.raises(_lib.GenCommHipError):
        U.boxes_iou_bev(torch.zeros(2, 7), torch.zeros(2, 7))


@pytest.mark.gpu
@pytest.mark.parametrize("n,vs,max_pts,max_vox", [(0, [0.4, 0.4, 4], 32, 100), (1, [0.4, 0.4, 4], 32, 100), (5000, [0.4, 0.4, 4], 32, 70000),
                                                  (120000, [0.4, 0.4, 4], 32, 32000), (60000, [0.4, 0.4, 4], 5, 3000),
                                                  (80000, [0.1, 0.1, 0.1], 5, 40000)])
def test_hip_voxelizer_bit_exact_vs_oracle(n, vs, max_pts, max_vox):
    from gencomm_amd.sp_voxel_preprocessor import SpVoxelPreprocessor
    rng = [-140.8, -40, -3, 140.8, 40, 1]
    pts = random_points(n, 20 + n)
    pp = SpVoxelPreprocessor({"cav_lidar_range": rng, "args": {"voxel_size": vs, "max_points_per_voxel": max_pts,
                                                                "max_voxel_train": max_vox, "max_voxel_test": max_vox}}, train=False)
    out = pp.preprocess(pts)
    v, c, k = N.points_to_voxel(pts, vs, rng, max_pts, max_vox)
    assert out["voxel_coords"].shape == c.shape and out["voxel_coords"].dtype == np.int32
    np.testing.assert_array_equal(out["voxel_coords"], c)          # bit-exact voxel indexing, in order of first appearance
    np.testing.assert_array_equal(out["voxel_num_points"], k)
    np.testing.assert_array_equal(out["voxel_features"], v)       # same points in the same slots, zero padding
    if n >= 60000:
        assert len(c) > 1000 and int(k.max()) == max_pts
    assert pp.grid_size.tolist() == np.round((np.array(rng[3:]) - np.array(rng[:3])) / np.array(vs)).astype(np.int64).tolist()
