"""``SpVoxelPreprocessor`` -- the reference's point-cloud voxeliser
(``opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:18-85``) without spconv: same constructor arguments
(``preprocess_params`` with ``cav_lidar_range`` and ``args.{voxel_size, max_points_per_voxel, max_voxel_train, max_voxel_test}``,
``train``), same ``preprocess(pcd_np)`` result (``voxel_features [M, max_points, 4]``, ``voxel_coords [M, 3] (z, y, x)``,
``voxel_num_points [M]`` as numpy arrays) and ``grid_size``, on the HIP voxeliser (``csrc/voxel_kernels.h``).
``preprocess_device`` keeps the result on the GPU (no host copy) for callers that feed the PointPillars encoder directly.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .runtime import ptr, stream_ptr, workspaces


class SpVoxelPreprocessor:
    def __init__(self, preprocess_params: dict, train: bool, device="cuda:0"):
        self.params = preprocess_params
        self.train = train
        self.device = torch.device(device)
        self.lidar_range = self.params['cav_lidar_range']
        self.voxel_size = self.params['args']['voxel_size']
        self.max_points_per_voxel = self.params['args']['max_points_per_voxel']
        self.max_voxels = self.params['args']['max_voxel_train'] if train else self.params['args']['max_voxel_test']
        grid_size = (np.array(self.lidar_range[3:6]) - np.array(self.lidar_range[0:3])) / np.array(self.voxel_size)
        self.grid_size = np.round(grid_size).astype(np.int64)

    def preprocess_device(self, points: torch.Tensor):
        """points [n, F] float32 on the GPU -> (voxels [M, max_points, F], coords [M, 3] int32 (z, y, x), num_points [M] int32)."""
        if not points.is_cuda:
            raise _lib.GenCommHipError("SpVoxelPreprocessor.preprocess_device needs a device tensor (no CPU fallback)")
        pts = points.contiguous().float()
        n, F = pts.shape
        l = _lib.lib()
        dev = pts.device
        voxels = torch.empty(self.max_voxels, self.max_points_per_voxel, F, dtype=torch.float32, device=dev)
        coords = torch.empty(self.max_voxels, 3, dtype=torch.int32, device=dev)
        npts = torch.empty(self.max_voxels, dtype=torch.int32, device=dev)
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        ws = workspaces.get(dev, _lib.check_size(l.gencomm_voxelize_workspace_bytes(n), "gencomm_voxelize_workspace_bytes"), "voxelize")
        vs = (C.c_float * 3)(*[float(v) for v in self.voxel_size])
        rg = (C.c_float * 6)(*[float(v) for v in self.lidar_range])
        _lib.check(l.gencomm_voxelize_fwd(ptr(pts), n, F, vs, rg, int(self.max_points_per_voxel), int(self.max_voxels), ptr(voxels),
                                          ptr(coords), ptr(npts), ptr(count), ptr(ws), ws.numel(), stream_ptr(dev)), "gencomm_voxelize_fwd")
        m = int(count.item())
        return voxels[:m], coords[:m], npts[:m]

    def preprocess(self, pcd_np: np.ndarray) -> dict:
        v, c, k = self.preprocess_device(torch.from_numpy(np.ascontiguousarray(pcd_np, dtype=np.float32)).to(self.device))
        return {'voxel_features': v.cpu().numpy(), 'voxel_coords': c.cpu().numpy(), 'voxel_num_points': k.cpu().numpy()}
