// Second translation unit of libgencomm_hip.so: iou3d_nms (reference extension semantics) and the point-cloud voxeliser.
// Kept apart from gencomm_abi.hip so that the rocPRIM templates do not lengthen the hot path's compile.
#include "../../include/gencomm_hip.h"

#include <algorithm>

#include "common.h"
#include "iou3d_kernels.h"
#include "voxel_kernels.h"

using namespace gc;

extern "C" {

int gencomm_iou3d_pairwise_fwd(const float* boxes_a, int num_a, const float* boxes_b, int num_b, int mode, float* out, void* stream) {
  GC_CHECK_ARG(num_a >= 0 && num_b >= 0 && (mode == 0 || mode == 1), "bad num_a / num_b / mode");
  if (num_a == 0 || num_b == 0) return GC_OK;
  GC_CHECK_ARG(boxes_a && boxes_b && out, "null pointer");
  const long long total = (long long)num_a * num_b;
  GC_CHECK_ARG((total + 255) / 256 < (1LL << 31), "num_a * num_b too large");
  iou3d_pairwise_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(boxes_a, num_a, boxes_b, num_b, mode, out);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_iou3d_max_boxes(void) { return kIou3dMaxBoxes; }
long long gencomm_iou3d_nms_workspace_bytes(int n) {
  if (n < 0 || n > kIou3dMaxBoxes) { fail(GC_ERR_ARG, "n out of range (gencomm_iou3d_max_boxes)"); return -1; }
  return (long long)align_up((size_t)std::max(n, 1) * ((std::max(n, 1) + 63) / 64) * sizeof(unsigned long long), 256);
}

int gencomm_iou3d_nms_fwd(const float* boxes, int n, float thresh, int normal, long long* keep, int* count,
                          void* workspace, long long workspace_bytes, void* stream) {
  GC_CHECK_ARG(n >= 0 && n <= kIou3dMaxBoxes, "n out of range (gencomm_iou3d_max_boxes)");
  GC_CHECK_ARG(count != nullptr, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    GC_HIP(hipMemsetAsync(count, 0, sizeof(int), st));
    return GC_OK;
  }
  GC_CHECK_ARG(boxes && keep && workspace, "null pointer");
  if (gencomm_iou3d_nms_workspace_bytes(n) > workspace_bytes) return fail(GC_ERR_WORKSPACE, "workspace too small (gencomm_iou3d_nms_workspace_bytes)");
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(workspace);
  const int cb = (n + 63) / 64;
  if (normal) iou3d_nms_mask_kernel<true><<<dim3(cb, cb), 64, 0, st>>>(n, thresh, boxes, mask);
  else iou3d_nms_mask_kernel<false><<<dim3(cb, cb), 64, 0, st>>>(n, thresh, boxes, mask);
  iou3d_nms_reduce_kernel<<<1, 64, 0, st>>>(n, mask, keep, count);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

static int voxel_args(VoxelArgs& a, const float* points, int n, int nfeat, const float* voxel_size3, const float* range6, int max_points, int max_voxels) {
  GC_CHECK_ARG(n >= 0 && nfeat >= 3 && max_points >= 1 && max_voxels >= 1 && voxel_size3 && range6, "bad n / nfeat / max_points / max_voxels");
  a.points = points; a.n = n; a.nfeat = nfeat; a.max_points = max_points; a.max_voxels = max_voxels;
  long long cells = 1;
  for (int j = 0; j < 3; ++j) {
    a.vs[j] = voxel_size3[j]; a.r0[j] = range6[j];
    GC_CHECK_ARG(voxel_size3[j] > 0.f, "voxel size must be positive");
    a.grid[j] = (int)lrintf((range6[3 + j] - range6[j]) / voxel_size3[j]);  // np.round((range[3:6] - range[0:3]) / voxel_size), sp_voxel_preprocessor.py:37-39
    GC_CHECK_ARG(a.grid[j] >= 1, "empty grid");
    cells *= a.grid[j];
  }
  GC_CHECK_ARG(cells < 0xFFFFFFFFLL, "grid has too many cells for 32-bit keys");
  return GC_OK;
}

long long gencomm_voxelize_workspace_bytes(int n) {
  if (n < 0) { fail(GC_ERR_ARG, "n must be non-negative"); return -1; }
  return (long long)voxel_ws(n).total;
}

int gencomm_voxelize_fwd(const float* points, int n, int nfeat, const float* voxel_size3, const float* range6, int max_points, int max_voxels,
                         float* voxels, int* coords_zyx, int* num_points, int* count, void* workspace, long long workspace_bytes, void* stream) {
  VoxelArgs a{};
  if (int rc = voxel_args(a, points, n, nfeat, voxel_size3, range6, max_points, max_voxels)) return rc;
  GC_CHECK_ARG(voxels && coords_zyx && num_points && count && workspace && (n == 0 || points), "null pointer");
  if ((long long)voxel_ws(n).total > workspace_bytes) return fail(GC_ERR_WORKSPACE, "workspace too small (gencomm_voxelize_workspace_bytes)");
  return voxelize_enqueue(a, voxels, coords_zyx, num_points, count, (char*)workspace, (hipStream_t)stream);
}

}  // extern "C"
