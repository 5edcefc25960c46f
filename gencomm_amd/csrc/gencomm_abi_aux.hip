// Second translation unit of libgencomm_hip.so: iou3d_nms (reference extension semantics) and the point-cloud voxeliser.
// Kept apart from gencomm_abi.hip so that the rocPRIM templates do not lengthen the hot path's compile.
#include "../../include/gencomm_hip.h"

#include <algorithm>

#include "common.h"
#include "iou3d_kernels.h"
#include "v2xvit_kernels.h"
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

int gencomm_warp_affine_fwd(const float* x, const double* theta, float* out, int n, int C, int H, int W, void* stream) {
  GC_CHECK_ARG(x && theta && out && n >= 1 && n <= 65535 && C >= 1 && H >= 1 && W >= 1, "bad arguments");
  WarpArgs a{x, theta, out, C, H, W};
  warp_affine_kernel<<<dim3((H * W + 255) / 256, n), 256, 0, (hipStream_t)stream>>>(a);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_hgt_attn_fwd(const float* qkv, const int* scene_off, float* out, int B, int heads, int dim_head, int HW, void* stream) {
  GC_CHECK_ARG(qkv && scene_off && out && B >= 1 && B <= 65535 && heads >= 1 && heads <= 65535 && HW >= 1, "bad arguments");
  HgtArgs a{qkv, scene_off, out, heads, HW, 1.0f / sqrtf((float)dim_head)};
  const dim3 grid((HW + 255) / 256, heads, B);
  hipStream_t st = (hipStream_t)stream;
  if (dim_head == 32) hgt_attn_kernel<32><<<grid, 256, 0, st>>>(a);
  else if (dim_head == 64) hgt_attn_kernel<64><<<grid, 256, 0, st>>>(a);
  else if (dim_head == 16) hgt_attn_kernel<16><<<grid, 256, 0, st>>>(a);
  else return fail(GC_ERR_ARG, "hgt attention: dim_head must be 16, 32 or 64");
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_win_attn_fwd(const float* qkv, const float* pos_embedding, float* out, int n, int heads, int dim_head, int window, int H, int W,
                         void* stream) {
  GC_CHECK_ARG(qkv && pos_embedding && out && n >= 1 && n <= 65535 && heads >= 1 && heads <= 65535, "bad arguments");
  GC_CHECK_ARG(H >= window && W >= window && H % window == 0 && W % window == 0, "H and W must be multiples of the window size");
  WinArgs a{qkv, pos_embedding, out, heads, H, W, 1.0f / sqrtf((float)dim_head)};
  hipStream_t st = (hipStream_t)stream;
  if (window == 4 && dim_head == 16) return win_attn_launch<16, 4>(a, n, st);
  if (window == 8 && dim_head == 32) return win_attn_launch<32, 8>(a, n, st);
  if (window == 16 && dim_head == 64) return win_attn_launch<64, 16>(a, n, st);
  if (window == 4 && dim_head == 32) return win_attn_launch<32, 4>(a, n, st);
  if (window == 8 && dim_head == 16) return win_attn_launch<16, 8>(a, n, st);
  if (window == 8 && dim_head == 64) return win_attn_launch<64, 8>(a, n, st);
  if (window == 16 && dim_head == 32) return win_attn_launch<32, 16>(a, n, st);
  return fail(GC_ERR_ARG, "window attention: supported (window, dim_head) pairs are (4,16) (4,32) (8,16) (8,32) (8,64) (16,32) (16,64)");
}

}  // extern "C"
