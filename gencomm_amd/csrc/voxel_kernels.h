// Point cloud -> voxels on the device, with the semantics of spconv's CPU point-to-voxel that the reference's dataloader
// calls (opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:25-29, :54-68): voxels in order
// of their FIRST point, points inside a voxel in input order, at most `max_points` per voxel and `max_voxels` voxels
// (points of later cells are dropped), coordinates (z, y, x), cell = floor((p - range_min) / voxel_size) in float32.
// The sequential definition is reproduced in parallel and deterministically:
//   1 key[i] = linear cell of point i (0xFFFFFFFF outside the grid), value[i] = i
//   2 STABLE radix sort by key (rocPRIM): equal cells become segments whose values ascend = input order
//   3 segment heads -> flag[first point index of the cell] = 1; exclusive scan of flag over the points = voxel number in
//     order of first appearance; inclusive max-scan of head positions = every element's segment start (its rank)
//   4 scatter: voxels[vid][rank] = points[value], coords / num_points at the heads
// Oracle: oracle/csrc/detect_port.c gc_oracle_points_to_voxel (PARITY UNPINNED: spconv is not under /root/reference).
#pragma once
#include <rocprim/rocprim.hpp>

#include "common.h"

namespace gc {

struct VoxelArgs {
  const float* points;  // [n][nfeat]
  int n, nfeat;
  float vs[3], r0[3];
  int grid[3];          // x, y, z cells
  int max_points, max_voxels;
};

__global__ __launch_bounds__(256) void voxel_key_kernel(const VoxelArgs a, unsigned int* __restrict__ key, unsigned int* __restrict__ val) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  unsigned int k = 0xFFFFFFFFu;
  int c[3];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float f = floorf((a.points[(size_t)i * a.nfeat + j] - a.r0[j]) / a.vs[j]);
    ok = ok && f >= 0.f && f < (float)a.grid[j];
    c[j] = (int)f;
  }
  if (ok) k = (unsigned int)((c[2] * a.grid[1] + c[1]) * a.grid[0] + c[0]);
  key[i] = k;
  val[i] = (unsigned int)i;
}

// on the sorted arrays: head position (for the max-scan) and the first-appearance flag in point-index space
__global__ __launch_bounds__(256) void voxel_head_kernel(int n, const unsigned int* __restrict__ skey, const unsigned int* __restrict__ sval,
                                                         int* __restrict__ headpos, int* __restrict__ flag) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const bool valid = skey[j] != 0xFFFFFFFFu;
  const bool head = valid && (j == 0 || skey[j - 1] != skey[j]);
  headpos[j] = head ? j : 0;
  if (head) flag[sval[j]] = 1;
}

__global__ __launch_bounds__(256) void voxel_scatter_kernel(const VoxelArgs a, const unsigned int* __restrict__ skey, const unsigned int* __restrict__ sval,
                                                            const int* __restrict__ segstart, const int* __restrict__ vid_of_point,
                                                            float* __restrict__ voxels, int* __restrict__ coords, int* __restrict__ num_points) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= a.n) return;
  const unsigned int k = skey[j];
  if (k == 0xFFFFFFFFu) return;
  const int hp = segstart[j], rank = j - hp;
  const int vid = vid_of_point[sval[hp]];
  if (vid >= a.max_voxels) return;
  if (rank == 0) {
    const int x = (int)(k % (unsigned)a.grid[0]), y = (int)((k / (unsigned)a.grid[0]) % (unsigned)a.grid[1]), z = (int)(k / ((unsigned)a.grid[0] * a.grid[1]));
    coords[vid * 3 + 0] = z; coords[vid * 3 + 1] = y; coords[vid * 3 + 2] = x;
  }
  if (rank < a.max_points) {
    const float* __restrict__ src = a.points + (size_t)sval[j] * a.nfeat;
    float* __restrict__ dst = voxels + ((size_t)vid * a.max_points + rank) * a.nfeat;
    for (int f = 0; f < a.nfeat; ++f) dst[f] = src[f];
    atomicMax(&num_points[vid], rank + 1);
  }
}

// number of voxels = min(number of first-appearance flags, max_voxels)
__global__ void voxel_count_kernel(int n, const int* __restrict__ vid_of_point, const int* __restrict__ flag, int max_voxels, int* __restrict__ count) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int total = n > 0 ? vid_of_point[n - 1] + flag[n - 1] : 0;
    *count = total < max_voxels ? total : max_voxels;
  }
}

struct VoxelWs {
  size_t key, val, skey, sval, headpos, segstart, flag, vid, temp, temp_bytes, total;
};
struct MaxOp {
  __device__ __host__ int operator()(int a, int b) const { return a > b ? a : b; }
};
inline VoxelWs voxel_ws(int n) {
  VoxelWs w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
  const size_t nn = (size_t)(n > 0 ? n : 1);
  w.key = take(nn * 4); w.val = take(nn * 4); w.skey = take(nn * 4); w.sval = take(nn * 4);
  w.headpos = take(nn * 4); w.segstart = take(nn * 4); w.flag = take(nn * 4); w.vid = take(nn * 4);
  size_t t1 = 0, t2 = 0, t3 = 0;
  (void)rocprim::radix_sort_pairs(nullptr, t1, (unsigned int*)nullptr, (unsigned int*)nullptr, (unsigned int*)nullptr, (unsigned int*)nullptr, nn, 0, 32, (hipStream_t)0);
  (void)rocprim::exclusive_scan(nullptr, t2, (int*)nullptr, (int*)nullptr, 0, nn, rocprim::plus<int>(), (hipStream_t)0);
  (void)rocprim::inclusive_scan(nullptr, t3, (int*)nullptr, (int*)nullptr, nn, MaxOp(), (hipStream_t)0);
  w.temp_bytes = std::max(t1, std::max(t2, t3)) + 256;
  w.temp = take(w.temp_bytes);
  w.total = off;
  return w;
}

inline int voxelize_enqueue(const VoxelArgs& a, float* voxels, int* coords, int* num_points, int* count, char* wsp, hipStream_t st) {
  const VoxelWs w = voxel_ws(a.n);
  auto U = [&](size_t o) { return reinterpret_cast<unsigned int*>(wsp + o); };
  auto I = [&](size_t o) { return reinterpret_cast<int*>(wsp + o); };
  GC_HIP(hipMemsetAsync(voxels, 0, (size_t)a.max_voxels * a.max_points * a.nfeat * sizeof(float), st));
  GC_HIP(hipMemsetAsync(num_points, 0, (size_t)a.max_voxels * sizeof(int), st));
  GC_HIP(hipMemsetAsync(coords, 0, (size_t)a.max_voxels * 3 * sizeof(int), st));
  GC_HIP(hipMemsetAsync(count, 0, sizeof(int), st));
  if (a.n == 0) return GC_OK;
  GC_HIP(hipMemsetAsync(I(w.flag), 0, (size_t)a.n * 4, st));
  const int nb = (a.n + 255) / 256;
  voxel_key_kernel<<<nb, 256, 0, st>>>(a, U(w.key), U(w.val));
  size_t tb = w.temp_bytes;
  GC_HIP(rocprim::radix_sort_pairs(wsp + w.temp, tb, U(w.key), U(w.skey), U(w.val), U(w.sval), (size_t)a.n, 0, 32, st));
  voxel_head_kernel<<<nb, 256, 0, st>>>(a.n, U(w.skey), U(w.sval), I(w.headpos), I(w.flag));
  tb = w.temp_bytes;
  GC_HIP(rocprim::inclusive_scan(wsp + w.temp, tb, I(w.headpos), I(w.segstart), (size_t)a.n, MaxOp(), st));
  tb = w.temp_bytes;
  GC_HIP(rocprim::exclusive_scan(wsp + w.temp, tb, I(w.flag), I(w.vid), 0, (size_t)a.n, rocprim::plus<int>(), st));
  voxel_scatter_kernel<<<nb, 256, 0, st>>>(a, U(w.skey), U(w.sval), I(w.segstart), I(w.vid), voxels, coords, num_points);
  voxel_count_kernel<<<1, 64, 0, st>>>(a.n, I(w.vid), I(w.flag), a.max_voxels, count);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

}  // namespace gc
