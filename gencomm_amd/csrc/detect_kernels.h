// Detection tail on the device (gfx950, wave64): anchor decoding + score filter + ordered compaction,
// rotated NMS (score sort, pairwise convex-quad IoU in fp64, greedy scan, range mask) and the axis-aligned
// overlap matrix of target assignment.  SURVEY.md 8f rank 3.
//
// Reference (all CPU/torch/numpy/shapely there):
//   VoxelPostprocessor.post_process     opencood/data_utils/post_processor/voxel_postprocessor.py:1084-1244
//   delta_to_boxes3d                    .../voxel_postprocessor.py:1351-1396
//   limit_period, rotate_points_along_z opencood/utils/common_utils.py:104-113, :139-161
//   boxes_to_corners_3d, project_box3d  opencood/utils/box_utils.py:152-204, :278-316
//   remove_large_pred_bbx, remove_bbx_abnormal_z, mask_boxes_outside_range_numpy   box_utils.py:1062-1112, :384-421
//   nms_rotated (+ shapely IoU)         box_utils.py:915-960, common_utils.py:230-270
//   bbox_overlaps                       opencood/utils/box_overlaps.pyx:17-57
//
// Arithmetic mirrors the reference op by op (fp32 for the boxes, fp64 for polygon areas) with fused
// multiply-add contraction switched off, so that the same candidates cross the same thresholds.
#pragma once
#include "common.h"

namespace gc {

#pragma clang fp contract(off)

struct DetArgs {
  const float* cls;      // [A][H][W]      (batch 1)
  const float* reg;      // [7A][H][W]
  const float* dir;      // [A*nb][H][W] or null
  const float* anchors;  // [H][W][A][7]
  const float* T;        // [4][4] row-major, agent -> ego
  float* score_tmp;      // [H*W*A] scratch: sigmoid score or -1
  int* block_count;      // [nblocks] scratch
  int* block_off;        // [nblocks] scratch
  float* corners;        // [cap][8][3] (appended at *count)
  float* scores;         // [cap]
  int* anchor_idx;       // [cap]
  int* count;            // device counter: candidates so far (in/out)
  int H, W, A, nb, cap, hwl;
  float thr, dir_offset;
};

__device__ __forceinline__ float limit_period_f(float val, float offset, float period) {
  return val - floorf(val / period + offset) * period;
}

// box7 of anchor i after direction fix -> 8 projected corners; returns false when a size / z filter rejects it
__device__ __forceinline__ bool det_decode_one(const DetArgs& a, int i, float (&c)[8][3]) {
  const int HW = a.H * a.W;
  const int an = i % a.A, pix = i / a.A;
  const float* __restrict__ ab = a.anchors + (size_t)i * 7;
  float d[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) d[k] = a.reg[(size_t)(an * 7 + k) * HW + pix];
  const float diag = sqrtf(ab[4] * ab[4] + ab[5] * ab[5]);
  float b[7];
  b[0] = d[0] * diag + ab[0];
  b[1] = d[1] * diag + ab[1];
  b[2] = d[2] * ab[3] + ab[2];
  b[3] = expf(d[3]) * ab[3];
  b[4] = expf(d[4]) * ab[4];
  b[5] = expf(d[5]) * ab[5];
  b[6] = d[6] + ab[6];
  if (a.dir != nullptr) {
    int label = 0;
    float best = a.dir[(size_t)(an * a.nb) * HW + pix];
    for (int k = 1; k < a.nb; ++k) {
      const float v = a.dir[(size_t)(an * a.nb + k) * HW + pix];
      if (v > best) { best = v; label = k; }  // first maximum wins, like torch.max
    }
    const float period = (float)(2.0 * 3.14159265358979323846 / (double)a.nb);
    const float rot = limit_period_f(b[6] - a.dir_offset, 0.f, period);
    b[6] = rot + a.dir_offset + period * (float)label;
    b[6] = limit_period_f(b[6], 0.5f, (float)(2.0 * 3.14159265358979323846));
  }
  // boxes_to_corners_3d: (l, w, h) extents; 'hwl' stores them as [h, w, l]
  const float ex = a.hwl ? b[5] : b[3], ey = b[4], ez = a.hwl ? b[3] : b[5];
  const float cosa = cosf(b[6]), sina = sinf(b[6]);
  const float sx[8] = {1, 1, -1, -1, 1, 1, -1, -1}, sy[8] = {-1, 1, 1, -1, -1, 1, 1, -1}, sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
  float xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float px = ex * (sx[k] / 2.f), py = ey * (sy[k] / 2.f), pz = ez * (sz[k] / 2.f);
    // row vector times [[cos, sin, 0], [-sin, cos, 0], [0, 0, 1]]
    const float rx = px * cosa + py * (-sina) + pz * 0.f + b[0];
    const float ry = px * sina + py * cosa + pz * 0.f + b[1];
    const float rz = px * 0.f + py * 0.f + pz * 1.f + b[2];
    const float X = a.T[0] * rx + a.T[1] * ry + a.T[2] * rz + a.T[3];
    const float Y = a.T[4] * rx + a.T[5] * ry + a.T[6] * rz + a.T[7];
    const float Z = a.T[8] * rx + a.T[9] * ry + a.T[10] * rz + a.T[11];
    c[k][0] = X; c[k][1] = Y; c[k][2] = Z;
    xmin = fminf(xmin, X); xmax = fmaxf(xmax, X);
    ymin = fminf(ymin, Y); ymax = fmaxf(ymax, Y);
    zmin = fminf(zmin, Z); zmax = fmaxf(zmax, Z);
  }
  const float x_len = xmax - xmin, y_len = ymax - ymin;
  // remove_large_pred_bbx: its "z extent" is the y extent again and is used as a truth value (non-zero)
  const bool size_ok = x_len <= 6.f && y_len <= 6.f && y_len != 0.f;
  const bool z_ok = zmin >= -3.f && zmax <= 1.f;
  return size_ok && z_ok;
}

__device__ __forceinline__ float det_score(const DetArgs& a, int i) {
  const int HW = a.H * a.W;
  const int an = i % a.A, pix = i / a.A;
  const float x = a.cls[(size_t)an * HW + pix];
  return 1.0f / (1.0f + expf(-x));
}

// pass 1: score + all filters -> score_tmp (or -1) and the number of survivors per workgroup
__global__ __launch_bounds__(256) void det_flag_kernel(const DetArgs a) {
  __shared__ int s_cnt[4];
  const int n = a.H * a.W * a.A;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float s = -1.f;
  if (i < n) {
    const float p = det_score(a, i);
    if (p > a.thr) {
      float c[8][3];
      if (det_decode_one(a, i, c)) s = p;
    }
    a.score_tmp[i] = s;
  }
  const unsigned long long m = __ballot(s >= 0.f);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) a.block_count[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// pass 2: exclusive scan of the workgroup counts (single workgroup), starting at *count; *count += total
__global__ __launch_bounds__(1024) void det_scan_kernel(const int* __restrict__ block_count, int* __restrict__ block_off,
                                                       int nblocks, int* __restrict__ count) {
  __shared__ int s_part[1024];
  __shared__ int s_base;
  if (threadIdx.x == 0) s_base = *count;
  __syncthreads();
  for (int b0 = 0; b0 < nblocks; b0 += 1024) {
    const int i = b0 + threadIdx.x;
    const int v = i < nblocks ? block_count[i] : 0;
    s_part[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan
      const int t = threadIdx.x >= o ? s_part[threadIdx.x - o] : 0;
      __syncthreads();
      s_part[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblocks) block_off[i] = s_base + s_part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 0) s_base += s_part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *count = s_base;
}

// pass 3: survivors are written in anchor order (the order torch.masked_select produces)
__global__ __launch_bounds__(256) void det_emit_kernel(const DetArgs a) {
  __shared__ int s_cnt[4];
  const int n = a.H * a.W * a.A;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const float s = i < n ? a.score_tmp[i] : -1.f;
  const bool keep = s >= 0.f;
  const unsigned long long m = __ballot(keep);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) s_cnt[wv] = __popcll(m);
  __syncthreads();
  int base = a.block_off[blockIdx.x];
  for (int w = 0; w < wv; ++w) base += s_cnt[w];
  if (!keep) return;
  const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
  if (slot >= a.cap) return;  // the host compares *count with cap and fails loudly
  float c[8][3];
  det_decode_one(a, i, c);
  float* __restrict__ dst = a.corners + (size_t)slot * 24;
#pragma unroll
  for (int k = 0; k < 8; ++k) { dst[3 * k] = c[k][0]; dst[3 * k + 1] = c[k][1]; dst[3 * k + 2] = c[k][2]; }
  a.scores[slot] = s;
  a.anchor_idx[slot] = i;
}

// ---------------------------------------------------------------------------------------------
// Rotated NMS
// ---------------------------------------------------------------------------------------------
constexpr int kNmsMaxN = 16384;  // candidates that fit the single-workgroup sort (128 KB of LDS)
constexpr int kNmsMaxTop = 1024;

// order[r] = index of the r-th candidate by (score descending, index descending) -- a stable ascending argsort read
// backwards -- for r < min(n, top).  One workgroup, bitonic sort in LDS.
__global__ __launch_bounds__(1024) void nms_sort_kernel(const float* __restrict__ scores, const int* __restrict__ n_dev,
                                                       int top, int* __restrict__ order, int* __restrict__ k_dev) {
  extern __shared__ unsigned long long s_key[];
  const int n = min(*n_dev, kNmsMaxN);
  int P = 1;
  while (P < n) P <<= 1;
  // key: larger = earlier.  score > 0 so its bit pattern is monotone; ties -> larger index first
  for (int i = threadIdx.x; i < P; i += 1024)
    s_key[i] = i < n ? ((unsigned long long)__float_as_uint(scores[i]) << 32) | (unsigned)i : 0ull;
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P; i += 1024) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long x = s_key[i], y = s_key[l];
          const bool desc = (i & k) == 0;  // descending overall
          if (desc ? x < y : x > y) { s_key[i] = y; s_key[l] = x; }
        }
      }
      __syncthreads();
    }
  }
  const int K = min(n, top);
  for (int i = threadIdx.x; i < K; i += 1024) order[i] = (int)(s_key[i] & 0xffffffffull);
  if (threadIdx.x == 0) *k_dev = K;
}

struct Pt { double x, y; };

__device__ __forceinline__ double poly_area(const Pt* p, int n) {
  if (n < 3) return 0.0;
  double s = 0.0;
  for (int i = 0; i < n; ++i) {
    const int j = i + 1 == n ? 0 : i + 1;
    s += p[i].x * p[j].y - p[j].x * p[i].y;
  }
  return 0.5 * s;
}

// Sutherland-Hodgman: part of `in` on the left of (or on) a->b
__device__ __forceinline__ int poly_clip(const Pt* in, int n, Pt a, Pt b, Pt* out) {
  if (n == 0) return 0;
  const double ex = b.x - a.x, ey = b.y - a.y;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    const int j = i + 1 == n ? 0 : i + 1;
    const double si = ex * (in[i].y - a.y) - ey * (in[i].x - a.x);
    const double sj = ex * (in[j].y - a.y) - ey * (in[j].x - a.x);
    if (si >= 0) out[m++] = in[i];
    if ((si > 0 && sj < 0) || (si < 0 && sj > 0)) {
      const double t = si / (si - sj);
      out[m].x = in[i].x + t * (in[j].x - in[i].x);
      out[m].y = in[i].y + t * (in[j].y - in[i].y);
      ++m;
    }
  }
  return m;
}

__device__ __forceinline__ double quad_iou_d(const float* __restrict__ ca, const float* __restrict__ cb) {
  Pt p[4], q[4];
  for (int k = 0; k < 4; ++k) { p[k] = {(double)ca[3 * k], (double)ca[3 * k + 1]}; q[k] = {(double)cb[3 * k], (double)cb[3 * k + 1]}; }
  double ap = poly_area(p, 4), aq = poly_area(q, 4);
  if (ap < 0) { Pt t = p[0]; p[0] = p[3]; p[3] = t; t = p[1]; p[1] = p[2]; p[2] = t; ap = -ap; }
  if (aq < 0) { Pt t = q[0]; q[0] = q[3]; q[3] = t; t = q[1]; q[1] = q[2]; q[2] = t; aq = -aq; }
  Pt buf0[12], buf1[12];
  int n = 4;
  for (int k = 0; k < 4; ++k) buf0[k] = p[k];
  Pt* cur = buf0; Pt* nxt = buf1;
  for (int e = 0; e < 4 && n > 0; ++e) {
    n = poly_clip(cur, n, q[e], q[(e + 1) & 3], nxt);
    Pt* t = cur; cur = nxt; nxt = t;
  }
  const double inter = fabs(poly_area(cur, n));
  const double uni = ap + aq - inter;
  return uni > 0 ? inter / uni : 0.0;
}

// mask[i][w] bit b: candidate order[64 w + b] (ranked after i) overlaps candidate order[i] above the threshold
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ corners, const int* __restrict__ order,
                                                     const int* __restrict__ k_dev, float thr,
                                                     unsigned long long* __restrict__ mask /*[kNmsMaxTop][kNmsMaxTop/64]*/) {
  const int K = *k_dev;
  const int i = blockIdx.x * 64 + threadIdx.x, w = blockIdx.y;
  if (i >= K) return;
  unsigned long long bits = 0ull;
  if (64 * w + 63 > i) {
    const float* __restrict__ ci = corners + (size_t)order[i] * 24;
    for (int b = 0; b < 64; ++b) {
      const int j = 64 * w + b;
      if (j > i && j < K) {
        const float iou = (float)quad_iou_d(ci, corners + (size_t)order[j] * 24);
        if (iou > thr) bits |= 1ull << b;
      }
    }
  }
  mask[(size_t)i * (kNmsMaxTop / 64) + w] = bits;
}

// greedy scan in rank order + range mask + ordered gather (single workgroup of 1024)
struct NmsOutArgs {
  const float* corners; const float* scores; const int* order; const int* k_dev;
  const unsigned long long* mask;
  const float* range6;  // may be null
  float* out_boxes; float* out_scores; int* out_index; int* out_count;
};
__global__ __launch_bounds__(1024) void nms_greedy_kernel(const NmsOutArgs a) {
  extern __shared__ unsigned long long s_mask[];  // [K][kNmsMaxTop / 64]: the serial scan reads LDS, not HBM
  __shared__ unsigned char s_keep[kNmsMaxTop];
  __shared__ int s_scan[1024];
  const int K = *a.k_dev;
  const int tid = threadIdx.x;
  for (int i = tid; i < K * (kNmsMaxTop / 64); i += 1024) s_mask[i] = a.mask[i];
  __syncthreads();
  if (tid < 64) {  // wave 0: lane l < 16 owns word l of the suppressed set
    unsigned long long removed = 0ull;
    for (int i = 0; i < K; ++i) {
      const unsigned long long wsel = __shfl(removed, i >> 6, 64);
      const bool kept = !((wsel >> (i & 63)) & 1ull);
      if (kept && tid < kNmsMaxTop / 64) removed |= s_mask[i * (kNmsMaxTop / 64) + tid];
      if (tid == 0) s_keep[i] = kept ? 1 : 0;
    }
  }
  __syncthreads();
  // range mask (all 8 corners inside, bounds inclusive), then ordered compaction
  bool out = false;
  int src = 0;
  if (tid < K && s_keep[tid]) {
    src = a.order[tid];
    out = true;
    if (a.range6 != nullptr) {
      const float* __restrict__ c = a.corners + (size_t)src * 24;
      for (int k = 0; k < 8; ++k)
        for (int d = 0; d < 3; ++d) out = out && c[3 * k + d] >= a.range6[d] && c[3 * k + d] <= a.range6[3 + d];
    }
  }
  s_scan[tid] = out ? 1 : 0;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int t = tid >= o ? s_scan[tid - o] : 0;
    __syncthreads();
    s_scan[tid] += t;
    __syncthreads();
  }
  if (out) {
    const int slot = s_scan[tid] - 1;
    const float* __restrict__ c = a.corners + (size_t)src * 24;
    for (int k = 0; k < 24; ++k) a.out_boxes[(size_t)slot * 24 + k] = c[k];
    a.out_scores[slot] = a.scores[src];
    a.out_index[slot] = src;
  }
  if (tid == 1023) *a.out_count = s_scan[1023];
}

// ---------------------------------------------------------------------------------------------
// bbox_overlaps (box_overlaps.pyx:17-57): precision follows the C Cython emits -- float differences, `+ 1.0` and the
// area products in double, float variables for box_area / iw / ih / ua, float product and division at the end.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bbox_overlaps_kernel(const float* __restrict__ boxes, const float* __restrict__ query,
                                                           float* __restrict__ out, int N, int K) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)N * K) return;
  const int n = (int)(t / K), k = (int)(t - (long long)n * K);
  const float* b = boxes + (size_t)n * 4;
  const float* q = query + (size_t)k * 4;
  const float box_area = (float)(((double)(q[2] - q[0]) + 1.0) * ((double)(q[3] - q[1]) + 1.0));
  float r = 0.f;
  const float iw = (float)((double)(fminf(b[2], q[2]) - fmaxf(b[0], q[0])) + 1.0);
  if (iw > 0) {
    const float ih = (float)((double)(fminf(b[3], q[3]) - fmaxf(b[1], q[1])) + 1.0);
    if (ih > 0) {
      const float ua = (float)(((double)(b[2] - b[0]) + 1.0) * ((double)(b[3] - b[1]) + 1.0) + (double)box_area - (double)(iw * ih));
      r = iw * ih / ua;
    }
  }
  out[t] = r;
}

#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
inline long long det_workspace_bytes(int H, int W, int A) {
  const long long n = (long long)H * W * A, nb = (n + 255) / 256;
  return align_up(n * 4, 256) + 2 * align_up(nb * 4, 256);
}
inline long long nms_workspace_bytes() {
  return align_up(kNmsMaxTop * 4, 256) + 256 + (long long)kNmsMaxTop * (kNmsMaxTop / 64) * 8;
}

inline int det_decode_enqueue(DetArgs a, void* workspace, hipStream_t st) {
  const long long n = (long long)a.H * a.W * a.A;
  const int nblocks = (int)((n + 255) / 256);
  char* ws = (char*)workspace;
  a.score_tmp = (float*)ws;
  a.block_count = (int*)(ws + align_up(n * 4, 256));
  a.block_off = (int*)(ws + align_up(n * 4, 256) + align_up((long long)nblocks * 4, 256));
  det_flag_kernel<<<nblocks, 256, 0, st>>>(a);
  det_scan_kernel<<<1, 1024, 0, st>>>(a.block_count, a.block_off, nblocks, a.count);
  det_emit_kernel<<<nblocks, 256, 0, st>>>(a);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

inline int nms_rotated_enqueue(const float* corners, const float* scores, const int* n_dev, float thr, int top, const float* range6,
                               float* out_boxes, float* out_scores, int* out_index, int* out_count, void* workspace, hipStream_t st) {
  char* ws = (char*)workspace;
  int* order = (int*)ws;
  int* k_dev = (int*)(ws + align_up(kNmsMaxTop * 4, 256));
  unsigned long long* mask = (unsigned long long*)(ws + align_up(kNmsMaxTop * 4, 256) + 256);
  static LdsAttrOnce attr_sort, attr_greedy;
  if (const int rc = attr_sort.set(nms_sort_kernel, kNmsMaxN * 8)) return rc;
  if (const int rc = attr_greedy.set(nms_greedy_kernel, kNmsMaxTop * (kNmsMaxTop / 64) * 8)) return rc;
  nms_sort_kernel<<<1, 1024, kNmsMaxN * 8, st>>>(scores, n_dev, top, order, k_dev);
  nms_mask_kernel<<<dim3(kNmsMaxTop / 64, kNmsMaxTop / 64), 64, 0, st>>>(corners, order, k_dev, thr, mask);
  NmsOutArgs o{corners, scores, order, k_dev, mask, range6, out_boxes, out_scores, out_index, out_count};
  nms_greedy_kernel<<<1, 1024, kNmsMaxTop * (kNmsMaxTop / 64) * 8, st>>>(o);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

}  // namespace gc
