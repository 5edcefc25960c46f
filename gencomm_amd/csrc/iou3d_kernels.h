// iou3d_nms on the device (gfx950, wave64) with the reference extension's own semantics -- OpenPCDet's rotated-box BEV
// overlap as vendored in opencood/pcdet_utils/iou3d_nms:
//   box_overlap / iou_bev            src/iou3d_nms_kernel.cu:104-243 (edge-pair intersections with strict crossing, corner
//                                    inclusion with MARGIN 1e-2, atan2 sort about the centroid, fan area) -- float32
//   boxes_overlap_kernel, boxes_iou_bev_kernel   :236-265    one (a, b) pair per thread
//   nms_kernel / nms_normal_kernel   :267-372    64 boxes per block, 64-bit suppression masks (one wave here)
//   greedy reduction of the masks    src/iou3d_nms.cpp:116-135 (host loop there; one wave on the device here, so the
//                                    call stays asynchronous and returns the keep list + count in device memory)
// Arithmetic follows the reference operation by operation in float32 with fused multiply-add contraction off, so that
// the same pairs cross the same thresholds as in the C oracle (oracle/csrc/detect_port.c).  Unlike the reference's
// CHECK_* macros nothing here can exit the process.
#pragma once
#include "common.h"

namespace gc {

#pragma clang fp contract(off)

struct Pt { float x, y; };
constexpr float kIouEps = 1e-8f;

__device__ __forceinline__ float pt_cross(Pt a, Pt b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ float pt_cross3(Pt p1, Pt p2, Pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }
__device__ __forceinline__ bool rect_cross(Pt p1, Pt p2, Pt q1, Pt q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}
__device__ __forceinline__ bool in_box2d(const float* box, Pt p) {
  const float MARGIN = 1e-2f;
  const float cx = box[0], cy = box[1];
  const float ac = cosf(-box[6]), as = sinf(-box[6]);
  const float rx = (p.x - cx) * ac + (p.y - cy) * (-as);
  const float ry = (p.x - cx) * as + (p.y - cy) * ac;
  return fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN;
}
__device__ __forceinline__ bool seg_intersection(Pt p1, Pt p0, Pt q1, Pt q0, Pt& ans) {
  if (!rect_cross(p0, p1, q0, q1)) return false;
  const float s1 = pt_cross3(q0, p1, p0), s2 = pt_cross3(p1, q1, p0), s3 = pt_cross3(p0, q1, q0), s4 = pt_cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  const float s5 = pt_cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > kIouEps) {
    ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans.x = (b0 * c1 - b1 * c0) / D;
    ans.y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}
__device__ __forceinline__ void rot_center(Pt c, float ac, float as, Pt& p) {
  const float nx = (p.x - c.x) * ac + (p.y - c.y) * (-as) + c.x;
  const float ny = (p.x - c.x) * as + (p.y - c.y) * ac + c.y;
  p.x = nx; p.y = ny;
}

__device__ float box_overlap_dev(const float* box_a, const float* box_b) {
  const float a_angle = box_a[6], b_angle = box_b[6];
  const float a_dx = box_a[3] / 2, b_dx = box_b[3] / 2, a_dy = box_a[4] / 2, b_dy = box_b[4] / 2;
  const float a_x1 = box_a[0] - a_dx, a_y1 = box_a[1] - a_dy, a_x2 = box_a[0] + a_dx, a_y2 = box_a[1] + a_dy;
  const float b_x1 = box_b[0] - b_dx, b_y1 = box_b[1] - b_dy, b_x2 = box_b[0] + b_dx, b_y2 = box_b[1] + b_dy;
  const Pt ca{box_a[0], box_a[1]}, cb{box_b[0], box_b[1]};
  Pt ac[5] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0.f, 0.f}};
  Pt bc[5] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0.f, 0.f}};
  const float a_cos = cosf(a_angle), a_sin = sinf(a_angle), b_cos = cosf(b_angle), b_sin = sinf(b_angle);
  for (int k = 0; k < 4; ++k) {
    rot_center(ca, a_cos, a_sin, ac[k]);
    rot_center(cb, b_cos, b_sin, bc[k]);
  }
  ac[4] = ac[0]; bc[4] = bc[0];
  Pt cp[16];
  Pt pc{0.f, 0.f};
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      Pt x;
      if (seg_intersection(ac[i + 1], ac[i], bc[j + 1], bc[j], x)) {
        cp[cnt] = x;
        pc.x = pc.x + x.x; pc.y = pc.y + x.y;
        ++cnt;
      }
    }
  for (int k = 0; k < 4; ++k) {
    if (in_box2d(box_a, bc[k])) { pc.x = pc.x + bc[k].x; pc.y = pc.y + bc[k].y; cp[cnt++] = bc[k]; }
    if (in_box2d(box_b, ac[k])) { pc.x = pc.x + ac[k].x; pc.y = pc.y + ac[k].y; cp[cnt++] = ac[k]; }
  }
  pc.x /= cnt; pc.y /= cnt;
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (atan2f(cp[i].y - pc.y, cp[i].x - pc.x) > atan2f(cp[i + 1].y - pc.y, cp[i + 1].x - pc.x)) {
        const Pt t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    const Pt u{cp[k].x - cp[0].x, cp[k].y - cp[0].y}, v{cp[k + 1].x - cp[0].x, cp[k + 1].y - cp[0].y};
    area += pt_cross(u, v);
  }
  return fabsf(area) / 2.0f;
}
__device__ __forceinline__ float iou_bev_dev(const float* a, const float* b) {
  const float sa = a[3] * a[4], sb = b[3] * b[4], s = box_overlap_dev(a, b);
  return s / fmaxf(sa + sb - s, kIouEps);
}
__device__ __forceinline__ float iou_normal_dev(const float* a, const float* b) {
  const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  const float interS = width * height, Sa = a[3] * a[4], Sb = b[3] * b[4];
  return interS / fmaxf(Sa + Sb - interS, kIouEps);
}

// mode 0: overlap area, 1: BEV IoU; one (a, b) pair per thread, b fastest (coalesced stores)
__global__ __launch_bounds__(256) void iou3d_pairwise_kernel(const float* __restrict__ boxes_a, int num_a, const float* __restrict__ boxes_b,
                                                             int num_b, int mode, float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)num_a * num_b) return;
  const int ia = (int)(t / num_b), ib = (int)(t - (long long)ia * num_b);
  float a[7], b[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) { a[k] = boxes_a[(size_t)ia * 7 + k]; b[k] = boxes_b[(size_t)ib * 7 + k]; }
  out[t] = mode ? iou_bev_dev(a, b) : box_overlap_dev(a, b);
}

// grid (col_blocks, row_blocks), 64 threads: thread = one row box against the 64 column boxes staged in LDS
template <bool NORMAL>
__global__ __launch_bounds__(64) void iou3d_nms_mask_kernel(int n, float thresh, const float* __restrict__ boxes, unsigned long long* __restrict__ mask) {
  const int row_start = blockIdx.y, col_start = blockIdx.x;
  const int row_size = min(n - row_start * 64, 64), col_size = min(n - col_start * 64, 64);
  __shared__ float blk[64 * 7];
  if ((int)threadIdx.x < col_size)
#pragma unroll
    for (int k = 0; k < 7; ++k) blk[threadIdx.x * 7 + k] = boxes[(size_t)(64 * col_start + threadIdx.x) * 7 + k];
  __syncthreads();
  if ((int)threadIdx.x < row_size) {
    const int cur = 64 * row_start + threadIdx.x;
    float me[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) me[k] = boxes[(size_t)cur * 7 + k];
    unsigned long long t = 0;
    const int start = row_start == col_start ? (int)threadIdx.x + 1 : 0;
    for (int i = start; i < col_size; ++i) {
      const float v = NORMAL ? iou_normal_dev(me, blk + i * 7) : iou_bev_dev(me, blk + i * 7);
      if (v > thresh) t |= 1ULL << i;
    }
    mask[(size_t)cur * ((n + 63) / 64) + col_start] = t;
  }
}

// iou3d_nms.cpp:116-135 on the device: one wave walks the boxes in order; `remv` (one bit per box) lives in LDS.
constexpr int kIou3dMaxBoxes = 32768;
__global__ __launch_bounds__(64) void iou3d_nms_reduce_kernel(int n, const unsigned long long* __restrict__ mask, long long* __restrict__ keep, int* __restrict__ count) {
  __shared__ unsigned long long remv[kIou3dMaxBoxes / 64];
  const int col_blocks = (n + 63) / 64, lane = threadIdx.x;
  for (int j = lane; j < col_blocks; j += 64) remv[j] = 0;
  __syncthreads();
  int num = 0;
  for (int i = 0; i < n; ++i) {
    const int nblock = i >> 6, inblock = i & 63;
    const bool gone = (remv[nblock] >> inblock) & 1ULL;  // uniform
    if (!gone) {
      if (lane == 0) keep[num] = i;
      ++num;
      for (int j = nblock + lane; j < col_blocks; j += 64) remv[j] |= mask[(size_t)i * col_blocks + j];
    }
    __syncthreads();
  }
  if (lane == 0) *count = num;
}

}  // namespace gc
