// Warp-to-ego + per-pixel cross-agent attention (gfx950, wave64, fp32; grid maths in fp64).
//
// Reference: AttFusion.forward (opencood/models/fuse_modules/fusion_in_one.py:131-151) =
//   warp_affine_simple (opencood/models/sub_modules/torch_transformation_utils.py:323-332:
//     F.affine_grid(M float64, align_corners=False).to(float32) -> F.grid_sample(bilinear, zeros,
//     align_corners=False))
//   -> x[HW, N, C]; softmax(x x^T / sqrt(C)) x; keep agent 0  (ScaledDotProductAttention :41-45).
// Only the ego row of the N x N attention is kept, so per pixel:
//   w_j = softmax_j(<x_0, x_j> / sqrt(C)),   out = sum_j w_j x_j        (SURVEY.md 8a).
// One lane per output pixel; the bilinear taps of every agent are computed once and reused for
// both channel passes (scores, then weighted sum).  Pure gather, HBM/L2-bound.
#pragma once
#include "common.h"

namespace gc {

struct FuseArgs {
  const float* x;        // [n][C][H][W]
  const double* theta;   // [n][2][3]
  const int* scene_off;  // [B+1]
  float* out;            // [B][C][H][W]
  int C, H, W;
};

template <int N>
__device__ __forceinline__ void fuse_body(const FuseArgs& a, int b, int off, int pix) {
  const int H = a.H, W = a.W, HW = H * W;
  const int h = pix / W, w = pix - h * W;
  // affine_grid base coordinates, align_corners=False: (2i+1)/size - 1, in float64 like theta
  const double xb = (2.0 * w + 1.0) / (double)W - 1.0;
  const double yb = (2.0 * h + 1.0) / (double)H - 1.0;
  int idx[N][4];
  float wt[N][4];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double* __restrict__ th = a.theta + (size_t)(off + j) * 6;
    const float gx = (float)(th[0] * xb + th[1] * yb + th[2]);
    const float gy = (float)(th[3] * xb + th[4] * yb + th[5]);
    // grid_sampler unnormalize (align_corners=False): ((g + 1) * size - 1) / 2
    const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
    const float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    // keep the integer conversion in range for far-away agents
    const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)W + 1.f);
    const int y0 = (int)fminf(fmaxf(fy, -2.f), (float)H + 1.f);
    const float tx = ix - fx, ty = iy - fy;  // == ix - ix_nw etc.
    const float wnw = (1.f - tx) * (1.f - ty), wne = tx * (1.f - ty), wsw = (1.f - tx) * ty, wse = tx * ty;
    const bool xl = x0 >= 0 && x0 < W, xr = x0 + 1 >= 0 && x0 + 1 < W;
    const bool yt = y0 >= 0 && y0 < H, yb_ = y0 + 1 >= 0 && y0 + 1 < H;
    const bool far = fx != (float)x0 || fy != (float)y0;  // clamped => everything out of range
    idx[j][0] = (xl && yt && !far) ? y0 * W + x0 : -1;
    idx[j][1] = (xr && yt && !far) ? y0 * W + x0 + 1 : -1;
    idx[j][2] = (xl && yb_ && !far) ? (y0 + 1) * W + x0 : -1;
    idx[j][3] = (xr && yb_ && !far) ? (y0 + 1) * W + x0 + 1 : -1;
    wt[j][0] = wnw; wt[j][1] = wne; wt[j][2] = wsw; wt[j][3] = wse;
  }
  auto sample = [&](int j, const float* __restrict__ plane) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) v = fmaf(idx[j][k] >= 0 ? plane[idx[j][k]] : 0.f, wt[j][k], v);
    return v;
  };
  const float* __restrict__ xs = a.x + (size_t)off * a.C * HW;
  float score[N];
#pragma unroll
  for (int j = 0; j < N; ++j) score[j] = 0.f;
  for (int c = 0; c < a.C; ++c) {
    const float v0 = sample(0, xs + (size_t)c * HW);
    score[0] = fmaf(v0, v0, score[0]);
#pragma unroll
    for (int j = 1; j < N; ++j) score[j] = fmaf(v0, sample(j, xs + ((size_t)j * a.C + c) * HW), score[j]);
  }
  const float inv = 1.0f / sqrtf((float)a.C);
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < N; ++j) { score[j] *= inv; mx = fmaxf(mx, score[j]); }
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < N; ++j) { score[j] = expf(score[j] - mx); den += score[j]; }
  const float rden = 1.0f / den;
#pragma unroll
  for (int j = 0; j < N; ++j) score[j] *= rden;
  float* __restrict__ op = a.out + (size_t)b * a.C * HW + pix;
  for (int c = 0; c < a.C; ++c) {
    float o = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) o = fmaf(score[j], sample(j, xs + ((size_t)j * a.C + c) * HW), o);
    op[(size_t)c * HW] = o;
  }
}

__global__ __launch_bounds__(256) void warp_attfuse_kernel(const FuseArgs a) {
  const int b = blockIdx.y;
  const int off = a.scene_off[b], N = a.scene_off[b + 1] - off;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= a.H * a.W) return;
  switch (N) {
    case 1: fuse_body<1>(a, b, off, pix); break;
    case 2: fuse_body<2>(a, b, off, pix); break;
    case 3: fuse_body<3>(a, b, off, pix); break;
    case 4: fuse_body<4>(a, b, off, pix); break;
    case 5: fuse_body<5>(a, b, off, pix); break;
    case 6: fuse_body<6>(a, b, off, pix); break;
    case 7: fuse_body<7>(a, b, off, pix); break;
    case 8: fuse_body<8>(a, b, off, pix); break;
    default: break;  // host validates 1 <= N <= 8 where it can; otherwise the scene is left untouched
  }
}

inline int warp_attfuse_enqueue(const float* x, const double* theta, const int* scene_off, float* out,
                                int B, int n, int C, int H, int W, hipStream_t st) {
  (void)n;
  FuseArgs a{x, theta, scene_off, out, C, H, W};
  TimedLaunch tl(KF_WARP_ATTFUSE, st);
  warp_attfuse_kernel<<<dim3((H * W + 255) / 256, B), 256, 0, st>>>(a);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

}  // namespace gc
