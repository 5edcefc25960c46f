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
  int mode;              // 0 = ego-row attention (AttFusion), 1 = element-wise max over agents (MaxFusion)
};

// Workgroup = 64 pixels x 4 channel quarters (wave cq owns channels cq, cq + 4, ...): four times the workgroups and a quarter of the
// dependent gathers per thread of the one-lane-per-pixel form (2 agents x 128 x 64 x 128: 117 -> see DESIGN section 7); the four partial
// scores of a pixel meet in LDS.
template <int N>
__device__ __forceinline__ void fuse_body(const FuseArgs& a, int b, int off, int pix_raw, int cq, int pl, float (*s_part)[4][64]) {
  const int H = a.H, W = a.W, HW = H * W;
  const bool live = pix_raw < HW;
  const int pix = live ? pix_raw : HW - 1;
  const int h = pix / W, w = pix - h * W;
  // affine_grid base coordinates, align_corners=False: (2i+1)/size - 1, in float64 like theta
  const double xb = (2.0 * w + 1.0) / (double)W - 1.0;
  const double yb = (2.0 * h + 1.0) / (double)H - 1.0;
  int idx[N][4];
  unsigned ok[N];
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
    // every tap is LOADED, from the corner clamped into the map, and an invalid one is replaced by an exact zero afterwards: with a
    // branch per tap the loads of a channel were issued one memory latency after the other (the token-major kernel's lesson)
    const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0 + 1, 0), W - 1), yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y0 + 1, 0), H - 1);
    idx[j][0] = yc0 * W + xc0; idx[j][1] = yc0 * W + xc1; idx[j][2] = yc1 * W + xc0; idx[j][3] = yc1 * W + xc1;
    ok[j] = (xl && yt && !far ? 1u : 0u) | (xr && yt && !far ? 2u : 0u) | (xl && yb_ && !far ? 4u : 0u) | (xr && yb_ && !far ? 8u : 0u);
    wt[j][0] = wnw; wt[j][1] = wne; wt[j][2] = wsw; wt[j][3] = wse;
  }
  auto sample = [&](int j, const float* __restrict__ plane) {
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = plane[idx[j][k]];
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) v = fmaf((ok[j] >> k) & 1u ? t[k] : 0.f, wt[j][k], v);
    return v;
  };
  const float* __restrict__ xs = a.x + (size_t)off * a.C * HW;
  if (a.mode == 1) {  // MaxFusion (fusion_in_one.py:87-124): max over the warped agents, zeros where an agent is out of range
    float* __restrict__ op = a.out + (size_t)b * a.C * HW + pix;
    for (int c = cq; c < a.C; c += 4) {
      float o = sample(0, xs + (size_t)c * HW);
#pragma unroll
      for (int j = 1; j < N; ++j) o = fmaxf(o, sample(j, xs + ((size_t)j * a.C + c) * HW));
      if (live) op[(size_t)c * HW] = o;
    }
    return;
  }
  float score[N];
#pragma unroll
  for (int j = 0; j < N; ++j) score[j] = 0.f;
  // C = 64 with up to four agents (the metric / training geometry): the 16 x N warped samples of this thread's channel quarter stay in
  // registers between the score pass and the weighted sum -- the second pass gathered every tap again (training forward, 4 x 200 x 704:
  // 220 us for 180 MB of algorithmic traffic)
#ifdef GC_DIAG_FUSE_NO_KEEP   // diagnostic build: both passes gather their taps (before the samples were kept)
  constexpr bool KEEP = false;
#else
  constexpr bool KEEP = N <= 4;
#endif
  float keep[KEEP ? N : 1][KEEP ? 16 : 1];
  const bool kept = KEEP && a.C == 64;
  if (kept) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = cq + 4 * i;
#pragma unroll
      for (int j = 0; j < N; ++j) keep[KEEP ? j : 0][KEEP ? i : 0] = sample(j, xs + ((size_t)j * 64 + c) * HW);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float v0 = keep[0][KEEP ? i : 0];
      score[0] = fmaf(v0, v0, score[0]);
#pragma unroll
      for (int j = 1; j < N; ++j) score[j] = fmaf(v0, keep[KEEP ? j : 0][KEEP ? i : 0], score[j]);
    }
  } else {
    for (int c = cq; c < a.C; c += 4) {
      const float v0 = sample(0, xs + (size_t)c * HW);
      score[0] = fmaf(v0, v0, score[0]);
#pragma unroll
      for (int j = 1; j < N; ++j) score[j] = fmaf(v0, sample(j, xs + ((size_t)j * a.C + c) * HW), score[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < N; ++j) s_part[j][cq][pl] = score[j];
  __syncthreads();  // every thread of the workgroup gets here: dead pixels were clamped, not retired
#pragma unroll
  for (int j = 0; j < N; ++j) score[j] = (s_part[j][0][pl] + s_part[j][1][pl]) + (s_part[j][2][pl] + s_part[j][3][pl]);
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
  if (kept) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < N; ++j) o = fmaf(score[j], keep[KEEP ? j : 0][KEEP ? i : 0], o);
      if (live) op[(size_t)(cq + 4 * i) * HW] = o;
    }
    return;
  }
  for (int c = cq; c < a.C; c += 4) {
    float o = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) o = fmaf(score[j], sample(j, xs + ((size_t)j * a.C + c) * HW), o);
    if (live) op[(size_t)c * HW] = o;
  }
}

__global__ __launch_bounds__(256) void warp_attfuse_kernel(const FuseArgs a) {
  __shared__ float s_part[8][4][64];
  const int b = blockIdx.y;
  const int off = a.scene_off[b], N = a.scene_off[b + 1] - off;   // block-uniform
  const int pl = threadIdx.x & 63, cq = threadIdx.x >> 6;
  const int pix = blockIdx.x * 64 + pl;
  switch (N) {
    case 1: fuse_body<1>(a, b, off, pix, cq, pl, s_part); break;
    case 2: fuse_body<2>(a, b, off, pix, cq, pl, s_part); break;
    case 3: fuse_body<3>(a, b, off, pix, cq, pl, s_part); break;
    case 4: fuse_body<4>(a, b, off, pix, cq, pl, s_part); break;
    case 5: fuse_body<5>(a, b, off, pix, cq, pl, s_part); break;
    case 6: fuse_body<6>(a, b, off, pix, cq, pl, s_part); break;
    case 7: fuse_body<7>(a, b, off, pix, cq, pl, s_part); break;
    case 8: fuse_body<8>(a, b, off, pix, cq, pl, s_part); break;
    default: break;  // host validates 1 <= N <= 8 where it can; otherwise the scene is left untouched
  }
}

inline int warp_attfuse_enqueue(const float* x, const double* theta, const int* scene_off, float* out,
                                int B, int n, int C, int H, int W, hipStream_t st, int mode = 0) {
  (void)n;
  FuseArgs a{x, theta, scene_off, out, C, H, W, mode};
  TimedLaunch tl(KF_WARP_ATTFUSE, st);
  warp_attfuse_kernel<<<dim3((H * W + 63) / 64, B), 256, 0, st>>>(a);
  GC_HIP(hipGetLastError());
  return GC_OK;
}



// ---------------------------------------------------------------------------------------------
// Backward of warp + ego-row attention (training: the detection loss back-propagates through AttFusion into the Enhancer
// and GenComm).  One lane per OUTPUT pixel, same taps as the forward:
//   w = softmax(s), s_j = <x_0, x_j> / sqrt(C), out = sum_j w_j x_j
//   d w_j = <g, x_j>,  d s_j = w_j (d w_j - sum_k w_k d w_k) / sqrt(C)
//   d x_j = w_j g + d s_j x_0  (j != 0),   d x_0 = w_0 g + sum_j d s_j x_j + d s_0 x_0
// and the bilinear gather's adjoint: every tap of d x_j is added (float atomics) to the source pixel it was read from.
// grad_x must be zero on entry.
// ---------------------------------------------------------------------------------------------
struct FuseBwdArgs {
  const float* x;        // [n][C][H][W] forward input
  const double* theta;   // [n][2][3]
  const int* scene_off;  // [B+1]
  const float* gout;     // [B][C][H][W]
  float* gx;             // [n][C][H][W], zeroed by the caller
  int C, H, W;
  float* ws;             // scratch [n][HW][2]: softmax weight and d score of (agent, output pixel), for the gather pass
  int* plan;             // scratch [n]: 1 = this agent's d x is formed by the gather pass, 0 = by the scatter (float atomics)
};

// Round 4 (last third): workgroup = 64 pixels x 4 channel quarters, as the forward (one wave per workgroup walked all C channels twice:
// 2 200 single-wave workgroups per scene, two waves per SIMD -- latency-bound, 250 us at 4 x 64 x 200 x 704); the partial scores / d weights
// of a pixel meet in LDS, and at C = 64 with up to four agents the 16 x N warped samples and the 16 output gradients of a thread stay in
// registers between the two passes.
template <int N>
__device__ __forceinline__ void fuse_bwd_body(const FuseBwdArgs& a, int b, int off, int pix_raw, int cq, int pl, float (*s_part)[4][64]) {
  const int H = a.H, W = a.W, HW = H * W;
  const bool live = pix_raw < HW;
  const int pix = live ? pix_raw : HW - 1;
  const int h = pix / W, w = pix - h * W;
  const double xb = (2.0 * w + 1.0) / (double)W - 1.0;
  const double yb = (2.0 * h + 1.0) / (double)H - 1.0;
  int idx[N][4];
  unsigned ok[N];
  float wt[N][4];
#pragma unroll
  for (int j = 0; j < N; ++j) {  // identical to fuse_body's tap computation
    const double* __restrict__ th = a.theta + (size_t)(off + j) * 6;
    const float gx = (float)(th[0] * xb + th[1] * yb + th[2]);
    const float gy = (float)(th[3] * xb + th[4] * yb + th[5]);
    const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
    const float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)W + 1.f);
    const int y0 = (int)fminf(fmaxf(fy, -2.f), (float)H + 1.f);
    const float tx = ix - fx, ty = iy - fy;
    const bool xl = x0 >= 0 && x0 < W, xr = x0 + 1 >= 0 && x0 + 1 < W;
    const bool yt = y0 >= 0 && y0 < H, yb_ = y0 + 1 >= 0 && y0 + 1 < H;
    const bool far = fx != (float)x0 || fy != (float)y0;
    // every tap is LOADED, from the corner clamped into the map, and an invalid one is replaced by an exact zero afterwards: with a
    // branch per tap the loads of a channel were issued one memory latency after the other (the token-major kernel's lesson)
    const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0 + 1, 0), W - 1), yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y0 + 1, 0), H - 1);
    idx[j][0] = yc0 * W + xc0; idx[j][1] = yc0 * W + xc1; idx[j][2] = yc1 * W + xc0; idx[j][3] = yc1 * W + xc1;
    ok[j] = (xl && yt && !far ? 1u : 0u) | (xr && yt && !far ? 2u : 0u) | (xl && yb_ && !far ? 4u : 0u) | (xr && yb_ && !far ? 8u : 0u);
    wt[j][0] = (1.f - tx) * (1.f - ty); wt[j][1] = tx * (1.f - ty); wt[j][2] = (1.f - tx) * ty; wt[j][3] = tx * ty;
  }
  auto sample = [&](int j, const float* __restrict__ plane) {
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = plane[idx[j][k]];
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) v = fmaf((ok[j] >> k) & 1u ? t[k] : 0.f, wt[j][k], v);
    return v;
  };
  const float* __restrict__ xs = a.x + (size_t)off * a.C * HW;
  const float* __restrict__ gp = a.gout + (size_t)b * a.C * HW + pix;
  float score[N], dw[N];
#pragma unroll
  for (int j = 0; j < N; ++j) { score[j] = 0.f; dw[j] = 0.f; }
#ifdef GC_DIAG_FUSE_NO_KEEP
  constexpr bool KEEP = false;
#else
  constexpr bool KEEP = N <= 4;
#endif
  float keep[KEEP ? N : 1][KEEP ? 16 : 1], gk[KEEP ? 16 : 1];
  const bool kept = KEEP && a.C == 64;
  if (kept) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = cq + 4 * i;
      gk[KEEP ? i : 0] = gp[(size_t)c * HW];
#pragma unroll
      for (int j = 0; j < N; ++j) keep[KEEP ? j : 0][KEEP ? i : 0] = sample(j, xs + ((size_t)j * 64 + c) * HW);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float g = gk[KEEP ? i : 0], v0 = keep[0][KEEP ? i : 0];
      score[0] = fmaf(v0, v0, score[0]);
      dw[0] = fmaf(g, v0, dw[0]);
#pragma unroll
      for (int j = 1; j < N; ++j) {
        const float vj = keep[KEEP ? j : 0][KEEP ? i : 0];
        score[j] = fmaf(v0, vj, score[j]);
        dw[j] = fmaf(g, vj, dw[j]);
      }
    }
  } else {
    for (int c = cq; c < a.C; c += 4) {
      const float g = gp[(size_t)c * HW];
      const float v0 = sample(0, xs + (size_t)c * HW);
      score[0] = fmaf(v0, v0, score[0]);
      dw[0] = fmaf(g, v0, dw[0]);
#pragma unroll
      for (int j = 1; j < N; ++j) {
        const float vj = sample(j, xs + ((size_t)j * a.C + c) * HW);
        score[j] = fmaf(v0, vj, score[j]);
        dw[j] = fmaf(g, vj, dw[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < N; ++j) { s_part[j][cq][pl] = score[j]; s_part[N + j][cq][pl] = dw[j]; }
  __syncthreads();  // every thread of the workgroup gets here: dead pixels were clamped, not retired
#pragma unroll
  for (int j = 0; j < N; ++j) {
    score[j] = (s_part[j][0][pl] + s_part[j][1][pl]) + (s_part[j][2][pl] + s_part[j][3][pl]);
    dw[j] = (s_part[N + j][0][pl] + s_part[N + j][1][pl]) + (s_part[N + j][2][pl] + s_part[N + j][3][pl]);
  }
  const float inv = 1.0f / sqrtf((float)a.C);
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < N; ++j) { score[j] *= inv; mx = fmaxf(mx, score[j]); }
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < N; ++j) { score[j] = expf(score[j] - mx); den += score[j]; }
  float sdot = 0.f;
#pragma unroll
  for (int j = 0; j < N; ++j) { score[j] /= den; sdot = fmaf(score[j], dw[j], sdot); }
  float ds[N];
#pragma unroll
  for (int j = 0; j < N; ++j) ds[j] = score[j] * (dw[j] - sdot) * inv;
  int gat[N];   // agent j's input gradient comes from the gather pass (fuse_bwd_gather_kernel): this pass only publishes (w_j, d s_j)
#pragma unroll
  for (int j = 0; j < N; ++j) {
    gat[j] = a.plan[off + j];
    if (cq == 0 && live) {
      float2* __restrict__ wp = reinterpret_cast<float2*>(a.ws) + (size_t)(off + j) * HW + pix;
      *wp = make_float2(score[j], ds[j]);
    }
  }
  if (!live) return;   // after the only barrier
  float* __restrict__ gxs = a.gx + (size_t)off * a.C * HW;
  auto emit = [&](int c, float g, const float (&v)[N]) {
    float d0 = fmaf(score[0], g, ds[0] * v[0]);
#pragma unroll
    for (int j = 0; j < N; ++j) d0 = fmaf(ds[j], v[j], d0);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (gat[j] == 1 && j != 0) continue;
      const float dj = j == 0 ? d0 : fmaf(score[j], g, ds[j] * v[0]);
      float* __restrict__ plane = gxs + ((size_t)j * a.C + c) * HW;
      if (j == 0 && gat[0] == 1) {   // the ego's warp is the identity: its own pixel, exactly once
        plane[pix] = dj;
        continue;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (((ok[j] >> k) & 1u) && wt[j][k] != 0.f) atomicAdd(plane + idx[j][k], wt[j][k] * dj);
    }
  };
  if (kept) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v[N];
#pragma unroll
      for (int j = 0; j < N; ++j) v[j] = keep[KEEP ? j : 0][KEEP ? i : 0];
      emit(cq + 4 * i, gk[KEEP ? i : 0], v);
    }
    return;
  }
  for (int c = cq; c < a.C; c += 4) {
    const float g = gp[(size_t)c * HW];
    float v[N];
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = sample(j, xs + ((size_t)j * a.C + c) * HW);
    emit(c, g, v);
  }
}

// Which agents take the gather pass (decided on the device: theta lives there and nothing synchronises the host).  Ego (first agent
// of a scene): 1 if its warp is exactly the identity -- d x_0 is then written once per pixel without atomics.  Others: 1 if the
// inverse map is tame (in pixel units every row of its matrix has an L1 norm <= 1.5, true for the rigid transforms of
// normalize_pairwise_tfm): at most KM output pixels sample a source pixel and they lie in a 5 x 5 window around its pre-image.
constexpr int FUSE_KM = 12;
__global__ __launch_bounds__(64) void fuse_bwd_plan_kernel(const FuseBwdArgs a, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int off = a.scene_off[b], N = a.scene_off[b + 1] - off;
  for (int j = 0; j < N; ++j) {
    const double* th = a.theta + (size_t)(off + j) * 6;
    int ok;
    if (N > 8) {   // the pixel pass instantiates 1..8 agents per scene: nothing of such a scene is computed (its gradients stay zero; the
      a.plan[off + j] = 0;   // Python modules refuse it before any launch, MAX_AGENTS_PER_SCENE), so the gather must not read its scratch
      continue;
    }
    if (j == 0) {
      ok = th[0] == 1.0 && th[1] == 0.0 && th[2] == 0.0 && th[3] == 0.0 && th[4] == 1.0 && th[5] == 0.0;
    } else {
      // pixel-space matrix of the forward map: ix = (W/2)(th0 xb + th1 yb + th2 + 1) - 1/2 with xb = (2 px + 1)/W - 1, ...
      const double m00 = th[0], m01 = th[1] * a.W / a.H, m10 = th[3] * a.H / a.W, m11 = th[4];
      const double det = m00 * m11 - m01 * m10;
      ok = 0;
      if (fabs(det) > 0.25) {
        const double r0 = (fabs(m11) + fabs(m01)) / fabs(det), r1 = (fabs(m10) + fabs(m00)) / fabs(det);
        ok = (r0 <= 1.5 && r1 <= 1.5 && fabs(det) > 0.6) ? 1 : 0;   // area of the pre-image of a 2 x 2 cell < 4 / 0.6: fewer than KM lattice points
      }
      if (a.plan[off] != 1) ok = 0;   // the gather needs x_0[c][p] = the ego's sample at p: only with an identity ego
    }
    a.plan[off + j] = ok;
  }
}

// d x_j for a non-ego agent without atomics: thread = one SOURCE pixel q of agent j.  The output pixels p whose bilinear cell
// contains q lie around the pre-image of q; each candidate's cell is recomputed with exactly the forward's arithmetic (float64
// affine grid cast to float32, floor, clamp), matches are kept in LDS as (p, wt w_j(p), wt d s_j(p)) and then, per channel,
//   d x_j[c][q] = sum over matches (wt w_j) g[c][p] + (wt d s_j) x_0[c][p]          (x_0[c][p]: the ego's warp is the identity)
// Deterministic (no float atomics), every element written exactly once.
__global__ __launch_bounds__(128) void fuse_bwd_gather_kernel(const FuseBwdArgs a, int B) {
  __shared__ int s_p[FUSE_KM][128];
  __shared__ float s_wa[FUSE_KM][128], s_wb[FUSE_KM][128];
  const int ag = blockIdx.y, tid = threadIdx.x;
  if (a.plan[ag] != 1) return;
  int b = 0;
  while (b + 1 < B && a.scene_off[b + 1] <= ag) ++b;
  const int off = a.scene_off[b];
  if (ag == off) return;                 // the ego: written by the pixel pass
  if (a.plan[off] != 1) return;          // needs x_0[c][p] = the ego's sample at p: only with an identity ego (else the scatter ran)
  const int H = a.H, W = a.W, HW = H * W;
  const int q = blockIdx.x * 128 + tid;
  const bool live = q < HW;
  const int qy = live ? q / W : 0, qx = live ? q - qy * W : 0;
  const double* __restrict__ th = a.theta + (size_t)ag * 6;
  int cnt = 0;
  const float2* __restrict__ wsp = reinterpret_cast<const float2*>(a.ws) + (size_t)ag * HW;
  // every output pixel p whose bilinear cell contains q, with its tap weight: f(p, wgt).  Candidates = a window around q's pre-image
  // (float64); each candidate's cell is recomputed with exactly fuse_body's arithmetic
  auto for_each_match = [&](auto&& f) {
    const double gx = (2.0 * qx + 1.0) / (double)W - 1.0, gy = (2.0 * qy + 1.0) / (double)H - 1.0;
    const double det = th[0] * th[4] - th[1] * th[3];
    const double xb = (th[4] * (gx - th[2]) - th[1] * (gy - th[5])) / det, yb = (-th[3] * (gx - th[2]) + th[0] * (gy - th[5])) / det;
    const double pxf = ((xb + 1.0) * W - 1.0) * 0.5, pyf = ((yb + 1.0) * H - 1.0) * 0.5;
    const int x_lo = max((int)floor(pxf - 1.6), 0), x_hi = min((int)ceil(pxf + 1.6), W - 1);
    const int y_lo = max((int)floor(pyf - 1.6), 0), y_hi = min((int)ceil(pyf + 1.6), H - 1);
    for (int py = y_lo; py <= y_hi; ++py)
      for (int px = x_lo; px <= x_hi; ++px) {
        const double oxb = (2.0 * px + 1.0) / (double)W - 1.0, oyb = (2.0 * py + 1.0) / (double)H - 1.0;
        const float sgx = (float)(th[0] * oxb + th[1] * oyb + th[2]);
        const float sgy = (float)(th[3] * oxb + th[4] * oyb + th[5]);
        const float ix = ((sgx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((sgy + 1.f) * (float)H - 1.f) * 0.5f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)W + 1.f), y0 = (int)fminf(fmaxf(fy, -2.f), (float)H + 1.f);
        if (fx != (float)x0 || fy != (float)y0) continue;   // clamped: everything out of range
        const int dx = qx - x0, dy = qy - y0;
        if (dx < 0 || dx > 1 || dy < 0 || dy > 1) continue;
        const float tx = ix - fx, ty = iy - fy;
        const float wgt = (dx ? tx : 1.f - tx) * (dy ? ty : 1.f - ty);
        if (wgt == 0.f) continue;
        f(py * W + px, wgt);
      }
  };
  if (live) {
    for_each_match([&](int p, float wgt) {
      if (cnt < FUSE_KM) {
        const float2 sd = wsp[p];
        s_p[cnt][tid] = p; s_wa[cnt][tid] = wgt * sd.x; s_wb[cnt][tid] = wgt * sd.y;
      }
      ++cnt;
    });
  }
  // More than FUSE_KM matches (the plan's bound is on the pre-image's AREA, not on its lattice-point count: a thin sheared cell can hold
  // more): the first FUSE_KM come from LDS, the rest are found again per channel -- slow, exact, and never silently dropped
  const bool overflow = cnt > FUSE_KM;
  const int kept = min(cnt, FUSE_KM);
  const float* __restrict__ x0p = a.x + (size_t)off * a.C * HW;
  const float* __restrict__ gp = a.gout + (size_t)b * a.C * HW;
  float* __restrict__ out = a.gx + (size_t)ag * a.C * HW + q;
  for (int c = 0; c < a.C; ++c) {
    float acc = 0.f;
    for (int k = 0; k < kept; ++k) {
      const int p = s_p[k][tid];
      acc = fmaf(s_wa[k][tid], gp[(size_t)c * HW + p], acc);
      acc = fmaf(s_wb[k][tid], x0p[(size_t)c * HW + p], acc);
    }
    if (overflow) {
      int m = 0;
      for_each_match([&](int p, float wgt) {
        if (m++ >= FUSE_KM) {
          const float2 sd = wsp[p];
          acc = fmaf(wgt * sd.x, gp[(size_t)c * HW + p], acc);
          acc = fmaf(wgt * sd.y, x0p[(size_t)c * HW + p], acc);
        }
      });
    }
    if (live) out[(size_t)c * HW] = acc;
  }
}

__global__ __launch_bounds__(256) void warp_attfuse_bwd_kernel(const FuseBwdArgs a) {
  __shared__ float s_part[16][4][64];   // [score of agent j | N + j: d weight of agent j][channel quarter][pixel]
  const int b = blockIdx.y;
  const int off = a.scene_off[b], N = a.scene_off[b + 1] - off;   // block-uniform
  const int pl = threadIdx.x & 63, cq = threadIdx.x >> 6;
  const int pix = blockIdx.x * 64 + pl;
  switch (N) {
    case 1: fuse_bwd_body<1>(a, b, off, pix, cq, pl, s_part); break;
    case 2: fuse_bwd_body<2>(a, b, off, pix, cq, pl, s_part); break;
    case 3: fuse_bwd_body<3>(a, b, off, pix, cq, pl, s_part); break;
    case 4: fuse_bwd_body<4>(a, b, off, pix, cq, pl, s_part); break;
    case 5: fuse_bwd_body<5>(a, b, off, pix, cq, pl, s_part); break;
    case 6: fuse_bwd_body<6>(a, b, off, pix, cq, pl, s_part); break;
    case 7: fuse_bwd_body<7>(a, b, off, pix, cq, pl, s_part); break;
    case 8: fuse_bwd_body<8>(a, b, off, pix, cq, pl, s_part); break;
    default: break;
  }
}

inline size_t warp_attfuse_bwd_scratch_floats(int n, int H, int W) { return (size_t)n * H * W * 2 + (size_t)n + 64; }
// scratch: warp_attfuse_bwd_scratch_floats(n, H, W) floats, or null -- every agent then takes the scatter with float atomics (round 2)
inline int warp_attfuse_bwd_enqueue(const float* x, const double* theta, const int* scene_off, const float* gout, float* gx, float* scratch,
                                    int B, int n, int C, int H, int W, hipStream_t st) {
  GC_HIP(hipMemsetAsync(gx, 0, (size_t)n * C * H * W * sizeof(float), st));
  FuseBwdArgs a{x, theta, scene_off, gout, gx, C, H, W, nullptr, nullptr};
  if (scratch != nullptr) {
    a.ws = scratch;
    a.plan = reinterpret_cast<int*>(scratch + (size_t)n * H * W * 2);
    fuse_bwd_plan_kernel<<<(B + 63) / 64, 64, 0, st>>>(a, B);
  } else {
    return fail(GC_ERR_ARG, "warp_attfuse_bwd: scratch is required");
  }
  warp_attfuse_bwd_kernel<<<dim3((H * W + 63) / 64, B), 256, 0, st>>>(a);
  fuse_bwd_gather_kernel<<<dim3((H * W + 127) / 128, n), 128, 0, st>>>(a, B);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

// ---------------------------------------------------------------------------------------------
// Token-major fast path (ScenePipeline): reads the Enhancer's token-major result O [n][HW][C] and
// its channel gate [n][C] straight from the Enhancer workspace (the NHWC->NCHW transpose launch is
// skipped and the gate multiply happens here). QL lanes share one output pixel, each owning C/QL
// consecutive channels as float4 quads, so a bilinear corner is ONE contiguous C*4-byte row fetched
// by QL adjacent lanes (fully coalesced however the agent is rotated); every agent's warped
// feature stays in registers (no second gather pass); the <x_0, x_j> dot products are reduced over
// the QL lanes with DPP-free xor shuffles inside the 16-lane group; results are transposed through
// LDS so that the NCHW output is written in 256-B runs.
// ---------------------------------------------------------------------------------------------
struct FuseTokArgs {
  const float* O;        // [n][HW][C]
  const float* gate;     // [n][C]
  const double* theta;   // [n][2][3]
  const int* scene_off;  // [B+1]
  float* out;            // [B][C][H][W]
  int C, H, W;
  int xcd;               // 1: XCD-aware block order (MODE_XCD_REMAP)
};

// bilinear cell of one (output pixel, agent): the four corner pixels CLAMPED into the map (every corner is loaded, without a branch:
// with a branch per corner the loads of a pass were issued one latency after the other), fractions, valid corners
struct __align__(16) FuseGeo { int idx[4]; float tx, ty; unsigned mask, pad; };

template <int N, int QPL /*float4 quads per lane*/>
__device__ __forceinline__ void fuse_tok_body(const FuseTokArgs& a, int bx, int b, int off, float* s_out /*[64][C+1]*/, FuseGeo (*s_geo)[64]) {
  const int tid = threadIdx.x;
  const int C = a.C, H = a.H, W = a.W, HW = H * W;
  // sampling geometry ONCE per (pixel, agent) -- one thread each -- instead of once per lane of the pixel's 16: the float64
  // affine grid (as F.affine_grid evaluates the reference's float64 theta) with its two divisions was 16 x redundant
  for (int e = tid; e < 64 * N; e += 256) {
    const int pl = e & 63, j = e >> 6;
    const int pix = bx * 64 + pl;
    const bool ok = pix < HW;
    const int h = ok ? pix / W : 0, w = ok ? pix - h * W : 0;
    const double xb = (2.0 * w + 1.0) / (double)W - 1.0, yb = (2.0 * h + 1.0) / (double)H - 1.0;
    const double* __restrict__ th = a.theta + (size_t)(off + j) * 6;
    const float gx = (float)(th[0] * xb + th[1] * yb + th[2]);
    const float gy = (float)(th[3] * xb + th[4] * yb + th[5]);
    const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)W + 1.f), y0 = (int)fminf(fmaxf(fy, -2.f), (float)H + 1.f);
    const bool far = fx != (float)x0 || fy != (float)y0;
    const bool xl = x0 >= 0 && x0 < W, xr = x0 + 1 >= 0 && x0 + 1 < W;
    const bool yt = y0 >= 0 && y0 < H, yb_ = y0 + 1 >= 0 && y0 + 1 < H;
    const bool in = ok && !far;
    FuseGeo g;
    const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0 + 1, 0), W - 1), yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y0 + 1, 0), H - 1);
    g.idx[0] = yc0 * W + xc0; g.idx[1] = yc0 * W + xc1; g.idx[2] = yc1 * W + xc0; g.idx[3] = yc1 * W + xc1;
    g.tx = ix - fx;
    g.ty = iy - fy;
    g.mask = (in && xl && yt ? 1u : 0u) | (in && xr && yt ? 2u : 0u) | (in && xl && yb_ ? 4u : 0u) | (in && xr && yb_ ? 8u : 0u);
    g.pad = 0u;
    s_geo[j][pl] = g;
  }
  __syncthreads();
  const int lp = tid >> 4, ql = tid & 15;       // 16 pixels per 256-thread block pass, 16 lanes per pixel
  const int c0 = ql * 4 * QPL;                  // first channel of this lane
  float4 gate[N][QPL];                          // channel gate of agent j (sigmoid output of SplitAttn), constant over pixels
#pragma unroll
  for (int j = 0; j < N; ++j)
#pragma unroll
    for (int q = 0; q < QPL; ++q) gate[j][q] = *reinterpret_cast<const float4*>(a.gate + (size_t)(off + j) * C + c0 + 4 * q);
  const float inv = 1.0f / sqrtf((float)C);
  for (int pass = 0; pass < 4; ++pass) {        // 64 pixels per block
    const int pl = pass * 16 + lp;              // local pixel 0..63
    float4 v[N][QPL];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const FuseGeo g = s_geo[j][pl];
      const float tx = g.tx, ty = g.ty;
      const float wgt[4] = {(1.f - tx) * (1.f - ty), tx * (1.f - ty), (1.f - tx) * ty, tx * ty};
      const float* __restrict__ base = a.O + (size_t)(off + j) * HW * C + c0;
#pragma unroll
      for (int q = 0; q < QPL; ++q) v[j][q] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool on = (g.mask >> k) & 1u;   // an invalid corner contributes an exact zero whatever the clamped pixel holds
#pragma unroll
        for (int q = 0; q < QPL; ++q) {
          float4 t = *reinterpret_cast<const float4*>(base + (size_t)g.idx[k] * C + 4 * q);
          t.x = on ? t.x : 0.f; t.y = on ? t.y : 0.f; t.z = on ? t.z : 0.f; t.w = on ? t.w : 0.f;
          v[j][q].x = fmaf(t.x, wgt[k], v[j][q].x); v[j][q].y = fmaf(t.y, wgt[k], v[j][q].y);
          v[j][q].z = fmaf(t.z, wgt[k], v[j][q].z); v[j][q].w = fmaf(t.w, wgt[k], v[j][q].w);
        }
      }
#pragma unroll
      for (int q = 0; q < QPL; ++q) {
        v[j][q].x *= gate[j][q].x; v[j][q].y *= gate[j][q].y; v[j][q].z *= gate[j][q].z; v[j][q].w *= gate[j][q].w;
      }
    }
    float sc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float d = 0.f;
#pragma unroll
      for (int q = 0; q < QPL; ++q)
        d += v[0][q].x * v[j][q].x + v[0][q].y * v[j][q].y + v[0][q].z * v[j][q].z + v[0][q].w * v[j][q].w;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) d += __shfl_xor(d, o, 16);   // over the 16 lanes of this pixel
      sc[j] = d;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < N; ++j) { sc[j] *= inv; mx = fmaxf(mx, sc[j]); }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) { sc[j] = expf(sc[j] - mx); den += sc[j]; }
    const float rden = 1.0f / den;
#pragma unroll
    for (int q = 0; q < QPL; ++q) {
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float wj = sc[j] * rden;
        o.x = fmaf(wj, v[j][q].x, o.x); o.y = fmaf(wj, v[j][q].y, o.y); o.z = fmaf(wj, v[j][q].z, o.z); o.w = fmaf(wj, v[j][q].w, o.w);
      }
      float* d = s_out + pl * (C + 1) + c0 + 4 * q;
      d[0] = o.x; d[1] = o.y; d[2] = o.z; d[3] = o.w;
    }
  }
  __syncthreads();
  // NCHW store: lanes over the 64 pixels of the block, waves over channels
  const int p = tid & 63;
  const int pix = bx * 64 + p;
  if (pix < HW)
    for (int c = tid >> 6; c < C; c += 4) a.out[((size_t)b * C + c) * HW + pix] = s_out[p * (C + 1) + c];
}

template <int QPL>
__global__ __launch_bounds__(256) void warp_attfuse_tok_kernel(const FuseTokArgs a) {
  extern __shared__ float s_out[];
  __shared__ FuseGeo s_geo[8][64];
  // XCD-aware order (common.h xcd_block): a block is 64 pixels of ONE output row and shares both source rows of its bilinear
  // cells with the blocks of the neighbouring rows, gridDim.x ids away -- dealt round-robin those land on other XCDs' L2s
  const BlockId bi = xcd_block(a.xcd);
  const int b = bi.y;
  const int off = a.scene_off[b], N = a.scene_off[b + 1] - off;
  switch (N) {
    case 1: fuse_tok_body<1, QPL>(a, bi.x, b, off, s_out, s_geo); break;
    case 2: fuse_tok_body<2, QPL>(a, bi.x, b, off, s_out, s_geo); break;
    case 3: fuse_tok_body<3, QPL>(a, bi.x, b, off, s_out, s_geo); break;
    case 4: fuse_tok_body<4, QPL>(a, bi.x, b, off, s_out, s_geo); break;
    case 5: fuse_tok_body<5, QPL>(a, bi.x, b, off, s_out, s_geo); break;
    case 6: fuse_tok_body<6, QPL>(a, bi.x, b, off, s_out, s_geo); break;
    case 7: fuse_tok_body<7, QPL>(a, bi.x, b, off, s_out, s_geo); break;
    case 8: fuse_tok_body<8, QPL>(a, bi.x, b, off, s_out, s_geo); break;
    default: break;
  }
}

inline int warp_attfuse_tok_enqueue(const float* O, const float* gate, const double* theta, const int* scene_off, float* out,
                                    int B, int C, int H, int W, int xcd, hipStream_t st) {
  FuseTokArgs a{O, gate, theta, scene_off, out, C, H, W, xcd};
  const dim3 grid((H * W + 63) / 64, B);
  const size_t sh = (size_t)64 * (C + 1) * sizeof(float);
  TimedLaunch tl(KF_WARP_ATTFUSE, st);
  if (C == 64) warp_attfuse_tok_kernel<1><<<grid, 256, sh, st>>>(a);
  else if (C == 128) warp_attfuse_tok_kernel<2><<<grid, 256, sh, st>>>(a);
  else if (C == 256) {
    GC_HIP(hipFuncSetAttribute((const void*)warp_attfuse_tok_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    warp_attfuse_tok_kernel<4><<<grid, 256, sh, st>>>(a);
  } else return fail(GC_ERR_ARG, "token-major fusion supports C in {64, 128, 256}");
  GC_HIP(hipGetLastError());
  return GC_OK;
}

}  // namespace gc
