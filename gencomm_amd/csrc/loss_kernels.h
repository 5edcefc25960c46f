// Detection-head part of the GenComm training criterion in ONE launch: classification (sigmoid focal), regression (smooth-L1 on the
// sin-difference encoding) and direction (softmax cross entropy over the heading bins) of PointPillarLoss
// (opencood/loss/point_pillar_loss.py:36-126, helpers :129-170, :216-245), forward AND the gradients of the three weighted sums with
// respect to the head maps.  The reference writes the criterion as ~70 framework elementwise operators on kilobyte-sized maps and
// autograd adds ~90 more in the backward: ~150 launches of a 1 150-launch training step for microseconds of arithmetic.
//
// Layouts as the heads and the collate produce them: cls [B][A][H][W], reg [B][7A][H][W], dir [B][A*A][H][W] (the reference's
// `view(-1, anchor_num)` groups the direction channels by anchor: logits of anchor a = channels a*A .. a*A+A-1);
// pos / neg [B][H][W][A], targets [B][H][W][7A].  Anchor k of a sample = (pixel hw, anchor a), k = hw * A + a.
// One thread per (sample, anchor a, pixel): consecutive threads walk the pixels of one channel (coalesced in the NCHW maps).
#pragma once
#include "common.h"

namespace gc {

constexpr int kLossMaxAnchors = 8;

struct HeadLossArgs {
  const float *cls, *reg, *dir;
  const float *pos, *neg, *tgt;
  float *gcls, *greg, *gdir;   // d (cls_loss + reg_loss + dir_loss) / d map, same layouts as cls / reg / dir
  double* sums;                // [4] += cls_loss, reg_loss, dir_loss (weighted, / batch size), their sum; the caller zeroes it
  int B, A, HW;
  int has_dir, num_bins;
  float pos_cls_weight, gamma, alpha, cls_weight, sigma, reg_weight, dir_weight, inv_bs;
  double dir_offset;
  double anchor_yaw[kLossMaxAnchors];   // radians, float64 like the reference's numpy table
};

__global__ __launch_bounds__(256) void head_loss_kernel(const HeadLossArgs a) {
  __shared__ float s_red[4][4];
  const int b = blockIdx.y, tid = threadIdx.x, A = a.A, HW = a.HW, NA = HW * A;
  // number of positive anchors of the sample (point_pillar_loss.py:72-74: clamp(min = 1)): every workgroup of the sample counts them
  // itself -- NA values, cache hits after the first workgroup -- instead of a launch of its own
  float cnt = 0.f;
  for (int k = tid; k < NA; k += 256) cnt += a.pos[(size_t)b * NA + k] > 0.f ? 1.f : 0.f;
  cnt = wave_sum(cnt);
  if ((tid & 63) == 0) s_red[tid >> 6][3] = cnt;
  __syncthreads();
  const float pos_norm = fmaxf(s_red[0][3] + s_red[1][3] + s_red[2][3] + s_red[3][3], 1.0f);

  float l_cls = 0.f, l_reg = 0.f, l_dir = 0.f;
  const int i = blockIdx.x * 256 + tid;   // (anchor a, pixel hw) of sample b
  if (i < NA) {
    const int an = i / HW, hw = i - an * HW, k = hw * A + an;
    const float t = a.pos[(size_t)b * NA + k];
    const bool positive = t > 0.f, negative = a.neg[(size_t)b * NA + k] > 0.f;
    // ---- classification: sigmoid focal loss (point_pillar_loss.py:230-245)
    {
      const size_t e = ((size_t)b * A + an) * HW + hw;
      const float x = a.cls[e];
      const float w = ((positive ? a.pos_cls_weight : 0.f) + (negative ? 1.f : 0.f)) / pos_norm;
      const float ex = expf(-fabsf(x));
      const float ce = fmaxf(x, 0.f) - x * t + log1pf(ex);
      const float p = 1.0f / (1.0f + expf(-x));
      const float pt = t * p + (1.f - t) * (1.f - p), q = 1.f - pt;
      const float mod = a.gamma == 2.0f ? q * q : powf(q, a.gamma);
      const float dmod = a.gamma == 2.0f ? 2.0f * q : (q > 0.f ? a.gamma * powf(q, a.gamma - 1.0f) : 0.f);   // d mod / d q
      const float aw = t * a.alpha + (1.f - t) * (1.f - a.alpha);
      l_cls = mod * aw * ce * w;
      const float dce = (x >= 0.f ? 1.f : 0.f) - t - (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f)) * ex / (1.f + ex);   // clamp(min = 0) passes the gradient at 0, |x| does not
      const float dq = -(2.f * t - 1.f) * p * (1.f - p);   // d q / d x
      a.gcls[e] = (dmod * dq * ce + mod * dce) * aw * w * a.cls_weight * a.inv_bs;
    }
    // ---- regression: smooth-L1 on (x, y, z, h, w, l, sin-difference of the yaw), positives only (:129-140, :216-226)
    const float rw = (positive ? 1.f : 0.f) / pos_norm;
    const float s2 = a.sigma * a.sigma, thr = 1.0f / s2;
    float t6 = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const size_t e = ((size_t)b * 7 * A + an * 7 + j) * HW + hw;
      const float pv = a.reg[e], tv = a.tgt[((size_t)b * NA + k) * 7 + j];
      float d, dd = 1.0f;   // d = encoded prediction - encoded target, dd = d d / d prediction
      if (j == 6) {
        float sp, cp, st, ct;
        sincosf(pv, &sp, &cp);
        sincosf(tv, &st, &ct);
        d = sp * ct - cp * st;
        dd = cp * ct + sp * st;
        t6 = tv;
      } else {
        d = pv - tv;
      }
      const float ad = fabsf(d);
      const bool lt = ad <= thr;
      const float as = ad * a.sigma;
      l_reg += (lt ? 0.5f * as * as : ad - 0.5f / s2) * rw;
      const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      a.greg[e] = (lt ? s2 * ad : 1.0f) * sgn * dd * rw * a.reg_weight * a.inv_bs;
    }
    // ---- direction: bin of the ground-truth heading (:142-170, float64 like the reference), softmax cross entropy, positives only
    if (a.has_dir) {
      const double two_pi = 6.283185307179586476925286766559;
      const double v = ((double)t6 + a.anchor_yaw[an]) - a.dir_offset;
      const double off = v - floor(v / two_pi) * two_pi;
      long long bin = (long long)floor(off / (two_pi / a.num_bins));
      bin = bin < 0 ? 0 : (bin > a.num_bins - 1 ? a.num_bins - 1 : bin);
      float lg[kLossMaxAnchors], mx = -INFINITY;
      for (int c = 0; c < A; ++c) {
        lg[c] = a.dir[((size_t)b * A * A + an * A + c) * HW + hw];
        mx = fmaxf(mx, lg[c]);
      }
      float se = 0.f, lt = 0.f;   // lt: the target bin's logit (selected in the loop: no dynamically indexed register array)
      for (int c = 0; c < A; ++c) {
        se += expf(lg[c] - mx);
        lt = c == (int)bin ? lg[c] : lt;
      }
      const float lse = mx + logf(se);
      l_dir = (lse - lt) * rw;
      for (int c = 0; c < A; ++c)
        a.gdir[((size_t)b * A * A + an * A + c) * HW + hw] = (expf(lg[c] - lse) - (c == (int)bin ? 1.f : 0.f)) * rw * a.dir_weight * a.inv_bs;
    }
  }
  l_cls = wave_sum(l_cls);
  l_reg = wave_sum(l_reg);
  l_dir = wave_sum(l_dir);
  if ((tid & 63) == 0) { s_red[tid >> 6][0] = l_cls; s_red[tid >> 6][1] = l_reg; s_red[tid >> 6][2] = l_dir; }
  __syncthreads();
  if (tid < 3 && (tid < 2 || a.has_dir)) {
    const float wsel = tid == 0 ? a.cls_weight : (tid == 1 ? a.reg_weight : a.dir_weight);
    const double v = ((double)s_red[0][tid] + (double)s_red[1][tid] + (double)s_red[2][tid] + (double)s_red[3][tid]) * (double)wsel * (double)a.inv_bs;
    atomicAdd(&a.sums[tid], v);
    atomicAdd(&a.sums[3], v);   // their sum: the differentiable output of the host-side Function
  }
}

inline int head_loss_enqueue(const HeadLossArgs& a, hipStream_t st) {
  const int NA = a.HW * a.A;
  head_loss_kernel<<<dim3((NA + 255) / 256, a.B), 256, 0, st>>>(a);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

}  // namespace gc
