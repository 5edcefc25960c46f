// PFNLayer of the PointPillars encoder in TRAINING mode (opencood/models/sub_modules/pillar_vfe.py:31-54: Linear(F -> C, no bias) ->
// BatchNorm1d with batch statistics -> ReLU -> max over the P point slots of a pillar), forward and backward, WITHOUT the [M P, C]
// intermediates.  As a 1x1 convolution + BatchNorm + slot max (gencomm_conv2d_fwd / gencomm_bn2d_train_* / gencomm_slot_max_*) the
// stage-1 training leg wrote and re-read five 393-MB tensors per step for it (48 000 pillars x 32 slots x 64 channels): 1.84 ms of
// a 19 ms step.  Everything BatchNorm needs of the Linear output z_p[c] = w_c . in_p is a function of two small moment sums of the
// INPUTS over all n = M P point slots,
//     S1 = sum_p in_p (F),   S2 = sum_p in_p in_p^T (F x F):    mean_c = w_c . S1 / n,   E[z_c^2] = w_c^T S2 w_c / n
// (exact algebra, evaluated in f64), and y = ReLU(gamma x^ + beta) is monotone in z, so the slot max of y is y at the arg-max (gamma > 0)
// or arg-min (gamma < 0) slot of z.  Forward = one moments launch + one launch that evaluates the 32 x F x C products of a pillar from
// the LDS and keeps the extreme slot.  Backward: with g_p = the incoming gradient at a pillar's arg slot (where y > 0), zero elsewhere,
//     d beta = sum g,  d gamma = sum g x^,   d z_p = gamma r (g_p - mean(g) - x^_p mean(g x^))   (dense over all slots), and
//     dW_c = sum_p dz_p in_p = gamma r [ sum_p g_p in_p  -  mean(g) S1  -  mean(g x^) r (S2 w_c - mean S1) ]
// -- one sparse gather per (pillar, channel) plus the same moment sums: one launch (per-block partial sums) + a finish launch.
// Slots beyond a pillar's point count carry zero features (pillar_vfe.py:96-100 masks them before the Linear): they take part in
// the statistics and in the max exactly as in the reference.
#pragma once
#include "common.h"

namespace gc {

constexpr int PFN_MAXF = 16, PFN_BWD_BLOCKS = 512;   // backward scratch: PFN_BWD_BLOCKS x C x (F + 2) doubles
struct PfnArgs {
  const float* feats;    // [M][P][F]
  const float* weight;   // [C][F]
  const float* gamma; const float* beta;
  double* moments;       // [F + F F] S1, S2 (row-major, full)
  float* save;           // [C][2] mean, rstd
  int M, P, F, C;
};

__device__ __forceinline__ double wave_sum_f64(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// S1, S2 over all M P slots: every thread walks slots with a grid stride and keeps the F + F (F + 1) / 2 sums in f64 registers
template <int F>
__global__ __launch_bounds__(256) void pfn_moments_kernel(const float* __restrict__ feats, double* __restrict__ moments, long long n) {
  constexpr int NT = F * (F + 1) / 2;
  __shared__ double s_red[4][F + NT];
  double s1[F], s2[NT];
#pragma unroll
  for (int k = 0; k < F; ++k) s1[k] = 0.0;
#pragma unroll
  for (int k = 0; k < NT; ++k) s2[k] = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    double v[F];
#pragma unroll
    for (int k = 0; k < F; ++k) v[k] = (double)feats[i * F + k];
    int t = 0;
#pragma unroll
    for (int a = 0; a < F; ++a) {
      s1[a] += v[a];
#pragma unroll
      for (int b = a; b < F; ++b) { s2[t] = fma(v[a], v[b], s2[t]); ++t; }
    }
  }
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < F; ++k) { const double r = wave_sum_f64(s1[k]); if (l == 0) s_red[wv][k] = r; }
#pragma unroll
  for (int k = 0; k < NT; ++k) { const double r = wave_sum_f64(s2[k]); if (l == 0) s_red[wv][F + k] = r; }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < F + NT) {
    const double tot = s_red[0][t] + s_red[1][t] + s_red[2][t] + s_red[3][t];
    if (t < F) atomicAdd(&moments[t], tot);
    else {   // upper-triangle index -> (a, b); both mirror cells of the full matrix
      int q = t - F, a = 0;
      while (q >= F - a) { q -= F - a; ++a; }
      const int b = a + q;
      atomicAdd(&moments[F + a * F + b], tot);
      if (a != b) atomicAdd(&moments[F + b * F + a], tot);
    }
  }
}

// mean and rstd of channel c's Linear output from the moments (f64)
template <int F>
__device__ __forceinline__ void pfn_channel_stats(const double* __restrict__ moments, const float (&w)[F], double n, float eps, double& mean, double& var) {
  double m = 0.0, q = 0.0;
#pragma unroll
  for (int a = 0; a < F; ++a) {
    m = fma((double)w[a], moments[a], m);
    double row = 0.0;
#pragma unroll
    for (int b = 0; b < F; ++b) row = fma(moments[F + a * F + b], (double)w[b], row);
    q = fma((double)w[a], row, q);
  }
  mean = m / n;
  var = fmax(q / n - mean * mean, 0.0);
}

// forward: block = 256 / C pillars x C channels; the pillars' P x F inputs in the LDS (every channel thread of a pillar reads the same words)
template <int F>
__global__ __launch_bounds__(256) void pfn_fwd_kernel(const PfnArgs a, float* __restrict__ out, unsigned char* __restrict__ arg, float* __restrict__ running_mean,
                                                      float* __restrict__ running_var, long long* __restrict__ nbt, float momentum, float eps) {
  extern __shared__ float pfn_in[];   // [pillars per block][P][F]
  const int ppb = 256 / a.C, tid = threadIdx.x, pl = tid / a.C, c = tid - pl * a.C;
  const int m0 = blockIdx.x * ppb;
  const int rows = min(ppb, a.M - m0) * a.P * F;
  for (int i = tid; i < rows; i += 256) pfn_in[i] = a.feats[(size_t)m0 * a.P * F + i];
  float w[F];
#pragma unroll
  for (int k = 0; k < F; ++k) w[k] = a.weight[c * F + k];
  const double n = (double)a.M * a.P;
  double md, var;
  pfn_channel_stats<F>(a.moments, w, n, eps, md, var);
  const float mean = (float)md, k = (float)(1.0 / sqrt(var + (double)eps)), g = a.gamma[c], bt = a.beta[c];
  if (blockIdx.x == 0 && pl == 0) {     // one writer per channel: what the backward reads, and nn.BatchNorm1d's running statistics
    a.save[c * 2] = mean;
    a.save[c * 2 + 1] = k;
    if (running_mean != nullptr) {
      const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
    if (nbt != nullptr && c == 0) *nbt += 1;
  }
  __syncthreads();
  const int m = m0 + pl;
  if (m >= a.M) return;
  const float* __restrict__ ip = pfn_in + pl * a.P * F;
  const float sgn = g > 0.f ? 1.f : (g < 0.f ? -1.f : 0.f);   // y is monotone in z with the sign of gamma (rstd > 0)
  float best = -INFINITY, zb = 0.f;
  int bi = 0;
  for (int p = 0; p < a.P; ++p) {
    float z = 0.f;
#pragma unroll
    for (int q = 0; q < F; ++q) z = fmaf(w[q], ip[p * F + q], z);
    const float key = sgn * z;
    if (key > best) { best = key; zb = z; bi = p; }           // strict: the FIRST extreme slot, as torch.max
  }
  out[(size_t)m * a.C + c] = fmaxf(fmaf((zb - mean) * k, g, bt), 0.f);
  arg[(size_t)m * a.C + c] = (unsigned char)bi;
}

// backward, sparse part: per channel sum g, sum g x^, sum g in[arg] over the pillars (f64 atomics into acc [C][F + 2], zeroed by the caller)
template <int F>
__global__ __launch_bounds__(256) void pfn_bwd_kernel(const PfnArgs a, const float* __restrict__ gout, const unsigned char* __restrict__ arg, double* __restrict__ acc) {
  extern __shared__ float pfn_in[];
  __shared__ double s_red[256];
  const int ppb = 256 / a.C, tid = threadIdx.x, pl = tid / a.C, c = tid - pl * a.C;
  float w[F];
#pragma unroll
  for (int k = 0; k < F; ++k) w[k] = a.weight[c * F + k];
  const float mean = a.save[c * 2], k = a.save[c * 2 + 1], g = a.gamma[c], bt = a.beta[c];
  double sb = 0.0, sg = 0.0, sa[F];
#pragma unroll
  for (int q = 0; q < F; ++q) sa[q] = 0.0;
  const int groups = (a.M + ppb - 1) / ppb;
  for (int gi = blockIdx.x; gi < groups; gi += gridDim.x) {
    const int m0 = gi * ppb;
    const int rows = min(ppb, a.M - m0) * a.P * F;
    __syncthreads();
    for (int i = tid; i < rows; i += 256) pfn_in[i] = a.feats[(size_t)m0 * a.P * F + i];
    __syncthreads();
    const int m = m0 + pl;
    if (m < a.M) {
      const float* __restrict__ ip = pfn_in + (pl * a.P + (int)arg[(size_t)m * a.C + c]) * F;
      float z = 0.f;
#pragma unroll
      for (int q = 0; q < F; ++q) z = fmaf(w[q], ip[q], z);
      const float xh = (z - mean) * k;
      const float gg = fmaf(xh, g, bt) > 0.f ? gout[(size_t)m * a.C + c] : 0.f;    // through the ReLU
      sb += (double)gg;
      sg = fma((double)gg, (double)xh, sg);
#pragma unroll
      for (int q = 0; q < F; ++q) sa[q] = fma((double)gg, (double)ip[q], sa[q]);
    }
  }
  // the block's pillar lanes of a channel, then the block's partial STORED at acc[block][c][F + 2] (the finish kernel adds the blocks up in a
  // fixed order: 786 k f64 atomics onto 768 addresses cost 0.2 ms, and their order changed from run to run)
  for (int t = 0; t < F + 2; ++t) {
    __syncthreads();
    s_red[tid] = t == 0 ? sb : (t == 1 ? sg : sa[t - 2 < F ? t - 2 : 0]);
    __syncthreads();
    if (pl == 0) {
      double tot = 0.0;
      for (int j = 0; j < ppb; ++j) tot += s_red[j * a.C + c];
      acc[((size_t)blockIdx.x * a.C + c) * (F + 2) + t] = tot;
    }
  }
}

// backward, closed form: dW [C][F], d gamma, d beta from the sparse sums and the moments (one 64-lane block per channel, f64)
template <int F>
__global__ void pfn_bwd_finish_kernel(const PfnArgs a, const double* __restrict__ part, int nblk, float* __restrict__ dweight, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  // one 64-lane block per channel: lane l adds blocks l, l + 64, ... (a fixed order), then a wave sum; lane 0 forms the results
  const int c = blockIdx.x, l = threadIdx.x;
  double acc[F + 2];
#pragma unroll
  for (int t = 0; t < F + 2; ++t) acc[t] = 0.0;
  for (int b = l; b < nblk; b += 64)
#pragma unroll
    for (int t = 0; t < F + 2; ++t) acc[t] += part[((size_t)b * a.C + c) * (F + 2) + t];
#pragma unroll
  for (int t = 0; t < F + 2; ++t) acc[t] = wave_sum_f64(acc[t]);
  if (l != 0) return;
  const double n = (double)a.M * a.P, mean = (double)a.save[c * 2], r = (double)a.save[c * 2 + 1], gm = (double)a.gamma[c];
  const double sb = acc[0], sg = acc[1];
  const double mg = sb / n, mgx = sg / n;
  if (dbeta != nullptr) dbeta[c] = (float)sb;
  if (dgamma != nullptr) dgamma[c] = (float)sg;
  if (dweight == nullptr) return;
  for (int q = 0; q < F; ++q) {
    double s2w = 0.0;
    for (int b = 0; b < F; ++b) s2w = fma(a.moments[F + q * F + b], (double)a.weight[c * F + b], s2w);
    const double sxh = r * (s2w - mean * a.moments[q]);            // sum_p x^_p in_p[q]
    dweight[c * F + q] = (float)(gm * r * (acc[2 + q] - mg * a.moments[q] - mgx * sxh));
  }
}

inline size_t pfn_moment_doubles(int F) { return (size_t)F + (size_t)F * F; }

template <int F>
inline int pfn_train_fwd_t(const PfnArgs& a, float* out, unsigned char* arg, float* rm, float* rv, long long* nbt, float momentum, float eps, hipStream_t st) {
  GC_HIP(hipMemsetAsync(a.moments, 0, pfn_moment_doubles(F) * sizeof(double), st));
  const long long n = (long long)a.M * a.P;
  pfn_moments_kernel<F><<<(unsigned)std::min<long long>((n + 2047) / 2048, 1024), 256, 0, st>>>(a.feats, a.moments, n);
  const int ppb = 256 / a.C;
  pfn_fwd_kernel<F><<<(a.M + ppb - 1) / ppb, 256, (size_t)ppb * a.P * F * sizeof(float), st>>>(a, out, arg, rm, rv, nbt, momentum, eps);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
template <int F>
inline int pfn_train_bwd_t(const PfnArgs& a, const float* gout, const unsigned char* arg, float* dw, float* dg, float* db, double* acc, hipStream_t st) {
  const int ppb = 256 / a.C, groups = (a.M + ppb - 1) / ppb, nblk = std::min(groups, PFN_BWD_BLOCKS);
  pfn_bwd_kernel<F><<<nblk, 256, (size_t)ppb * a.P * F * sizeof(float), st>>>(a, gout, arg, acc);
  pfn_bwd_finish_kernel<F><<<a.C, 64, 0, st>>>(a, acc, nblk, dw, dg, db);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

}  // namespace gc
