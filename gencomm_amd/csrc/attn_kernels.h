// BEV self-attention block of the UNet (AttnBlock, opencood/models/gencomm_modules/unet.py:141-193):
//   h = GroupNorm(x); q,k,v = 1x1 convs(h); A = softmax_j(q_i . k_j / sqrt(c)); o_i = sum_j A_ij v_j;
//   y = x + proj_out(o).           single head, c = 8 channels, N = H*W tokens.
// The reference materialises the N x N matrix (unusable beyond a few thousand tokens; no shipped
// yaml instantiates the block). Here: flash-style streaming over key tiles with an online softmax,
// so memory is O(N). Two launches: (1) GN + q/k/v projection to token-major [n][N][8];
// (2) attention + proj_out + residual + statistics of y for the next GroupNorm.
#pragma once
#include "unet_kernels.h"

namespace gc {

struct AttnQkvArgs {
  const float* x;       // [n][8][HW]
  const double* sstat;  // [n][8][2]
  const float* gamma; const float* beta;            // norm [8]
  const float* wq; const float* bq; const float* wk; const float* bk; const float* wv; const float* bv;  // [8][8] (oc, ic), [8]
  float* Q; float* K; float* V;                     // [n][HW][8]
  double inv_cnt;
  int HW;
};

__global__ __launch_bounds__(256) void attn_qkv_kernel(const AttnQkvArgs a) {
  __shared__ float s_ab[8][2];
  const int n = blockIdx.y, tid = threadIdx.x;
  if (tid < 8) {
    float A, B;
    gn_coeff(a.sstat + (size_t)n * 16, tid, 2, a.inv_cnt, a.gamma[tid], a.beta[tid], &A, &B);
    s_ab[tid][0] = A; s_ab[tid][1] = B;
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + tid;
  if (i >= a.HW) return;
  float h[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) h[c] = fmaf(s_ab[c][0], a.x[((size_t)n * 8 + c) * a.HW + i], s_ab[c][1]);  // no SiLU in AttnBlock
  const float scale = 0.35355339059327373f;  // 8^-0.5, folded into q (unet.py:178)
  float q[8], k[8], v[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    float sq = as_const(a.bq)[o], sk = as_const(a.bk)[o], sv = as_const(a.bv)[o];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      sq = fmaf(as_const(a.wq)[o * 8 + c], h[c], sq);
      sk = fmaf(as_const(a.wk)[o * 8 + c], h[c], sk);
      sv = fmaf(as_const(a.wv)[o * 8 + c], h[c], sv);
    }
    q[o] = sq * scale; k[o] = sk; v[o] = sv;
  }
  const size_t t = ((size_t)n * a.HW + i) * 8;
  *reinterpret_cast<float4*>(a.Q + t) = make_float4(q[0], q[1], q[2], q[3]);
  *reinterpret_cast<float4*>(a.Q + t + 4) = make_float4(q[4], q[5], q[6], q[7]);
  *reinterpret_cast<float4*>(a.K + t) = make_float4(k[0], k[1], k[2], k[3]);
  *reinterpret_cast<float4*>(a.K + t + 4) = make_float4(k[4], k[5], k[6], k[7]);
  *reinterpret_cast<float4*>(a.V + t) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(a.V + t + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

struct AttnFlashArgs {
  const float* Q; const float* K; const float* V;  // [n][HW][8]
  const float* x;                                   // residual [n][8][HW]
  const float* wp; const float* bp;                 // proj_out [8][8], [8]
  float* y;                                         // [n][8][HW]
  double* dstat;                                    // [n][8][2]
  int HW;
};

// One lane per query; key/value tiles of 256 tokens staged in LDS and read as wave-uniform
// broadcasts; online softmax over chunks of 8 keys (one rescale per chunk).
__global__ __launch_bounds__(256) void attn_flash_kernel(const AttnFlashArgs a) {
  constexpr int KT = 256, CH = 8;
  __shared__ __align__(16) float sk[KT][8];
  __shared__ __align__(16) float sv[KT][8];
  __shared__ float s_red[4][16];
  const int n = blockIdx.y, tid = threadIdx.x;
  const int i = blockIdx.x * 256 + tid;
  const bool ok = i < a.HW;
  const size_t base = (size_t)n * a.HW * 8;
  float q[8];
  {
    const float4 q0 = ok ? *reinterpret_cast<const float4*>(a.Q + base + (size_t)i * 8) : make_float4(0, 0, 0, 0);
    const float4 q1 = ok ? *reinterpret_cast<const float4*>(a.Q + base + (size_t)i * 8 + 4) : make_float4(0, 0, 0, 0);
    q[0] = q0.x; q[1] = q0.y; q[2] = q0.z; q[3] = q0.w; q[4] = q1.x; q[5] = q1.y; q[6] = q1.z; q[7] = q1.w;
  }
  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;

  for (int k0 = 0; k0 < a.HW; k0 += KT) {
    __syncthreads();
    {
      const int j = k0 + tid;
      float4 ka = make_float4(0, 0, 0, 0), kb = ka, va = ka, vb = ka;
      if (j < a.HW) {
        ka = *reinterpret_cast<const float4*>(a.K + base + (size_t)j * 8);
        kb = *reinterpret_cast<const float4*>(a.K + base + (size_t)j * 8 + 4);
        va = *reinterpret_cast<const float4*>(a.V + base + (size_t)j * 8);
        vb = *reinterpret_cast<const float4*>(a.V + base + (size_t)j * 8 + 4);
      }
      *reinterpret_cast<float4*>(&sk[tid][0]) = ka; *reinterpret_cast<float4*>(&sk[tid][4]) = kb;
      *reinterpret_cast<float4*>(&sv[tid][0]) = va; *reinterpret_cast<float4*>(&sv[tid][4]) = vb;
    }
    __syncthreads();
    const int nk = min(KT, a.HW - k0);
    for (int j0 = 0; j0 < nk; j0 += CH) {
      float s[CH];
      float cm = -INFINITY;
#pragma unroll
      for (int jj = 0; jj < CH; ++jj) {
        float d = -INFINITY;
        if (j0 + jj < nk) {
          d = 0.f;
#pragma unroll
          for (int c = 0; c < 8; ++c) d = fmaf(q[c], sk[j0 + jj][c], d);
        }
        s[jj] = d;
        cm = fmaxf(cm, d);
      }
      const float mn = fmaxf(m, cm);
      const float resc = __expf(m - mn);  // exp(-inf) = 0 on the first chunk
      l *= resc;
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] *= resc;
#pragma unroll
      for (int jj = 0; jj < CH; ++jj) {
        const float p = (j0 + jj < nk) ? __expf(s[jj] - mn) : 0.f;
        l += p;
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = fmaf(p, sv[j0 + jj][c], acc[c]);
      }
      m = mn;
    }
  }

  const float rl = 1.0f / l;
  float part[16];
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    float v = as_const(a.bp)[o];
#pragma unroll
    for (int c = 0; c < 8; ++c) v = fmaf(as_const(a.wp)[o * 8 + c], acc[c] * rl, v);
    if (ok) {
      v += a.x[((size_t)n * 8 + o) * a.HW + i];
      a.y[((size_t)n * 8 + o) * a.HW + i] = v;
    }
    part[o] = ok ? v : 0.f;
    part[8 + o] = ok ? v * v : 0.f;
  }
  if (a.dstat != nullptr) block_stats_commit<256>(part, s_red, a.dstat + (size_t)n * 16);
}

}  // namespace gc
