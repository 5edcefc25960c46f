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

// The same attention on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products and sums, like the lane-per-query form
// above), flash style.  Workgroup = 64 queries x 4 key quarters = 8 waves (lanes l and l + 32 share a query); key / value tiles of 256 tokens in LDS ([key][9] and [key][33]: odd
// strides, conflict-free operand reads; value columns 8 .. 31 stay zero so that V^T fills the 32 MFMA rows).  Per 32-key block:
//   S^T[key][query] = K Q^T    4 MFMAs (8 channels = 4 k-pairs), A = K rows from LDS, B = Q from registers; a lane then owns ONE query
//                              column and 16 of the 32 keys (the rest in lane ^ 32): softmax statistics are register-local + one exchange;
//   O^T[d][query] += V^T P     16 MFMAs, A = V rows from LDS in the key order the P registers already have, B = P from the registers.
// 2 agents x 2 048 tokens: 250 -> see DESIGN section 4 (the lane-per-query form keeps running maps below 128 tokens).
using f32x16a = __attribute__((ext_vector_type(16))) float;
__global__ __launch_bounds__(512) void attn_flash_mfma_kernel(const AttnFlashArgs a) {
  constexpr int KT = 256;
  __shared__ float sk[KT][9];
  __shared__ float sv[KT][33];
  __shared__ float s_partial[4][64][10];   // [key quarter][query of the workgroup][running max, denominator, 8 channels]
  __shared__ float s_red[8][16];
  // 8 waves: wave = (query block qb of 32, key quarter kq): the four waves of a query block each walk two of the eight 32-key
  // blocks of every staged tile and their (max, denominator, O) triples meet in LDS -- a quarter of the dependent
  // MFMA -> softmax -> MFMA chain per wave, four times the waves on the machine
  const int n = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, l = tid & 63, qi = l & 31, h = l >> 5;
  const int qb = wave & 1, kq = wave >> 1;
  const size_t base = (size_t)n * a.HW * 8;
  const int ql = qb * 32 + qi, i = blockIdx.x * 64 + ql;   // this lane's query (lanes l and l + 32 share it)
  const bool ok = i < a.HW;
  for (int j = tid; j < KT * 33; j += 512) (&sv[0][0])[j] = 0.f;   // columns 8 .. 31 are never written again
  float qv[4];
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) qv[s_] = ok ? a.Q[base + (size_t)i * 8 + 2 * s_ + h] : 0.f;
  f32x16a o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float mrun = -INFINITY, den = 0.f;

  for (int k0 = 0; k0 < a.HW; k0 += KT) {
    __syncthreads();   // the previous tile is no longer read (first pass: the zero fill is complete)
    if (tid < KT) {
      const int j = tid;
      float4 ka = make_float4(0, 0, 0, 0), kb = ka, va = ka, vb = ka;
      if (k0 + j < a.HW) {
        ka = *reinterpret_cast<const float4*>(a.K + base + (size_t)(k0 + j) * 8);
        kb = *reinterpret_cast<const float4*>(a.K + base + (size_t)(k0 + j) * 8 + 4);
        va = *reinterpret_cast<const float4*>(a.V + base + (size_t)(k0 + j) * 8);
        vb = *reinterpret_cast<const float4*>(a.V + base + (size_t)(k0 + j) * 8 + 4);
      }
      sk[j][0] = ka.x; sk[j][1] = ka.y; sk[j][2] = ka.z; sk[j][3] = ka.w; sk[j][4] = kb.x; sk[j][5] = kb.y; sk[j][6] = kb.z; sk[j][7] = kb.w;
      sv[j][0] = va.x; sv[j][1] = va.y; sv[j][2] = va.z; sv[j][3] = va.w; sv[j][4] = vb.x; sv[j][5] = vb.y; sv[j][6] = vb.z; sv[j][7] = vb.w;
    }
    __syncthreads();
    const int nkb = (min(KT, a.HW - k0) + 31) / 32;
    for (int kb_ = 2 * kq; kb_ < min(nkb, 2 * kq + 2); ++kb_) {
      f32x16a sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(sk[32 * kb_ + qi][2 * s_ + h], qv[s_], sc, 0, 0, 0);
      float bm = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + 32 * kb_ + 8 * (r >> 2) + 4 * h + (r & 3);
        sc[r] = key < a.HW ? sc[r] : -INFINITY;
        bm = fmaxf(bm, sc[r]);
      }
      bm = fmaxf(bm, __shfl_xor(bm, 32, 64));   // every 32-key block holds at least one valid key: bm is finite
      const float nm = fmaxf(mrun, bm), corr = __expf(mrun - nm);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = __expf(sc[r] - nm); ps += sc[r]; }
      den = den * corr + ps;
      mrun = nm;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] *= corr;
#pragma unroll
      for (int t = 0; t < 16; ++t)
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[32 * kb_ + 8 * (t >> 2) + 4 * h + (t & 3)][qi], sc[t], o, 0, 0, 0);
    }
  }
  den += __shfl_xor(den, 32, 64);
  // O^T rows 0 .. 7 are the channels: registers 0 .. 3 of a lane hold channels 4 h .. 4 h + 3 of its query
  if (h == 0) { s_partial[kq][ql][0] = mrun; s_partial[kq][ql][1] = den; }
#pragma unroll
  for (int c = 0; c < 4; ++c) s_partial[kq][ql][2 + 4 * h + c] = o[c];
  __syncthreads();
  float part[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) part[j] = 0.f;
  if (kq == 0) {   // the query block's first wave merges the four key quarters and finishes channels 4 h .. 4 h + 3
    float m = s_partial[0][ql][0];
#pragma unroll
    for (int j = 1; j < 4; ++j) m = fmaxf(m, s_partial[j][ql][0]);
    float lsum = 0.f, acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float wgt = __expf(s_partial[j][ql][0] - m);   // a quarter without keys: exp(-inf) = 0
      lsum = fmaf(s_partial[j][ql][1], wgt, lsum);
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = fmaf(s_partial[j][ql][2 + c], wgt, acc[c]);
    }
    const float rl = 1.0f / lsum;
#pragma unroll
    for (int oo = 0; oo < 4; ++oo) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if (hh != h) continue;
        const int ch = 4 * hh + oo;
        float v = as_const(a.bp)[ch];
#pragma unroll
        for (int c = 0; c < 8; ++c) v = fmaf(as_const(a.wp)[ch * 8 + c], acc[c] * rl, v);
        if (ok) {
          v += a.x[((size_t)n * 8 + ch) * a.HW + i];
          a.y[((size_t)n * 8 + ch) * a.HW + i] = v;
          part[ch] = v;
          part[8 + ch] = v * v;
        }
      }
    }
  }
  if (a.dstat != nullptr) block_stats_commit<512>(part, s_red, a.dstat + (size_t)n * 16);
}

}  // namespace gc
