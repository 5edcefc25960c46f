// Device kernels of the Enhancer (gfx950, wave64, fp32).
//
// Reference: opencood/models/gencomm_modules/enhancer.py -- Enhancer.forward :367-383,
// Enhancer_block.forward :346-357 (attention commented out :352), FRFN.forward :222-250,
// SplitAttn.forward :315-333 with RadixSoftmax(radix=1) = sigmoid :287-300.
//
// Token-major (NHWC) intermediates: the two Linear layers are [pixels x C] GEMMs and run on the
// f32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate).
#pragma once
#include "common.h"

namespace gc {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using ef32x4 = __attribute__((ext_vector_type(4))) float;

// ---------------------------------------------------------------------------------------------
// K1: x NCHW -> Y = x + LN1(x), Z = LN2(Y) (token-major), Zc = Z[:, :dc] (compact copy that the
// partial 3x3 conv reads its halo from).  64 pixels x all channels per workgroup, LDS transpose.
// ---------------------------------------------------------------------------------------------
struct EnhLnArgs {
  const float* x;                 // [n][C][HW]
  const float* g1; const float* b1; const float* g2; const float* b2;  // [C]
  float* Y; float* Z; float* Zc;  // [n][HW][C], [n][HW][C], [n][HW][dc]
  int C, dc, HW;
};

__global__ __launch_bounds__(256) void enh_ln_kernel(const EnhLnArgs a) {
  extern __shared__ float smem[];
  const int C = a.C, S = 65;
  float* tile = smem;               // [C][65]
  float* part = smem + (size_t)C * S;  // [4][64]
  float* mu = part + 256;           // [64]
  float* rs = mu + 64;              // [64]
  const int tid = threadIdx.x, q = tid >> 6, p = tid & 63;
  const int n = blockIdx.y, p0 = blockIdx.x * 64;
  const int np = min(64, a.HW - p0);
  const bool pv = p < np;
  const float* __restrict__ xp = a.x + (size_t)n * C * a.HW + p0;
  for (int c = q; c < C; c += 4) tile[c * S + p] = pv ? xp[(size_t)c * a.HW + p] : 0.f;
  __syncthreads();
  const float invC = 1.0f / (float)C;

  auto moments = [&](float eps) {
    float s = 0.f;
    for (int c = q; c < C; c += 4) s += tile[c * S + p];
    part[q * 64 + p] = s;
    __syncthreads();
    const float m = (part[p] + part[64 + p] + part[128 + p] + part[192 + p]) * invC;
    __syncthreads();
    float v = 0.f;
    for (int c = q; c < C; c += 4) { const float d = tile[c * S + p] - m; v = fmaf(d, d, v); }
    part[q * 64 + p] = v;
    __syncthreads();
    const float var = (part[p] + part[64 + p] + part[128 + p] + part[192 + p]) * invC;
    __syncthreads();
    if (q == 0) { mu[p] = m; rs[p] = 1.0f / sqrtf(var + eps); }
    __syncthreads();
  };

  moments(1e-5f);
  {
    const float m = mu[p], r = rs[p];
    for (int c = q; c < C; c += 4) {
      const float v = tile[c * S + p];
      tile[c * S + p] = v + fmaf((v - m) * r, a.g1[c], a.b1[c]);  // x + LN1(x)
    }
  }
  __syncthreads();
  moments(1e-5f);

  float* __restrict__ Yp = a.Y + ((size_t)n * a.HW + p0) * C;
  float* __restrict__ Zp = a.Z + ((size_t)n * a.HW + p0) * C;
  float* __restrict__ Zcp = a.Zc + ((size_t)n * a.HW + p0) * a.dc;
  const int tot = np * C;
  for (int i = tid; i < tot; i += 256) {
    const int pp = i / C, c = i - pp * C;
    const float y = tile[c * S + pp];
    const float z = fmaf((y - mu[pp]) * rs[pp], a.g2[c], a.b2[c]);
    Yp[i] = y;
    Zp[i] = z;
    if (c < a.dc) Zcp[pp * a.dc + c] = z;
  }
}

// The same for C = 64 (the benchmark geometry) without workgroup barriers in the arithmetic: a lane owns ONE pixel and keeps its 64
// channels in registers (64 coalesced 256-byte loads per wave), both LayerNorms are in-lane reductions, and the token-major rows go
// out through a per-wave LDS transpose so that every store instruction writes 1 KB contiguous.  The barrier-per-moment kernel above
// spent 0.86 of its wave cycles parked (profiles/r2_pmc_sq.json).
__global__ __launch_bounds__(256) void enh_ln64_kernel(const EnhLnArgs a) {
  constexpr int C = 64, S = 65;
  __shared__ float tile[4][64 * S];   // per wave: [pixel][channel] padded
  const int tid = threadIdx.x, wv = tid >> 6, p = tid & 63;
  const int n = blockIdx.y, p0 = (blockIdx.x * 4 + wv) * 64;
  if (p0 >= a.HW) return;             // whole wave out of range (no barriers below)
  const int np = min(64, a.HW - p0);
  const bool pv = p < np;
  const float* __restrict__ xp = a.x + (size_t)n * C * a.HW + p0 + p;
  float v[C];
#pragma unroll
  for (int c = 0; c < C; ++c) v[c] = pv ? xp[(size_t)c * a.HW] : 0.f;
  const float invC = 1.0f / (float)C;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) s += v[c];
  float m = s * invC, q = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { const float d = v[c] - m; q = fmaf(d, d, q); }
  float r = 1.0f / sqrtf(q * invC + 1e-5f);
#pragma unroll
  for (int c = 0; c < C; ++c) v[c] = v[c] + fmaf((v[c] - m) * r, as_const(a.g1)[c], as_const(a.b1)[c]);   // x + LN1(x)
  s = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) s += v[c];
  m = s * invC; q = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) { const float d = v[c] - m; q = fmaf(d, d, q); }
  r = 1.0f / sqrtf(q * invC + 1e-5f);
  float* __restrict__ tw = tile[wv];
  float* __restrict__ Yp = a.Y + ((size_t)n * a.HW + p0) * C;
  float* __restrict__ Zp = a.Z + ((size_t)n * a.HW + p0) * C;
  float* __restrict__ Zcp = a.Zc + ((size_t)n * a.HW + p0) * a.dc;
  const int tot = np * C;
  // Y: through the wave's LDS tile (row = pixel), read back flat so that lanes write consecutive words
#pragma unroll
  for (int c = 0; c < C; ++c) tw[p * S + c] = v[c];
  __builtin_amdgcn_wave_barrier();
  for (int i = p; i < tot; i += 64) Yp[i] = tw[(i >> 6) * S + (i & 63)];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int c = 0; c < C; ++c) tw[p * S + c] = fmaf((v[c] - m) * r, as_const(a.g2)[c], as_const(a.b2)[c]);   // LN2
  __builtin_amdgcn_wave_barrier();
  for (int i = p; i < tot; i += 64) Zp[i] = tw[(i >> 6) * S + (i & 63)];
  const int totc = np * a.dc;
  for (int i = p; i < totc; i += 64) { const int pp = i / a.dc, c = i - pp * a.dc; Zcp[i] = tw[pp * S + c]; }
}

// ---------------------------------------------------------------------------------------------
// K2: FRFN.partial_conv3 -- 3x3 conv (no bias) on the first dc = C/4 channels, the rest untouched
// (enhancer.py:232-234).  Reads Zc with halo, overwrites Z[:, :dc] in place.
// Weights pre-transposed to [tap][ic][oc].
// ---------------------------------------------------------------------------------------------
struct EnhPconvArgs {
  const float* Zc;  // [n][H][W][dc]
  const float* wT;  // [9][dc][dcp]  (dcp = dc rounded up to 16, zero padded)
  float* Z;         // [n][H][W][C]
  int C, dc, dcp, H, W;
};

template <int TW, int TH>
__global__ __launch_bounds__(TW * TH) void enh_pconv_kernel(const EnhPconvArgs a) {
  extern __shared__ float smem[];
  const int dc = a.dc, PS = dc + 1;  // padded pixel stride: conflict-free ds_read_b32 across pixels
  constexpr int LW = TW + 2, LH = TH + 2;
  const int tid = threadIdx.x, n = blockIdx.z;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const float* __restrict__ zp = a.Zc + (size_t)n * a.H * a.W * dc;
  for (int i = tid; i < LH * LW * dc; i += TW * TH) {
    const int pix = i / dc, c = i - pix * dc;
    const int r = pix / LW, col = pix - r * LW;
    const int gy = y0 - 1 + r, gx = x0 - 1 + col;
    float v = 0.f;
    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = zp[((size_t)gy * a.W + gx) * dc + c];
    smem[pix * PS + c] = v;
  }
  __syncthreads();
  const int tx = tid % TW, ty = tid / TW;
  const int gy = y0 + ty, gx = x0 + tx;
  const bool ok = gy < a.H && gx < a.W;
  float* __restrict__ op = a.Z + ((size_t)n * a.H * a.W + (size_t)gy * a.W + gx) * a.C;
  for (int ob = 0; ob < dc; ob += 16) {
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = 0.f;
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
      const float* sp = smem + ((ty + dy) * LW + tx + dx) * PS;
      const float* __restrict__ wp = a.wT + (size_t)tap * dc * a.dcp + ob;
#pragma unroll 4
      for (int ic = 0; ic < dc; ++ic) {
        const float v = sp[ic];
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] = fmaf(wp[(size_t)ic * a.dcp + o], v, acc[o]);
      }
    }
    if (ok) {
#pragma unroll
      for (int o = 0; o < 16; ++o) if (ob + o < dc) op[ob + o] = acc[o];
    }
  }
}

// [OC][IC][3][3] -> [9][IC][OCP] zero padded
__global__ void enh_prep_pconv_kernel(const float* __restrict__ w, float* __restrict__ wT, int dc, int dcp) {
  const int total = 9 * dc * dcp;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int tap = i / (dc * dcp), rem = i - tap * dc * dcp, ic = rem / dcp, oc = rem - ic * dcp;
    wT[i] = oc < dc ? w[((size_t)oc * dc + ic) * 9 + tap] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// fp32 GEMM on the matrix cores:  out[m][j] = epi( sum_k A[m][k] * Bw[j][k] + bias[j] )
// A [M][K] row-major (tokens), Bw [N][K] row-major (nn.Linear weight layout), per agent (grid.z).
// Workgroup 256 threads = 4 waves; tile 128 (M) x 64 (N); each wave 32 x 64 = two 32x32
// accumulators; K streamed through LDS in chunks of 32.
//   EPI 0: GELU(erf)                                   (FRFN.linear1, enhancer.py:240)
//   EPI 1: + res[m][j], and accumulate column sums     (FRFN.linear2 + residual :355, GAP :325)
// MFMA operand maps (cdna_hip_programming.md section 3): A[i = lane&31][k = lane>>5],
// B[k = lane>>5][j = lane&31]; D col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// The k order inside a chunk is permuted (lane-half h takes k = 4j+2h, 4j+2h+1) identically for
// A and B so that each lane fetches its two k values with one ds_read_b64.
// ---------------------------------------------------------------------------------------------
struct GemmArgs {
  const float* A; const float* Bw; const float* bias; const float* res;
  float* out; float* colsum;  // colsum [agents][N] (EPI 1)
  int M, N, K;                // per agent
};

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(const GemmArgs a) {
  constexpr int BM = 128, BN = 64, KC = 32, S = KC + 2;
  __shared__ __align__(16) float As[BM * S];
  __shared__ __align__(16) float Bs[BN * S];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
  const int ag = blockIdx.z;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const float* __restrict__ Ap = a.A + (size_t)ag * a.M * a.K;
  const bool kvec = (a.K & 3) == 0;

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

  for (int k0 = 0; k0 < a.K; k0 += KC) {
    __syncthreads();
    {
      const int kq = tid & 7, gk = k0 + 4 * kq;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i, gm = m0 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gm < a.M) {
          const float* src = Ap + (size_t)gm * a.K + gk;
          if (kvec && gk + 3 < a.K) v = *reinterpret_cast<const float4*>(src);
          else {
            if (gk + 0 < a.K) v.x = src[0];
            if (gk + 1 < a.K) v.y = src[1];
            if (gk + 2 < a.K) v.z = src[2];
            if (gk + 3 < a.K) v.w = src[3];
          }
        }
        float* d = &As[row * S + 4 * kq];
        *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
        *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 3) + 32 * i, gn = n0 + row;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gn < a.N) {
          const float* src = a.Bw + (size_t)gn * a.K + gk;
          if (kvec && gk + 3 < a.K) v = *reinterpret_cast<const float4*>(src);
          else {
            if (gk + 0 < a.K) v.x = src[0];
            if (gk + 1 < a.K) v.y = src[1];
            if (gk + 2 < a.K) v.z = src[2];
            if (gk + 3 < a.K) v.w = src[3];
          }
        }
        float* d = &Bs[row * S + 4 * kq];
        *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
        *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KC / 4; ++j) {
      const float2 af = *reinterpret_cast<const float2*>(&As[(32 * w + r) * S + 4 * j + 2 * h]);
      const float2 b0 = *reinterpret_cast<const float2*>(&Bs[r * S + 4 * j + 2 * h]);
      const float2 b1 = *reinterpret_cast<const float2*>(&Bs[(32 + r) * S + 4 * j + 2 * h]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, b0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, b1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, b0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, b1.y, acc1, 0, 0, 0);
    }
  }

  float* __restrict__ op = a.out + (size_t)ag * a.M * a.N;
  const float* __restrict__ rp = EPI == 1 ? a.res + (size_t)ag * a.M * a.N : nullptr;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = n0 + 32 * t + r;
    const bool cok = col < a.N;
    const float b = cok ? a.bias[col] : 0.f;
    float cs = 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = m0 + 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      float v = (t == 0 ? acc0[reg] : acc1[reg]) + b;
      if (cok && m < a.M) {
        if (EPI == 0) v = gelu_erf_f(v);
        else { v += rp[(size_t)m * a.N + col]; cs += v; }
        op[(size_t)m * a.N + col] = v;
      }
    }
    if (EPI == 1) {
      cs += __shfl_xor(cs, 32, 64);
      if (h == 0 && cok) atomicAdd(&a.colsum[(size_t)ag * a.N + col], cs);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same GEMM on the f16 matrix pipe with EXACT THREE-TERM splits of both operands (the arithmetic of conv8h_kernels.h,
// round 3): a token x = hi + lo + t (fp16, fp16, the last bit or two as a power of two -> bf8 after scaling by 2^20), a
// weight w 2^12 = w1 + w2 + w3 (fp16) and its bf8 rounding wb = bf8(w 2^-8).  Per product block five
// v_mfma_f32_32x32x16_f16 (hi w1, lo w1, hi w2, lo w2, hi w3) into the main accumulators and one
// v_mfma_f32_32x32x16_bf8_bf8 (t wb) into accumulators OF ITS OWN, added in the epilogue: matrix instructions of different
// input type never share registers (gfx950 does not forward SrcC between them; hipcc assumes it does --
// tools/probes/mfma_mixed_dep_probe.hip).  Products to 2^-26; fp32 accumulation.  Tokens and weights are split while they are
// staged into LDS; K = 64 costs 48 matrix instructions of 32 cycles per wave instead of 64 fp32 ones of 64, beside -- not in
// front of -- the GELU / residual epilogue's VALU work.
// LDS rows are 32 halves + 8 pad (80 B; 40 B in the bf8 planes): the 16 lanes of a read phase hit distinct banks.
// Operand maps of v_mfma_f32_32x32x16_{f16, bf8_bf8}: A[i = lane & 31][k = 8 * (lane >> 5) .. +7], B[k same][j = lane & 31],
// D as in the fp32 kernel.  K must be a multiple of 4 (it is C or 2C).
// ---------------------------------------------------------------------------------------------
typedef _Float16 eh8_t __attribute__((ext_vector_type(8)));
typedef _Float16 eh2_t __attribute__((ext_vector_type(2)));
typedef float ef2_t __attribute__((ext_vector_type(2)));
constexpr float ENH_TSCALE = 1048576.0f;  // 2^20: third terms are stored as bf8(t 2^20), bf8 weights as bf8(w scale 2^-20)
constexpr float ENH_WS = 4096.0f;         // static weight scale of the Linear layers (2^12; |w| < 16 stays inside fp16)
__device__ __forceinline__ uint32_t enh_bf8x4(float a, float b, float c, float d) {
  int v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
  v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true);
  return (uint32_t)v;
}
// exact three-term split of four values: hi / lo as packed fp16, the third terms (x 2^20) as four bf8 bytes
__device__ __forceinline__ void enh_split4(float4 v, float mul, uint2& hi, uint2& lo, uint32_t& t8) {
  const float x[4] = {v.x * mul, v.y * mul, v.z * mul, v.w * mul};
  const eh2_t h0 = __builtin_convertvector((ef2_t){x[0], x[1]}, eh2_t), h1 = __builtin_convertvector((ef2_t){x[2], x[3]}, eh2_t);
  const float r[4] = {x[0] - (float)h0[0], x[1] - (float)h0[1], x[2] - (float)h1[0], x[3] - (float)h1[1]};
  const eh2_t l0 = __builtin_convertvector((ef2_t){r[0], r[1]}, eh2_t), l1 = __builtin_convertvector((ef2_t){r[2], r[3]}, eh2_t);
  hi = make_uint2(__builtin_bit_cast(uint32_t, h0), __builtin_bit_cast(uint32_t, h1));
  lo = make_uint2(__builtin_bit_cast(uint32_t, l0), __builtin_bit_cast(uint32_t, l1));
  t8 = enh_bf8x4((r[0] - (float)l0[0]) * ENH_TSCALE, (r[1] - (float)l0[1]) * ENH_TSCALE, (r[2] - (float)l1[0]) * ENH_TSCALE,
                 (r[3] - (float)l1[1]) * ENH_TSCALE);
}
// a weight row segment: w1 / w2 / w3 as packed fp16 and the bf8 rounding of w scale 2^-20
__device__ __forceinline__ void enh_split4_w(float4 v, float mul, uint2& w1, uint2& w2, uint2& w3, uint32_t& wb) {
  const float x[4] = {v.x * mul, v.y * mul, v.z * mul, v.w * mul};
  const eh2_t a0 = __builtin_convertvector((ef2_t){x[0], x[1]}, eh2_t), a1 = __builtin_convertvector((ef2_t){x[2], x[3]}, eh2_t);
  const float r[4] = {x[0] - (float)a0[0], x[1] - (float)a0[1], x[2] - (float)a1[0], x[3] - (float)a1[1]};
  const eh2_t b0 = __builtin_convertvector((ef2_t){r[0], r[1]}, eh2_t), b1 = __builtin_convertvector((ef2_t){r[2], r[3]}, eh2_t);
  const eh2_t c0 = __builtin_convertvector((ef2_t){r[0] - (float)b0[0], r[1] - (float)b0[1]}, eh2_t);
  const eh2_t c1 = __builtin_convertvector((ef2_t){r[2] - (float)b1[0], r[3] - (float)b1[1]}, eh2_t);
  w1 = make_uint2(__builtin_bit_cast(uint32_t, a0), __builtin_bit_cast(uint32_t, a1));
  w2 = make_uint2(__builtin_bit_cast(uint32_t, b0), __builtin_bit_cast(uint32_t, b1));
  w3 = make_uint2(__builtin_bit_cast(uint32_t, c0), __builtin_bit_cast(uint32_t, c1));
  const float k = 1.0f / ENH_TSCALE;
  wb = enh_bf8x4(x[0] * k, x[1] * k, x[2] * k, x[3] * k);
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f16s_mfma_kernel(const GemmArgs a) {
  constexpr int BM = 128, BN = 64, KC = 32, RB = 80, RT = 40;  // row bytes of the fp16 / bf8 planes
  constexpr float WS = ENH_WS, WSI = 1.0f / ENH_WS;
  __shared__ __align__(16) unsigned char Ah[BM * RB], Al[BM * RB], At[BM * RT], B1[BN * RB], B2[BN * RB], B3[BN * RB], Bb[BN * RT];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
  const int ag = blockIdx.z;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const float* __restrict__ Ap = a.A + (size_t)ag * a.M * a.K;

  f32x16 acc0, acc1, act0, act1;  // act*: the bf8 third-term products, accumulators of their own (header)
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; act0[i] = 0.f; act1[i] = 0.f; }

  // software pipeline: the next K chunk's tokens and weights travel from global memory into registers while the current chunk
  // is on the matrix cores (the unpipelined loop spent 0.89 of its wave cycles parked: profiles/r2_pmc_sq.json)
  const int kq = tid & 7;
  float4 va[4], vb[2];
  auto fetch = [&](int k0) {
    const int gk = k0 + 4 * kq;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gm = m0 + (tid >> 3) + 32 * i;
      va[i] = (gm < a.M && gk + 3 < a.K) ? *reinterpret_cast<const float4*>(Ap + (size_t)gm * a.K + gk) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int gn = n0 + (tid >> 3) + 32 * i;
      vb[i] = (gn < a.N && gk + 3 < a.K) ? *reinterpret_cast<const float4*>(a.Bw + (size_t)gn * a.K + gk) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < a.K; k0 += KC) {
    __syncthreads();
    {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        uint2 hi, lo;
        uint32_t t8;
        enh_split4(va[i], 1.0f, hi, lo, t8);
        *reinterpret_cast<uint2*>(Ah + row * RB + 8 * kq) = hi;
        *reinterpret_cast<uint2*>(Al + row * RB + 8 * kq) = lo;
        *reinterpret_cast<uint32_t*>(At + row * RT + 4 * kq) = t8;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 3) + 32 * i;
        uint2 w1, w2, w3;
        uint32_t wb;
        enh_split4_w(vb[i], WS, w1, w2, w3, wb);
        *reinterpret_cast<uint2*>(B1 + row * RB + 8 * kq) = w1;
        *reinterpret_cast<uint2*>(B2 + row * RB + 8 * kq) = w2;
        *reinterpret_cast<uint2*>(B3 + row * RB + 8 * kq) = w3;
        *reinterpret_cast<uint32_t*>(Bb + row * RT + 4 * kq) = wb;
      }
    }
    __syncthreads();
    if (k0 + KC < a.K) fetch(k0 + KC);
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      const int ko = 32 * ks + 16 * h;  // byte offset of this lane's 8 halves in the row (half of it in the bf8 planes)
      const eh8_t ah = *reinterpret_cast<const eh8_t*>(Ah + (32 * w + r) * RB + ko);
      const eh8_t al = *reinterpret_cast<const eh8_t*>(Al + (32 * w + r) * RB + ko);
      const long at = *reinterpret_cast<const long*>(At + (32 * w + r) * RT + (ko >> 1));
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x16& acc = t == 0 ? acc0 : acc1;
        f32x16& act = t == 0 ? act0 : act1;
        const int row = 32 * t + r;
        const eh8_t b1 = *reinterpret_cast<const eh8_t*>(B1 + row * RB + ko);
        const eh8_t b2 = *reinterpret_cast<const eh8_t*>(B2 + row * RB + ko);
        const eh8_t b3 = *reinterpret_cast<const eh8_t*>(B3 + row * RB + ko);
        const long bb = *reinterpret_cast<const long*>(Bb + row * RT + (ko >> 1));
        act = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(at, bb, act, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b3, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1, acc, 0, 0, 0);
      }
    }
  }

  float* __restrict__ op = a.out + (size_t)ag * a.M * a.N;
  const float* __restrict__ rp = EPI == 1 ? a.res + (size_t)ag * a.M * a.N : nullptr;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = n0 + 32 * t + r;
    const bool cok = col < a.N;
    const float b = cok ? a.bias[col] : 0.f;
    float cs = 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = m0 + 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      float v = fmaf((t == 0 ? acc0[reg] : acc1[reg]) + (t == 0 ? act0[reg] : act1[reg]), WSI, b);
      if (cok && m < a.M) {
        if (EPI == 0) v = gelu_erf_f(v);
        else { v += rp[(size_t)m * a.N + col]; cs += v; }
        op[(size_t)m * a.N + col] = v;
      }
    }
    if (EPI == 1) {
      cs += __shfl_xor(cs, 32, 64);
      if (h == 0 && cok) atomicAdd(&a.colsum[(size_t)ag * a.N + col], cs);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K2 on the f16 matrix pipe (dc = 16 or 32): the partial 3x3 convolution as an implicit GEMM with the three-term
// arithmetic of the GEMM above.  M = 16 output channels, N = 16 consecutive pixels of a row, K = 32 consecutive
// (tap, input channel) pairs; the tile (32x16 pixels + halo) is staged token-major in fp16 hi / lo planes and a bf8
// third-term plane, so a B operand is one 16-B (8-B) read at (pixel + tap shift, channel offset).
// Table: [oc block][k slice] x {[term 3][lane][4 dwords] fp16, [lane][2 dwords] bf8} = 896 dwords, then 64 floats
// (even 1 / scale, odd scale).
// ---------------------------------------------------------------------------------------------
struct EnhPconvHArgs {
  const float* Zc;   // [n][H][W][dc]
  const float* tab;  // prepared A operands
  float* Z;          // [n][H][W][C]
  int C, dc, H, W, nslice;
};

__global__ __launch_bounds__(256) void enh_prep_pconv_h_kernel(const float* __restrict__ w /*[dc][dc][3][3]*/, float* __restrict__ tab,
                                                              int dc, int nslice) {
  __shared__ float s_max[256];
  const int tid = threadIdx.x;
  float m = 0.f;
  for (int i = tid; i < dc * dc * 9; i += 256) m = fmaxf(m, fabsf(w[i]));
  s_max[tid] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) s_max[tid] = fmaxf(s_max[tid], s_max[tid + s]);
    __syncthreads();
  }
  const float wmax = s_max[0];
  int ex = 0;
  if (wmax > 0.f) (void)frexpf(wmax, &ex);
  const float scale = wmax > 0.f ? ldexpf(1.0f, 14 - ex) : 1.0f;  // largest weight in [2^13, 2^14): conv8h_kernels.h
  const int nob = dc / 16, total = nob * nslice * 896;
  uint32_t* __restrict__ out = reinterpret_cast<uint32_t*>(tab);
  for (int i = blockIdx.x * 256 + tid; i < total; i += gridDim.x * 256) {   // every workgroup derives the same scale
    const int blk = i / 896, q = i - blk * 896, sl = blk % nslice, ob = blk / nslice;
    const int l = q < 768 ? (q >> 2) & 63 : (q - 768) >> 1;
    const int mrow = l & 15, kg = l >> 4, oc = 16 * ob + mrow;
    float wv[8];
    for (int e = 0; e < 8; ++e) {
      const int kidx = 32 * sl + 8 * kg + e, tap = kidx / dc, ic = kidx - tap * dc;
      wv[e] = tap < 9 ? w[((size_t)oc * dc + ic) * 9 + tap] * scale : 0.f;
    }
    if (q < 768) {
      const int d = q & 3, term = q >> 8;
      uint16_t v[2];
      for (int e = 0; e < 2; ++e) {
        const float x = wv[2 * d + e];
        const _Float16 w1 = (_Float16)x;
        const float r1 = x - (float)w1;
        const _Float16 w2 = (_Float16)r1;
        const _Float16 w3 = (_Float16)(r1 - (float)w2);
        v[e] = __builtin_bit_cast(uint16_t, term == 0 ? w1 : term == 1 ? w2 : w3);
      }
      out[i] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
    } else {
      const int d = (q - 768) & 1;
      const float k = 1.0f / ENH_TSCALE;
      out[i] = enh_bf8x4(wv[4 * d] * k, wv[4 * d + 1] * k, wv[4 * d + 2] * k, wv[4 * d + 3] * k);
    }
  }
  if (blockIdx.x == 0 && tid < 64) tab[total + tid] = (tid & 1) ? scale : 1.0f / scale;
}

// TW = 32 (two 16-pixel column groups per wave row) or 16 (one): the tile's LDS image is 18 x (TW + 2) x dc x 5 bytes, so dc = 64
// (C = 256) fits with TW = 16 (104 KB) where TW = 32 would need 196 KB, and small maps get twice the workgroups.
template <int TW>
__global__ __launch_bounds__(256) void enh_pconv_h_kernel(const EnhPconvHArgs a) {
  constexpr int TH = 16, LW = TW + 2, LH = TH + 2, NG = TW / 4;   // NG accumulator groups per wave: 4 rows x TW / 16 column groups
  fp16_ovfl_clamp();  // operands are LayerNorm outputs (bounded by the affine); saturate rather than overflow (common.h)
  extern __shared__ __align__(16) unsigned char pch_smem[];
  const int dc = a.dc, PB = dc * 2, plane = LH * LW * PB, q4 = dc >> 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.z;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const float* __restrict__ zp = a.Zc + (size_t)n * a.H * a.W * dc;
  for (int i = tid; i < LH * LW * q4; i += 256) {
    const int pix = i / q4, q = i - pix * q4;
    const int r = pix / LW, col = pix - r * LW;
    const int gy = y0 - 1 + r, gx = x0 - 1 + col;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = *reinterpret_cast<const float4*>(zp + ((size_t)gy * a.W + gx) * dc + 4 * q);
    uint2 hi, lo;
    uint32_t t8;
    enh_split4(v, 1.0f, hi, lo, t8);
    *reinterpret_cast<uint2*>(pch_smem + pix * PB + 8 * q) = hi;
    *reinterpret_cast<uint2*>(pch_smem + plane + pix * PB + 8 * q) = lo;
    *reinterpret_cast<uint32_t*>(pch_smem + 2 * plane + ((pix * PB + 8 * q) >> 1)) = t8;
  }
  __syncthreads();
  const int ln = lane & 15, kg = lane >> 4;
  const int nob = dc >> 4;
  const float inv_s = a.tab[nob * a.nslice * 896];
  for (int ob = 0; ob < nob; ++ob) {
    ef32x4 acc[NG], act[NG];  // act: the bf8 third-term products, accumulators of their own (see the GEMM above)
#pragma unroll
    for (int g = 0; g < NG; ++g) { acc[g] = ef32x4{0.f, 0.f, 0.f, 0.f}; act[g] = ef32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
    for (int sl = 0; sl < a.nslice; ++sl) {
      const int kidx0 = 32 * sl + 8 * kg;
      const int tap = min(kidx0 / dc, 8), ic0 = kidx0 - (kidx0 / dc) * dc;  // slices past tap 8 carry zero weights
      const int dy = tap / 3, dx = tap - 3 * dy;
      const float* __restrict__ tb = a.tab + (size_t)(ob * a.nslice + sl) * 896;
      eh8_t wa[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) wa[k] = __builtin_bit_cast(eh8_t, *reinterpret_cast<const uint4*>(tb + (k * 64 + lane) * 4));
      const long wb = *reinterpret_cast<const long*>(tb + 768 + lane * 2);
      const int base = ((4 * wave + dy) * LW + ln + dx) * PB + ic0 * 2;
      eh8_t bh[NG], bl[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ad = base + (TW == 32 ? ((g >> 1) * LW + 16 * (g & 1)) : g * LW) * PB;
        bh[g] = *reinterpret_cast<const eh8_t*>(pch_smem + ad);
        bl[g] = *reinterpret_cast<const eh8_t*>(pch_smem + plane + ad);
        const long bt = *reinterpret_cast<const long*>(pch_smem + 2 * plane + (ad >> 1));
        act[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(wb, bt, act[g], 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[2], bh[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[1], bl[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[1], bh[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[0], bl[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[0], bh[g], acc[g], 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] += act[g];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int gy = y0 + 4 * wave + (TW == 32 ? (g >> 1) : g), gx = x0 + (TW == 32 ? 16 * (g & 1) : 0) + ln;
      if (gy < a.H && gx < a.W) {
        float* __restrict__ op = a.Z + ((size_t)n * a.H * a.W + (size_t)gy * a.W + gx) * a.C + 16 * ob + 4 * kg;
        *reinterpret_cast<float4*>(op) = make_float4(acc[g][0] * inv_s, acc[g][1] * inv_s, acc[g][2] * inv_s, acc[g][3] * inv_s);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K3 + K4 fused on the f16 matrix pipe (C = 64): Linear1 + GELU -> depthwise 3x3 + GELU on the first half -> times the
// second half, per 8x8-pixel tile with the Linear1 outputs of its 10x10 halo region RECOMPUTED instead of written to and
// read back from HBM (the hidden tensor is 4C floats per pixel: 2.3 GB written and 2.3 GB read per 16 agents at the
// benchmark geometry; with the GEMM on the f16 pipe 1.56x of it is ~0.1 ms of MFMA time).
//   per workgroup (256 threads): stage the 100 tokens (C channels) as fp16 hi / lo planes; for each of the hid / 16 chunks
//   of 16 gate-branch channels q: M = 32 rows = Linear1 rows {16q..16q+15} (x1) and {hid + 16q ..} (x2), N = 4 blocks of
//   32 region pixels (one per wave), K = C: 4 k-steps x 3 split products of v_mfma_f32_32x32x16_f16; + bias, GELU, zero
//   outside the image (the depthwise conv pads its INPUT with zeros) -> LDS [100][32]; then 64 pixels x 16 channels:
//   dw3x3 + bias, GELU, times x2 at the centre -> G[pixel][16q + ch].
// Table (prep kernel): [chunk][k-step 4] x {[term 3][lane 64][4 dwords] fp16, [lane][2 dwords] bf8} = 896 dwords, weights
// pre-multiplied by 2^12 (three-term arithmetic of the GEMM above).
// ---------------------------------------------------------------------------------------------
struct EnhFrontArgs {
  const float* Z;     // [n][H][W][C] tokens (LayerNorm2 output, partial conv applied)
  const float* tab;   // prepared Linear1 A operands
  const float* b1;    // [2*hid]
  const float* dww;   // [hid][9]
  const float* dwb;   // [hid]
  float* G;           // [n][H][W][hid]   (written when Linear2 is not fused)
  int C, hid, H, W;
  // Linear2 fused (enh_front_h_kernel<true>): out = G W2^T + b2 + res, column sums for the global average pool
  const float* tab2;  // prepared Linear2 B operands (enh_prep_back_kernel)
  const float* b2;    // [C]
  const float* res;   // [n][H][W][C]  x + LayerNorm1(x) (enhancer.py:355)
  float* O;           // [n][H][W][C]  must not alias Z (neighbouring workgroups still read their halos)
  float* colsum;      // [n][C]
  const float* wsc;   // {scale of W1, 1 / it, scale of W2, 1 / it} (enh_wscale_kernel)
};

// Per-tensor power-of-two scales of the two Linear weight matrices for the f16 operand tables: ENH_WS = 2^12 for ordinary weights
// (max |w| < 4: bit-identical to the static scale of round 3), smaller -- the largest power of two that keeps max |w| scale below 2^14 --
// for larger ones: round 3 cast w * 4096 to fp16 unchecked, so a weight of magnitude >= 16 became inf in the table.
// out = {s1, 1 / s1, s2, 1 / s2}; all-zero or non-finite maxima keep ENH_WS.  grid = 2 (one workgroup per tensor).
__global__ __launch_bounds__(256) void enh_wscale_kernel(const float* __restrict__ w1, long long n1, const float* __restrict__ w2, long long n2,
                                                         float* __restrict__ out) {
  __shared__ float s_max[256];
  const float* __restrict__ w = blockIdx.x == 0 ? w1 : w2;
  const long long cnt = blockIdx.x == 0 ? n1 : n2;
  float m = 0.f;
  for (long long i = threadIdx.x; i < cnt; i += 256) m = fmaxf(m, fabsf(w[i]));
  s_max[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) s_max[threadIdx.x] = fmaxf(s_max[threadIdx.x], s_max[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float scale = ENH_WS;
    const float wmax = s_max[0];
    if (wmax > 0.f && wmax < INFINITY) {
      int ex = 0;
      (void)frexpf(wmax, &ex);                       // wmax = f 2^ex, f in [0.5, 1)
      scale = fminf(ENH_WS, ldexpf(1.0f, 14 - ex));  // max |w| scale < 2^14
    }
    out[2 * blockIdx.x] = scale;
    out[2 * blockIdx.x + 1] = 1.0f / scale;
  }
}

__global__ __launch_bounds__(256) void enh_prep_front_kernel(const float* __restrict__ w1 /*[2*hid][C]*/, float* __restrict__ tab, int C, int hid,
                                                             const float* __restrict__ wsc) {
  const float WS1 = wsc[0];
  const int ksteps = C / 16, total = (hid / 16) * ksteps * 896;
  uint32_t* __restrict__ out = reinterpret_cast<uint32_t*>(tab);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int blk = i / 896, qq = i - blk * 896, ks = blk % ksteps, q = blk / ksteps;
    const int l = qq < 768 ? (qq >> 2) & 63 : (qq - 768) >> 1;
    const int m = l & 31, kg = l >> 5;
    const int row = m < 16 ? 16 * q + m : hid + 16 * q + (m - 16);
    float wv[8];
    for (int e = 0; e < 8; ++e) wv[e] = w1[(size_t)row * C + 16 * ks + 8 * kg + e] * WS1;
    if (qq < 768) {
      const int d = qq & 3, term = qq >> 8;
      uint16_t v[2];
      for (int e = 0; e < 2; ++e) {
        const float x = wv[2 * d + e];
        const _Float16 a1 = (_Float16)x;
        const float r1 = x - (float)a1;
        const _Float16 a2 = (_Float16)r1;
        const _Float16 a3 = (_Float16)(r1 - (float)a2);
        v[e] = __builtin_bit_cast(uint16_t, term == 0 ? a1 : term == 1 ? a2 : a3);
      }
      out[i] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
    } else {
      const int d = (qq - 768) & 1;
      const float k = 1.0f / ENH_TSCALE;
      out[i] = enh_bf8x4(wv[4 * d] * k, wv[4 * d + 1] * k, wv[4 * d + 2] * k, wv[4 * d + 3] * k);
    }
  }
}

// Linear2's weights [C][hid] as B operands of v_mfma_f32_32x32x16_*: block (chunk q of 16 hidden channels, output-channel block ob
// of 32) = {[term 3][lane 64][4 dwords] fp16, [lane][2 dwords] bf8} = 896 dwords; lane l holds W2[32 ob + (l & 31)][16 q + 8 (l >> 5) ..+7]
__global__ __launch_bounds__(256) void enh_prep_back_kernel(const float* __restrict__ w2 /*[C][hid]*/, float* __restrict__ tab, int C, int hid,
                                                            const float* __restrict__ wsc) {
  const float WS2 = wsc[2];
  const int nob = C / 32, total = (hid / 16) * nob * 896;
  uint32_t* __restrict__ out = reinterpret_cast<uint32_t*>(tab);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int blk = i / 896, qq = i - blk * 896, ob = blk % nob, q = blk / nob;
    const int l = qq < 768 ? (qq >> 2) & 63 : (qq - 768) >> 1;
    const int oc = 32 * ob + (l & 31), kg = l >> 5;
    float wv[8];
    for (int e = 0; e < 8; ++e) wv[e] = w2[(size_t)oc * hid + 16 * q + 8 * kg + e] * WS2;
    if (qq < 768) {
      const int d = qq & 3, term = qq >> 8;
      uint16_t v[2];
      for (int e = 0; e < 2; ++e) {
        const float x = wv[2 * d + e];
        const _Float16 a1 = (_Float16)x;
        const float r1 = x - (float)a1;
        const _Float16 a2 = (_Float16)r1;
        const _Float16 a3 = (_Float16)(r1 - (float)a2);
        v[e] = __builtin_bit_cast(uint16_t, term == 0 ? a1 : term == 1 ? a2 : a3);
      }
      out[i] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
    } else {
      const int d = (qq - 768) & 1;
      const float k = 1.0f / ENH_TSCALE;
      out[i] = enh_bf8x4(wv[4 * d] * k, wv[4 * d + 1] * k, wv[4 * d + 2] * k, wv[4 * d + 3] * k);
    }
  }
}

// L2 = true: Linear2 + bias + residual + the column sums of the global average pool run here as well.  The gated
// hidden chunk g [64 pixels][16 channels] goes through 5 KB of LDS as fp32, is split into its three terms while it is read as the A
// operand (M = 32 pixels per wave, K = 16 = the chunk), W2's prepared B operands give N = 32 output channels per wave: wave w owns
// pixel block w >> 1, output-channel block w & 1, one 32x32 accumulator for the whole kernel.  The 4C-per-pixel gated tensor
// (1.15 GB per 16 agents written in 64-byte runs, read back by the Linear2 GEMM) and that GEMM's launch disappear.
// The bf8 third-term product shares the accumulator with the f16 products: 16 wait states stand between the two instruction types
// (tools/probes/mfma32_mixed_dep_probe.hip; tests/test_abi.py checks the code object).
template <bool L2>
__global__ __launch_bounds__(256, 3) void enh_front_h_kernel(const EnhFrontArgs a) {
  fp16_ovfl_clamp();
  constexpr int TP = 8, RP = TP + 2, NPX = RP * RP, C = 64, RB = C * 2 + 16, RT = C + 8, HS = 17, GS = 20;  // region 10x10, row bytes, strides
  __shared__ __align__(16) unsigned char zh[NPX * RB], zl[NPX * RB], zt[NPX * RT];  // hi / lo fp16 planes, bf8 third-term plane
  __shared__ float hb1[NPX * HS];       // GELU(x1) on the 10x10 region, 16 channels of the chunk
  __shared__ float hb2[TP * TP * HS];   // GELU(x2) on the tile's own 8x8 pixels (the gate is read at the centre only)
  __shared__ __align__(16) float gbuf[L2 ? TP * TP * GS : 4];  // the gated chunk [pixel][16] (+4 pad: conflict-free 128-bit reads)
  __shared__ float s_b1[4 * C];  // Linear1 bias; the depthwise weights / bias are read from global memory per chunk (a few KB, cache
                                 // hits): with the third plane their 5 KB in LDS would cost the third resident workgroup
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.z;
  const int x0 = blockIdx.x * TP, y0 = blockIdx.y * TP;
  const int hid = a.hid;  // == 2 * C
  const float inv_ws1 = a.wsc[1], inv_ws2 = a.wsc[3];   // 1 / the tables' weight scales (enh_wscale_kernel)
  for (int i = tid; i < 4 * C; i += 256) s_b1[i] = a.b1[i];
  // ---- stage the region's tokens: item = (pixel, float4 of 4 channels): 100 * 16 items
  for (int i = tid; i < NPX * (C / 4); i += 256) {
    const int p = i >> 4, q4 = i & 15;
    const int gy = y0 - 1 + p / RP, gx = x0 - 1 + p % RP;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = *reinterpret_cast<const float4*>(a.Z + ((size_t)n * a.H * a.W + (size_t)gy * a.W + gx) * C + 4 * q4);
    uint2 hi, lo;
    uint32_t t8;
    enh_split4(v, 1.0f, hi, lo, t8);
    *reinterpret_cast<uint2*>(zh + p * RB + 8 * q4) = hi;
    *reinterpret_cast<uint2*>(zl + p * RB + 8 * q4) = lo;
    *reinterpret_cast<uint32_t*>(zt + p * RT + 4 * q4) = t8;
  }
  __syncthreads();
  const int r = lane & 31, h = lane >> 5;
  const int pb = min(32 * wave + r, NPX - 1);  // this lane's region pixel as B column (padding columns repeat the last pixel)
  const int pcol = 32 * wave + r;
  const bool pvalid = pcol < NPX;
  const int pgy = y0 - 1 + pcol / RP, pgx = x0 - 1 + pcol % RP;
  const bool inimg = pvalid && pgy >= 0 && pgy < a.H && pgx >= 0 && pgx < a.W;
  // the gate branch x2 is only read at the tile's own 8x8 pixels: no GELU for it on the halo ring
  const int prow = pcol / RP, pcl = pcol - prow * RP;
  const bool ring = !(pvalid && prow >= 1 && prow <= TP && pcl >= 1 && pcl <= TP);
  const bool centre = inimg && !ring;
  const int cidx = ring ? 0 : (prow - 1) * TP + (pcl - 1);
  // second phase ownership: channel tid % 16, tile row (tid / 16) % 8, pixels 4 * (tid / 128) .. +3 of that row
  const int dch = tid & 15, dpy = (tid >> 4) & 7, dx0 = 4 * (tid >> 7);
  const int nchunk = hid / 16;
  // The A operands of a chunk (4 k-steps x {3 fp16 terms, 1 bf8} = 56 registers) are requested one chunk AHEAD, right behind the
  // matrix phase that consumed the previous set: they arrive during the GELU / depthwise phases.  The table (114 KB) does not
  // fit the 32 KB L1: fetched just before use, every k-step waited for an L2 round trip (2.07 ms per 16-agent launch; 32 waits
  // per workgroup).
  eh8_t wa[C / 16][3];
  long wbq[C / 16];
  auto load_chunk_operands = [&](int q) {
#pragma unroll
    for (int ks = 0; ks < C / 16; ++ks) {
      const float* __restrict__ tb = a.tab + (size_t)(q * (C / 16) + ks) * 896;
#pragma unroll
      for (int k = 0; k < 3; ++k) wa[ks][k] = __builtin_bit_cast(eh8_t, *reinterpret_cast<const uint4*>(tb + (k * 64 + lane) * 4));
      wbq[ks] = *reinterpret_cast<const long*>(tb + 768 + lane * 2);
    }
  };
  load_chunk_operands(0);
  f32x16 acc2;  // Linear2: [32 pixels of block wave >> 1] x [32 output channels of block wave & 1]
#pragma unroll
  for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
#pragma unroll 1
  for (int q = 0; q < nchunk; ++q) {
    // one accumulator for the f16 and the bf8 products: the 32x32x16 forms do not show the stale-accumulator behaviour of the
    // 16x16x32 pair (tools/probes/mfma32_mixed_dep_probe.hip); the bf8 instruction of a k-step goes first, five f16 ones follow
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // this chunk's depthwise weights / bias: requested now, used behind the matrix phase and a barrier
    float wd[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wd[t] = a.dww[(16 * q + dch) * 9 + t];
    const float bd = a.dwb[16 * q + dch];
#pragma unroll
    for (int ks = 0; ks < C / 16; ++ks) {
      const int ko = 32 * ks + 16 * h;
      const eh8_t bh = *reinterpret_cast<const eh8_t*>(zh + pb * RB + ko);
      const eh8_t bl = *reinterpret_cast<const eh8_t*>(zl + pb * RB + ko);
      const long bt = *reinterpret_cast<const long*>(zt + pb * RT + (ko >> 1));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(wbq[ks], bt, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[ks][2], bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[ks][1], bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[ks][1], bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[ks][0], bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[ks][0], bh, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (q + 1 < nchunk) load_chunk_operands(q + 1);
    // !L2: the previous chunk's depthwise phase must have finished reading hb1 / hb2.  L2: the barrier in front of the Linear2 step
    // (below) already separates every wave's depthwise reads of chunk q from these writes of chunk q + 1, and the barrier behind
    // them separates the Linear2 reads of gbuf from the next depthwise phase's writes: two barriers per chunk, not three
    if (!L2 && q > 0) __syncthreads();
    if (pvalid) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = (reg & 3) + 8 * (reg >> 2) + 4 * h;  // row of the chunk: < 16 gate-branch x1, >= 16 x2
        const int row = m < 16 ? 16 * q + m : hid + 16 * q + (m - 16);
        const float v = fmaf(acc[reg], inv_ws1, s_b1[row]);
        if (m < 16) hb1[pcol * HS + m] = inimg ? gelu_erf_f(v) : 0.f;
        else if (!ring) hb2[cidx * HS + (m - 16)] = centre ? gelu_erf_f(v) : 0.f;
      }
    }
    __syncthreads();
    // Linear2's operands of this chunk: requested now, they arrive during the depthwise phase
    eh8_t w2a[3];
    long w2b = 0;
    if (L2) {
      const float* __restrict__ tb = a.tab2 + (size_t)(q * 2 + (wave & 1)) * 896;
#pragma unroll
      for (int k = 0; k < 3; ++k) w2a[k] = __builtin_bit_cast(eh8_t, *reinterpret_cast<const uint4*>(tb + (k * 64 + lane) * 4));
      w2b = *reinterpret_cast<const long*>(tb + 768 + lane * 2);
    }
    {
      const int ch = 16 * q + dch;
      const int gy = y0 + dpy;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = dx0 + j, gx = x0 + x;
        float s = bd;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) s = fmaf(wd[dy * 3 + dx], hb1[((dpy + dy) * RP + x + dx) * HS + dch], s);
        const float g = gelu_erf_f(s) * hb2[(dpy * TP + x) * HS + dch];
        if (L2) gbuf[(dpy * TP + x) * GS + dch] = g;
        else if (gy < a.H && gx < a.W) a.G[((size_t)n * a.H * a.W + (size_t)gy * a.W + gx) * hid + ch] = g;
      }
    }
    if (L2) {
      __syncthreads();
      const float* __restrict__ gp = gbuf + (32 * (wave >> 1) + r) * GS + 8 * h;
      uint2 h0, l0, h1, l1;
      uint32_t t0, t1;
      enh_split4(*reinterpret_cast<const float4*>(gp), 1.0f, h0, l0, t0);
      enh_split4(*reinterpret_cast<const float4*>(gp + 4), 1.0f, h1, l1, t1);
      const eh8_t ah = __builtin_bit_cast(eh8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
      const eh8_t al = __builtin_bit_cast(eh8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
      const long at = (long)(((unsigned long)t1 << 32) | (unsigned long)t0);
      __builtin_amdgcn_sched_barrier(0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(at, w2b, acc2, 0, 0, 0);
      // 24 wait states between matrix instructions of different input type on one accumulator (tied to it: nothing moves across)
      asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc2));
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w2a[2], acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, w2a[1], acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w2a[1], acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, w2a[0], acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w2a[0], acc2, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (L2) {  // out[pixel][oc] = acc2 / scale + b2 + res; column sums over the tile's pixels -> one atomic per (workgroup half, oc)
    const int oc = 32 * (wave & 1) + r;
    const float b = a.b2[oc];
    float cs = 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int p = 32 * (wave >> 1) + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      const int gy = y0 + (p >> 3), gx = x0 + (p & 7);
      if (gy < a.H && gx < a.W) {
        const size_t o = ((size_t)n * a.H * a.W + (size_t)gy * a.W + gx) * C + oc;
        const float v = fmaf(acc2[reg], inv_ws2, b) + a.res[o];
        a.O[o] = v;
        cs += v;
      }
    }
    cs += __shfl_xor(cs, 32, 64);
    if (h == 0) atomicAdd(&a.colsum[(size_t)n * C + oc], cs);
  }
}

// ---------------------------------------------------------------------------------------------
// K4: depthwise 3x3 (+bias) + GELU on the first 2C hidden channels, times the other 2C
// (enhancer.py:241-246).  Lanes run over channels (contiguous in NHWC); each thread slides a
// 3x3 register window along a strip of SL pixels.
// ---------------------------------------------------------------------------------------------
struct EnhDwArgs {
  const float* Hd;  // [n][H][W][2*hid]
  const float* w;   // [hid][9]
  const float* b;   // [hid]
  float* G;         // [n][H][W][hid]
  int hid, H, W;
};

template <int SL>
__global__ __launch_bounds__(256) void enh_dwgate_kernel(const EnhDwArgs a) {
  const int n = blockIdx.y;
  const int nstrip_x = (a.W + SL - 1) / SL;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)a.H * nstrip_x * a.hid;
  if (idx >= total) return;
  const int j = (int)(idx % a.hid);
  const int s = (int)(idx / a.hid);
  const int y = s / nstrip_x, xs = (s - y * nstrip_x) * SL;
  const int ld = 2 * a.hid;
  const float* __restrict__ hp = a.Hd + (size_t)n * a.H * a.W * ld;
  float wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = a.w[j * 9 + t];
  const float bias = a.b[j];
  auto ld3 = [&](int x, float (&col)[3]) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = y - 1 + dy;
      col[dy] = (yy >= 0 && yy < a.H && x >= 0 && x < a.W) ? hp[((size_t)yy * a.W + x) * ld + j] : 0.f;
    }
  };
  float c0[3], c1[3], c2[3];
  ld3(xs - 1, c0);
  ld3(xs, c1);
#pragma unroll
  for (int i = 0; i < SL; ++i) {
    const int x = xs + i;
    if (x >= a.W) break;
    ld3(x + 1, c2);
    float d = bias;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) d = fmaf(wv[dy * 3 + 0], c0[dy], fmaf(wv[dy * 3 + 1], c1[dy], fmaf(wv[dy * 3 + 2], c2[dy], d)));
    const float gate = hp[((size_t)y * a.W + x) * ld + a.hid + j];
    a.G[((size_t)n * a.H * a.W + (size_t)y * a.W + x) * a.hid + j] = gelu_erf_f(d) * gate;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) { c0[dy] = c1[dy]; c1[dy] = c2[dy]; }
  }
}

// ---------------------------------------------------------------------------------------------
// K6: SplitAttn gate per agent: mean over H,W -> fc1 -> LayerNorm -> ReLU -> fc2 -> sigmoid.
// ---------------------------------------------------------------------------------------------
struct EnhGateArgs {
  const float* colsum;  // [n][C] sums over pixels
  const float* fc1; const float* lnw; const float* lnb; const float* fc2;  // [C][C],[C],[C],[C][C]
  float* gate;          // [n][C]
  int C; float inv_hw;
};

__global__ __launch_bounds__(256) void enh_gate_kernel(const EnhGateArgs a) {
  extern __shared__ float smem[];
  const int C = a.C, tid = threadIdx.x, n = blockIdx.x;
  float* g = smem;       // [C]
  float* t1 = smem + C;  // [C]
  float* red = smem + 2 * C;  // [8]
  for (int c = tid; c < C; c += 256) g[c] = a.colsum[(size_t)n * C + c] * a.inv_hw;
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float s = 0.f;
    for (int k = 0; k < C; ++k) s = fmaf(a.fc1[(size_t)c * C + k], g[k], s);
    t1[c] = s;
  }
  __syncthreads();
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const float r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
  };
  float s = 0.f;
  for (int c = tid; c < C; c += 256) s += t1[c];
  const float mean = block_sum(s) / (float)C;
  float v = 0.f;
  for (int c = tid; c < C; c += 256) { const float d = t1[c] - mean; v = fmaf(d, d, v); }
  const float rstd = 1.0f / sqrtf(block_sum(v) / (float)C + 1e-5f);
  for (int c = tid; c < C; c += 256) g[c] = fmaxf(fmaf((t1[c] - mean) * rstd, a.lnw[c], a.lnb[c]), 0.f);
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float s2 = 0.f;
    for (int k = 0; k < C; ++k) s2 = fmaf(a.fc2[(size_t)c * C + k], g[k], s2);
    a.gate[(size_t)n * C + c] = 1.0f / (1.0f + expf(-s2));
  }
}

// ---------------------------------------------------------------------------------------------
// K7: out[n][c][p] = O[n][p][c] * gate[n][c]   (channel gate + NHWC -> NCHW, enhancer.py:332, :380)
// ---------------------------------------------------------------------------------------------
struct EnhOutArgs {
  const float* O; const float* gate; float* out; int C, HW;
};
__global__ __launch_bounds__(256) void enh_scale_transpose_kernel(const EnhOutArgs a) {
  extern __shared__ float smem[];  // [32][C+1]
  const int C = a.C, S = C + 1, tid = threadIdx.x, n = blockIdx.y, p0 = blockIdx.x * 32;
  const int np = min(32, a.HW - p0);
  const float* __restrict__ ip = a.O + ((size_t)n * a.HW + p0) * C;
  for (int i = tid; i < np * C; i += 256) { const int pp = i / C, c = i - pp * C; smem[pp * S + c] = ip[i]; }
  __syncthreads();
  const int p = tid & 31;
  for (int c = tid >> 5; c < C; c += 8)
    if (p < np) a.out[((size_t)n * C + c) * a.HW + p0 + p] = smem[p * S + c] * a.gate[(size_t)n * C + c];
}

}  // namespace gc
