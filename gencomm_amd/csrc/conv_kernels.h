// General 2-D convolution for the layers AROUND the hot path (gfx950, wave64): BaseBEVBackbone blocks/deblocks, DownsampleConv,
// the detection heads, the Linear layers of the fusion transformers.  Kernels behind conv2d_enqueue:
//   conv2d_h3l_kernel     (conv_h3_kernels.h) DEFAULT for 3x3 stride 1 / 2x2 with Cin >= 16, Cin % 8 == 0, >= 32 GEMM rows: f16 pipe, six
//   conv2d_h3_kernel       matrix instructions per product block from exact three-term operand splits (2^-26 products); _h3_: 3x3 stride 2
//   conv2d_igemm_kernel   exact fp32 on v_mfma_f32_32x32x2_f32 -- every other shape (1x1 / Linear / transposed convolutions among them); the only one in GENCOMM_MODE_ARITH = 1
//   conv3x3_f16s_kernel   GENCOMM_MODE_ARITH = 3 (opt-in): 3x3 stride 1 / 2 with Cin % 8 == 0 on the f16 pipe from two-term splits (22-bit products)
//   conv1x1_f16s_kernel   the same for 1x1 (and ConvTranspose2d, kernel == stride) with >= 128 GEMM rows
//
// Reference call sites: opencood/models/sub_modules/base_bev_backbone.py:40-92 (ZeroPad2d(1) + 3x3
// stride-s conv, BatchNorm2d(eps 1e-3), ReLU; ConvTranspose2d(k = stride) deblocks),
// opencood/models/sub_modules/downsample_conv.py:17-24 (conv + ReLU twice),
// opencood/models/heter_model_baseline_w_gencomm_stage1.py:137-142 (1x1 heads).
//
// Implicit GEMM  out[co][pixel] = sum_k W[co][k] * im2col[k][pixel],  k = (ci, ky, kx), on
// v_mfma_f32_32x32x2_f32 with the WEIGHTS as the A operand (rows = output channels) and the PIXELS
// as the B operand (columns), so that for a fixed accumulator register the 32 lanes of a half-wave
// hold 32 consecutive pixels of one output channel: NCHW stores are 128-byte runs.
// Workgroup = 4 waves = 64 output channels x (4 rows x 16 columns) pixels; K is walked in chunks of
// CC input channels x KH x KW taps staged through LDS (input patch + weight slab), channel pairs
// interleaved so that the two k-lanes of the MFMA (lane / 32) read adjacent words.
// Epilogue: y = acc * scale[co] + shift[co] (folded BatchNorm or bias), optional ReLU, optional
// pixel-shuffle store for ConvTranspose2d with kernel == stride (a 1x1 conv to Cout*s*s channels).
#pragma once
#include "common.h"

namespace gc {

using f32x16c = __attribute__((ext_vector_type(16))) float;

struct Conv2dArgs {
  const float* x;      // [N][Cin][H][W]
  const float* w;      // prepared [Cin*KH*KW][CoutP]  (k-major; CoutP = GEMM rows = Cout * ups^2)
  const float* scale;  // [Cout]
  const float* shift;  // [Cout]
  float* y;            // [N][out_ctotal][Ho*ups][Wo*ups], this layer writes channels [out_coff, out_coff + Cout)
  int Cin, H, W, CoutP, Ho, Wo;
  int stride, pad, relu /* 0 none, 1 ReLU, 2 erf-GELU, 3 ReLU after the residual */, ups, out_ctotal, out_coff;
  const float* res = nullptr;  // optional residual, same layout as y, added after the activation
  // three-term operand form of the weights + per-row unscale (conv_h3_kernels.h), set by the ABI entries when the prepared buffer carries it
  const unsigned char* w3 = nullptr;
  const float* wsc = nullptr;
};

// TY = 8 (round 4): a wave owns TWO 32 x 32 accumulators (the same 32 output channels on two groups of four output rows), so the
// weight slab of a chunk -- 18 of the 22 floats a thread stages per 3 x 3 chunk -- serves twice the matrix work.
template <int KH, int KW, int CC, int STRIDE, int TY = 4>
__global__ __launch_bounds__(256) void conv2d_igemm_kernel(const Conv2dArgs a) {
  constexpr int TX = 16, KHW = KH * KW, NA = TY / 4;
  static_assert(TY == 4 || TY == 8, "row groups of four");
  constexpr int PH = (TY - 1) * STRIDE + KH, PW = (TX - 1) * STRIDE + KW;
  constexpr int PS = PH * PW;            // patch plane (one channel)
  static_assert(CC % 2 == 0, "channel pairs");
  __shared__ float Ws[KHW * (CC / 2) * 64 * 2];  // [tap][cpair][co][2]
  __shared__ float Ps[(CC / 2) * PS * 2];        // [cpair][y][x][2]

  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
  const int tiles_x = (a.Wo + TX - 1) / TX;
  const int ty0 = (blockIdx.x / tiles_x) * TY, tx0 = (blockIdx.x % tiles_x) * TX;
  const int co0 = blockIdx.y * 64, n = blockIdx.z;
  const int iy0 = ty0 * STRIDE - a.pad, ix0 = tx0 * STRIDE - a.pad;
  const size_t plane = (size_t)a.H * a.W;
  const float* __restrict__ xn = a.x + (size_t)n * a.Cin * plane;

  // lane bases: A = weights (row = co), B = pixels (column)
  const int a_base = (32 * (wv & 1) + r) * 2 + h;
  const int pyl = 2 * (wv >> 1) + (r >> 4), pxl = r & 15;
  const int b_base = ((pyl * STRIDE) * PW + pxl * STRIDE) * 2 + h;

  constexpr int G_STRIDE = 4 * STRIDE * PW * 2;   // words between the operand bases of two row groups
  f32x16c acc[NA];
#pragma unroll
  for (int g = 0; g < NA; ++g)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;

  // Software pipeline: chunk c0 + CC travels from global memory into registers while chunk c0 is on the
  // matrix cores; it is written to LDS after the MFMAs.
  constexpr int NP = (CC * PS + 255) / 256, NW = (KHW * CC * 64 + 255) / 256;
  float rp[NP], rw[NW];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int i = tid + 256 * j;
      const int c = i / PS, rem = i - c * PS, py = rem / PW, px = rem - py * PW;
      const int gy = iy0 + py, gx = ix0 + px, gc_ = c0 + c;
      rp[j] = 0.f;
      if (i < CC * PS && gc_ < a.Cin && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) rp[j] = xn[(size_t)gc_ * plane + (size_t)gy * a.W + gx];
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int i = tid + 256 * j;
      const int co = i & 63, kc = i >> 6, c = kc / KHW, tap = kc - c * KHW;
      const int gc_ = c0 + c, gco = co0 + co;
      rw[j] = 0.f;
      if (i < KHW * CC * 64 && gc_ < a.Cin && gco < a.CoutP) rw[j] = a.w[((size_t)gc_ * KHW + tap) * a.CoutP + gco];
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int i = tid + 256 * j;
      const int c = i / PS, rem = i - c * PS;
      if (i < CC * PS) Ps[((c >> 1) * PS + rem) * 2 + (c & 1)] = rp[j];
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int i = tid + 256 * j;
      const int co = i & 63, kc = i >> 6, c = kc / KHW, tap = kc - c * KHW;
      if (i < KHW * CC * 64) Ws[((tap * (CC / 2) + (c >> 1)) * 64 + co) * 2 + (c & 1)] = rw[j];
    }
  };

  fetch(0);
  for (int c0 = 0; c0 < a.Cin; c0 += CC) {
    __syncthreads();  // every wave is done with the previous chunk in LDS
    commit();
    __syncthreads();
    if (c0 + CC < a.Cin) fetch(c0 + CC);
#pragma unroll
    for (int tap = 0; tap < KHW; ++tap) {
      const int ky = tap / KW, kx = tap - ky * KW;
#pragma unroll
      for (int cp = 0; cp < CC / 2; ++cp) {
        const float av = Ws[(tap * (CC / 2) + cp) * 128 + a_base];
#pragma unroll
        for (int g = 0; g < NA; ++g) {
          const float bv = Ps[(cp * PS + ky * PW + kx) * 2 + b_base + g * G_STRIDE];
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[g], 0, 0, 0);
        }
      }
    }
  }

  // epilogue: lane = pixel (4 g + pyl, pxl), register = output channel row
  const int s = a.ups, s2 = s * s;
  const size_t oplane = (size_t)a.Ho * s * a.Wo * s;
  float* __restrict__ yn = a.y + ((size_t)n * a.out_ctotal + a.out_coff) * oplane;
#pragma unroll
  for (int g = 0; g < NA; ++g) {
    const int oy = ty0 + 4 * g + pyl, ox = tx0 + pxl;
    if (oy >= a.Ho || ox >= a.Wo) continue;
    if (s == 1) {  // plain convolution: no sub-pixel arithmetic (integer divisions by a run-time value, 16 times per lane)
      const size_t po = (size_t)oy * a.Wo + ox;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int co = co0 + 32 * (wv & 1) + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (co >= a.CoutP) continue;
        float v = fmaf(acc[g][reg], a.scale[co], a.shift[co]);
        if (a.relu == 1) v = fmaxf(v, 0.f);
        else if (a.relu == 2) v = gelu_erf_f(v);
        const size_t oi = (size_t)co * oplane + po;
        if (a.res != nullptr) v += a.res[((size_t)n * a.out_ctotal + a.out_coff) * oplane + oi];
        if (a.relu == 3) v = fmaxf(v, 0.f);
        yn[oi] = v;
      }
      continue;
    }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int gco = co0 + 32 * (wv & 1) + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (gco >= a.CoutP) continue;
      const int co = gco / s2, sub = gco - co * s2, dy = sub / s, dx = sub - dy * s;
      float v = fmaf(acc[g][reg], a.scale[co], a.shift[co]);
      if (a.relu == 1) v = fmaxf(v, 0.f);
      else if (a.relu == 2) v = gelu_erf_f(v);
      const size_t oi = (size_t)co * oplane + (size_t)(oy * s + dy) * (a.Wo * s) + (ox * s + dx);
      if (a.res != nullptr) v += a.res[((size_t)n * a.out_ctotal + a.out_coff) * oplane + oi];
      if (a.relu == 3) v = fmaxf(v, 0.f);   // ReLU AFTER the residual add (ResNet BasicBlock)
      yn[oi] = v;
    }
  }
}

// BatchNorm (eval) folded to scale/shift on the device: scale = g / sqrt(var + eps), shift = b - mean * scale
// (+ conv bias * scale when the conv has one).  No BN: scale = 1, shift = bias (or 0).
struct FoldArgs {
  const float* gamma; const float* beta; const float* mean; const float* var;  // null => no BN
  const float* bias;  // may be null
  float* scale; float* shift;
  float eps;
  int C;
};
__global__ void conv_fold_kernel(const FoldArgs a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.C) return;
  float sc = 1.f, sh = 0.f;
  if (a.gamma != nullptr) {
    sc = a.gamma[c] / sqrtf(a.var[c] + a.eps);
    sh = a.beta[c] - a.mean[c] * sc;
  }
  if (a.bias != nullptr) sh = fmaf(a.bias[c], sc, sh);
  a.scale[c] = sc;
  a.shift[c] = sh;
}

// OIHW (Conv2d) or IOHW (ConvTranspose2d, kernel == stride) -> prepared [k][CoutP]
struct PrepWArgs {
  const float* w; float* out;
  int Cin, Cout, KH, KW, transposed;
};
__global__ void conv_prep_w_kernel(const PrepWArgs a) {
  const int khw = a.KH * a.KW;
  const long long total = (long long)a.Cin * a.Cout * khw;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  if (!a.transposed) {
    // out[(ci*KHW + tap)][co] = w[co][ci][tap]
    const int co = (int)(i % a.Cout);
    const long long k = i / a.Cout;
    const int ci = (int)(k / khw), tap = (int)(k - (long long)ci * khw);
    a.out[i] = a.w[((size_t)co * a.Cin + ci) * khw + tap];
  } else if (a.transposed == 1) {
    // 1x1 GEMM with CoutP = Cout*khw rows: out[ci][co*khw + tap] = w[ci][co][tap]  (already that order)
    a.out[i] = a.w[i];
  } else {
    // transposed == 2, the INPUT-GRADIENT convolution of a stride-1 layer with forward weights w[Cout][Cin][KH][KW] (here "Cin" / "Cout" are
    // the gradient convolution's own: its input channels = the forward's outputs): out[(g*KHW + tap)][c] = w[g][c][KHW - 1 - tap] with
    // g over a.Cin (forward outputs) and c over a.Cout (forward inputs) -- the flip + transpose + contiguous + prepare of four launches in one
    const int c = (int)(i % a.Cout);
    const long long k = i / a.Cout;
    const int g = (int)(k / khw), tap = (int)(k - (long long)g * khw);
    a.out[i] = a.w[((size_t)g * a.Cout + c) * khw + (khw - 1 - tap)];
  }
}

// ---------------------------------------------------------------------------------------------
// 1x1 stride-1 convolution (the Linear layers of the fusion transformers, message-extractor / head projections, the
// training helpers) on the f16 matrix pipe with the exact hi/lo split arithmetic of conv8h_kernels.h / enhancer_kernels.h:
// three v_mfma_f32_32x32x16_f16 per product block (hi*hi, hi*lo, lo*hi), fp32 accumulation, 22-bit products.
//   out[co][p] = act(scale[co] * sum_ci W[co][ci] x[ci][p] + shift[co]) (+ res)
// GEMM rows M = output channels (A = weights), columns N = pixels (B = x): for a fixed accumulator register the 32 lanes of a
// half-wave hold 32 consecutive pixels of one channel, so NCHW stores are 128-byte runs.  Both operands are K-major in
// memory (x is [ci][p], the prepared weights [ci][co]) while the MFMA wants 8 consecutive k per lane: the loader gives
// every thread 8 consecutive k of ONE column (dword loads, coalesced across the column index), splits them and writes one
// 16-byte record per plane -- the transposition costs nothing.  Workgroup = 128 channels x 64 pixels, 4 waves (32 channels
// each), K in chunks of 32 with the next chunk in flight during the MFMAs.
// Range: the weights are pre-multiplied by 2^6 (|w| < 1e3 assumed, as in the Enhancer GEMM); the activations get a RUNNING
// power-of-two scale: every chunk's max|x| is reduced over the workgroup, the scale puts the largest |x| seen so far into
// [2^13, 2^14) -- up for small inputs such as gradients, down for large ones -- and when it has to drop the accumulators are
// rescaled (exact).  Any finite fp32 input gives finite, fp32-grade results relative to the tile's largest input.
// ---------------------------------------------------------------------------------------------
typedef _Float16 c1h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 c1h2_t __attribute__((ext_vector_type(2)));
typedef float c1f2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void c1_split8(const float (&v)[8], float mul, uint4& hi, uint4& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a0 = v[2 * i] * mul, a1 = v[2 * i + 1] * mul;
    const c1h2_t hh = __builtin_convertvector((c1f2_t){a0, a1}, c1h2_t);
    const c1h2_t ll = __builtin_convertvector((c1f2_t){a0 - (float)hh[0], a1 - (float)hh[1]}, c1h2_t);
    h[i] = __builtin_bit_cast(uint32_t, hh);
    l[i] = __builtin_bit_cast(uint32_t, ll);
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}

template <int BN>  // pixels per workgroup: 64, or 32 when the launch would otherwise have too few workgroups to hide its load latency
__global__ __launch_bounds__(256) void conv1x1_f16s_kernel(const Conv2dArgs a) {
  constexpr int BM = 128, KC = 32, RB = 80;  // row bytes: 32 halves + 8 pad (conflict-free ds_read_b128 phases)
  constexpr float WS = 64.0f;
  __shared__ __align__(16) unsigned char Ah[BM * RB], Al[BM * RB], Bh[BN * RB], Bl[BN * RB];
  __shared__ float s_max[2][4];
  fp16_ovfl_clamp();  // a weight beyond the assumed range saturates instead of turning into inf
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
  const int HW = a.H * a.W, n = blockIdx.z;
  const int p0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const float* __restrict__ xn = a.x + (size_t)n * a.Cin * HW;

  f32x16c acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

  // loader roles: B: thread = (pixel tid & 63, k group tid >> 6) -> 8 consecutive k; A: (channel tid & 127, k groups (tid >> 7) and + 2)
  const int bp = tid & (BN - 1), bk = (tid / BN) & 3, am = tid & 127, ak = tid >> 7;
  const bool bload = tid < 4 * BN;
  // two chunks of operands in flight: vb / va hold the chunk about to be staged, wb / wa the one after it (requested a whole iteration
  // before it is needed, so that only the first chunk's memory latency is exposed)
  float vb[8], va[2][8], wb[8], wa[2][8];
  auto fetch = [&](int k0, float (&b)[8], float (&aw)[2][8]) {
    const int gp = p0 + bp;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + 8 * bk + j;
      b[j] = (bload && gp < HW && k < a.Cin) ? xn[(size_t)k * HW + gp] : 0.f;
    }
    const int gm = m0 + am;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + 8 * (ak + 2 * i) + j;
        aw[i][j] = (gm < a.CoutP && k < a.Cin) ? a.w[(size_t)k * a.CoutP + gm] : 0.f;
      }
  };
  float xs = 1.0f, run_max = 0.f;  // running activation scale (power of two, workgroup-uniform) and the max|x| behind it
  fetch(0, vb, va);
  fetch(KC, wb, wa);  // all zeros beyond Cin
  int par = 0;
  for (int k0 = 0; k0 < a.Cin; k0 += KC, par ^= 1) {
    // max|x| of this chunk over the workgroup
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) mx = fmaxf(mx, fabsf(vb[j]));
    mx = wave_max_nonneg(mx);  // DPP ladder: no LDS round trips in the chunk loop's critical path
    if (l == 0) s_max[par][w] = mx;
    __syncthreads();  // also: every wave is done with the previous chunk in LDS
    run_max = fmaxf(run_max, fmaxf(fmaxf(s_max[par][0], s_max[par][1]), fmaxf(s_max[par][2], s_max[par][3])));
    if (run_max > 0.f) {  // scale = the power of two that puts the largest |x| seen so far into [2^13, 2^14) (fp16 max 65504; small
      int e;              // inputs -- gradients -- are scaled UP so that their low parts stay normal fp16 numbers)
      (void)frexpf(run_max, &e);  // run_max = f * 2^e, f in [0.5, 1)
      const float ns = ldexpf(1.0f, min(14 - e, 100));
      if (ns != xs) {     // the exponent grew (or this is the first non-zero chunk): rescale what is accumulated (exact)
        const float ratio = ns / xs;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[i] *= ratio; acc1[i] *= ratio; }
        xs = ns;
      }
    }
    {
      uint4 hi, lo;
      c1_split8(vb, xs, hi, lo);
      if (bload) {
        *reinterpret_cast<uint4*>(Bh + bp * RB + 16 * bk) = hi;
        *reinterpret_cast<uint4*>(Bl + bp * RB + 16 * bk) = lo;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        c1_split8(va[i], WS, hi, lo);
        *reinterpret_cast<uint4*>(Ah + am * RB + 16 * (ak + 2 * i)) = hi;
        *reinterpret_cast<uint4*>(Al + am * RB + 16 * (ak + 2 * i)) = lo;
      }
    }
    __syncthreads();
    // rotate: the chunk after next travels during this chunk's MFMAs and the whole next iteration
#pragma unroll
    for (int j = 0; j < 8; ++j) { vb[j] = wb[j]; va[0][j] = wa[0][j]; va[1][j] = wa[1][j]; }
    if (k0 + 2 * KC < a.Cin) fetch(k0 + 2 * KC, wb, wa);
#pragma unroll
    for (int ks = 0; ks < KC / 16; ++ks) {
      const int ko = 32 * ks + 16 * h;  // byte offset of this lane's 8 halves in the row
      const c1h8_t ah = *reinterpret_cast<const c1h8_t*>(Ah + (32 * w + r) * RB + ko);
      const c1h8_t al = *reinterpret_cast<const c1h8_t*>(Al + (32 * w + r) * RB + ko);
      const c1h8_t b0h = *reinterpret_cast<const c1h8_t*>(Bh + r * RB + ko);
      const c1h8_t b0l = *reinterpret_cast<const c1h8_t*>(Bl + r * RB + ko);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0h, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b0l, acc0, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b0h, acc0, 0, 0, 0);
      if constexpr (BN == 64) {
        const c1h8_t b1h = *reinterpret_cast<const c1h8_t*>(Bh + (32 + r) * RB + ko);
        const c1h8_t b1l = *reinterpret_cast<const c1h8_t*>(Bl + (32 + r) * RB + ko);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1h, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b1l, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b1h, acc1, 0, 0, 0);
      }
    }
  }

  // epilogue: register = GEMM row (output channel, or channel x sub-pixel of a ConvTranspose2d with kernel == stride), lane (r) = pixel column
  const float unscale = 1.0f / (WS * xs);
  const int ups = a.ups, s2 = ups * ups;
  const size_t oplane = (size_t)HW * s2;
  float* __restrict__ yn = a.y + ((size_t)n * a.out_ctotal + a.out_coff) * oplane;
  const float* __restrict__ rn = a.res != nullptr ? a.res + ((size_t)n * a.out_ctotal + a.out_coff) * oplane : nullptr;
#pragma unroll
  for (int t = 0; t < BN / 32; ++t) {
    const int gp = p0 + 32 * t + r;
    if (gp >= HW) continue;
    if (ups == 1) {  // plain 1x1 convolution: no index arithmetic beyond row * HW + pixel (the divisions below cost 13 us on a 44 us launch)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int co = m0 + 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (co >= a.CoutP) continue;
        float v = fmaf((t == 0 ? acc0[reg] : acc1[reg]) * unscale, a.scale[co], a.shift[co]);
        if (a.relu == 1) v = fmaxf(v, 0.f);
        else if (a.relu == 2) v = gelu_erf_f(v);
        const size_t oi = (size_t)co * HW + gp;
        if (rn != nullptr) v += rn[oi];
        if (a.relu == 3) v = fmaxf(v, 0.f);
        yn[oi] = v;
      }
      continue;
    }
    const int oy = gp / a.W, ox = gp - oy * a.W;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int gco = m0 + 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (gco >= a.CoutP) continue;
      const int co = gco / s2, sub = gco - co * s2, dy = sub / ups, dx = sub - dy * ups;
      float v = fmaf((t == 0 ? acc0[reg] : acc1[reg]) * unscale, a.scale[co], a.shift[co]);
      if (a.relu == 1) v = fmaxf(v, 0.f);
      else if (a.relu == 2) v = gelu_erf_f(v);
      const size_t oi = (size_t)co * oplane + (size_t)(oy * ups + dy) * (a.W * ups) + (ox * ups + dx);
      if (rn != nullptr) v += rn[oi];
      if (a.relu == 3) v = fmaxf(v, 0.f);
      yn[oi] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 3x3 convolution (stride 1 or 2) on the f16 matrix pipe with the same exact hi/lo split arithmetic: the BEV backbones' layers.
// Same tile as conv2d_igemm_kernel (64 output channels x 4x16 output pixels, 4 waves = 2 channel halves x 2 row pairs) and the same
// epilogue; K is walked in chunks of 16 input channels x 9 taps = 9 MFMA steps of v_mfma_f32_32x32x16_f16 x 3 split terms
// (27 x 32 cycles per wave and chunk against 72 x 64 for the fp32 form).  LDS holds the input patch pixel-major and the weight slab
// [tap][channel-out], both as 80-byte records {16 hi halves | 16 lo halves | pad}: an MFMA operand is one 16-byte read per plane and
// the odd record stride (5 x 16 B) keeps the 16 lanes of a ds_read_b128 phase on distinct banks.  NCHW is channel-strided, the MFMA
// wants 8 consecutive channels per lane: every loader item is 8 channel values of ONE pixel (or one tap / output channel), dword
// loads coalesced across the pixel / channel-out index, split and written as one record half.  Activations carry the running
// power-of-two scale of conv1x1_f16s_kernel; weights are pre-multiplied by 2^6.  Cin % 8 == 0 (a last half-filled chunk is zero-padded).
// ---------------------------------------------------------------------------------------------
template <int STRIDE>
__global__ __launch_bounds__(256) void conv3x3_f16s_kernel(const Conv2dArgs a) {
  constexpr int TY = 4, TX = 16, CC = 16, REC = 80;
  constexpr int PH = (TY - 1) * STRIDE + 3, PW = (TX - 1) * STRIDE + 3, NPX = PH * PW;
  constexpr int NPI = NPX * 2, NWI = 9 * 64 * 2;                       // loader items: (pixel, channel octet), (tap, channel-out, channel octet)
  constexpr int PIT = (NPI + 255) / 256, WIT = (NWI + 255) / 256;     // items per thread
  constexpr float WS = 64.0f;
  __shared__ __align__(16) unsigned char Ps[NPX * REC], Wsl[9 * 64 * REC];
  __shared__ float s_max[2][4];
  fp16_ovfl_clamp();
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
  const int tiles_x = (a.Wo + TX - 1) / TX;
  const int ty0 = (blockIdx.x / tiles_x) * TY, tx0 = (blockIdx.x % tiles_x) * TX;
  const int co0 = blockIdx.y * 64, n = blockIdx.z;
  const int iy0 = ty0 * STRIDE - a.pad, ix0 = tx0 * STRIDE - a.pad;
  const size_t plane = (size_t)a.H * a.W;
  const float* __restrict__ xn = a.x + (size_t)n * a.Cin * plane;

  const int pyl = 2 * (wv >> 1) + (r >> 4), pxl = r & 15;
  const int b_base = ((pyl * STRIDE) * PW + pxl * STRIDE) * REC + 16 * h;   // + (ky * PW + kx) * REC per tap; lo plane at + 32
  const int a_base = (32 * (wv & 1) + r) * REC + 16 * h;                    // + tap * 64 * REC

  f32x16c acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  float rp[PIT][8], rw[WIT][8];
  auto fetch = [&](int c0) {
#pragma unroll
    for (int j = 0; j < PIT; ++j) {
      const int it = tid + 256 * j, px = it >> 1, g = it & 1;
      const int py = px / PW, pxx = px - py * PW;
      const int gy = iy0 + py, gx = ix0 + pxx;
      const bool ok = it < NPI && c0 + 8 * g < a.Cin && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;   // Cin % 8 == 0: the last chunk may hold one octet
      const float* __restrict__ src = xn + (size_t)(ok ? c0 + 8 * g : 0) * plane + (ok ? (size_t)gy * a.W + gx : 0);
#pragma unroll
      for (int e = 0; e < 8; ++e) rp[j][e] = ok ? src[(size_t)e * plane] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < WIT; ++j) {
      const int it = tid + 256 * j, co = it & 63, tg = it >> 6, tap = tg >> 1, g = tg & 1;
      const bool ok = it < NWI && c0 + 8 * g < a.Cin && co0 + co < a.CoutP;
      const float* __restrict__ src = a.w + ((size_t)(ok ? c0 + 8 * g : 0) * 9 + tap) * a.CoutP + (ok ? co0 + co : 0);
#pragma unroll
      for (int e = 0; e < 8; ++e) rw[j][e] = ok ? src[(size_t)e * 9 * a.CoutP] : 0.f;
    }
  };
  float xs = 1.0f, run_max = 0.f;
  fetch(0);
  int par = 0;
  for (int c0 = 0; c0 < a.Cin; c0 += CC, par ^= 1) {
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < PIT; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(rp[j][e]));
    mx = wave_max_nonneg(mx);
    if (l == 0) s_max[par][wv] = mx;
    __syncthreads();  // also: every wave is done with the previous chunk in LDS
    run_max = fmaxf(run_max, fmaxf(fmaxf(s_max[par][0], s_max[par][1]), fmaxf(s_max[par][2], s_max[par][3])));
    if (run_max > 0.f) {
      int e2;
      (void)frexpf(run_max, &e2);
      const float ns = ldexpf(1.0f, min(14 - e2, 100));
      if (ns != xs) {
        const float ratio = ns / xs;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] *= ratio;
        xs = ns;
      }
    }
#pragma unroll
    for (int j = 0; j < PIT; ++j) {
      const int it = tid + 256 * j, px = it >> 1, g = it & 1;
      if (it < NPI) {
        uint4 hi, lo;
        c1_split8(rp[j], xs, hi, lo);
        *reinterpret_cast<uint4*>(Ps + px * REC + 16 * g) = hi;
        *reinterpret_cast<uint4*>(Ps + px * REC + 32 + 16 * g) = lo;
      }
    }
#pragma unroll
    for (int j = 0; j < WIT; ++j) {
      const int it = tid + 256 * j, co = it & 63, tg = it >> 6, tap = tg >> 1, g = tg & 1;
      if (it < NWI) {
        uint4 hi, lo;
        c1_split8(rw[j], WS, hi, lo);
        *reinterpret_cast<uint4*>(Wsl + (tap * 64 + co) * REC + 16 * g) = hi;
        *reinterpret_cast<uint4*>(Wsl + (tap * 64 + co) * REC + 32 + 16 * g) = lo;
      }
    }
    __syncthreads();
    if (c0 + CC < a.Cin) fetch(c0 + CC);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - 3 * ky;
      const unsigned char* ap = Wsl + tap * 64 * REC + a_base;
      const unsigned char* bp = Ps + (ky * PW + kx) * REC + b_base;
      const c1h8_t ah = *reinterpret_cast<const c1h8_t*>(ap), al = *reinterpret_cast<const c1h8_t*>(ap + 32);
      const c1h8_t bh = *reinterpret_cast<const c1h8_t*>(bp), bl = *reinterpret_cast<const c1h8_t*>(bp + 32);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    }
  }

  // epilogue: lane = pixel (pyl, pxl), register = output channel row (as conv2d_igemm_kernel, ups == 1)
  const int oy = ty0 + pyl, ox = tx0 + pxl;
  if (oy >= a.Ho || ox >= a.Wo) return;
  const float unscale = 1.0f / (WS * xs);
  const size_t oplane = (size_t)a.Ho * a.Wo;
  float* __restrict__ yn = a.y + ((size_t)n * a.out_ctotal + a.out_coff) * oplane;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int co = co0 + 32 * (wv & 1) + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (co >= a.CoutP) continue;
    float v = fmaf(acc[reg] * unscale, a.scale[co], a.shift[co]);
    if (a.relu == 1) v = fmaxf(v, 0.f);
    else if (a.relu == 2) v = gelu_erf_f(v);
    const size_t oi = (size_t)co * oplane + (size_t)oy * a.Wo + ox;
    if (a.res != nullptr) v += a.res[((size_t)n * a.out_ctotal + a.out_coff) * oplane + oi];
    if (a.relu == 3) v = fmaxf(v, 0.f);
    yn[oi] = v;
  }
}

}  // namespace gc
#include "conv_h3_kernels.h"
namespace gc {

// default arithmetic mode: the three-term f16-pipe kernels for every shape whose prepared buffer carries the operand form
template <int KH, int KW, int TY, int KS, int NSUB>
inline int conv2d_h3l_launch(const Conv2dArgs& a, dim3 grid, hipStream_t st) {   // grid = (tiles, channel blocks, samples): launched 1-D, decoded XCD-aware
  using G = H3L<KH, KW, TY, KS, NSUB>;
  static LdsAttrOnce attr;
  if (const int rc = attr.set(conv2d_h3l_kernel<KH, KW, TY, KS, NSUB>, G::SMEM)) return rc;
  conv2d_h3l_kernel<KH, KW, TY, KS, NSUB><<<dim3(grid.x * grid.y * grid.z), 256, G::SMEM, st>>>(a, (int)grid.x, (int)grid.z);
  return GC_OK;
}
inline int conv2d_h3_enqueue(const Conv2dArgs& a, int N, int KH, int KW, hipStream_t st) {
  const int cb = (a.CoutP + 63) / 64;
  const long long tiles8 = (long long)((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
  const bool s2 = a.stride == 2;
  const bool ty8 = !s2 && a.Ho >= 8 && tiles8 * cb * N >= 512;   // 8-row tiles while they still give every CU its two resident workgroups
  const int tiles = ty8 ? (int)tiles8 : ((a.Ho + 3) / 4) * ((a.Wo + 15) / 16);
  const dim3 grid(tiles, cb, N);
  if (grid.y > 65535 || grid.z > 65535) return fail(GC_ERR_ARG, "conv2d: too many channel tiles / samples");
  GC_KLOG("conv2d_h3_kernel");
  TimedLaunch tl(KF_CONV2D, st, 2.0 * N * a.Ho * a.Wo * (double)a.CoutP * a.Cin * KH * KW);   // algorithmic FLOPs (bench.py --workload train)
  int rc = GC_OK;
  if (KH == 3 && s2) conv2d_h3_kernel<3, 3, 2, 4, 1><<<grid, 256, 0, st>>>(a);
  else if (KH == 3) rc = ty8 ? conv2d_h3l_launch<3, 3, 8, 1, 3>(a, grid, st) : conv2d_h3l_launch<3, 3, 4, 1, 3>(a, grid, st);
  else rc = ty8 ? conv2d_h3l_launch<2, 2, 8, 1, 1>(a, grid, st) : conv2d_h3l_launch<2, 2, 4, 1, 1>(a, grid, st);   // one sub-stage of four steps: 80 KB, still two workgroups per CU
  if (rc != GC_OK) return rc;
  GC_HIP(hipGetLastError());
  return GC_OK;
}

inline int conv2d_enqueue(const Conv2dArgs& a, int N, int KH, int KW, hipStream_t st) {
  {
    const Modes md = modes_snapshot();
    const bool shape = (KH == 3 && KW == 3 && (a.stride == 1 || a.stride == 2)) || (KH == 2 && KW == 2 && a.stride == 1);
    if (a.w3 != nullptr && shape && md.split() && !md.split2()) return conv2d_h3_enqueue(a, N, KH, KW, st);
  }
  if (KH == 3 && KW == 3 && (a.stride == 1 || a.stride == 2) && a.ups == 1 && a.Cin % 8 == 0 && a.CoutP >= 32 && modes_snapshot().split2()) {
    const int tiles3 = ((a.Ho + 3) / 4) * ((a.Wo + 15) / 16);
    const dim3 g3(tiles3, (a.CoutP + 63) / 64, N);
    if (g3.y > 65535 || g3.z > 65535) return fail(GC_ERR_ARG, "conv2d: too many channel tiles / samples");
    if (a.stride == 1) conv3x3_f16s_kernel<1><<<g3, 256, 0, st>>>(a);
    else conv3x3_f16s_kernel<2><<<g3, 256, 0, st>>>(a);
    GC_HIP(hipGetLastError());
    return GC_OK;
  }
  // measured on MI355X (V2X-ViT / Where2comm Linear layers, 2-5 agents, 64x128 .. 96x352): against the exact-fp32 kernel below the
  // split kernel wins clearly from 256 output channels up (qkv 384 / 768: 71 -> 44 us, 544 -> 315 us); at 128 it is level on the
  // smallest shape and ahead on the larger ones (V2X-ViT forward, 4 agents x 96x352: 10.1 -> 9.6 ms) since it got two chunks in flight
  if (KH == 1 && KW == 1 && a.stride == 1 && a.pad == 0 && a.Cin >= 16 && a.CoutP >= 128 && modes_snapshot().split2()) {  // ups > 1: ConvTranspose2d deblocks
    const int HW = a.H * a.W, mb = (a.CoutP + 127) / 128;
    const bool narrow = (long long)((HW + 63) / 64) * mb * N < 1024;  // fewer than 4 workgroups per CU: halve the pixel tile
    const dim3 g1(narrow ? (HW + 31) / 32 : (HW + 63) / 64, mb, N);
    if (g1.y > 65535 || g1.z > 65535) return fail(GC_ERR_ARG, "conv2d: too many channel tiles / samples");
    if (narrow) conv1x1_f16s_kernel<32><<<g1, 256, 0, st>>>(a);
    else conv1x1_f16s_kernel<64><<<g1, 256, 0, st>>>(a);
    GC_HIP(hipGetLastError());
    return GC_OK;
  }
  // Measured and NOT adopted (second half of round 4, profiles/r4_train_lib_ab.txt): a dedicated Linear kernel over consecutive pixels
  // (workgroup = 128 pixels x up to 128 output channels, 128-bit loads of 512-byte rows, 128-byte store runs, the input read once per
  // 128 channels).  Its launches were shorter in the kernel trace (the Enhancer's four Linear launches at 4 x 200 x 704: 1 041 us against
  // 1 190 us here), but the training step on one box was 0.45 ms SLOWER with it (13.65 against 13.2 ms per scene, three alternating
  // rounds; cause not established -- the side stream's weight-gradient launches run beside these layers).
  // 256 channels per workgroup: 702 us for Linear1 alone; several tiles per workgroup: 688 us (the next barrier waits for the stores).
  // 8-row tiles (two accumulators per wave) whenever they still give the launch >= 2 workgroups per CU
  const long long tiles8 = (long long)((a.Ho + 7) / 8) * ((a.Wo + 15) / 16);
  // (not for 1 x 1: its weight slab is small against the pixel patch, and the doubled patch staging cost 9 % on the Enhancer's Linear layers)
  const bool ty8 = a.Ho >= 8 && KH > 1 && tiles8 * ((a.CoutP + 63) / 64) * N >= 512;
  const int tiles = ty8 ? (int)tiles8 : ((a.Ho + 3) / 4) * ((a.Wo + 15) / 16);
  const dim3 grid(tiles, (a.CoutP + 63) / 64, N);
  if (grid.y > 65535 || grid.z > 65535) return fail(GC_ERR_ARG, "conv2d: too many channel tiles / samples");
  TimedLaunch tl(KF_CONV2D, st, 2.0 * N * a.Ho * a.Wo * (double)a.CoutP * a.Cin * KH * KW);   // algorithmic FLOPs (bench.py --workload train)
  if (KH == 3 && KW == 3 && a.stride == 1) { if (ty8) conv2d_igemm_kernel<3, 3, 8, 1, 8><<<grid, 256, 0, st>>>(a); else conv2d_igemm_kernel<3, 3, 8, 1><<<grid, 256, 0, st>>>(a); }
  else if (KH == 3 && KW == 3 && a.stride == 2) { if (ty8) conv2d_igemm_kernel<3, 3, 8, 2, 8><<<grid, 256, 0, st>>>(a); else conv2d_igemm_kernel<3, 3, 8, 2><<<grid, 256, 0, st>>>(a); }
  else if (KH == 1 && KW == 1 && a.stride == 1) conv2d_igemm_kernel<1, 1, 32, 1><<<grid, 256, 0, st>>>(a);
  else if (KH == 2 && KW == 2 && a.stride == 1) { if (ty8) conv2d_igemm_kernel<2, 2, 8, 1, 8><<<grid, 256, 0, st>>>(a); else conv2d_igemm_kernel<2, 2, 8, 1><<<grid, 256, 0, st>>>(a); }   // sub-pixel form of a transposed 3x3 stride-2 conv
  else return fail(GC_ERR_ARG, "conv2d: supported shapes are 3x3 stride 1/2 and 1x1 stride 1 (ConvTranspose2d with kernel == stride runs as 1x1)");
  GC_HIP(hipGetLastError());
  return GC_OK;
}

}  // namespace gc
