// General 2-D convolution for the layers AROUND the hot path (gfx950, wave64, exact fp32 on the
// matrix cores): BaseBEVBackbone blocks/deblocks, DownsampleConv, the detection heads.
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
};

template <int KH, int KW, int CC, int STRIDE>
__global__ __launch_bounds__(256) void conv2d_igemm_kernel(const Conv2dArgs a) {
  constexpr int TY = 4, TX = 16, KHW = KH * KW;
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

  f32x16c acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

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
        const float bv = Ps[(cp * PS + ky * PW + kx) * 2 + b_base];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
      }
    }
  }

  // epilogue: lane = pixel (pyl, pxl), register = output channel row
  const int oy = ty0 + pyl, ox = tx0 + pxl;
  if (oy >= a.Ho || ox >= a.Wo) return;
  const int s = a.ups, s2 = s * s;
  const size_t oplane = (size_t)a.Ho * s * a.Wo * s;
  float* __restrict__ yn = a.y + ((size_t)n * a.out_ctotal + a.out_coff) * oplane;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int gco = co0 + 32 * (wv & 1) + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (gco >= a.CoutP) continue;
    const int co = gco / s2, sub = gco - co * s2, dy = sub / s, dx = sub - dy * s;
    float v = fmaf(acc[reg], a.scale[co], a.shift[co]);
    if (a.relu == 1) v = fmaxf(v, 0.f);
    else if (a.relu == 2) v = gelu_erf_f(v);
    const size_t oi = (size_t)co * oplane + (size_t)(oy * s + dy) * (a.Wo * s) + (ox * s + dx);
    if (a.res != nullptr) v += a.res[((size_t)n * a.out_ctotal + a.out_coff) * oplane + oi];
    if (a.relu == 3) v = fmaxf(v, 0.f);   // ReLU AFTER the residual add (ResNet BasicBlock)
    yn[oi] = v;
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
  } else {
    // 1x1 GEMM with CoutP = Cout*khw rows: out[ci][co*khw + tap] = w[ci][co][tap]  (already that order)
    a.out[i] = a.w[i];
  }
}

inline int conv2d_enqueue(const Conv2dArgs& a, int N, int KH, int KW, hipStream_t st) {
  const int tiles = ((a.Ho + 3) / 4) * ((a.Wo + 15) / 16);
  const dim3 grid(tiles, (a.CoutP + 63) / 64, N);
  if (grid.y > 65535 || grid.z > 65535) return fail(GC_ERR_ARG, "conv2d: too many channel tiles / samples");
  if (KH == 3 && KW == 3 && a.stride == 1) conv2d_igemm_kernel<3, 3, 8, 1><<<grid, 256, 0, st>>>(a);
  else if (KH == 3 && KW == 3 && a.stride == 2) conv2d_igemm_kernel<3, 3, 8, 2><<<grid, 256, 0, st>>>(a);
  else if (KH == 1 && KW == 1 && a.stride == 1) conv2d_igemm_kernel<1, 1, 32, 1><<<grid, 256, 0, st>>>(a);
  else return fail(GC_ERR_ARG, "conv2d: supported shapes are 3x3 stride 1/2 and 1x1 stride 1 (ConvTranspose2d with kernel == stride runs as 1x1)");
  GC_HIP(hipGetLastError());
  return GC_OK;
}

}  // namespace gc
