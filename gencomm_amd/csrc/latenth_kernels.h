// Latent sampler step (latent_kernels.h: conv_out -> ancestral update -> conv_in of one step, collapsed by linearity)
// on the f16 matrix pipe with the exact three-term operand splits, the arithmetic and LDS layout of conv8h_kernels.h
// (hi / lo fp16 planes + a bf8 third-term plane; weights w1 + w2 + w3 in fp16 and a bf8 copy; six matrix instructions
// per product block, five for the fp16-exact step noise).
//
//   hs0_{t-1} = k + c2*(hs0_t - k) + c1*[ Wc5 (*) A_t + bsum + fix ] + s*( W_x (*) eps_t )
//
// Phase 2 (5x5 composite, 8 -> 8): a row pair sees a 6x5 window = 30 taps = 8 MFMAs of 4 taps x 8 channels (the last
// two tap slots carry zero weights); phase 3 (3x3, C -> 8 on the noise field, 8 channels per pass through LDS) is
// conv8h's inner loop unchanged.  Per tile-wave: 192 + C/8 * 72 MFMAs of 16 cycles instead of 1600 + C/8 * 576 fp32
// MFMAs of ~9 cycles, and they no longer compete with the Philox / Box-Muller VALU work for the issue port.
#pragma once
#include "conv8b_kernels.h"
#include "conv8h_kernels.h"
#include "latent_kernels.h"

namespace gc {

constexpr int HL_LH5 = HC_TH + 4;            // 20 tile rows (2-pixel halo)
constexpr int HL_PLANE5 = HL_LH5 * HC_ROW;   // 23040 bytes per hi / lo plane
constexpr int HL_W5TAB3 = 8 * HC_WC3;        // dwords: 8 tap groups in conv8h's three-term layout, then 64 floats (even 1/scale, odd scale)
constexpr int HL_TPLANE5 = HL_PLANE5 / 2;    // 11520 bytes: bf8 third-term plane of the 20-row tile
constexpr int HL_RING = (2 * (HC_TW + 4) + 2 * HL_LH5) * 8;  // floats: exact A' of the image's outermost ring inside the tile (border fix)

// Three-term A-operand tables (conv8h_kernels.h: per tap group [term 3][lane][4 dwords] fp16 + [lane][2 dwords] bf8) of the
// composite kernel, of conv_in's x part and of its message chunk, with ONE power-of-two scale (they accumulate into the
// same registers).  wc5: prepared [8 i][25 d][8 o] (prep_latent_kernel, same stream, earlier).
__global__ __launch_bounds__(256) void prep_latent_h_kernel(const float* __restrict__ wc5, const float* __restrict__ w_in /*[8][C+2][3][3]*/,
                                                           float* __restrict__ dst5, float* __restrict__ dstx,
                                                           float* __restrict__ dstc, int C) {
  __shared__ float s_max[256];
  const int tid = threadIdx.x;
  const int CI = C + 2;
  float m = 0.f;
  for (int i = tid; i < 1600; i += 256) m = fmaxf(m, fabsf(wc5[i]));
  for (int i = tid; i < 8 * CI * 9; i += 256) m = fmaxf(m, fabsf(w_in[i]));  // message channels included (conv_in_h_kernel)
  s_max[tid] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) s_max[tid] = fmaxf(s_max[tid], s_max[tid + s]);
    __syncthreads();
  }
  const float wmax = s_max[0];
  int ex = 0;
  if (wmax > 0.f) (void)frexpf(wmax, &ex);
  const float scale = wmax > 0.f ? ldexpf(1.0f, 14 - ex) : 1.0f;  // largest weight in [2^13, 2^14), see prep_conv8h_kernel
  // entry q of a tap group's HC_WC3 dwords from the scaled weights of the eight channels of (group, lane)
  auto entry = [&](int q, const float (&wv)[8]) -> uint32_t {
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
      return (uint32_t)v[0] | ((uint32_t)v[1] << 16);
    }
    const int d = (q - 768) & 1;
    const float k = 1.0f / HC_TSCALE;
    return bf8x4(wv[4 * d] * k, wv[4 * d + 1] * k, wv[4 * d + 2] * k, wv[4 * d + 3] * k);
  };
  auto lane_of = [](int q) { return q < 768 ? (q >> 2) & 63 : (q - 768) >> 1; };
  uint32_t* __restrict__ o5 = reinterpret_cast<uint32_t*>(dst5);
  for (int i = tid; i < HL_W5TAB3; i += 256) {
    const int c = i / HC_WC3, q = i - c * HC_WC3, l = lane_of(q);
    const int mrow = l & 15, kg = l >> 4, r = mrow >> 3, oc = mrow & 7;
    const int t = 4 * c + kg, dyp = t / 5, dx = t - 5 * dyp, dy = dyp - r;
    const bool live = t < 30 && dy >= 0 && dy <= 4;
    float wv[8];
    for (int ch = 0; ch < 8; ++ch) wv[ch] = live ? wc5[(ch * 25 + dy * 5 + dx) * 8 + oc] * scale : 0.f;
    o5[i] = entry(q, wv);
  }
  if (tid < 64) dst5[HL_W5TAB3 + tid] = (tid & 1) ? scale : 1.0f / scale;
  uint32_t* __restrict__ ox = reinterpret_cast<uint32_t*>(dstx);
  for (int i = tid; i < (C / 8) * HC_WTAB3; i += 256) {
    const int s = i / HC_WTAB3, rem = i - s * HC_WTAB3;
    const int c = rem / HC_WC3, q = rem - c * HC_WC3, l = lane_of(q);
    const int mrow = l & 15, kg = l >> 4, r = mrow >> 3, oc = mrow & 7;
    const int dyp = hc_tap_row(kg), dx = c, dy = dyp - r;
    const bool live = dy >= 0 && dy <= 2;
    float wv[8];
    for (int ch = 0; ch < 8; ++ch) wv[ch] = live ? w_in[(((size_t)oc * CI + 2 + s * 8 + ch) * 3 + dy) * 3 + dx] * scale : 0.f;
    ox[i] = entry(q, wv);
  }
  // the two message channels as an 8-channel chunk (channels 2..7 zero): chunk 0 of conv_in_h_kernel
  uint32_t* __restrict__ oc_ = reinterpret_cast<uint32_t*>(dstc);
  for (int i = tid; i < HC_WTAB3; i += 256) {
    const int c = i / HC_WC3, q = i - c * HC_WC3, l = lane_of(q);
    const int mrow = l & 15, kg = l >> 4, r = mrow >> 3, oc = mrow & 7;
    const int dyp = hc_tap_row(kg), dx = c, dy = dyp - r;
    const bool live = dy >= 0 && dy <= 2;
    float wv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < 2; ++ch) wv[ch] = live ? w_in[(((size_t)oc * CI + ch) * 3 + dy) * 3 + dx] * scale : 0.f;
    oc_[i] = entry(q, wv);
  }
}

// Exact fp32 A' values of the image's outermost ring inside the tile, for the border correction (the tile itself holds A' as
// hi + lo + a deferred third term): rows [0] = image row 0, [1] = image row H-1, indexed by tile pixel + 2; then columns
// [0] = image column 0, [1] = image column W-1, indexed by tile row; 8 channels each.  Every position the correction reads
// lies on that ring (it is an in-image neighbour of an out-of-image position).
__device__ __forceinline__ int hl_ring_slot(int gy, int gx, int trow, int tpx, int H, int W) {
  if (gy == 0) return tpx + 2;
  if (gy == H - 1) return (HC_TW + 4) + tpx + 2;
  if (gx == 0) return 2 * (HC_TW + 4) + trow;
  return 2 * (HC_TW + 4) + HL_LH5 + trow;  // gx == W - 1
}
// Border correction of one output row strip (row gy, pixels gx..gx+3, channels oc0..oc0+3) -- the terms of the 5x5
// composite that would pass through x0_hat positions outside the image (latent_kernels.h, phase 2b, same algebra).
// Applied IN PLACE to the accumulators of the row pair (accp[pixel group j][channel o], accumulator units: x sc) -- a separate
// 32-register correction array beside the 32 accumulators and the deferred third terms made the kernel spill.
__device__ __forceinline__ void hl_border_fix(const LatentArgs& a, const float* ring, int x0, int y0, f32x4 (&accp)[4], int gy, int gx,
                                              int trow /* gy - y0 */, int px0 /* gx - x0 */, int oc0, float c1, float sc) {
  const int H = a.H, W = a.W;
  const bool mine = gy < H && gx < W && (gy == 0 || gy == H - 1 || gx == 0 || gx + 4 >= W);
  if (!__any(mine)) return;
#pragma unroll 1
  for (int t1 = 0; t1 < 9; ++t1) {
    const int t1y = t1 / 3, t1x = t1 - t1y * 3;
    const int qy = gy + t1y - 1;
    bool outp[4];
    bool anyo = false;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int qx = gx + p + t1x - 1;
      outp[p] = mine && (gx + p < W) && (qy < 0 || qy >= H || qx < 0 || qx >= W);
      anyo |= outp[p];
    }
    if (!__any(anyo)) continue;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const float b = a.bring[t1 * 8 + oc0 + o] * c1 * sc;
#pragma unroll
      for (int p = 0; p < 4; ++p) accp[p][o] -= outp[p] ? b : 0.f;
    }
#pragma unroll 1
    for (int t2 = 0; t2 < 9; ++t2) {
      const int t2y = t2 / 3, t2x = t2 - t2y * 3;
      const int ry = qy + t2y - 1;
      bool inp[4];
      bool anyi = false;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int rx = gx + p + t1x + t2x - 2;
        inp[p] = outp[p] && ry >= 0 && ry < H && rx >= 0 && rx < W;
        anyi |= inp[p];
      }
      if (!__any(anyi)) continue;
      const float* __restrict__ w = a.wc1 + (t1 * 9 + t2) * 64;
#pragma unroll 2
      for (int i = 0; i < 8; ++i) {
        float av[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {  // always inside the halo-2 tile and on the image's outermost ring
          const int tr = trow + t1y + t2y, tp = px0 + p + t1x + t2x - 2;  // tile row (0 = image row y0 - 2), tile pixel (0 = image column x0)
          av[p] = inp[p] ? ring[hl_ring_slot(y0 - 2 + tr, x0 + tp, tr, tp, H, W) * 8 + i] : 0.f;
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const float wv = w[i * 8 + oc0 + o] * sc;
#pragma unroll
          for (int p = 0; p < 4; ++p) accp[p][o] = fmaf(-wv, av[p], accp[p][o]);
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// conv_in on the f16 matrix pipe: cat[cond(2), x_t(C)] -> 8 channels, 3x3 pad 1 (unet.py:229-233; channel order cond
// first, cond_diff.py:318) -- conv8h's inner loop over 1 + C/8 eight-channel chunks streamed through LDS (the message
// chunk carries six zero channels), next chunk's tile and weights in flight during the current chunk's MFMAs.
// C == 0: message channels only (the sampler's constant map k).  Needs W % 4 == 0.
// ---------------------------------------------------------------------------------------------
struct ConvInHArgs {
  const float* cond;    // [n][2][H][W]
  const float* x;       // [n][C][H][W]
  const float* wch;     // message-chunk table (three-term layout, HC_WTAB3 dwords)
  const float* wxh;     // C/8 tables of the x part
  const float* scales;  // [0] = 1 / scale, [1] = scale
  const float* bias;    // [8]
  float* dst;           // [n][8][H][W]
  double* dstat;
  int C, H, W;
  int xcd;
  const float* amax;    // optional device bounds {max|cond|, max|x|}: exact power-of-two range reduction (common.h)
};

__global__ __launch_bounds__(HC_NT, 3) void conv_in_h_kernel(const ConvInHArgs a) {
  constexpr int NT = HC_NT, TW = HC_TW, TH = HC_TH;
  __shared__ __align__(16) unsigned char tile[2 * HC_PLANE + HC_TPLANE];
  __shared__ float s_red[NT / 64][16];
  fp16_ovfl_clamp();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const BlockId bid = xcd_block(a.xcd);
  const int n = bid.z;
  const int x0 = bid.x * TW, y0 = bid.y * TH;
  const int H = a.H, W = a.W;
  const unsigned plane = (unsigned)(H * W);
  const int ln = lane & 15, g = lane >> 4, ch = g & 1, rr = g >> 1;
  const int gx = x0 + 4 * ln, gy0 = y0 + 4 * wave + rr;
  const bool wave_live = y0 + 4 * wave < H;
  const int r0 = tid >> 4, qx = tid & 15;
  const float mul = a.amax != nullptr ? act_scale(fmaxf(a.amax[0], a.C > 0 ? a.amax[1] : 0.f)) : 1.0f;
  const float inv_s = a.scales[0] / mul, sc = a.scales[1] * mul;

  f32x4 acc[2][4];
  {
    const float4 b4 = *reinterpret_cast<const float4*>(a.bias + 4 * ch);
    const f32x4 b0 = {b4.x * sc, b4.y * sc, b4.z * sc, b4.w * sc};
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[p][j] = b0;
  }
  int off[4][3];
  hc_lane_offsets(off, wave, lane);
  const int nchunk = a.C / 8;
  const float* __restrict__ xp = a.x + (size_t)n * a.C * plane;
  TileRegs<TW, TH, NT, 8> R;
  float2 hreg = make_float2(0.f, 0.f);
  {  // chunk 0: the two message channels
    const float* __restrict__ sp = a.cond + (size_t)n * 2 * plane;
    TileRegs<TW, TH, NT, 2> Rc;
    stage_load<TW, TH, NT, 2, false>(Rc, sp, plane, W, H, W, x0, y0, tid);
    float2 hc = make_float2(0.f, 0.f);
    if (tid < HC_LH * 8 && (tid & 3) == 0) {
      const int side = (tid >> 2) & 1, r = tid >> 3;
      const int gy = y0 - 1 + r, gxh = side ? x0 + TW : x0 - 1;
      if (gy >= 0 && gy < H && gxh >= 0 && gxh < W) {
        hc.x = sp[(unsigned)gy * (unsigned)W + (unsigned)gxh];
        hc.y = sp[plane + (unsigned)gy * (unsigned)W + (unsigned)gxh];
      }
    }
    if (nchunk > 0) {  // first x chunk requested behind the message tile
      stage_load<TW, TH, NT, 8, false>(R, xp, plane, W, H, W, x0, y0, tid);
      hreg = halo_load_h<false>(xp, plane, W, H, W, x0, y0, tid);
    }
    float e[8][4];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) e[c][j] = 0.f;
    e[0][0] = Rc.v[0].x * mul; e[0][1] = Rc.v[0].y * mul; e[0][2] = Rc.v[0].z * mul; e[0][3] = Rc.v[0].w * mul;
    e[1][0] = Rc.v[1].x * mul; e[1][1] = Rc.v[1].y * mul; e[1][2] = Rc.v[1].z * mul; e[1][3] = Rc.v[1].w * mul;
    hc_store_main3(tile, r0, qx, e);
    const float er[4] = {Rc.vr.x * mul, Rc.vr.y * mul, Rc.vr.z * mul, Rc.vr.w * mul};  // zero for threads whose channel (tid / 32) is not 0 or 1
    hc_store_rem3(tile, tid, er);
    if (tid < HC_LH * 8) hc_store_halo3(tile, tid, hc.x * mul, hc.y * mul);  // pairs 1..3: zeros
    __syncthreads();
    if (wave_live) conv_tile_mfma3<false, true>(tile, a.wch, acc, off, lane);
  }
#pragma unroll 1
  for (int cc = 0; cc < nchunk; ++cc) {
    __syncthreads();
    stage_store_h<false, true>(tile, R, hreg, H, W, x0, y0, nullptr, tid, mul);
    if (cc + 1 < nchunk) {
      stage_load<TW, TH, NT, 8, false>(R, xp + (size_t)(cc + 1) * 8 * plane, plane, W, H, W, x0, y0, tid);
      hreg = halo_load_h<false>(xp + (size_t)(cc + 1) * 8 * plane, plane, W, H, W, x0, y0, tid);
    }
    __syncthreads();
    if (wave_live) conv_tile_mfma3<false, true>(tile, a.wxh + (size_t)cc * HC_WTAB3, acc, off, lane);
  }

  float part[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) part[i] = 0.f;
  if (wave_live) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int gy = gy0 + 2 * p;
      if (gy < H && gx + 3 < W) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = acc[p][j][i] * inv_s;
            part[i] += v[j];
            part[4 + i] = fmaf(v[j], v[j], part[4 + i]);
          }
          *reinterpret_cast<float4*>(a.dst + ((size_t)n * 8 + 4 * ch + i) * plane + (size_t)gy * W + gx) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  }
  if (a.dstat != nullptr) hc_stats_commit(part, s_red, a.dstat + (size_t)n * 16, tid);
}

// ---------------------------------------------------------------------------------------------
// conv_out on the f16 matrix pipe: GroupNorm(norm_out)+SiLU -> 3x3 8 -> C with the sampler's update in the epilogue
// (conv_out_kernel's contract, unet_kernels.h: POST 0 x0_hat, 1 explicit noise, 2 in-kernel Philox).  M = 16 output
// channels (no row pairing needed), N = 16 pixels (lane n of group j owns pixel 4n + j), K = 4 taps x 8 channels: the 9
// taps are 3 MFMAs (three zero tap slots).  One workgroup stages its 64x16 tile ONCE and walks all C/16 channel blocks
// (the fp32 kernel stages it once per block).  Table: [block][c 3][hi/lo][lane][4 dwords], then 64 scale floats.
// Needs W % 4 == 0.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_conv_out_h_kernel(const float* __restrict__ w /*[C][8][3][3]*/, float* __restrict__ tab, int C) {
  __shared__ float s_max[256];
  const int tid = threadIdx.x;
  float m = 0.f;
  for (int i = tid; i < C * 72; i += 256) m = fmaxf(m, fabsf(w[i]));
  s_max[tid] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) s_max[tid] = fmaxf(s_max[tid], s_max[tid + s]);
    __syncthreads();
  }
  const float wmax = s_max[0];
  int ex = 0;
  if (wmax > 0.f) (void)frexpf(wmax, &ex);
  const float scale = wmax > 0.f ? ldexpf(1.0f, 14 - ex) : 1.0f;  // largest weight in [2^13, 2^14), see prep_conv8h_kernel
  const int nocb = (C + 15) / 16;
  uint32_t* __restrict__ out = reinterpret_cast<uint32_t*>(tab);
  for (int i = tid; i < nocb * HC_WTAB3; i += 256) {
    const int ob = i / HC_WTAB3, rem = i - ob * HC_WTAB3;
    const int c = rem / HC_WC3, q = rem - c * HC_WC3;
    const int l = q < 768 ? (q >> 2) & 63 : (q - 768) >> 1;
    const int oc = 16 * ob + (l & 15), t = 4 * c + (l >> 4);
    float wv[8];
    for (int ch = 0; ch < 8; ++ch) wv[ch] = (t < 9 && oc < C) ? w[((size_t)oc * 8 + ch) * 9 + t] * scale : 0.f;
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
      const float k = 1.0f / HC_TSCALE;
      out[i] = bf8x4(wv[4 * d] * k, wv[4 * d + 1] * k, wv[4 * d + 2] * k, wv[4 * d + 3] * k);
    }
  }
  if (tid < 64) tab[nocb * HC_WTAB3 + tid] = (tid & 1) ? scale : 1.0f / scale;
}

template <int POST>
__global__ __launch_bounds__(HC_NT, 3) void conv_out_h_kernel(const ConvOutArgs a) {
  constexpr int NT = HC_NT, TW = HC_TW, TH = HC_TH;
  __shared__ __align__(16) unsigned char tile[2 * HC_PLANE + HC_TPLANE];
  __shared__ float s_ab[8][2];
  fp16_ovfl_clamp();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const BlockId bid = xcd_block(a.xcd);
  const int n = bid.z;
  const int x0 = bid.x * TW, y0 = bid.y * TH;
  const int H = a.H, W = a.W, C = a.C;
  const unsigned plane = (unsigned)(H * W);
  const int ln = lane & 15, g = lane >> 4;
  const int gx = x0 + 4 * ln;
  const bool wave_live = y0 + 4 * wave < H;
  const int nocb = (C + 15) / 16;

  TileRegs<TW, TH, NT, 8> R;
  float2 hreg;
  stage_load_b<false>(R, hreg, a.src, !a.src_bf16, (size_t)n * 8 * plane, plane, W, H, W, x0, y0, tid);
  if (tid < 8) {
    float A, B;
    gn_coeff(a.sstat + (size_t)n * 16, tid, 2, a.inv_cnt, a.gamma[tid], a.beta[tid], &A, &B);
    s_ab[tid][0] = HC_NL2E * A;   // stage_store_h<true, true> takes the coefficients times -log2(e) (gn_silu_zr)
    s_ab[tid][1] = HC_NL2E * B;
  }
  __syncthreads();
  stage_store_h<true, true>(tile, R, hreg, H, W, x0, y0, s_ab, tid);
  __syncthreads();
  if (!wave_live) return;  // no barrier below

  // B addressing: tap t = 4c + lane/16 (slots 9..11 carry zero weights: clamp), output row 4*wave + rrow
  int off[4][3];
  {
    const int kg = lane >> 4;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int t = min(4 * c + kg, 8), dy = t / 3, dx = t - 3 * dy;
#pragma unroll
      for (int j = 0; j < 4; ++j) off[j][c] = hc_addr(4 * wave + dy, 4 * ln + j + dx - 1);
    }
  }
  const float inv_s = a.wh[nocb * HC_WTAB3], sc = a.wh[nocb * HC_WTAB3 + 1];
  const unsigned long long seed = (POST == 2 && a.seed_dev) ? *a.seed_dev : a.seed;
  float c1 = 0.f, c2 = 0.f, sg = 0.f;
  if (POST != 0) { c1 = a.sched[2]; c2 = a.sched[3]; sg = a.sched[4]; }
  float amax = 0.f;

#pragma unroll 1
  for (int ob = 0; ob < nocb; ++ob) {
    const float* __restrict__ tab = a.wh + (size_t)ob * HC_WTAB3;
    const int oc0 = 16 * ob + 4 * g;
    f32x4 b0;
#pragma unroll
    for (int i = 0; i < 4; ++i) b0[i] = (oc0 + i < C) ? a.bias[oc0 + i] * sc : 0.f;
    // two output rows at a time (32 accumulator registers live; the whole 4-row block at once spilled 1.6 KB per thread):
    // the bf8 third-term instructions of both rows first, then the f16 passes row by row -- instructions of different input
    // type on the same registers are at least four matrix instructions apart (conv8h_kernels.h conv_tile_mfma3)
#pragma unroll 1
    for (int rp = 0; rp < 2; ++rp) {
      f32x4 acc[2][4] = {{b0, b0, b0, b0}, {b0, b0, b0, b0}};
      const unsigned char* trow = tile + 2 * rp * HC_ROW;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        WA3 wa;
        load_wa3(wa, tab, c, lane);
        long bt[2][4];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int j = 0; j < 4; ++j) bt[r][j] = *reinterpret_cast<const long*>(tile + HC_TOFF + (((2 * rp + r) * HC_ROW + off[j][c]) >> 1));
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(wa.wb, bt[r][j], acc[r][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          half8_t bh[4], bl[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bh[j] = *reinterpret_cast<const half8_t*>(trow + r * HC_ROW + off[j][c]);
            bl[j] = *reinterpret_cast<const half8_t*>(trow + r * HC_ROW + HC_PLANE + off[j][c]);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa.w[2], bh[j], acc[r][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa.w[1], bl[j], acc[r][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa.w[1], bh[j], acc[r][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa.w[0], bl[j], acc[r][j], 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[r][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa.w[0], bh[j], acc[r][j], 0, 0, 0);
          if (r == 0) __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int gy = y0 + 4 * wave + 2 * rp + r;
        if (gy >= H || gx + 3 >= W) continue;
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {  // channel pairs (oc0 + 2 ip, oc0 + 2 ip + 1): one noise call each
          if (oc0 + 2 * ip >= C) continue;
          float z8[8];
          if (POST == 2)  // canonical step-noise field, already scaled by sigma_t and rounded to fp16 (common.h)
            noise_pair_quad((uint64_t)(((size_t)n * C + oc0 + 2 * ip) * plane + (size_t)gy * W + gx), a.stream_id, seed, bm_k2(sg), z8);
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * ip + ii, oc = oc0 + i;
            const size_t e = ((size_t)n * C + oc) * plane + (size_t)gy * W + gx;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = acc[r][j][i] * inv_s;
            if (POST != 0) {
              const float4 t4 = *reinterpret_cast<const float4*>(a.xt + e);
              const float xt[4] = {t4.x, t4.y, t4.z, t4.w};
              if (POST == 1) {
                const float4 z4 = *reinterpret_cast<const float4*>(a.noise + e);
                const float z[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaf(sg, z[j], fmaf(c1, v[j], c2 * xt[j]));
              } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = z8[4 * ii + j] + fmaf(c1, v[j], c2 * xt[j]);
              }
            }
            *reinterpret_cast<float4*>(a.out + e) = make_float4(v[0], v[1], v[2], v[3]);
            if (POST != 0) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
          }
        }
      }
    }
  }
  if (POST != 0 && a.amax_out != nullptr) wave_amax_commit(amax, a.amax_out);
}

template <int NOISE>
__global__ __launch_bounds__(HC_NT, 3) void latent_step_h_kernel(const LatentArgs a) {
  constexpr int NT = HC_NT, TW = HC_TW, TH = HC_TH;
  // in-kernel noise: the step noise is an fp16 number (hi plane only) and the composite's third-term plane lies in the lo
  // region during phase 3: 2 x 23040 bytes.  Explicit noise (tests): three planes of the 18-row noise tile, 51840 bytes.
  constexpr int TILE_BYTES = NOISE == 1 ? 2 * HC_PLANE + HC_TPLANE : 2 * HL_PLANE5;
  __shared__ __align__(16) unsigned char tile[TILE_BYTES];
  __shared__ float s_ring[HL_RING];
  __shared__ float s_ab[8][2];
  __shared__ float s_red[NT / 64][16];

  fp16_ovfl_clamp();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const BlockId bid = xcd_block(a.xcd);
  const int n = bid.z;
  const int x0 = bid.x * TW, y0 = bid.y * TH;
  const int H = a.H, W = a.W;
  const unsigned plane = (unsigned)(H * W);
  const float c1 = a.sched[2], c2 = a.sched[3], sg = a.sched[4];
  const unsigned long long seed = (NOISE == 2 && a.seed_dev) ? *a.seed_dev : a.seed;
  const int ln = lane & 15, g = lane >> 4, ch = g & 1, rr = g >> 1;
  const int gx = x0 + 4 * ln;
  const int gy0 = y0 + 4 * wave + rr;
  const bool wave_live = y0 + 4 * wave < H;

  // ---------------- phase 1: A' = c1 * SiLU(GN(a)) with a 2-pixel halo -> LDS: hi / lo planes now, third terms kept in
  // registers (11 dwords of bf8) until the lo region is free for them (phase 3, first chunk) ----------------
  const int af32 = !a.a_bf16;
  const size_t abase = (size_t)n * 8 * plane;
  float4 qm[8], qr[2];
  float hh[2][2];
  const int r0 = tid >> 4, qx = tid & 15;
  const bool ok_m = (y0 - 2 + r0) >= 0 && (y0 - 2 + r0) < H && (x0 + 4 * qx) < W;
#pragma unroll
  for (int c = 0; c < 8; ++c)
    qm[c] = ok_m ? ld4(a.a_src, af32, abase + ((unsigned)c * plane + (unsigned)(y0 - 2 + r0) * (unsigned)W + (unsigned)(x0 + 4 * qx)))
                 : make_float4(0.f, 0.f, 0.f, 0.f);
  // rows 16..19: thread = (row 16 + (tid & 63) / 16, quad tid % 16, channel pair tid / 64)
  const int rrow = TH + ((tid & 63) >> 4), cpr = tid >> 6;
  const bool ok_r = (y0 - 2 + rrow) >= 0 && (y0 - 2 + rrow) < H && (x0 + 4 * qx) < W;
#pragma unroll
  for (int k = 0; k < 2; ++k)
    qr[k] = ok_r ? ld4(a.a_src, af32, abase + ((unsigned)(2 * cpr + k) * plane + (unsigned)(y0 - 2 + rrow) * (unsigned)W + (unsigned)(x0 + 4 * qx)))
                 : make_float4(0.f, 0.f, 0.f, 0.f);
  // halo columns x0-2, x0-1, x0+64, x0+65: item = (row idx / 16, column (idx / 4) & 3, channel pair idx & 3), 320 items
  bool ok_h[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = tid + j * NT;
    const int cp = idx & 3, s = (idx >> 2) & 3, r = idx >> 4;
    const int gy = y0 - 2 + r, gxh = s < 2 ? x0 - 2 + s : x0 + TW + (s - 2);
    ok_h[j] = idx < HL_LH5 * 16 && gy >= 0 && gy < H && gxh >= 0 && gxh < W;
    hh[j][0] = hh[j][1] = 0.f;
    if (ok_h[j]) {
      hh[j][0] = ld1(a.a_src, af32, abase + (unsigned)(2 * cp) * plane + (unsigned)gy * (unsigned)W + (unsigned)gxh);
      hh[j][1] = ld1(a.a_src, af32, abase + (unsigned)(2 * cp + 1) * plane + (unsigned)gy * (unsigned)W + (unsigned)gxh);
    }
  }
  if (tid < 8) {
    float A, B;
    gn_coeff(a.a_stat + (size_t)n * 16, tid, 2, a.inv_cnt, a.gamma[tid], a.beta[tid], &A, &B);
    s_ab[tid][0] = A;
    s_ab[tid][1] = B;
  }
  __syncthreads();
  // (the z r product form of conv8h's staging -- gn_silu_zr -- was tried here in round 5 and measured 2 % SLOWER in this kernel, whose
  // register budget is full: 162 -> 168 VGPRs; the plain form stays)
  auto act = [&](int c, float v, bool ok) { return ok ? c1 * silu_f(fmaf(s_ab[c][0], v, s_ab[c][1])) : 0.f; };
  uint2 tq[4];      // third terms of the main pass: pixel j -> 8 channels
  uint32_t tr2[2];  // rows 16..19: pixels (0, 1) and (2, 3) of the thread's channel pair, 16 bits each
  uint32_t th2;     // halo columns: the two items' channel pairs, 16 bits each
  {
    float e[8][4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      e[c][0] = act(c, qm[c].x, ok_m); e[c][1] = act(c, qm[c].y, ok_m); e[c][2] = act(c, qm[c].z, ok_m); e[c][3] = act(c, qm[c].w, ok_m);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4 hi, lo;
      float t[8];
      split3_pair(e[0][j], e[1][j], hi.x, lo.x, t[0], t[1]);
      split3_pair(e[2][j], e[3][j], hi.y, lo.y, t[2], t[3]);
      split3_pair(e[4][j], e[5][j], hi.z, lo.z, t[4], t[5]);
      split3_pair(e[6][j], e[7][j], hi.w, lo.w, t[6], t[7]);
      const int addr = r0 * HC_ROW + j * HC_PHASE + (qx + 1) * 16;
      *reinterpret_cast<uint4*>(tile + addr) = hi;
      *reinterpret_cast<uint4*>(tile + HL_PLANE5 + addr) = lo;
      tq[j] = make_uint2(bf8x4s(t[0], t[1], t[2], t[3]), bf8x4s(t[4], t[5], t[6], t[7]));
    }
  }
  {
    const float e0[4] = {qr[0].x, qr[0].y, qr[0].z, qr[0].w}, e1[4] = {qr[1].x, qr[1].y, qr[1].z, qr[1].w};
    tr2[0] = tr2[1] = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t hi, lo;
      float ta, tb;
      const float v0 = act(2 * cpr, e0[j], ok_r), v1 = act(2 * cpr + 1, e1[j], ok_r);
      split3_pair(v0, v1, hi, lo, ta, tb);
      const int addr = rrow * HC_ROW + j * HC_PHASE + (qx + 1) * 16 + cpr * 4;
      *reinterpret_cast<uint32_t*>(tile + addr) = hi;
      *reinterpret_cast<uint32_t*>(tile + HL_PLANE5 + addr) = lo;
      tr2[j >> 1] |= bf8x2s(ta, tb) << (16 * (j & 1));
    }
  }
  th2 = 0u;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = tid + j * NT;
    if (idx < HL_LH5 * 16) {
      const int cp = idx & 3, s = (idx >> 2) & 3, r = idx >> 4;
      uint32_t hi, lo;
      float ta, tb;
      const float v0 = act(2 * cp, hh[j][0], ok_h[j]), v1 = act(2 * cp + 1, hh[j][1], ok_h[j]);
      split3_pair(v0, v1, hi, lo, ta, tb);
      const int tpx = s < 2 ? s - 2 : TW + (s - 2);
      const int addr = hc_addr(r, tpx) + cp * 4;
      *reinterpret_cast<uint32_t*>(tile + addr) = hi;
      *reinterpret_cast<uint32_t*>(tile + HL_PLANE5 + addr) = lo;
      th2 |= bf8x2s(ta, tb) << (16 * j);
    }
  }
  // exact fp32 A' of the image's outermost ring inside this tile's window, for the border correction: only tiles that meet
  // the ring (wave-uniform test) run this, one (slot, channel) value per thread and round, recomputed from the source map
  if (y0 - 2 <= 0 || y0 + TH + 1 >= H - 1 || x0 == 0 || x0 + TW + 1 >= W - 1) {
    constexpr int NSLOT = 2 * (TW + 4) + 2 * HL_LH5;
    for (int i = tid; i < NSLOT * 8; i += NT) {
      const int slot = i >> 3, c = i & 7;
      int gy, gxp;
      if (slot < 2 * (TW + 4)) {
        gy = slot < TW + 4 ? 0 : H - 1;
        gxp = x0 - 2 + (slot < TW + 4 ? slot : slot - (TW + 4));
      } else {
        const int k = slot - 2 * (TW + 4);
        gxp = k < HL_LH5 ? 0 : W - 1;
        gy = y0 - 2 + (k < HL_LH5 ? k : k - HL_LH5);
      }
      const bool in_win = gy >= y0 - 2 && gy < y0 + TH + 2 && gxp >= x0 - 2 && gxp < x0 + TW + 2;
      if (in_win && gy >= 0 && gy < H && gxp >= 0 && gxp < W)
        s_ring[i] = act(c, ld1(a.a_src, af32, abase + (unsigned)c * plane + (unsigned)gy * (unsigned)W + (unsigned)gxp), true);
    }
  }
  // the third-term plane of the 20-row tile: 8-byte records at half the byte offsets, based at `tbase`
  auto write_t_plane = [&](unsigned char* tbase) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<uint2*>(tbase + ((r0 * HC_ROW + j * HC_PHASE + (qx + 1) * 16) >> 1)) = tq[j];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<uint16_t*>(tbase + ((rrow * HC_ROW + j * HC_PHASE + (qx + 1) * 16 + cpr * 4) >> 1)) = (uint16_t)(tr2[j >> 1] >> (16 * (j & 1)));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = tid + j * NT;
      if (idx < HL_LH5 * 16) {
        const int cp = idx & 3, s = (idx >> 2) & 3, r = idx >> 4;
        *reinterpret_cast<uint16_t*>(tbase + ((hc_addr(r, s < 2 ? s - 2 : TW + (s - 2)) + cp * 4) >> 1)) = (uint16_t)(th2 >> (16 * j));
      }
    }
  };
  // eps chunk 0 is requested now so that it arrives during the composite phase
  const float* __restrict__ np_ = NOISE == 1 ? a.noise + (size_t)n * a.C * plane : nullptr;
  TileRegs<TW, TH, NT, 8> R;
  float2 hreg = make_float2(0.f, 0.f);
  if (NOISE == 1) {
    stage_load<TW, TH, NT, 8, false>(R, np_, plane, W, H, W, x0, y0, tid);
    hreg = halo_load_h<false>(np_, plane, W, H, W, x0, y0, tid);
  }
  __syncthreads();

  // ---------------- phase 2: 5x5 composite on the f16 matrix pipe: the five f16 terms now, the bf8 term in phase 3 ----------------
  const float inv_s = a.wc5h[HL_W5TAB3], sc = a.wc5h[HL_W5TAB3 + 1];
  f32x4 acc[2][4];
  {
    const float4 b4 = *reinterpret_cast<const float4*>(a.bsum + 4 * ch);
    const float k = c1 * sc;
    const f32x4 b0 = {b4.x * k, b4.y * k, b4.z * k, b4.w * k};  // + c1 * bsum, in accumulator units
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[p][j] = b0;
  }
  // byte offset of the B operand of tap group c, row pair p, pixel group j in the 20-row tile
  auto addr5 = [&](int c, int p, int j) {
    const int t = min(4 * c + g, 29);  // tap slots 30, 31 carry zero weights: any valid address
    const int dyp = t / 5, dx = t - 5 * dyp;
    const int po = j + dx - 2;  // pixel 4*ln + po
    return (4 * wave + dyp + 2 * p) * HC_ROW + (ln + 1) * 16 + (po & 3) * HC_PHASE + (po >> 2) * 16;
  };
  if (wave_live) {
#pragma unroll 2   // fully unrolled, hipcc hoists the LDS reads of several tap groups and spills the deferred third terms
    for (int c = 0; c < 8; ++c) {
      half8_t w[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) w[k] = __builtin_bit_cast(half8_t, *reinterpret_cast<const uint4*>(a.wc5h + c * HC_WC3 + (k * 64 + lane) * 4));
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        // 16 operand registers in flight: the hi records feed three passes, the lo records are fetched behind them (the third
        // terms of phase 1 are live across this phase: with 32 the kernel spills)
        half8_t b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const half8_t*>(tile + addr5(c, p, j));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2], b[j], acc[p][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[1], b[j], acc[p][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], b[j], acc[p][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const half8_t*>(tile + HL_PLANE5 + addr5(c, p, j));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[1], b[j], acc[p][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], b[j], acc[p][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---------------- phase 2b: border fix (lanes whose strips touch the image border), from the exact ring values ----------------
  if (wave_live) {
    hl_border_fix(a, s_ring, x0, y0, acc[0], gy0, gx, gy0 - y0, gx - x0, 4 * ch, c1, sc);
    hl_border_fix(a, s_ring, x0, y0, acc[1], gy0 + 2, gx, gy0 + 2 - y0, gx - x0, 4 * ch, c1, sc);
  }

  // the composite's bf8 term: two fenced passes per tap group (row pair 0, row pair 1); every instruction of another input
  // type on the same registers is a whole pass or more away (conv8h_kernels.h conv_tile_mfma3)
  auto composite_third_term = [&](const unsigned char* tbase) {
    static_for<0, 8>([&](auto CC) {
      constexpr int c = decltype(CC)::value;
      const long wb = *reinterpret_cast<const long*>(a.wc5h + c * HC_WC3 + 768 + lane * 2);
      long bt[2][4];
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) bt[p][j] = *reinterpret_cast<const long*>(tbase + (addr5(c, p, j) >> 1));
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(wb, bt[p][j], acc[p][j], 0, 0, 0);
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  };
  // The third terms move into the lo region (free once phase 2 has read it) and are applied from there, before the noise
  // chunks: inside the chunk loop (first iteration only) the 64 operand offsets of the pass are hoisted out of the loop as
  // invariants and the kernel spills 44 registers across every iteration's noise generation.
  __syncthreads();
  write_t_plane(tile + HL_PLANE5);
  __syncthreads();
  if (wave_live) composite_third_term(tile + HL_PLANE5);

  // ---------------- phase 3: s * (W_x (*) eps), eps in chunks of 8 channels through LDS ----------------
  int off[4][3];
  hc_lane_offsets(off, wave, lane);
  const int nchunk = a.C / 8;
  if (NOISE == 2) {
    // the composite's third terms move into the lo region (free since phase 2; the noise tile is one 18-row hi plane).
    // Outside the chunk loop: written inside it (first iteration only) the 11 registers stay live across every iteration's
    // noise generation and the kernel spills 66 registers there.
    __syncthreads();
    write_t_plane(tile + HL_PLANE5);
  }
#pragma unroll 1
  for (int cc = 0; cc < nchunk; ++cc) {
    __syncthreads();  // previous LDS contents (third-term plane / previous chunk) are no longer read
    const float* __restrict__ wtab = a.wxh + (size_t)cc * HC_WTAB3;
    if (NOISE == 1) {
      stage_store_h<false, true>(tile, R, hreg, H, W, x0, y0, nullptr, tid, sg);
      if (cc + 1 < nchunk) {
        stage_load<TW, TH, NT, 8, false>(R, np_ + (size_t)(cc + 1) * 8 * plane, plane, W, H, W, x0, y0, tid);
        hreg = halo_load_h<false>(np_ + (size_t)(cc + 1) * 8 * plane, plane, W, H, W, x0, y0, tid);
      }
    } else {
      // canonical step-noise field (common.h): nu = fp16(sigma_t * z), one Philox call per (channel pair, aligned quad)
      // gives the packed (even, odd channel) dword of each of the quad's four pixel records.  Only the hi plane is
      // written: nu IS an fp16 number, so the noise convolution needs one operand term (conv_tile_mfma3_hionly).
      // No branch around the generator: lanes whose quad lies outside the image run it with k2 = 0 and get (+-0, +-0).
      // The remainder rows and the halo columns need only two / one of a quad's four words: Box-Muller on those only.
      const float k2 = bm_k2(sg);
      auto words = [&](int cp, int r, int gx2, NoiseWords& w) -> float {
        const int gy2 = y0 - 1 + r;
        noise_words((uint64_t)(((size_t)n * a.C + (size_t)cc * 8 + 2 * cp) * plane + (size_t)gy2 * W + gx2), a.stream_id, seed, w);
        return (gy2 >= 0 && gy2 < H && gx2 >= 0 && gx2 < W) ? k2 : 0.f;
      };
      {
        uint32_t h[4][4];
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
          NoiseWords w;
          const float kk = words(cp, r0, x0 + 4 * qx, w);
#pragma unroll
          for (int j = 0; j < 4; ++j) h[cp][j] = bm_pair_h(w.ur[j], w.w[j], kk);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<uint4*>(tile + r0 * HC_ROW + j * HC_PHASE + (qx + 1) * 16) = make_uint4(h[0][j], h[1][j], h[2][j], h[3][j]);
      }
      {  // rows 16, 17: thread = (channel tid / 32, row 16 + (tid / 16 & 1), quad); the even channel's thread writes
         // the pair's dwords of pixels 0, 1, the odd channel's thread those of pixels 2, 3 (whole dwords, see hc_store_rem)
        const int cr = tid >> 5, rrw = TH + ((tid >> 4) & 1);
        NoiseWords w;
        const float kk = words(cr >> 1, rrw, x0 + 4 * qx, w);
        const bool odd = (cr & 1) != 0;
        const int jb = odd ? 2 : 0;
        const uint32_t wsel[2] = {odd ? w.w[2] : w.w[0], odd ? w.w[3] : w.w[1]};
        const float usel[2] = {odd ? w.ur[2] : w.ur[0], odd ? w.ur[3] : w.ur[1]};
#pragma unroll
        for (int k = 0; k < 2; ++k)
          *reinterpret_cast<uint32_t*>(tile + rrw * HC_ROW + (jb + k) * HC_PHASE + (qx + 1) * 16 + (cr >> 1) * 4) = bm_pair_h(usel[k], wsel[k], kk);
      }
      if (tid < HC_LH * 8) {
        const int cp = tid & 3, side = (tid >> 2) & 1, r = tid >> 3;
        NoiseWords w;  // the aligned quad that owns the halo pixel: x0-4..x0-1 (pixel 3) or x0+64.. (pixel 0)
        const float kk = words(cp, r, side ? x0 + TW : x0 - 4, w);
        *reinterpret_cast<uint32_t*>(tile + hc_addr(r, side ? HC_TW : -1) + cp * 4) = bm_pair_h(side ? w.ur[0] : w.ur[3], side ? w.w[0] : w.w[3], kk);
      }
    }
    __syncthreads();
    if (wave_live) {
      if (NOISE == 1) {
        conv_tile_mfma3(tile, wtab, acc, off, lane);
      } else {
        conv_tile_mfma3_hionly(tile, wtab, acc, off, lane);
      }
    }
  }

  // ---------------- epilogue: combine, store in place, statistics for the next GroupNorm ----------------
  float part[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) part[i] = 0.f;
  if (wave_live) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int gy = gy0 + 2 * p;
      if (gy < H && gx + 3 < W) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const size_t e = ((size_t)n * 8 + 4 * ch + i) * plane + (size_t)gy * W + gx;
          const float4 k4 = *reinterpret_cast<const float4*>(a.kmap + e);
          const float4 h4 = *reinterpret_cast<const float4*>(a.hs0 + e);
          const float kk[4] = {k4.x, k4.y, k4.z, k4.w}, hv[4] = {h4.x, h4.y, h4.z, h4.w};
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = kk[j] + c2 * (hv[j] - kk[j]) + acc[p][j][i] * inv_s;
          *reinterpret_cast<float4*>(a.hs0 + e) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
          for (int j = 0; j < 4; ++j) { part[i] += v[j]; part[4 + i] = fmaf(v[j], v[j], part[4 + i]); }
        }
      }
    }
  }
  hc_stats_commit(part, s_red, a.hs0_stat + (size_t)n * 16, tid);
}

}  // namespace gc
