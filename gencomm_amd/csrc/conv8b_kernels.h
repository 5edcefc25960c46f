// bf16 denoise mode (BASELINE.json config 5: "bf16 denoise"; the reference's --half / autocast path, train_ddp.py:139-141, :173-176):
// the 8-channel intermediates of the UNet are STORED as bf16 (16 B per pixel instead of 32: these layers are HBM-bound, so
// bytes are time) and every 3x3 product is ONE v_mfma_f32_16x16x32_bf16 (fp32 accumulation) instead of three fp16 split
// products.  GroupNorm statistics stay f64 sums of the values actually stored; GroupNorm + SiLU, bias, residual adds and the
// 1x1 nin_shortcut are fp32.  The sampler's carried state (hs0 = conv_in(x_t), the k map) stays fp32 -- it is re-read for 20
// steps -- so sources / residuals / destination each carry an "is fp32" flag.  Same mapping, LDS layout (one plane instead of
// hi + lo) and tile shape as conv8h_kernels.h.  Needs W % 4 == 0 at every level (no scalar staging path in this mode).
// Accuracy is that of bf16 storage (2^-9 relative per stored value), reported against the fp32 oracle in
// tests/test_gpu_bf16.py -- a separate `dtype: "bf16"` benchmark line, never the fp32 headline.
#pragma once
#include "conv8h_kernels.h"

namespace gc {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
constexpr int BC_WTAB = 3 * 64 * 4;  // dwords of one prepared 8-input-channel bf16 weight table

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((float2_t){a, b}, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float round_bf16(float a) { return bf_lo(pk_bf16(a, 0.f)); }

// element loads from a map that is fp32 (F32) or bf16; idx = element index.  The dtype is a template parameter of the loaders
// so that the (uniform) fp32 / bf16 decision is taken ONCE per tile, outside the unrolled load sequences: with the branch
// inside, every load sat behind its own s_cbranch and the 9 .. 32 loads of a phase could not be issued back to back.
template <bool F32>
__device__ __forceinline__ float4 ld4t(const void* __restrict__ p, size_t idx) {
  if constexpr (F32) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + idx);
  } else {
    const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p) + idx);
    return make_float4(bf_lo(v.x), bf_hi(v.x), bf_lo(v.y), bf_hi(v.y));
  }
}
template <bool F32>
__device__ __forceinline__ float2 ld2t(const void* __restrict__ p, size_t idx) {
  if constexpr (F32) {
    return *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(p) + idx);
  } else {
    const uint32_t v = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(p) + idx);
    return make_float2(bf_lo(v), bf_hi(v));
  }
}
template <bool F32>
__device__ __forceinline__ float ld1t(const void* __restrict__ p, size_t idx) {
  if constexpr (F32) return reinterpret_cast<const float*>(p)[idx];
  else return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(p)[idx] << 16);
}
__device__ __forceinline__ float4 ld4(const void* __restrict__ p, int f32, size_t idx) { return f32 ? ld4t<true>(p, idx) : ld4t<false>(p, idx); }
__device__ __forceinline__ float ld1(const void* __restrict__ p, int f32, size_t idx) { return f32 ? ld1t<true>(p, idx) : ld1t<false>(p, idx); }
__device__ __forceinline__ void st4(void* __restrict__ p, int f32, size_t idx, float a, float b, float c, float d) {
  if (f32) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + idx) = make_float4(a, b, c, d);
  else *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p) + idx) = make_uint2(pk_bf16(a, b), pk_bf16(c, d));
}

struct Conv8BArgs {
  const void* src[2];     // [n][8][Hin][Win], fp32 or bf16
  int src_f32[2];
  const double* sstat[2];
  const float* gamma;
  const float* beta;
  const float* wb;        // prepared bf16 A-operand tables, NSRC * BC_WTAB dwords
  const float* bias;      // [8]
  const void* res[2];
  int res_f32[2];
  const float* ninw;      // prepared nin_shortcut [16][8] = (ic, oc), fp32
  void* dst;              // [n][8][H][W]
  int dst_f32;
  double* dstat;
  double inv_cnt;
  int H, W, Hin, Win, xcd;
};

// this thread's share of the 18 x 66 x 8 tile (same ownership as stage_load<64, 16, 256, 8, UP> + halo_load_h)
template <bool UP, bool F32>
__device__ __forceinline__ void stage_load_t(TileRegs<HC_TW, HC_TH, HC_NT, 8>& R, float2& hreg, const void* __restrict__ sp,
                                             size_t base, unsigned plane_in, int Win, int H, int W, int x0, int y0, int tid) {
  auto quad = [&](int c, int r, int qx) {
    const int gy = y0 - 1 + r, gx = x0 + 4 * qx;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gy >= 0 && gy < H && gx < W) {
      if (!UP) {
        out = ld4t<F32>(sp, base + (size_t)c * plane_in + (size_t)gy * Win + gx);
      } else {
        const float2 t = ld2t<F32>(sp, base + (size_t)c * plane_in + (size_t)(gy >> 1) * Win + (gx >> 1));
        out = make_float4(t.x, t.x, t.y, t.y);
      }
    }
    return out;
  };
  const int r0 = tid >> 4, qx = tid & 15;
  if constexpr (UP && !F32) {
    // two source pixels per quad: one dword load each, all eight issued before the first use
    const int gy = y0 - 1 + r0, gx = x0 + 4 * qx;
    const bool ok = gy >= 0 && gy < H && gx < W;
    uint32_t u[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
      u[c] = ok ? *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(sp) + base + (size_t)c * plane_in + (size_t)(gy >> 1) * Win + (gx >> 1)) : 0u;
#pragma unroll
    for (int c = 0; c < 8; ++c) { R.v[c].x = bf_lo(u[c]); R.v[c].y = bf_lo(u[c]); R.v[c].z = bf_hi(u[c]); R.v[c].w = bf_hi(u[c]); }
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) R.v[c] = quad(c, r0, qx);
  }
  R.vr = quad(tid >> 5, HC_TH + ((tid >> 4) & 1), qx);
  hreg = make_float2(0.f, 0.f);
  if (tid < HC_LH * 8) {
    const int cp = tid & 3, side = (tid >> 2) & 1, r = tid >> 3;
    const int gy = y0 - 1 + r, gx = side ? x0 + HC_TW : x0 - 1;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      const size_t o = UP ? (size_t)(gy >> 1) * Win + (gx >> 1) : (size_t)gy * Win + gx;
      hreg.x = ld1t<F32>(sp, base + (size_t)(2 * cp) * plane_in + o);
      hreg.y = ld1t<F32>(sp, base + (size_t)(2 * cp + 1) * plane_in + o);
    }
  }
}
template <bool UP>
__device__ __forceinline__ void stage_load_b(TileRegs<HC_TW, HC_TH, HC_NT, 8>& R, float2& hreg, const void* __restrict__ sp, int f32,
                                             size_t base, unsigned plane_in, int Win, int H, int W, int x0, int y0, int tid) {
  if (f32) stage_load_t<UP, true>(R, hreg, sp, base, plane_in, Win, H, W, x0, y0, tid);
  else stage_load_t<UP, false>(R, hreg, sp, base, plane_in, Win, H, W, x0, y0, tid);
}

// GroupNorm + SiLU (GN) or plain copy, round to bf16, write the single-plane tile
template <bool GN>
__device__ __forceinline__ void stage_store_b(unsigned char* tile, const TileRegs<HC_TW, HC_TH, HC_NT, 8>& R, float2 hreg, int H, int W,
                                              int x0, int y0, const float (*ab)[2], int tid) {
  {
    const int r0 = tid >> 4, qx = tid & 15;
    const int gy = y0 - 1 + r0, gx = x0 + 4 * qx;
    const bool ok = gy >= 0 && gy < H && gx < W;
    if constexpr (GN) {
      float e[8][4];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        e[c][0] = R.v[c].x; e[c][1] = R.v[c].y; e[c][2] = R.v[c].z; e[c][3] = R.v[c].w;
        const float A = ok ? ab[c][0] : 0.f, B = ok ? ab[c][1] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) e[c][j] = silu_f(fmaf(A, e[c][j], B));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<uint4*>(tile + r0 * HC_ROW + j * HC_PHASE + (qx + 1) * 16) =
            make_uint4(pk_bf16(e[0][j], e[1][j]), pk_bf16(e[2][j], e[3][j]), pk_bf16(e[4][j], e[5][j]), pk_bf16(e[6][j], e[7][j]));
    } else {
      // plain copy: straight from the load registers, one pixel record at a time (a per-thread array of copies was parked in
      // scratch by the optimiser: 128 B per thread in the Upsample variant)
      auto comp = [](const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); };
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<uint4*>(tile + r0 * HC_ROW + j * HC_PHASE + (qx + 1) * 16) =
            make_uint4(pk_bf16(comp(R.v[0], j), comp(R.v[1], j)), pk_bf16(comp(R.v[2], j), comp(R.v[3], j)),
                       pk_bf16(comp(R.v[4], j), comp(R.v[5], j)), pk_bf16(comp(R.v[6], j), comp(R.v[7], j)));
    }
  }
  {  // rows 16, 17: thread = (channel tid / 32, row, quad); channel pairs meet through lane ^ 32, whole dwords are written
    const int cr = tid >> 5, rr = HC_TH + ((tid >> 4) & 1), qx = tid & 15;
    const int gy = y0 - 1 + rr, gx = x0 + 4 * qx;
    const bool ok = gy >= 0 && gy < H && gx < W;
    float e[4] = {R.vr.x, R.vr.y, R.vr.z, R.vr.w};
    if (GN) {
      const float A = ok ? ab[cr][0] : 0.f, B = ok ? ab[cr][1] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] = silu_f(fmaf(A, e[j], B));
    }
    const bool odd = (cr & 1) != 0;
    const float s0 = odd ? e[0] : e[2], s1 = odd ? e[1] : e[3];
    const float p0 = __shfl_xor(s0, 32, 64), p1 = __shfl_xor(s1, 32, 64);
    const float m0 = odd ? e[2] : e[0], m1 = odd ? e[3] : e[1];
    const int jb = odd ? 2 : 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float mine = k ? m1 : m0, theirs = k ? p1 : p0;
      *reinterpret_cast<uint32_t*>(tile + rr * HC_ROW + (jb + k) * HC_PHASE + (qx + 1) * 16 + (cr >> 1) * 4) =
          pk_bf16(odd ? theirs : mine, odd ? mine : theirs);  // low half = even channel
    }
  }
  if (tid < HC_LH * 8) {
    const int cp = tid & 3, side = (tid >> 2) & 1, r = tid >> 3;
    float e0 = hreg.x, e1 = hreg.y;
    if (GN) {
      const int gy = y0 - 1 + r, gx = side ? x0 + HC_TW : x0 - 1;
      const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
      e0 = ok ? silu_f(fmaf(ab[2 * cp][0], e0, ab[2 * cp][1])) : 0.f;
      e1 = ok ? silu_f(fmaf(ab[2 * cp + 1][0], e1, ab[2 * cp + 1][1])) : 0.f;
    }
    *reinterpret_cast<uint32_t*>(tile + hc_addr(r, side ? HC_TW : -1) + cp * 4) = pk_bf16(e0, e1);
  }
}

__device__ __forceinline__ void load_wb(bf16x8_t (&wa)[3], const float* __restrict__ tab, int lane) {
#pragma unroll
  for (int c = 0; c < 3; ++c) wa[c] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(tab + (c * 64 + lane) * 4));
}
__device__ __forceinline__ void conv_tile_mfma_b(const unsigned char* tile, const bf16x8_t (&wa)[3], f32x4 (&acc)[2][4], const int (&off)[4][3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      bf16x8_t b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(tile + off[j][c] + p * 2 * HC_ROW);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c], b[j], acc[p][j], 0, 0, 0);
    }
}

template <int NSRC, bool GN, bool UP, int RES>
__global__ __launch_bounds__(HC_NT, RES == 2 ? 3 : 4) void conv8b_kernel(const Conv8BArgs a) {
  constexpr int NT = HC_NT, TW = HC_TW, TH = HC_TH;
  __shared__ __align__(16) unsigned char tile[HC_PLANE];
  __shared__ float s_ab[16][2];
  __shared__ float s_red[NT / 64][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const BlockId bid = xcd_block(a.xcd);
  const int n = bid.z;
  const int x0 = bid.x * TW, y0 = bid.y * TH;
  const unsigned plane_in = (unsigned)(a.Hin * a.Win);
  const size_t plane = (size_t)a.H * a.W;
  const int ln = lane & 15, g = lane >> 4, ch = g & 1, rr = g >> 1;
  const int gx = x0 + 4 * ln, gy0 = y0 + 4 * wave + rr;
  const bool wave_live = y0 + 4 * wave < a.H;
  const bool col_ok = gx + 3 < a.W;

  bf16x8_t wa[3];
  load_wb(wa, a.wb, lane);
  const float4 bias4 = *reinterpret_cast<const float4*>(a.bias + 4 * ch);
  TileRegs<TW, TH, NT, 8> R;
  float2 hreg;
  if (UP) stage_load_t<UP, false>(R, hreg, a.src[0], (size_t)n * 8 * plane_in, plane_in, a.Win, a.H, a.W, x0, y0, tid);  // never the fp32 state map
  else stage_load_b<UP>(R, hreg, a.src[0], a.src_f32[0], (size_t)n * 8 * plane_in, plane_in, a.Win, a.H, a.W, x0, y0, tid);
  if (GN) {
    if (tid < NSRC * 8) {
      const int s = tid >> 3, c = tid & 7;
      float A, B;
      gn_coeff(a.sstat[s] + (size_t)n * 16, c, 2 * NSRC, a.inv_cnt, a.gamma[tid], a.beta[tid], &A, &B);
      s_ab[tid][0] = A;
      s_ab[tid][1] = B;
    }
    __syncthreads();
  }
  f32x4 acc[2][4];
  {
    const f32x4 b0 = {bias4.x, bias4.y, bias4.z, bias4.w};
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[p][j] = b0;
  }
  stage_store_b<GN>(tile, R, hreg, a.H, a.W, x0, y0, &s_ab[0], tid);
  int off[4][3];
  hc_lane_offsets(off, wave, lane);
  float resv[RES == 1 ? 2 : 1][4][4];
  if (RES == 1 && wave_live) {
    auto load_res = [&](auto F) {
      constexpr bool F32 = decltype(F)::value;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int gy = gy0 + 2 * p;
          float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
          if (col_ok && gy < a.H) r = ld4t<F32>(a.res[0], ((size_t)n * 8 + 4 * ch + i) * plane + (size_t)gy * a.W + gx);
          resv[p][i][0] = r.x; resv[p][i][1] = r.y; resv[p][i][2] = r.z; resv[p][i][3] = r.w;
        }
    };
    if (a.res_f32[0]) load_res(std::true_type{}); else load_res(std::false_type{});
  }
  if (NSRC == 2) stage_load_b<UP>(R, hreg, a.src[1], a.src_f32[1], (size_t)n * 8 * plane_in, plane_in, a.Win, a.H, a.W, x0, y0, tid);
  __syncthreads();
  if (wave_live) conv_tile_mfma_b(tile, wa, acc, off);
  if (NSRC == 2) {
    __syncthreads();
    stage_store_b<GN>(tile, R, hreg, a.H, a.W, x0, y0, &s_ab[8], tid);
    load_wb(wa, a.wb + BC_WTAB, lane);
    __syncthreads();
    if (wave_live) conv_tile_mfma_b(tile, wa, acc, off);
  }

  float part[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) part[i] = 0.f;
  if (wave_live) {
    // the epilogue works in place on the accumulators (acc[p][j][i] = row pair p, pixel j, channel i): no second copy
    if (RES == 1) {
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[p][j][i] += resv[p][i][j];
    }
    if (RES == 2) {  // 1x1 nin_shortcut over the 16 raw input channels of the block
      auto nin_src = [&](int s, auto F) {
        constexpr bool F32 = decltype(F)::value;
#pragma unroll 8
        for (int c8 = 0; c8 < 8; ++c8) {
          const float4 wv4 = *reinterpret_cast<const float4*>(a.ninw + (8 * s + c8) * 8 + 4 * ch);
          const float wv[4] = {wv4.x, wv4.y, wv4.z, wv4.w};
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int gy = gy0 + 2 * p;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col_ok && gy < a.H) t = ld4t<F32>(a.res[s], ((size_t)n * 8 + c8) * plane + (size_t)gy * a.W + gx);
            const float r[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[p][j][i] = fmaf(wv[i], r[j], acc[p][j][i]);
          }
        }
      };
      for (int s = 0; s < 2; ++s) {
        if (a.res_f32[s]) nin_src(s, std::true_type{}); else nin_src(s, std::false_type{});
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int gy = gy0 + 2 * p;
      if (!(col_ok && gy < a.H)) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = a.dst_f32 ? acc[p][j][i] : round_bf16(acc[p][j][i]);  // statistics of what is stored
        st4(a.dst, a.dst_f32, ((size_t)n * 8 + 4 * ch + i) * plane + (size_t)gy * a.W + gx, v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { part[i] += v[j]; part[4 + i] = fmaf(v[j], v[j], part[4 + i]); }
      }
    }
  }
  if (a.dstat != nullptr) hc_stats_commit(part, s_red, a.dstat + (size_t)n * 16, tid);
}

// OIHW [8][IC][3][3] -> IC/8 bf16 A-operand tables [c 3][lane 64][4 dwords] (conv8h's row / tap mapping, no scale: bf16 has the
// fp32 exponent range)
__global__ __launch_bounds__(256) void prep_conv8b_kernel(const float* __restrict__ w, float* __restrict__ dst, int IC) {
  uint32_t* __restrict__ out = reinterpret_cast<uint32_t*>(dst);
  for (int i = threadIdx.x; i < (IC / 8) * BC_WTAB; i += 256) {
    const int s = i / BC_WTAB, rem = i - s * BC_WTAB;
    const int d = rem & 3, l = (rem >> 2) & 63, c = rem >> 8;
    const int mrow = l & 15, kg = l >> 4, r = mrow >> 3, oc = mrow & 7;
    const int dyp = hc_tap_row(kg), dx = c, dy = dyp - r;
    float x[2];
    for (int e = 0; e < 2; ++e) {
      const int ic = s * 8 + 2 * d + e;
      x[e] = (dy >= 0 && dy <= 2) ? w[((oc * IC + ic) * 3 + dy) * 3 + dx] : 0.f;
    }
    out[i] = pk_bf16(x[0], x[1]);
  }
}

}  // namespace gc
