// Weight gradient of a WIDE 3x3 stride-1 convolution on the f16 matrix pipe with three-term operands (round 5): the default behind
// wgrad3x3_wide_enqueue when the caller gives scratch (train_kernels.h; the exact-fp32 kernel stays for stride 2, for
// GENCOMM_MODE_ARITH = 1 and for the atomic form).  Same decomposition as wgrad3x3_wide_kernel<1, 2> -- a workgroup owns 64 output x 64
// input channels and walks `chunk` pixel tiles of 2 rows x 32 columns of one sample, nine 32 x 32 accumulators per wave (one per tap),
// partial sums stored per group and added up by wgrad3x3_wide_reduce_kernel -- with the matrix work of a tile cut from
// 288 x 64 cycles (v_mfma_f32_32x32x2_f32) to 216 x 32 (v_mfma_f32_32x32x16_f16, six per product block).
//
//   dW[co][ci][ky][kx] = sum_{n, y, x} dY[n][co][y][x] X[n][ci][y + ky - pad][x + kx - pad]:  GEMM rows = co (A = dY), columns = ci (B = X),
//   K = the tile's 64 pixels in four steps of 16 (a lane holds 8 CONSECUTIVE pixels of its row / column: both operands are pixel-contiguous).
// Arithmetic as conv_h3_kernels.h with the third activation term carried in fp16 instead of bf8: A takes the weight role (fp16 w1 + w2 + w3
// exact, wq = fp16(v 2^-20)), B the activation role (fp16 hi + lo, tq = fp16(rest 2^20): the rest is a power of two, exact);
// acc += w1 hi + w1 lo + w2 hi + w2 lo + w3 hi + wq tq -- six instructions of ONE type per product block, the third term exact, all
// seven operand planes of one shape (one staging / read / shift path).  (The nine accumulators of a wave take every term; the stale-SrcC
// hazard between matrix instructions of different input types that conv8h_kernels.h orders its 16x16x32 instructions around does not
// exist for the 32x32x16 forms -- tools/probes/mfma_mixed_dep32_probe.hip -- so this is a simplification, not a workaround.)
// Both operands ride under RUNNING
// power-of-two scales (largest |dY| resp. |X| seen so far by the workgroup in [2^13, 2^14): gradients are scaled up); when an
// exponent grows the 144 accumulators are brought to the new scale with v_ldexp (exact).
// The horizontal tap shift: a lane's 8 pixels for kx = 1 are one aligned 16-byte record of the X image (rows stored with an 8-pixel
// left margin); for kx = 0 / 2 they straddle it by one pixel -- the record, the dword left and the dword right of it are read ONCE per
// (ky, step) and the two shifted operands are formed with v_alignbit (15 vector instructions per 18 matrix instructions).
#pragma once
#include "conv8h_kernels.h"   // split3_pair, half8_t, HC_TSCALE
#include "train_kernels.h"

namespace gc {

struct WgH3 {
  static constexpr int SA = 144;                           // bytes per dY row of an fp16 plane (64 px + pad)
  static constexpr int A_PLANE = 64 * SA;                  // planes w1, w2, w3, wq
  static constexpr int SBR = 96, SB = 4 * SBR + 16;        // X: bytes per image row (48 px) and per channel (4 rows + pad) of an fp16 plane
  static constexpr int B_PLANE = 64 * SB;                  // planes hi, lo, tq
  static constexpr int OFF_A2 = A_PLANE, OFF_A3 = 2 * A_PLANE, OFF_A4 = 3 * A_PLANE, OFF_BH = 4 * A_PLANE, OFF_BL = OFF_BH + B_PLANE,
                       OFF_BQ = OFF_BL + B_PLANE, OFF_MAX = OFF_BQ + B_PLANE, SMEM = OFF_MAX + 2 * 4 * 2 * 4;
};

__global__ __launch_bounds__(256, 1) void wgrad3x3_h3_kernel(const Wgrad3x3Args a) {
  using L = WgH3;
  constexpr int TR = 2, TC = 32;
  extern __shared__ __align__(16) unsigned char wg_smem[];
  float (*s_max)[4][2] = reinterpret_cast<float (*)[4][2]>(wg_smem + L::OFF_MAX);   // [parity][wave][dY, X]
  using f32x16t = __attribute__((ext_vector_type(16))) float;
  fp16_ovfl_clamp();
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
  const int n = blockIdx.x / a.chunks, ck = blockIdx.x - n * a.chunks;
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
  const int mh = wv & 1, nh = wv >> 1;
  const float* __restrict__ dyn = a.dy + (size_t)n * a.Cout * a.Ho * a.Wo;
  const float* __restrict__ xn = a.x + (size_t)n * a.Cin * a.Hi * a.Wi;
  f32x16t acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float bsum = 0.f;
  const int t_end = min((ck + 1) * a.chunk, a.tiles);

  // staging: dY -- thread = (channel row tid / 4, 16 consecutive pixels of the tile's K order); X -- the 256 (channel, image row) lines of the
  // tile, 32 aligned input columns ix0 + 1 .. ix0 + 32 (ix0 = ox0 - pad) each: EIGHT lanes per line (a lane = 4 columns: one 128-byte
  // line per 8 lanes, 8 lines per wave instruction -- one thread per line made every lane of a wave walk a cache line of its own and
  // cost 12 k of the tile's 19 k cycles), line = 32 k + tid / 8 for k = 0 .. 7; the two halo columns (ix0, ix0 + 33) of line tid as scalars
  float va[16], vx[8][4], vh[2];
  unsigned xmask = 0u;   // bit k: line k of this thread is live (set by load_tile, applied where the values are used)
  const int srow = tid >> 2, seg = tid & 3;
  const int xl = tid >> 3, xq = tid & 7;
  auto load_tile = [&](int tile) {
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = ty * TR, ox0 = tx * TC;
    {
      const int co = co0 + srow, oy = oy0 + (seg >> 1), ox = ox0 + 16 * (seg & 1);
      const bool row_ok = co < a.Cout && oy < a.Ho;
      const float* __restrict__ pa = dyn + ((size_t)(row_ok ? co : 0) * a.Ho + (row_ok ? oy : 0)) * a.Wo + ox;
      if (row_ok && ox + 16 <= a.Wo && (a.Wo & 3) == 0) {
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
          const float4 u = *reinterpret_cast<const float4*>(pa + e);
          va[e] = u.x; va[e + 1] = u.y; va[e + 2] = u.z; va[e + 3] = u.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) va[e] = (row_ok && ox + e < a.Wo) ? pa[e] : 0.f;
      }
    }
    {
      const int ix0 = ox0 - a.pad, iyb = oy0 - a.pad;
      xmask = 0u;
      const bool fast = ix0 + 1 >= 0 && ix0 + 33 <= a.Wi && ((ix0 + 1) & 3) == 0 && (a.Wi & 3) == 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int line = 32 * k + xl, ci = ci0 + (line >> 2), iy = iyb + (line & 3);
        const bool row_ok = ci < a.Cin && iy >= 0 && iy < a.Hi;
        const float* __restrict__ px = xn + ((size_t)(row_ok ? ci : 0) * a.Hi + (row_ok ? iy : 0)) * a.Wi + (ix0 + 1 + 4 * xq);
        if (fast) {          // branch-free: a line outside the image / beyond Cin reads (0, 0) of its sample and is zeroed when USED (xmask)
          const float4 u = *reinterpret_cast<const float4*>(px);
          vx[k][0] = u.x; vx[k][1] = u.y; vx[k][2] = u.z; vx[k][3] = u.w;
          xmask |= (row_ok ? 1u : 0u) << k;
        } else {
          xmask |= 1u << k;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ix = ix0 + 1 + 4 * xq + e;
            vx[k][e] = (row_ok && ix >= 0 && ix < a.Wi) ? px[e] : 0.f;
          }
        }
      }
      {
        const int ci = ci0 + (tid >> 2), iy = iyb + (tid & 3);
        const bool row_ok = ci < a.Cin && iy >= 0 && iy < a.Hi;
        const float* __restrict__ px = xn + ((size_t)(row_ok ? ci : 0) * a.Hi + (row_ok ? iy : 0)) * a.Wi + ix0;
        vh[0] = (row_ok && ix0 >= 0 && ix0 < a.Wi) ? px[0] : 0.f;
        vh[1] = (row_ok && ix0 + 33 >= 0 && ix0 + 33 < a.Wi) ? px[33] : 0.f;
      }
    }
  };

  int eA = 0, eB = 0;            // exponents of the running scales 2^eA (dY), 2^eB (X); the accumulators hold sums of (dY 2^eA)(X 2^eB)
  float runA = 0.f, runB = 0.f;
  bool scaled = false;
  auto exp_for = [](float run, int prev) {
    if (!(run > 0.f) || !(run < 3.0e38f)) return prev;
    int e;
    (void)frexpf(run, &e);
    return max(min(14 - e, 126), -126);
  };

  int tile = ck * a.chunk;
  if (tile < t_end) load_tile(tile);
  for (; tile < t_end; ++tile) {
    // this tile's largest |dY| and |X| over the workgroup (published before the barrier that also retires the previous tile's matrix phase)
    {
      float ma = 0.f, mb = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) ma = fmaxf(ma, fabsf(va[e]));
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float m4 = fmaxf(fmaxf(fabsf(vx[k][0]), fabsf(vx[k][1])), fmaxf(fabsf(vx[k][2]), fabsf(vx[k][3])));
        mb = fmaxf(mb, ((xmask >> k) & 1u) ? m4 : 0.f);
      }
      mb = fmaxf(mb, fmaxf(fabsf(vh[0]), fabsf(vh[1])));
      ma = wave_max_nonneg(ma);
      mb = wave_max_nonneg(mb);
      if (l == 0) { s_max[tile & 1][wv][0] = ma; s_max[tile & 1][wv][1] = mb; }
    }
    __syncthreads();
    {
      const float (*m)[2] = s_max[tile & 1];
      runA = fmaxf(runA, fmaxf(fmaxf(m[0][0], m[1][0]), fmaxf(m[2][0], m[3][0])));
      runB = fmaxf(runB, fmaxf(fmaxf(m[0][1], m[1][1]), fmaxf(m[2][1], m[3][1])));
      const int nA = exp_for(runA, eA), nB = exp_for(runB, eB);
      if (scaled && (nA != eA || nB != eB)) {     // an exponent grew: what is accumulated follows (exact)
        const int d = (nA - eA) + (nB - eB);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[t][i] = ldexpf(acc[t][i], d);
      }
      eA = nA; eB = nB; scaled = true;
    }
    const float sA = ldexpf(1.0f, eA), sB = ldexpf(1.0f, eB);
    {   // dY -> weight-role planes; the bias gradient's share of this tile
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) s += va[e];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      bsum += s;
      uint32_t p1[8], p2[8], p3[8], p4[8];
      constexpr float TS = 1.0f / HC_TSCALE;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float x0 = va[2 * i] * sA, x1 = va[2 * i + 1] * sA;
        const half2_t h1 = __builtin_convertvector((float2_t){x0, x1}, half2_t);
        const float r0 = x0 - (float)h1[0], r1 = x1 - (float)h1[1];
        const half2_t h2 = __builtin_convertvector((float2_t){r0, r1}, half2_t);
        const half2_t h3 = __builtin_convertvector((float2_t){r0 - (float)h2[0], r1 - (float)h2[1]}, half2_t);
        p1[i] = __builtin_bit_cast(uint32_t, h1);
        p2[i] = __builtin_bit_cast(uint32_t, h2);
        p3[i] = __builtin_bit_cast(uint32_t, h3);
        p4[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector((float2_t){x0 * TS, x1 * TS}, half2_t));
      }
      unsigned char* pa = wg_smem + srow * L::SA + 32 * seg;
      *reinterpret_cast<uint4*>(pa) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
      *reinterpret_cast<uint4*>(pa + 16) = make_uint4(p1[4], p1[5], p1[6], p1[7]);
      *reinterpret_cast<uint4*>(pa + L::OFF_A2) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
      *reinterpret_cast<uint4*>(pa + L::OFF_A2 + 16) = make_uint4(p2[4], p2[5], p2[6], p2[7]);
      *reinterpret_cast<uint4*>(pa + L::OFF_A3) = make_uint4(p3[0], p3[1], p3[2], p3[3]);
      *reinterpret_cast<uint4*>(pa + L::OFF_A3 + 16) = make_uint4(p3[4], p3[5], p3[6], p3[7]);
      *reinterpret_cast<uint4*>(pa + L::OFF_A4) = make_uint4(p4[0], p4[1], p4[2], p4[3]);
      *reinterpret_cast<uint4*>(pa + L::OFF_A4 + 16) = make_uint4(p4[4], p4[5], p4[6], p4[7]);
    }
    {   // X -> activation-role planes: stored pixel p = input column - ix0 + 7 (p = 7 .. 40; the 32 aligned ones at p = 8 .. 39)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int line = 32 * k + xl, ci = line >> 2, row = line & 3;
        uint32_t h0, h1, l0, l1;
        float t0, t1, t2, t3;
        const bool lv = ((xmask >> k) & 1u) != 0u;
        split3_pair(lv ? vx[k][0] * sB : 0.f, lv ? vx[k][1] * sB : 0.f, h0, l0, t0, t1);
        split3_pair(lv ? vx[k][2] * sB : 0.f, lv ? vx[k][3] * sB : 0.f, h1, l1, t2, t3);
        *reinterpret_cast<uint2*>(wg_smem + L::OFF_BH + ci * L::SB + row * L::SBR + 16 + 8 * xq) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(wg_smem + L::OFF_BL + ci * L::SB + row * L::SBR + 16 + 8 * xq) = make_uint2(l0, l1);
        *reinterpret_cast<uint2*>(wg_smem + L::OFF_BQ + ci * L::SB + row * L::SBR + 16 + 8 * xq) =
            make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector((float2_t){t0 * HC_TSCALE, t1 * HC_TSCALE}, half2_t)),
                       __builtin_bit_cast(uint32_t, __builtin_convertvector((float2_t){t2 * HC_TSCALE, t3 * HC_TSCALE}, half2_t)));
      }
      {   // the halo columns of line tid: p = 7 (the high half of its dword) and p = 40 (the low half)
        uint32_t hh, ll;
        float ta, tb;
        split3_pair(vh[0] * sB, vh[1] * sB, hh, ll, ta, tb);
        unsigned char* ph = wg_smem + L::OFF_BH + srow * L::SB + seg * L::SBR;
        unsigned char* pl = wg_smem + L::OFF_BL + srow * L::SB + seg * L::SBR;
        unsigned char* pq = wg_smem + L::OFF_BQ + srow * L::SB + seg * L::SBR;
        const uint32_t qq = __builtin_bit_cast(uint32_t, __builtin_convertvector((float2_t){ta * HC_TSCALE, tb * HC_TSCALE}, half2_t));
        *reinterpret_cast<uint16_t*>(ph + 14) = (uint16_t)(hh & 0xffffu);
        *reinterpret_cast<uint16_t*>(pl + 14) = (uint16_t)(ll & 0xffffu);
        *reinterpret_cast<uint16_t*>(pq + 14) = (uint16_t)(qq & 0xffffu);
        *reinterpret_cast<uint16_t*>(ph + 80) = (uint16_t)(hh >> 16);
        *reinterpret_cast<uint16_t*>(pl + 80) = (uint16_t)(ll >> 16);
        *reinterpret_cast<uint16_t*>(pq + 80) = (uint16_t)(qq >> 16);
      }
    }
    __syncthreads();
    if (tile + 1 < t_end) load_tile(tile + 1);

    const unsigned char* __restrict__ pA = wg_smem + (32 * mh + r) * L::SA + 16 * h;            // + 32 s: 8 pixels of step s
    const unsigned char* __restrict__ pBh = wg_smem + L::OFF_BH + (32 * nh + r) * L::SB + 16 * h;
    const unsigned char* __restrict__ pBl = wg_smem + L::OFF_BL + (32 * nh + r) * L::SB + 16 * h;
    const unsigned char* __restrict__ pBq = wg_smem + L::OFF_BQ + (32 * nh + r) * L::SB + 16 * h;
    // 12 groups g = 3 s + ky of 18 matrix instructions; the LDS reads of group g + 1 (and the dY operands of the next step) are issued
    // BETWEEN group g's matrix instructions (sched_group_barrier pins the interleave: one wave per SIMD, nobody else covers an LDS round
    // trip -- left to the scheduler every group waited lgkmcnt for its own reads: ~11 k of a tile's 18 k cycles were not matrix work)
    struct Raw { uint4 ch, cl, cq; uint32_t hL, hR, lL, lR, qL, qR; };
    struct AOp { half8_t w1, w2, w3, wq; };
    auto load_raw = [&](int g, Raw& q) {
      const int s = g / 3, ky = g - 3 * s, yr = s >> 1, xb = 32 * (s & 1);     // xb: byte offset of the step's first column in an fp16 row
      // aligned record (kx = 1) at p = xc + 8, the dword left of it (p = xc + 6, xc + 7) and right of it (p = xc + 16, xc + 17)
      const unsigned char* qh = pBh + (yr + ky) * L::SBR + xb + 16;
      const unsigned char* ql = pBl + (yr + ky) * L::SBR + xb + 16;
      const unsigned char* qq = pBq + (yr + ky) * L::SBR + xb + 16;
      q.ch = *reinterpret_cast<const uint4*>(qh);
      q.cl = *reinterpret_cast<const uint4*>(ql);
      q.cq = *reinterpret_cast<const uint4*>(qq);
      q.hL = *reinterpret_cast<const uint32_t*>(qh - 4);
      q.hR = *reinterpret_cast<const uint32_t*>(qh + 16);
      q.lL = *reinterpret_cast<const uint32_t*>(ql - 4);
      q.lR = *reinterpret_cast<const uint32_t*>(ql + 16);
      q.qL = *reinterpret_cast<const uint32_t*>(qq - 4);
      q.qR = *reinterpret_cast<const uint32_t*>(qq + 16);
    };
    auto load_a = [&](int s, AOp& o) {
      o.w1 = *reinterpret_cast<const half8_t*>(pA + 32 * s);
      o.w2 = *reinterpret_cast<const half8_t*>(pA + L::OFF_A2 + 32 * s);
      o.w3 = *reinterpret_cast<const half8_t*>(pA + L::OFF_A3 + 32 * s);
      o.wq = *reinterpret_cast<const half8_t*>(pA + L::OFF_A4 + 32 * s);
    };
    Raw raw[2];
    AOp aop[2];
    load_a(0, aop[0]);
    load_raw(0, raw[0]);
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      const int s = g / 3, ky = g - 3 * s;
      if (g + 1 < 12) load_raw(g + 1, raw[(g + 1) & 1]);
      if (ky == 2 && s + 1 < 4) load_a(s + 1, aop[(s + 1) & 1]);
      const Raw& q = raw[g & 1];
      const AOp& o = aop[s & 1];
      const uint4 ch = q.ch, cl = q.cl, cq = q.cq;
      const uint32_t q01 = __builtin_amdgcn_alignbit(cq.y, cq.x, 16), q12 = __builtin_amdgcn_alignbit(cq.z, cq.y, 16), q23 = __builtin_amdgcn_alignbit(cq.w, cq.z, 16);
      const uint32_t h01 = __builtin_amdgcn_alignbit(ch.y, ch.x, 16), h12 = __builtin_amdgcn_alignbit(ch.z, ch.y, 16), h23 = __builtin_amdgcn_alignbit(ch.w, ch.z, 16);
      const uint32_t l01 = __builtin_amdgcn_alignbit(cl.y, cl.x, 16), l12 = __builtin_amdgcn_alignbit(cl.z, cl.y, 16), l23 = __builtin_amdgcn_alignbit(cl.w, cl.z, 16);
      half8_t bh[3], bl[3], bq[3];
      bh[0] = __builtin_bit_cast(half8_t, make_uint4(__builtin_amdgcn_alignbit(ch.x, q.hL, 16), h01, h12, h23));
      bh[1] = __builtin_bit_cast(half8_t, ch);
      bh[2] = __builtin_bit_cast(half8_t, make_uint4(h01, h12, h23, __builtin_amdgcn_alignbit(q.hR, ch.w, 16)));
      bl[0] = __builtin_bit_cast(half8_t, make_uint4(__builtin_amdgcn_alignbit(cl.x, q.lL, 16), l01, l12, l23));
      bl[1] = __builtin_bit_cast(half8_t, cl);
      bl[2] = __builtin_bit_cast(half8_t, make_uint4(l01, l12, l23, __builtin_amdgcn_alignbit(q.lR, cl.w, 16)));
      bq[0] = __builtin_bit_cast(half8_t, make_uint4(__builtin_amdgcn_alignbit(cq.x, q.qL, 16), q01, q12, q23));
      bq[1] = __builtin_bit_cast(half8_t, cq);
      bq[2] = __builtin_bit_cast(half8_t, make_uint4(q01, q12, q23, __builtin_amdgcn_alignbit(q.qR, cq.w, 16)));
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wq, bq[kx], acc[ky * 3 + kx], 0, 0, 0);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.w1, bh[kx], acc[ky * 3 + kx], 0, 0, 0);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.w1, bl[kx], acc[ky * 3 + kx], 0, 0, 0);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.w2, bh[kx], acc[ky * 3 + kx], 0, 0, 0);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.w2, bl[kx], acc[ky * 3 + kx], 0, 0, 0);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.w3, bh[kx], acc[ky * 3 + kx], 0, 0, 0);
    }
    // the interleave, in program order of the region: the first group's reads, then per group its 15 operand-forming instructions, (matrix, read)
    // pairs for the next group's reads, the remaining matrix instructions
    __builtin_amdgcn_sched_group_barrier(0x100, 13, 0);
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x002, 15, 0);
      if (g == 11) {
        __builtin_amdgcn_sched_group_barrier(0x008, 18, 0);
      } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (g % 3 == 2) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
        }
      }
    }
  }
  // acc[t][reg]: row (co) = (reg & 3) + 8 (reg >> 2) + 4 h, column (ci) = r; stored per group as wgrad3x3_wide_kernel does
  const int ci = ci0 + nh * 32 + r;
  const size_t cop = (size_t)gridDim.y * 64, cip = (size_t)gridDim.z * 64;
  float* __restrict__ pg = a.part + (size_t)blockIdx.x * cop * (9 * cip + 1);
  const int un = -(eA + eB);
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int co = co0 + mh * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
#pragma unroll
    for (int t = 0; t < 9; ++t) pg[((size_t)co * 9 + t) * cip + ci] = ldexpf(acc[t][reg], un);   // 32 lanes = 128 contiguous bytes
  }
  if (blockIdx.z == 0 && seg == 0) pg[cop * 9 * cip + co0 + srow] = bsum;
}

inline int wgrad3x3_h3_launch(const Wgrad3x3Args& a, dim3 grid, hipStream_t st) {
  static LdsAttrOnce attr;
  if (const int rc = attr.set(wgrad3x3_h3_kernel, WgH3::SMEM)) return rc;
  GC_KLOG("wgrad3x3_h3_kernel");
  wgrad3x3_h3_kernel<<<grid, 256, WgH3::SMEM, st>>>(a);
  return GC_OK;
}

}  // namespace gc
