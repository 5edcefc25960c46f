// 8 (or 8+8) -> 8 channel 3x3 convolution of the UNet on the f16 matrix pipe with fp32-grade products.
//
// Why: on gfx950 the fp32 MFMA forms (4x4x1, 16x16x4, 32x32x2) and ordinary fp32 VALU work do not overlap --
// measured with tools/probes/mfma16_probe.hip: a wave's VALU instructions issued beside another wave's fp32 MFMA
// stream, or interleaved into it, cost their full issue time on top of the MFMA's (32 + 6n cycles for 16x16x4 + n
// v_fma) -- so conv8_kernel's 576 MFMAs and ~700 VALU instructions per tile-wave simply add up.  The f16/bf16
// matrix pipe is separate (an MFMA holds the vector issue port for 8 of its 16 cycles) and 16x faster per MAC.
//
// Arithmetic (round 3: every operand carries all 24 bits of its fp32 value).  An activation x is split EXACTLY into three
// terms x = hi + lo + t: hi = fp16(x), lo = fp16(x - hi), t = x - hi - lo (each difference is exact in fp32; t is 0 or
// +-1 unit in the last place of x, i.e. a power of two, because two 11-bit terms and the sign of lo cover 23 of the 24
// bits).  hi and lo live in fp16 planes of the LDS tile, t -- scaled by 2^20 -- in a bf8 (e5m2) plane, where a power of
// two is exact.  A weight w (pre-scaled by a power of two so that the largest lies in [2^13, 2^14)) is split the same way
// into three fp16 terms w1 + w2 + w3 (exact) and additionally rounded once to bf8, wb = bf8(w 2^-20).  A product block is
// SIX matrix instructions into the same fp32 accumulators:
//     hi w1 + lo w1 + hi w2 + lo w2 + hi w3        (five v_mfma_f32_16x16x32_f16: every fp16 x fp16 product exact)
//   + t  wb                                        (one v_mfma_f32_16x16x32_bf8_bf8: the 2^-23-sized term to 3 bits)
// The terms left out are lo w3, t w2, t w3 (<= 2^-33 |x w|) and the bf8 rounding of w in the last one
// (2^-23 * 2^-3 = 2^-26 |x w|): a product is accurate to 2^-26 -- finer than the 2^-24 rounding of the fp32 accumulation
// that the reference's own arithmetic has.  Round 2 used three instructions (hi w1 + lo w1 + hi w2) on two-term splits:
// 2^-22 |x w| per product, "22-bit products".  Ranges: |x| < 65504 (range guard, common.h); t is carried for
// |x| >= 2^-12 (below: absolute error <= 2^-36 per product), w3 for |w| >= 2^-14 max|w|.
// GENCOMM_MODE_ARITH = 1 selects the exact-fp32 conv8_kernel instead (unet_host.h).
//
// Mapping (one workgroup = 64x16 output pixels, 4 waves, wave w = rows 4w..4w+3):
//   MFMA M = 16 = 8 output channels x 2 vertically adjacent output rows (r = 0, 1),
//        N = 16 pixels: lane n of group j owns pixel x = 4n + j, so a lane ends up with 4 consecutive pixels
//            (dwordx4 stores / residual loads),
//        K = 32 = 4 taps x 8 input channels; the 4x3 input window of a row pair is 12 taps = 3 MFMAs with no K
//            padding (rows of A that a tap does not reach are zero: 75 % of the MACs are useful).
//   LDS tile: [hi|lo][18 rows][4 phases][18 slots][8 ch] fp16 -- pixel x of a row lives in phase x & 3, slot
//   (x >> 2) + 1, its 8 channels contiguous (16 B): the B operand of a lane is ONE ds_read_b128, 16 lanes read 256
//   contiguous bytes (conflict-free) for every tap shift, and the staging pass (a thread holds a 4-pixel quad of all
//   8 channels) writes one 16-B record per pixel, again 256 contiguous bytes per 16 lanes.
#pragma once
#include "unet_kernels.h"
#include "unet_plan.h"

namespace gc {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

constexpr int HC_TW = 64, HC_TH = 16, HC_NT = 256, HC_LH = 18;
constexpr int HC_SLOTS = 18;
constexpr int HC_PHASE = HC_SLOTS * 16;  // bytes
constexpr int HC_ROW = 4 * HC_PHASE;     // 1152
constexpr int HC_PLANE = HC_LH * HC_ROW; // 20736: hi plane, then lo plane
constexpr int HC_WTAB = 3 * 2 * 64 * 4;  // dwords of one prepared two-term 8-input-channel weight table (+ 64 for the scale)
// three-term tables (conv8h_kernel): per tap group c: [term 3][lane 64][4 dwords] fp16, then [lane 64][2 dwords] bf8
constexpr int HC_WC3 = 3 * 64 * 4 + 64 * 2;  // 896 dwords per tap group
constexpr int HC_WTAB3 = 3 * HC_WC3;         // 2688 dwords per 8-input-channel source (+ 64 floats for the scale after the last)
static_assert(HC_WTAB3 == kWtab3, "unet_plan.h sizes the prepared blob with kWtab3");
constexpr int HC_TPLANE = HC_PLANE / 2;      // 10368 bytes: the bf8 third-term plane (8-byte pixel records, same pixel order)
constexpr int HC_TOFF = 2 * HC_PLANE;        // its byte offset in the tile
constexpr float HC_TSCALE = 1048576.0f;      // 2^20: t is stored as bf8(t 2^20), the bf8 weight as bf8(w 2^-20)

// exact two-term fp16 split of a pair of floats: hi = rne16(x), lo = rne16(x - hi)
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
  const half2_t h = __builtin_convertvector((float2_t){a, b}, half2_t);
  const float ra = a - (float)h[0], rb = b - (float)h[1];
  const half2_t l = __builtin_convertvector((float2_t){ra, rb}, half2_t);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}
__device__ __forceinline__ void split_one(float a, uint16_t& hi, uint16_t& lo) {
  const _Float16 h = (_Float16)a;
  const _Float16 l = (_Float16)(a - (float)h);
  hi = __builtin_bit_cast(uint16_t, h);
  lo = __builtin_bit_cast(uint16_t, l);
}

// 1.0f the optimiser cannot see through: fma(x, one, -fp16) stays a fused multiply-add and is selected as ONE v_fma_mix_f32 (an fp16
// operand read in place); written as x - (float)h it becomes v_cvt_f32_f16 + v_sub_f32.  One s_mov per kernel (the asm is pure: CSE'd).
__device__ __forceinline__ float hc_one() {
  float o;
  asm("s_mov_b32 %0, 1.0" : "=s"(o));
  return o;
}
// exact three-term split of a pair: hi / lo packed fp16 pairs, ta / tb = the third terms, UNSCALED: the 2^20 rides inside the
// conversion to bf8 (bf8x2s / bf8x4s: v_cvt_scalef32_pk_bf8_f32 divides by its scale operand 2^-20 -- one multiply per element less)
// 7 vector instructions per pair: 2 packed conversions + 4 v_fma_mix_f32 (+ the bf8 conversion at the caller).
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& hi, uint32_t& lo, float& ta, float& tb) {
  const float one = hc_one();
  const half2_t h = __builtin_convertvector((float2_t){a, b}, half2_t);
  const float ra = fmaf(a, one, -(float)h[0]), rb = fmaf(b, one, -(float)h[1]);
  const half2_t l = __builtin_convertvector((float2_t){ra, rb}, half2_t);
  ta = fmaf(ra, one, -(float)l[0]);
  tb = fmaf(rb, one, -(float)l[1]);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}
// The same for PRODUCTS za ra, zb rb (the GroupNorm + SiLU staging below forms its activation as such a product): hi = fp16 of the
// rounded product, first residual = fma(z, r, -hi) -- the exact product minus hi, rounded once -- and on from there as above.
__device__ __forceinline__ void split3_prod_pair(float za, float ra, float zb, float rb, uint32_t& hi, uint32_t& lo, float& ta, float& tb) {
  const float one = hc_one();
  const half2_t h = __builtin_convertvector((float2_t){za * ra, zb * rb}, half2_t);
  const float qa = fmaf(za, ra, -(float)h[0]), qb = fmaf(zb, rb, -(float)h[1]);
  const half2_t l = __builtin_convertvector((float2_t){qa, qb}, half2_t);
  ta = fmaf(qa, one, -(float)l[0]);
  tb = fmaf(qb, one, -(float)l[1]);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}
// GroupNorm + SiLU in FIVE vector instructions per element (six before): the coefficients arrive pre-multiplied by -log2(e)
// (hc_gn_coeff2: a2 = -log2e A, b2 = -log2e B), so z = a2 x + b2 = -log2e y with y = A x + B, e = 2^z = exp(-y),
// d = -log2e (1 + e) as ONE fma, r = 1 / d, and SiLU(y) = y / (1 + e) = z r.  Out-of-image positions carry a2 = b2 = 0: z = 0 and
// the product is 0, as zero padding needs; y -> -inf gives e = inf, r = -0, z r = -0; y -> +inf gives r = 1 / -log2e, z r = y.
constexpr float HC_NL2E = -1.4426950408889634f;
// cs = -log2e / k gives k SiLU(y) = z r (the latent step scales its activation by the posterior coefficient: no extra multiply).
__device__ __forceinline__ void gn_silu_zr(float a2, float b2, float x, float& z, float& r, float cs = HC_NL2E) {
  z = fmaf(a2, x, b2);
  r = __builtin_amdgcn_rcpf(fmaf(cs, __builtin_amdgcn_exp2f(z), cs));
}
// two / four scaled third terms -> bf8 bytes (v_cvt_pk_bf8_f32: OCP e5m2, round to nearest even; powers of two are exact)
__device__ __forceinline__ uint32_t bf8x2(float a, float b) {
  return (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false) & 0xffffu;
}
__device__ __forceinline__ uint32_t bf8x4(float a, float b, float c, float d) {
  int v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
  v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true);
  return (uint32_t)v;
}

// the same for UNSCALED third terms: bf8(t 2^20) through the scaled conversion of gfx950 (value / scale, scale = 2^-20; checked
// bit for bit against multiply + convert on hardware)
typedef short hc_s2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf8x2s(float a, float b) {
  const hc_s2_t v = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32((hc_s2_t){0, 0}, a, b, 1.0f / HC_TSCALE, false);
  return (uint32_t)(uint16_t)v[0];
}
__device__ __forceinline__ uint32_t bf8x4s(float a, float b, float c, float d) {
  // (an undefined "old" operand instead of the zero fill -- an empty volatile asm -- saves two v_mov per pixel and was measured 3.5 %
  // SLOWER in latent_step_h_kernel, where the volatile statements pin the schedule: profiles/r5_conv8h_valu_diet.txt)
  hc_s2_t u = {0, 0};
  hc_s2_t v = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(u, a, b, 1.0f / HC_TSCALE, false);
  v = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(v, c, d, 1.0f / HC_TSCALE, true);
  return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ int hc_addr(int row, int px /* -1 .. 64 */) {
  return row * HC_ROW + (px & 3) * HC_PHASE + ((px >> 2) + 1) * 16;
}

// ---- LDS writers of the 18-row tile (1-pixel halo); LO = byte offset of the lo plane ----
// main pass: thread (row r0 = tid / 16, quad qx = tid % 16) holds 4 pixels x 8 channels
template <int LO>
__device__ __forceinline__ void hc_store_main(unsigned char* tile, int r0, int qx, const float (&e)[8][4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint4 hi, lo;
    split_pair(e[0][j], e[1][j], hi.x, lo.x);
    split_pair(e[2][j], e[3][j], hi.y, lo.y);
    split_pair(e[4][j], e[5][j], hi.z, lo.z);
    split_pair(e[6][j], e[7][j], hi.w, lo.w);
    const int addr = r0 * HC_ROW + j * HC_PHASE + (qx + 1) * 16;
    *reinterpret_cast<uint4*>(tile + addr) = hi;
    *reinterpret_cast<uint4*>(tile + LO + addr) = lo;
  }
}
// rows 16, 17: a thread holds ONE channel's quad (channel tid / 32, row 16 + (tid / 16 & 1), quad tid % 16); lane ^ 32
// holds the other channel of the pair.  Sub-dword LDS writes of two lanes of one instruction to the same dword must
// not be relied on, so the pair is brought together first and whole dwords are written: the even channel's thread
// writes pixels 0, 1 of the quad, the odd channel's thread pixels 2, 3.
template <int LO>
__device__ __forceinline__ void hc_store_rem(unsigned char* tile, int tid, const float (&e)[4]) {
  const int cr = tid >> 5, rr = HC_TH + ((tid >> 4) & 1), qx = tid & 15;
  const bool odd = (cr & 1) != 0;
  const float s0 = odd ? e[0] : e[2], s1 = odd ? e[1] : e[3];
  const float p0 = __shfl_xor(s0, 32, 64), p1 = __shfl_xor(s1, 32, 64);
  const float m0 = odd ? e[2] : e[0], m1 = odd ? e[3] : e[1];
  const int jb = odd ? 2 : 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float mine = k ? m1 : m0, theirs = k ? p1 : p0;
    uint32_t hi, lo;
    split_pair(odd ? theirs : mine, odd ? mine : theirs, hi, lo);  // low half = even channel
    const int addr = rr * HC_ROW + (jb + k) * HC_PHASE + (qx + 1) * 16 + (cr >> 1) * 4;
    *reinterpret_cast<uint32_t*>(tile + addr) = hi;
    *reinterpret_cast<uint32_t*>(tile + LO + addr) = lo;
  }
}
// halo columns: thread tid < 144 = (row tid / 8, side (tid / 4) & 1, channel pair tid % 4) writes pixel -1 (phase 3,
// slot 0) or pixel 64 (phase 0, slot 17)
template <int LO>
__device__ __forceinline__ void hc_store_halo(unsigned char* tile, int tid, float e0, float e1) {
  const int cp = tid & 3, side = (tid >> 2) & 1, r = tid >> 3;
  uint32_t hi, lo;
  split_pair(e0, e1, hi, lo);
  const int addr = hc_addr(r, side ? HC_TW : -1) + cp * 4;
  *reinterpret_cast<uint32_t*>(tile + addr) = hi;
  *reinterpret_cast<uint32_t*>(tile + LO + addr) = lo;
}

// ---- three-term writers (hi / lo fp16 planes + bf8 third-term plane at HC_TOFF, 8-byte records at half the byte offsets) ----
__device__ __forceinline__ void hc_store_main3(unsigned char* tile, int r0, int qx, const float (&e)[8][4]) {
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
    *reinterpret_cast<uint4*>(tile + HC_PLANE + addr) = lo;
    *reinterpret_cast<uint2*>(tile + HC_TOFF + (addr >> 1)) = make_uint2(bf8x4s(t[0], t[1], t[2], t[3]), bf8x4s(t[4], t[5], t[6], t[7]));
  }
}
// the same from (z, r) pairs: element = z r (gn_silu_zr / split3_prod_pair)
__device__ __forceinline__ void hc_store_main3_zr(unsigned char* tile, int r0, int qx, const float (&z)[8][4], const float (&r)[8][4]) {
  const int base = r0 * HC_ROW + (qx + 1) * 16;   // even: the third-term plane's offset is base / 2 + j * HC_PHASE / 2, no shift per pixel
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint4 hi, lo;
    float t[8];
    split3_prod_pair(z[0][j], r[0][j], z[1][j], r[1][j], hi.x, lo.x, t[0], t[1]);
    split3_prod_pair(z[2][j], r[2][j], z[3][j], r[3][j], hi.y, lo.y, t[2], t[3]);
    split3_prod_pair(z[4][j], r[4][j], z[5][j], r[5][j], hi.z, lo.z, t[4], t[5]);
    split3_prod_pair(z[6][j], r[6][j], z[7][j], r[7][j], hi.w, lo.w, t[6], t[7]);
    *reinterpret_cast<uint4*>(tile + base + j * HC_PHASE) = hi;
    *reinterpret_cast<uint4*>(tile + HC_PLANE + base + j * HC_PHASE) = lo;
    *reinterpret_cast<uint2*>(tile + HC_TOFF + (base >> 1) + j * (HC_PHASE / 2)) = make_uint2(bf8x4s(t[0], t[1], t[2], t[3]), bf8x4s(t[4], t[5], t[6], t[7]));
  }
}
// rows 16, 17 (thread = one channel's quad, lane ^ 32 = the other channel of the pair; see hc_store_rem): the pair's two
// bf8 bytes of a pixel are one 16-bit store, written by ONE lane (different lanes of an instruction never share a dword)
__device__ __forceinline__ void hc_store_rem3(unsigned char* tile, int tid, const float (&e)[4]) {
  const int cr = tid >> 5, rr = HC_TH + ((tid >> 4) & 1), qx = tid & 15;
  const bool odd = (cr & 1) != 0;
  const float s0 = odd ? e[0] : e[2], s1 = odd ? e[1] : e[3];
  const float p0 = __shfl_xor(s0, 32, 64), p1 = __shfl_xor(s1, 32, 64);
  const float m0 = odd ? e[2] : e[0], m1 = odd ? e[3] : e[1];
  const int jb = odd ? 2 : 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float mine = k ? m1 : m0, theirs = k ? p1 : p0;
    uint32_t hi, lo;
    float ta, tb;
    split3_pair(odd ? theirs : mine, odd ? mine : theirs, hi, lo, ta, tb);  // low half = even channel
    const int addr = rr * HC_ROW + (jb + k) * HC_PHASE + (qx + 1) * 16 + (cr >> 1) * 4;
    *reinterpret_cast<uint32_t*>(tile + addr) = hi;
    *reinterpret_cast<uint32_t*>(tile + HC_PLANE + addr) = lo;
    *reinterpret_cast<uint16_t*>(tile + HC_TOFF + (addr >> 1)) = (uint16_t)bf8x2s(ta, tb);
  }
}
__device__ __forceinline__ void hc_store_halo3(unsigned char* tile, int tid, float e0, float e1) {
  const int cp = tid & 3, side = (tid >> 2) & 1, r = tid >> 3;
  uint32_t hi, lo;
  float ta, tb;
  split3_pair(e0, e1, hi, lo, ta, tb);
  const int addr = hc_addr(r, side ? HC_TW : -1) + cp * 4;
  *reinterpret_cast<uint32_t*>(tile + addr) = hi;
  *reinterpret_cast<uint32_t*>(tile + HC_PLANE + addr) = lo;
  *reinterpret_cast<uint16_t*>(tile + HC_TOFF + (addr >> 1)) = (uint16_t)bf8x2s(ta, tb);
}

// GroupNorm+SiLU (GN) or a plain scale (!GN), split, and write this thread's share of the tile (registers filled by
// stage_load / halo_load_h).
template <bool GN, bool T3 = false>
__device__ __forceinline__ void stage_store_h(unsigned char* tile, const TileRegs<HC_TW, HC_TH, HC_NT, 8>& R,
                                              float2 hreg, int H, int W, int x0, int y0, const float (*ab)[2], int tid,
                                              float mul = 1.0f) {
  using TR = TileRegs<HC_TW, HC_TH, HC_NT, 8>;
  // GN && T3 (the shipped path): `ab` holds the coefficients times -log2(e) (hc_gn_coeff2) and an element is formed as the product
  // z r of gn_silu_zr -- five vector instructions for GroupNorm + SiLU and seven per pair for the split
  constexpr bool ZR = GN && T3;
  {
    const int r0 = tid >> 4, qx = tid & 15;
    const int gy = y0 - 1 + r0, gx = x0 + 4 * qx;
    const bool ok = gy >= 0 && gy < H && gx < W;
    float e[8][4], rr[ZR ? 8 : 1][4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      e[c][0] = R.v[c].x; e[c][1] = R.v[c].y; e[c][2] = R.v[c].z; e[c][3] = R.v[c].w;
      if (GN) {
        // zero padding through the coefficients: out-of-image quads were loaded as 0 and silu(0*0+0) == 0
        const float A = ok ? ab[c][0] : 0.f, B = ok ? ab[c][1] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (ZR) gn_silu_zr(A, B, e[c][j], e[c][j], rr[c][j]);
          else e[c][j] = silu_f(fmaf(A, e[c][j], B));
        }
      } else if (mul != 1.0f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) e[c][j] *= mul;
      }
    }
    if constexpr (ZR) hc_store_main3_zr(tile, r0, qx, e, rr);
    else if constexpr (T3) hc_store_main3(tile, r0, qx, e);
    else hc_store_main<HC_PLANE>(tile, r0, qx, e);
  }
  {
    const int cr = tid >> 5, rr = TR::RPP + ((tid >> 4) & 1), qx = tid & 15;
    const int gy = y0 - 1 + rr, gx = x0 + 4 * qx;
    const bool ok = gy >= 0 && gy < H && gx < W;
    float e[4] = {R.vr.x, R.vr.y, R.vr.z, R.vr.w};
    if (GN) {
      const float A = ok ? ab[cr][0] : 0.f, B = ok ? ab[cr][1] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (ZR) { float z, r; gn_silu_zr(A, B, e[j], z, r); e[j] = z * r; }
        else e[j] = silu_f(fmaf(A, e[j], B));
      }
    } else if (mul != 1.0f) {
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] *= mul;
    }
    if constexpr (T3) hc_store_rem3(tile, tid, e);
    else hc_store_rem<HC_PLANE>(tile, tid, e);
  }
  if (tid < HC_LH * 8) {
    const int cp = tid & 3, side = (tid >> 2) & 1, r = tid >> 3;
    float e0 = hreg.x, e1 = hreg.y;
    if (GN) {
      const int gy = y0 - 1 + r, gx = side ? x0 + HC_TW : x0 - 1;
      const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
      if constexpr (ZR) {
        float z0, r0, z1, r1;
        gn_silu_zr(ok ? ab[2 * cp][0] : 0.f, ok ? ab[2 * cp][1] : 0.f, e0, z0, r0);
        gn_silu_zr(ok ? ab[2 * cp + 1][0] : 0.f, ok ? ab[2 * cp + 1][1] : 0.f, e1, z1, r1);
        e0 = z0 * r0;
        e1 = z1 * r1;
      } else {
        e0 = ok ? silu_f(fmaf(ab[2 * cp][0], e0, ab[2 * cp][1])) : 0.f;
        e1 = ok ? silu_f(fmaf(ab[2 * cp + 1][0], e1, ab[2 * cp + 1][1])) : 0.f;
      }
    } else if (mul != 1.0f) {
      e0 *= mul;
      e1 *= mul;
    }
    if constexpr (T3) hc_store_halo3(tile, tid, e0, e1);
    else hc_store_halo<HC_PLANE>(tile, tid, e0, e1);
  }
}

// Halo columns of the tile for stage_store_h: thread tid < 144 loads the channel pair (2cp, 2cp+1) of halo pixel
// (row tid / 8, side (tid / 4) & 1).  (TileRegs::hv, one channel per thread, is not used by this kernel.)
template <bool UP>
__device__ __forceinline__ float2 halo_load_h(const float* __restrict__ sp, unsigned plane_in, int Win, int H, int W,
                                              int x0, int y0, int tid) {
  float2 out = make_float2(0.f, 0.f);
  if (tid < HC_LH * 8) {
    const int cp = tid & 3, side = (tid >> 2) & 1, r = tid >> 3;
    const int gy = y0 - 1 + r, gx = side ? x0 + HC_TW : x0 - 1;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      const unsigned o = UP ? (unsigned)(gy >> 1) * (unsigned)Win + (unsigned)(gx >> 1) : (unsigned)gy * (unsigned)Win + (unsigned)gx;
      out.x = sp[(unsigned)(2 * cp) * plane_in + o];
      out.y = sp[(unsigned)(2 * cp + 1) * plane_in + o];
    }
  }
  return out;
}

// Slow path for widths that are not a multiple of 4 (tests only): one element at a time.
template <bool GN, bool UP, bool T3 = false>
__device__ __noinline__ void stage_tile_scalar_h(unsigned char* __restrict__ tile, const float* __restrict__ sp, unsigned plane_in,
                                                 int Win, int H, int W, int x0, int y0, const float (*ab)[2], int tid,
                                                 float mul = 1.0f) {
  constexpr int LW = HC_TW + 2;
  for (int i = tid; i < 8 * HC_LH * LW; i += HC_NT) {
    const int c = i / (HC_LH * LW), rem = i - c * (HC_LH * LW);
    const int r = rem / LW, col = rem - r * LW;
    const int gy = y0 - 1 + r, gx = x0 - 1 + col;
    float e = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
      e = UP ? sp[(unsigned)c * plane_in + (unsigned)(gy >> 1) * (unsigned)Win + (unsigned)(gx >> 1)]
             : sp[(unsigned)c * plane_in + (unsigned)gy * (unsigned)Win + (unsigned)gx];
      if (GN && T3) { float z, r; gn_silu_zr(ab[c][0], ab[c][1], e, z, r); e = z * r; }   // ab = coefficients x -log2(e), as in stage_store_h
      else if (GN) e = silu_f(fmaf(ab[c][0], e, ab[c][1]));
      else e *= mul;
    }
    uint16_t hi, lo;
    split_one(e, hi, lo);
    const int addr = hc_addr(r, col - 1) + c * 2;
    *reinterpret_cast<uint16_t*>(tile + addr) = hi;
    *reinterpret_cast<uint16_t*>(tile + HC_PLANE + addr) = lo;
    if constexpr (T3) {
      const float t = (e - (float)__builtin_bit_cast(_Float16, hi)) - (float)__builtin_bit_cast(_Float16, lo);
      tile[HC_TOFF + (addr >> 1)] = (unsigned char)(bf8x2s(t, 0.f) & 0xffu);
    }
  }
}

// Which of the 12 taps of a row pair's 4-row x 3-column window a K group of a matrix instruction holds: instruction c = column
// dx = c, K group kg (= lane / 16) = window row 0, 2, 1, 3.  The order is chosen for the LDS banks.  ds_read_b128 serves a wave in
// four groups of 16 lanes that are NOT the K groups: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- half of K group
// 2h and half of K group 2h + 1 at complementary pixel indices -- so a read is conflict-free exactly when the two K groups' 256-byte
// spans start at the same offset modulo 256.  Window rows two apart do (2 HC_ROW = 9 x 256 bytes); neighbouring rows are 128 bytes
// off and neighbouring columns 32 (the order t = 4c + kg -> row t / 3, column t % 3 of rounds 2-5 paired exactly those: every
// operand read took 8 LDS cycles instead of 4, SQ_LDS_BANK_CONFLICT = 0.39-0.44 of SQ_LDS_IDX_ACTIVE).  The bf8 plane's ds_read_b64
// (two groups of 32 lanes, 128-byte spans per K group) wants the spans 128 bytes apart modulo 256: half of 2 HC_ROW is.
__host__ __device__ constexpr int hc_tap_row(int kg) { return ((kg & 1) << 1) | (kg >> 1); }

// Per-lane byte offsets of the B operand: off[j][c] for pixel group j (pixel 4n + j) and MFMA c (column shift c - 1, this lane's
// window row hc_tap_row(lane / 16)), row pair 0 of wave `wave`.
__device__ __forceinline__ void hc_lane_offsets(int (&off)[4][3], int wave, int lane) {
  const int n = lane & 15, kg = lane >> 4;
  // hc_addr(row, 4n + j + s), s = dx - 1, is linear in j (one phase = HC_PHASE bytes per pixel step) except where the pixel
  // index leaves the lane's own slot: s = -1 at j = 0 (phase 3 of the slot before) and s = +1 at j = 3 (phase 0 of the slot
  // after).  One base per tap group and two wrap fixes instead of twelve full address computations.
  constexpr int WRAP = 4 * HC_PHASE - 16;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int dyp = hc_tap_row(kg), sx = c - 1;
    const int base = (4 * wave + dyp) * HC_ROW + (n + 1) * 16 + sx * HC_PHASE;
    off[0][c] = base + (sx < 0 ? WRAP : 0);
    off[1][c] = base + HC_PHASE;
    off[2][c] = base + 2 * HC_PHASE;
    off[3][c] = base + 3 * HC_PHASE - (sx > 0 ? WRAP : 0);
  }
}

// One 8-input-channel source: acc[p][j] (row pair p, pixel group j) += W (*) tile.  wa[c][0/1] = hi/lo A operands.
// Per (tap group c, row pair p): 8 ds_read_b128 (hi and lo records of the four pixel groups), then three passes of four
// MFMAs (hi*hi, hi*lo, lo*hi) so that consecutive MFMAs never share an accumulator.
template <bool LOWREG = false>
__device__ __forceinline__ void conv_tile_mfma_h(const unsigned char* tile, const half8_t (&wa)[3][2],
                                                 f32x4 (&acc)[2][4], const int (&off)[4][3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if constexpr (!LOWREG) {
        half8_t bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bh[j] = *reinterpret_cast<const half8_t*>(tile + off[j][c] + p * 2 * HC_ROW);
          bl[j] = *reinterpret_cast<const half8_t*>(tile + HC_PLANE + off[j][c] + p * 2 * HC_ROW);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[c][0], bh[j], acc[p][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[c][0], bl[j], acc[p][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[c][1], bh[j], acc[p][j], 0, 0, 0);
      } else {
        // 16 instead of 32 operand registers in flight: the hi records feed two passes, the lo records are fetched behind
        // them (fenced, or the scheduler hoists the reads and the kernel that asks for this spills instead)
        half8_t b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const half8_t*>(tile + off[j][c] + p * 2 * HC_ROW);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[c][0], b[j], acc[p][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[c][1], b[j], acc[p][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const half8_t*>(tile + HC_PLANE + off[j][c] + p * 2 * HC_ROW);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[c][0], b[j], acc[p][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// Three-term form (see the header): six matrix instructions per product block.  tab = the prepared table of this 8-channel
// source (prep_conv8h_kernel): the A operands of tap group c + 1 are requested while group c is on the matrix pipe (the
// tables are a few KB shared by every workgroup: cache hits), so only two groups' worth of operand registers are live.
struct WA3 {
  half8_t w[3];
  long wb;
};
__device__ __forceinline__ void load_wa3(WA3& o, const float* __restrict__ tab, int c, int lane) {
  const float* __restrict__ t = tab + c * HC_WC3;
#pragma unroll
  for (int k = 0; k < 3; ++k) o.w[k] = __builtin_bit_cast(half8_t, *reinterpret_cast<const uint4*>(t + (k * 64 + lane) * 4));
  o.wb = *reinterpret_cast<const long*>(t + 768 + lane * 2);
}
// DIAG (unit-test instantiation only): `mask` selects which of the six terms are issued -- bit 0 hi w1, 1 lo w1, 2 hi w2,
// 3 lo w2, 4 hi w3, 5 t wb -- so that every operand plane and table is checked on its own (tests/test_gpu_conv8.py).
// LOWREG (kernels that hold the NEXT chunk's tile in registers during this phase): no operand prefetch of the next tap group
// and the lo records fetched behind the hi passes -- 30 fewer live registers.
// INIT: the accumulators are not read -- the first matrix instruction on each of them (the bf8 one of tap group 0) takes `init` as its
// C operand instead (bias x scale for all eight: 28 register copies less per wave).  Not with DIAG, whose term mask may skip it.
// `early` (round 5): work that REQUESTS memory for later -- the identity residual's loads, the second source's tile -- runs here, BEHIND the
// first weight-operand loads.  Return counters are in order: requested in front of the matrix phase (as until now) those HBM loads stood
// in front of this phase's L1 / L2 weight loads, and the first matrix instruction waited for them (s_waitcnt vmcnt(8) with eight
// weight loads behind eight residual loads) -- the "early request" bought no overlap at all.
struct HcNoEarly { __device__ __forceinline__ void operator()() const {} };
template <bool DIAG = false, bool LOWREG = false, bool INIT = false, class Early = HcNoEarly>
__device__ __forceinline__ void conv_tile_mfma3(const unsigned char* tile, const float* __restrict__ tab, f32x4 (&acc)[2][4],
                                                const int (&off)[4][3], int lane, int mask = 63, f32x4 init = f32x4{0.f, 0.f, 0.f, 0.f},
                                                Early early = Early()) {
  static_assert(!(DIAG && INIT), "the diagnostic term mask needs initialised accumulators");
  // ORDER MATTERS.  On gfx950 a v_mfma_f32_16x16x32_f16 issued fewer than 6 wait states after a v_mfma_f32_16x16x32_bf8_bf8
  // whose result it accumulates onto (or the other way round) reads a STALE half of the accumulator: the hardware forwards
  // SrcC only between matrix instructions of one input type, hipcc (ROCm 7.2) assumes it always does and schedules such a
  // pair back to back (tools/probes/mfma_mixed_dep_probe.hip: half of the results wrong at 0..4 wait states, none from 6;
  // in this kernel it showed as a lost bias on two of an accumulator's four registers).  Per tap group the two row pairs'
  // bf8 instructions therefore come first, as two fenced passes, then the f16 passes of row pair 0, then those of row pair 1:
  // whatever order hipcc picks INSIDE a pass, at least four matrix instructions (16 wait states) lie between instructions of
  // different type on the same registers.  tests/test_abi.py checks the code object for violations.
  WA3 cur, nxt;
  load_wa3(cur, tab, 0, lane);
  if (!LOWREG) load_wa3(nxt, tab, 1, lane);
  early();
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (!LOWREG && c == 1) load_wa3(nxt, tab, 2, lane);
    long bt[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) bt[p][j] = *reinterpret_cast<const long*>(tile + HC_TOFF + ((off[j][c] + p * 2 * HC_ROW) >> 1));
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      __builtin_amdgcn_sched_barrier(0);
      if (!DIAG || (mask & 32))
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf8_bf8(cur.wb, bt[p][j], (INIT && c == 0) ? init : acc[p][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if constexpr (LOWREG) {
        half8_t b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const half8_t*>(tile + off[j][c] + p * 2 * HC_ROW);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.w[2], b[j], acc[p][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.w[1], b[j], acc[p][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.w[0], b[j], acc[p][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const half8_t*>(tile + HC_PLANE + off[j][c] + p * 2 * HC_ROW);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.w[1], b[j], acc[p][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.w[0], b[j], acc[p][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        continue;
      }
      half8_t bh[4], bl[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = off[j][c] + p * 2 * HC_ROW;
        bh[j] = *reinterpret_cast<const half8_t*>(tile + o);
        bl[j] = *reinterpret_cast<const half8_t*>(tile + HC_PLANE + o);
      }
      // passes of four independent accumulators, the small terms first so that they meet before they meet the large ones
      if (!DIAG || (mask & 16))
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.w[2], bh[j], acc[p][j], 0, 0, 0);
      if (!DIAG || (mask & 8))
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.w[1], bl[j], acc[p][j], 0, 0, 0);
      if (!DIAG || (mask & 4))
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.w[1], bh[j], acc[p][j], 0, 0, 0);
      if (!DIAG || (mask & 2))
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.w[0], bl[j], acc[p][j], 0, 0, 0);
      if (!DIAG || (mask & 1))
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.w[0], bh[j], acc[p][j], 0, 0, 0);
      if (p == 0) __builtin_amdgcn_sched_barrier(0);   // row pair 1's f16 instructions stay behind row pair 0's
    }
    if (c < 2) {
      if constexpr (LOWREG) load_wa3(cur, tab, c + 1, lane);
      else cur = nxt;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// B operand is an fp16 number already (the sampler's step noise): hi plane only, two MFMA passes (w_hi, w_lo).
__device__ __forceinline__ void conv_tile_mfma_hionly(const unsigned char* tile, const half8_t (&wa)[3][2],
                                                      f32x4 (&acc)[2][4], const int (&off)[4][3]) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      half8_t b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const half8_t*>(tile + off[j][c] + p * 2 * HC_ROW);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[c][0], b[j], acc[p][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[c][1], b[j], acc[p][j], 0, 0, 0);
    }
  }
}

// The same with a B operand that IS an fp16 number (the sampler's step noise): hi plane only, the three weight terms --
// three f16 instructions per product block, exact products (no bf8 instruction: nothing to keep apart).
__device__ __forceinline__ void conv_tile_mfma3_hionly(const unsigned char* tile, const float* __restrict__ tab, f32x4 (&acc)[2][4],
                                                       const int (&off)[4][3], int lane) {
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    half8_t w[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) w[k] = __builtin_bit_cast(half8_t, *reinterpret_cast<const uint4*>(tab + c * HC_WC3 + (k * 64 + lane) * 4));
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      half8_t b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const half8_t*>(tile + off[j][c] + p * 2 * HC_ROW);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[2], b[j], acc[p][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[1], b[j], acc[p][j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[p][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[0], b[j], acc[p][j], 0, 0, 0);
    }
  }
}

// prepared weight table of one 8-input-channel source: [c 3][hi/lo 2][lane 64][4 dwords]; after the last table
// 64 floats: even entries 1 / scale, odd entries scale
__device__ __forceinline__ void load_wa(half8_t (&wa)[3][2], const float* __restrict__ tab, int lane) {
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint4 v = *reinterpret_cast<const uint4*>(tab + ((c * 2 + h) * 64 + lane) * 4);
      wa[c][h] = __builtin_bit_cast(half8_t, v);
    }
}

// Per-(sample, channel) sum / sum of squares of the workgroup's output: part[i] / part[4 + i] are this lane's partial
// sums for channel 4 * ((lane / 16) & 1) + i.  Sum over the 16 lanes of a row (same channels, different pixels), then
// over the two rows that hold the same channels; lane 15 (channels 0..3) and lane 31 (channels 4..7) publish the wave
// totals, 16 f64 atomics per workgroup.
__device__ __forceinline__ void hc_stats_commit(float (&part)[8], float (*s_red)[16], double* __restrict__ dstat, int tid) {
  const int lane = tid & 63, wave = tid >> 6, ch = (lane >> 4) & 1;
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    float x = part[v];
    x = dpp_add<0x111, 0xf>(x);
    x = dpp_add<0x112, 0xf>(x);
    x = dpp_add<0x114, 0xf>(x);
    x = dpp_add<0x118, 0xf>(x);
    x += __shfl_xor(x, 32, 64);
    part[v] = x;
  }
  if (lane == 15 || lane == 31) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s_red[wave][4 * ch + i] = part[i];
      s_red[wave][8 + 4 * ch + i] = part[4 + i];
    }
  }
  __syncthreads();
  if (tid < 16) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < HC_NT / 64; ++w) v += s_red[w][tid];
    atomicAdd(&dstat[(tid & 7) * 2 + (tid >> 3)], (double)v);
  }
}

// One 64x16 output tile of one agent (bid = tile x, tile y, agent): the body of conv8h_kernel, also called per work item by
// the persistent dataflow kernel (dataflow_kernels.h).  LDS is the caller's: tile (2 * HC_PLANE + HC_TPLANE bytes, 16-byte
// aligned), s_ab [16][2], s_red [4][16].
constexpr int HC_TILE_BYTES = 2 * HC_PLANE + HC_TPLANE;
// PERSIST (conv8hp_kernel below: a workgroup walks several tiles): `R` / `hreg` arrive holding THIS tile's first source (loaded by the
// caller's prologue or by the previous tile's call), and once they are free -- after the last staging pass of this tile -- the NEXT
// tile's first source is requested into them (`have_next`, `next`) behind the last matrix phase, so that its loads are in flight during
// this tile's epilogue, stores and statistics (requested earlier -- across a matrix phase -- the 38 registers spill).
struct HcCarry {
  TileRegs<HC_TW, HC_TH, HC_NT, 8> R;
  float2 hreg;
};
template <int NSRC, bool GN, bool UP, int RES, bool DIAG = false, bool PERSIST = false>
__device__ __forceinline__ void conv8h_tile(const Conv8Args& a, const BlockId bid, unsigned char* tile, float (*s_ab)[2], float (*s_red)[16],
                                            size_t bias_off, HcCarry& carry, bool have_next, const BlockId next) {
  constexpr int NT = HC_NT, TW = HC_TW, TH = HC_TH;
  TileRegs<TW, TH, NT, 8>& R = carry.R;
  float2& hreg = carry.hreg;
  fp16_ovfl_clamp();
  int tid_ = threadIdx.x;
  // PERSIST: everything derived from the thread index (lane offsets, tile addresses, ...) is re-derived per tile -- hoisted out of the
  // tile loop as invariants these ~50 values are spilled to scratch and reloaded in the middle of the matrix phase
  if (PERSIST) asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63, wave = tid >> 6;
  const int n = bid.z;
  const int x0 = bid.x * TW, y0 = bid.y * TH;
  const size_t plane_in = (size_t)a.Hin * a.Win;
  const size_t plane = (size_t)a.H * a.W;
  const bool wvec = UP ? ((a.Win & 1) == 0) : ((a.W & 3) == 0);
  // output ownership: pixels x0 + 4*ln .. +3, rows y0 + 4*wave + 2p + rr (p = 0, 1), channels 4*ch + i (i = 0..3)
  const int ln = lane & 15, g = lane >> 4, ch = g & 1, rr = g >> 1;
  const int gx = x0 + 4 * ln;
  const int gy0 = y0 + 4 * wave + rr;
  const bool vec_ok = (gx + 3 < a.W) && ((a.W & 3) == 0);
  const bool wave_live = y0 + 4 * wave < a.H;

  GC_STAMP(0);
  // raw (not normalised) input: exact power-of-two range reduction from a device-side bound on max|src| (common.h)
  float mul = 1.0f;
  if (!GN) {
    float bound = 0.f;
    if (a.amax != nullptr) {
      bound = *a.amax;
    } else if (a.sstat[0] != nullptr) {  // max|x| <= sqrt(sum x^2) of the largest channel of this sample
      // vector loads at agent scope: in the dataflow kernel another workgroup of this launch wrote these sums, and a
      // wave-uniform address would otherwise go through the scalar cache, which an acquire does not refresh
      const double* st = a.sstat[0] + (size_t)n * 16;
      double q = 0.0;
#pragma unroll
      for (int c = 0; c < 8; ++c) q = fmax(q, __hip_atomic_load(st + 2 * c + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      bound = sqrtf((float)q) * 1.0001f;
    }
    mul = act_scale(bound);
  }
  const float inv_s = a.wh[NSRC * HC_WTAB3] / mul;  // one scale for the whole (concatenated) weight tensor
  const float4 bias4 = *reinterpret_cast<const float4*>(a.bias + bias_off + 4 * ch);
  const float bias[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
  if (!PERSIST) {
    hreg = make_float2(0.f, 0.f);
    if (wvec) {
      stage_load<TW, TH, NT, 8, UP, GN>(R, a.src[0] + (size_t)n * 8 * plane_in, (unsigned)plane_in, a.Win, a.H, a.W, x0, y0, tid);
      hreg = halo_load_h<UP>(a.src[0] + (size_t)n * 8 * plane_in, (unsigned)plane_in, a.Win, a.H, a.W, x0, y0, tid);
    }
  }
  auto prefetch_next = [&]() {   // PERSIST: the next tile's first source into the (now free) staging registers
    if (PERSIST && have_next) {
      const float* np = a.src[0] + (size_t)next.z * 8 * plane_in;
      stage_load<TW, TH, NT, 8, UP, GN>(R, np, (unsigned)plane_in, a.Win, a.H, a.W, next.x * TW, next.y * TH, tid);
      hreg = halo_load_h<UP>(np, (unsigned)plane_in, a.Win, a.H, a.W, next.x * TW, next.y * TH, tid);
    }
  };
  if (GN) {
    if (tid < NSRC * 8) {
      const int s = tid >> 3, c = tid & 7;
      float A, B;
      gn_coeff(a.sstat[s] + (size_t)n * 16, c, 2 * NSRC, a.inv_cnt, a.gamma[tid], a.beta[tid], &A, &B);
      s_ab[tid][0] = HC_NL2E * A;   // stage_store_h<true, true> / gn_silu_zr take the coefficients times -log2(e)
      s_ab[tid][1] = HC_NL2E * B;
    }
    __syncthreads();
  }

  GC_STAMP(1);
  // the accumulators start at bias * scale (exact: the scale is a power of two), so the epilogue is one multiply
  f32x4 acc[2][4];
  const float bsc = a.wh[NSRC * HC_WTAB3 + 1] * mul;
  const f32x4 b0 = {bias[0] * bsc, bias[1] * bsc, bias[2] * bsc, bias[3] * bsc};
  if (DIAG || !wave_live) {   // otherwise the first matrix instruction of every accumulator reads b0 directly (conv_tile_mfma3<.., INIT>)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[p][j] = b0;
  }

  if (wvec) stage_store_h<GN, true>(tile, R, hreg, a.H, a.W, x0, y0, &s_ab[0], tid, mul);
  else stage_tile_scalar_h<GN, UP, true>(tile, a.src[0] + (size_t)n * 8 * plane_in, (unsigned)plane_in, a.Win, a.H, a.W, x0, y0, &s_ab[0], tid, mul);
  int off[4][3];
  hc_lane_offsets(off, wave, lane);
  // identity residual: requested once the staging registers are free, so that it arrives during the matrix phase
  float resv[RES == 1 ? 2 : 1][4][4];
  auto request_residual = [&]() {
  if (RES == 1 && wave_live) {
    const float* const lane_res = a.res[0] + ((size_t)n * 8 + 4 * ch) * plane + (size_t)gy0 * a.W + gx;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gy = gy0 + 2 * p;
        const float* __restrict__ rp = lane_res + ((size_t)i * plane + (size_t)(2 * p) * a.W);   // lane pointer + uniform
        if (vec_ok && gy < a.H) {
          const float4 r = *reinterpret_cast<const float4*>(rp);
          resv[p][i][0] = r.x; resv[p][i][1] = r.y; resv[p][i][2] = r.z; resv[p][i][3] = r.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) resv[p][i][j] = (gy < a.H && gx + j < a.W) ? rp[j] : 0.f;
        }
      }
  }
  };
  // the second source is requested AFTER the first matrix phase in the persistent form (full-register matrix phase; measured equal to
  // the early request in round 4: profiles/r4_conv8h_ab.txt) -- early, the loop-carried state on top of it spills
  constexpr bool LATE2 = PERSIST;
  // requests for later (PERSIST: the residual goes behind the matrix phase -- the prefetched next tile needs the registers): issued
  // behind the matrix phase's first weight loads (conv_tile_mfma3 `early`), not in front of the phase
  auto early = [&]() {
    if (!PERSIST) request_residual();
    if (NSRC == 2 && wvec && !LATE2) {  // prefetch the skip tensor's tile while the first half is on the matrix cores
      stage_load<TW, TH, NT, 8, UP, GN>(R, a.src[1] + (size_t)n * 8 * plane_in, (unsigned)plane_in, a.Win, a.H, a.W, x0, y0, tid);
      hreg = halo_load_h<UP>(a.src[1] + (size_t)n * 8 * plane_in, (unsigned)plane_in, a.Win, a.H, a.W, x0, y0, tid);
    }
  };
#ifdef HC_EARLY_IN_FRONT   // A/B builds (tools/diag/early_ab.sh): the requests in front of the phase, as until round 5
  early();
#endif
  __syncthreads();
  GC_STAMP(2);
#ifdef HC_PRIO_MFMA   // diagnostic builds (tools/diag/prio_ab.sh): static wave priority from the matrix phase on
  __builtin_amdgcn_s_setprio(HC_PRIO_MFMA);
#endif
#ifdef HC_EARLY_IN_FRONT
  if (wave_live) conv_tile_mfma3<DIAG, NSRC == 2 && (!LATE2 || PERSIST), !DIAG>(tile, a.wh, acc, off, lane, a.term_mask, b0);
#else
  if (wave_live) conv_tile_mfma3<DIAG, NSRC == 2 && (!LATE2 || PERSIST), !DIAG>(tile, a.wh, acc, off, lane, a.term_mask, b0, early);
  else early();   // waves below the image issue no matrix instructions but take part in the second source's staging
#endif
  if (NSRC == 2 && wvec && LATE2) {
    stage_load<TW, TH, NT, 8, UP, GN>(R, a.src[1] + (size_t)n * 8 * plane_in, (unsigned)plane_in, a.Win, a.H, a.W, x0, y0, tid);
    hreg = halo_load_h<UP>(a.src[1] + (size_t)n * 8 * plane_in, (unsigned)plane_in, a.Win, a.H, a.W, x0, y0, tid);
  }
  // diagnostic instantiation, MODE_RESFUSE_EMU: the matrix work of a second convolution on the same tile (its result is added: the
  // numbers are meaningless, the instruction count is that of a fused conv1 + conv2 workgroup)
  if (DIAG && (a.term_mask & 128) && wave_live) conv_tile_mfma3<DIAG, NSRC == 2>(tile, a.wh, acc, off, lane, a.term_mask);
  GC_STAMP(3);
  if (NSRC == 2) {
    __syncthreads();
    if (wvec) stage_store_h<GN, true>(tile, R, hreg, a.H, a.W, x0, y0, &s_ab[8], tid);
    else stage_tile_scalar_h<GN, UP, true>(tile, a.src[1] + (size_t)n * 8 * plane_in, (unsigned)plane_in, a.Win, a.H, a.W, x0, y0, &s_ab[8], tid);
    __syncthreads();
    if (wave_live) conv_tile_mfma3<false, PERSIST>(tile, a.wh + HC_WTAB3, acc, off, lane);
  }

  if (PERSIST) request_residual();
  if (RES != 2) prefetch_next();   // behind the last matrix phase: live across the epilogue only (held across a matrix phase it spills)
  GC_STAMP(4);
#ifdef HC_PRIO_EPI    // diagnostic builds: static wave priority for the epilogue (the workgroup's last phase)
  __builtin_amdgcn_s_setprio(HC_PRIO_EPI);
#endif
  float part[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) part[i] = 0.f;
  if (wave_live) {
    float* const lane_dst = a.dst + ((size_t)n * 8 + 4 * ch) * plane + (size_t)gy0 * a.W + gx;
    float out[2][4][4];  // [row pair][channel i][pixel j]
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) out[p][i][j] = RES == 1 ? fmaf(acc[p][j][i], inv_s, resv[p][i][j]) : acc[p][j][i] * inv_s;

    if (RES == 2) {  // 1x1 nin_shortcut over the 16 raw input channels of the block
      const size_t lane_off = (size_t)n * 8 * plane;
      if ((a.W & 3) == 0) {
        // Branch-free, batched loads: with W % 4 == 0 a quad that starts inside the image lies inside it, and a lane whose outputs fall
        // outside the image (never stored, never counted) reads the nearest in-image quad instead.  No exec-mask branch stands between
        // the loads, so a batch of them is in flight while the previous batch's FMAs issue -- written with a conditional load per
        // (channel, row pair) the 32 loads of this block each waited for an L2 round trip of their own (s_waitcnt vmcnt(0) in front of
        // every FMA group: 23 us of a 57 us full-resolution launch).
        int gxl = gx, gyl = gy0;
        asm volatile("" : "+v"(gxl), "+v"(gyl));   // the address arithmetic below starts HERE: hoisted above the matrix phase it costs spills
        const int cx = min(gxl, a.W - 4);
        const int cy0 = min(gyl, a.H - 1), d1 = (min(gyl + 2, a.H - 1) - cy0) * a.W;   // second row pair: d1 floats further (0 when clamped)
        const size_t o0 = lane_off + (size_t)cy0 * a.W + cx;
        const float* const lp[2][2] = {{a.res[0] + o0, a.res[0] + o0 + d1}, {a.res[1] + o0, a.res[1] + o0 + d1}};
        constexpr int CB = 4;   // channels per batch: 8 quads = 32 registers in flight per batch
        float4 rq[2][CB][2];
        auto issue = [&](int b, int st) {
#pragma unroll
          for (int c = 0; c < CB; ++c)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              const int cc = b * CB + c;
              rq[st][c][p] = *reinterpret_cast<const float4*>(lp[cc >> 3][p] + (size_t)(cc & 7) * plane);
            }
        };
        issue(0, 0);
#pragma unroll
        for (int b = 0; b < 16 / CB; ++b) {
          __builtin_amdgcn_sched_barrier(0);   // at most two batches in flight: the scheduler would otherwise hoist every load and spill
          if (b + 1 < 16 / CB) issue(b + 1, (b + 1) & 1);
#pragma unroll
          for (int c = 0; c < CB; ++c) {
            const float4 wv4 = *reinterpret_cast<const float4*>(a.ninw + (b * CB + c) * 8 + 4 * ch);
            const float wv[4] = {wv4.x, wv4.y, wv4.z, wv4.w};
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              const float4 t = rq[b & 1][c][p];
              const float r[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) out[p][i][j] = fmaf(wv[i], r[j], out[p][i][j]);
            }
          }
        }
      } else {   // widths that are not a multiple of 4 (tests): element by element, kept small (no unrolling: its register needs must not spill the fast path)
#pragma unroll 1
        for (int c = 0; c < 16; ++c) {
          const float4 wv4 = *reinterpret_cast<const float4*>(a.ninw + c * 8 + 4 * ch);
          const float wv[4] = {wv4.x, wv4.y, wv4.z, wv4.w};
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int gy = gy0 + 2 * p;
            const float* __restrict__ rp = a.res[c >> 3] + ((size_t)n * 8 + (c & 7)) * plane + (size_t)gy * a.W + gx;
            float r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = (gy < a.H && gx + j < a.W) ? rp[j] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) out[p][i][j] = fmaf(wv[i], r[j], out[p][i][j]);
          }
        }
      }
    }

    if (RES == 2) prefetch_next();   // behind the shortcut's loads (in-order completion: in front of them they would wait for these)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int gy = gy0 + 2 * p;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float* __restrict__ dp = lane_dst + ((size_t)i * plane + (size_t)(2 * p) * a.W);   // lane pointer + uniform: one 64-bit add per store
        float s = 0.f, q = 0.f;
        if (vec_ok && gy < a.H) {
          if (!(DIAG && (a.term_mask & 64)))   // diagnostic, MODE_RESFUSE_EMU: statistics-only pass (no store)
            *reinterpret_cast<float4*>(dp) = make_float4(out[p][i][0], out[p][i][1], out[p][i][2], out[p][i][3]);
          s = out[p][i][0] + out[p][i][1];  // no "0 + x" / "x * x + 0" instructions in the common path
          q = out[p][i][0] * out[p][i][0];
          q = fmaf(out[p][i][1], out[p][i][1], q);
#pragma unroll
          for (int j = 2; j < 4; ++j) { s += out[p][i][j]; q = fmaf(out[p][i][j], out[p][i][j], q); }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (gy < a.H && gx + j < a.W) { dp[j] = out[p][i][j]; s += out[p][i][j]; q = fmaf(out[p][i][j], out[p][i][j], q); }
        }
        part[i] = p == 0 ? s : part[i] + s;
        part[4 + i] = p == 0 ? q : part[4 + i] + q;
      }
    }
  } else if (RES == 2) {
    prefetch_next();   // waves below the image (last tile row) stage the next tile like any other
  }
  GC_STAMP(5);
  if (a.dstat != nullptr) hc_stats_commit(part, s_red, a.dstat + (size_t)n * 16, tid);
  GC_STAMP(6);
  if (PERSIST) __syncthreads();   // the next tile's coefficient / tile writes must not overtake this tile's LDS reads (s_red, last matrix phase)
}
// the one-tile form (per-layer launches, the dataflow kernel): registers of its own, nothing carried
template <int NSRC, bool GN, bool UP, int RES, bool DIAG = false>
__device__ __forceinline__ void conv8h_tile(const Conv8Args& a, const BlockId bid, unsigned char* tile, float (*s_ab)[2], float (*s_red)[16],
                                            size_t bias_off = 0) {
  HcCarry carry;
  conv8h_tile<NSRC, GN, UP, RES, DIAG, false>(a, bid, tile, s_ab, s_red, bias_off, carry, false, bid);
}

template <int NSRC, bool GN, bool UP, int RES, bool DIAG = false>
__global__ __launch_bounds__(HC_NT, 3) void conv8h_kernel(const Conv8Args a) {
  __shared__ __align__(16) unsigned char tile[HC_TILE_BYTES];  // hi | lo (fp16) | third term (bf8)
  __shared__ float s_ab[16][2];
  __shared__ float s_red[HC_NT / 64][16];
#ifdef HC_LDS_PAD  // diagnostic builds: inflate the LDS footprint to limit workgroups per CU
  __shared__ float s_pad[HC_LDS_PAD / 4];
  if (a.H < 0) s_pad[threadIdx.x] = 1.f;
#endif
  conv8h_tile<NSRC, GN, UP, RES, DIAG>(a, xcd_block(a.xcd), tile, s_ab, s_red);
}

// Persistent form (round 5): a 1-D grid of at most the resident slots (3 per CU); workgroup b walks the tiles b, b + G, b + 2G, ... of the
// (tx, ty, n) tile grid in xcd_block order (G % 8 == 0: a workgroup's tiles stay on its XCD's share).  Per tile the chain is the
// one-tile kernel's, minus what a workgroup's slot otherwise sits through without issuing anything: the wait for the tile's first
// loads (requested one tile ahead) and the drain of its stores (the workgroup moves on while they complete).  Vector widths only
// (W % 4 == 0; Win % 2 == 0 for the nearest-x2 form): the host launches the one-tile kernel otherwise.
template <int NSRC, bool GN, bool UP, int RES>
__global__ __launch_bounds__(HC_NT, 3) void conv8hp_kernel(const Conv8Args a, int tx, int ty, int nz) {
  __shared__ __align__(16) unsigned char tile[HC_TILE_BYTES];
  __shared__ float s_ab[16][2];
  __shared__ float s_red[HC_NT / 64][16];
  const unsigned total = (unsigned)tx * ty * nz;
  const size_t plane_in = (size_t)a.Hin * a.Win;
  unsigned lin = blockIdx.x;
  if (lin >= total) return;
  BlockId bid = xcd_block_dims(a.xcd, lin, (unsigned)tx, (unsigned)ty, total);
  HcCarry carry;
  {
    const float* sp = a.src[0] + (size_t)bid.z * 8 * plane_in;
    stage_load<HC_TW, HC_TH, HC_NT, 8, UP, GN>(carry.R, sp, (unsigned)plane_in, a.Win, a.H, a.W, bid.x * HC_TW, bid.y * HC_TH, threadIdx.x);
    carry.hreg = halo_load_h<UP>(sp, (unsigned)plane_in, a.Win, a.H, a.W, bid.x * HC_TW, bid.y * HC_TH, threadIdx.x);
  }
#pragma unroll 1
  for (; lin < total; lin += gridDim.x) {
    const unsigned nl = lin + gridDim.x;
    const bool have_next = nl < total;
    const BlockId nb = have_next ? xcd_block_dims(a.xcd, nl, (unsigned)tx, (unsigned)ty, total) : bid;
    conv8h_tile<NSRC, GN, UP, RES, false, true>(a, bid, tile, s_ab, s_red, 0, carry, have_next, nb);
    bid = nb;
  }
}

// Weight preparation: OIHW [8][IC][3][3] (IC = 8 or 16) -> IC/8 three-term tables of HC_WTAB3 dwords + 64 floats (1 / scale, scale).
// A operand of MFMA c, lane l: row m = l % 16 = (r = m / 8, oc = m % 8), K group l / 16 -> the tap of the 4x3 window in
// column dx = c, window row dyp = hc_tap_row(l / 16), 8 input channels; tap row of the kernel = dyp - r (zero outside 0..2).
// Per tap group: fp16 terms w1, w2, w3 of w * scale ([term][lane][4 dwords], channels 2d, 2d+1 in dword d), then the
// bf8 rounding of w * scale * 2^-20 ([lane][2 dwords], channel k in byte k).
__global__ __launch_bounds__(256) void prep_conv8h_kernel(const float* __restrict__ w, float* __restrict__ dst, int IC) {
  __shared__ float s_max[256];
  const int tid = threadIdx.x;
  float m = 0.f;
  for (int i = tid; i < 8 * IC * 9; i += 256) m = fmaxf(m, fabsf(w[i]));
  s_max[tid] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) s_max[tid] = fmaxf(s_max[tid], s_max[tid + s]);
    __syncthreads();
  }
  // power-of-two scale that puts the largest weight in [2^13, 2^14): the third terms of weights down to 2^-14 of the
  // largest stay on fp16's 2^-24 grid (exact), the bf8 copies (x 2^-20) of weights down to 2^-7 of it stay normal;
  // all-zero weights -> scale 1
  const float wmax = s_max[0];
  int ex = 0;
  if (wmax > 0.f) (void)frexpf(wmax, &ex);  // wmax = f * 2^ex, f in [0.5, 1)
  const float scale = wmax > 0.f ? ldexpf(1.0f, 14 - ex) : 1.0f;
  const int nsrc = IC / 8;
  uint32_t* __restrict__ out = reinterpret_cast<uint32_t*>(dst);
  auto weight = [&](int s, int c, int l, int ch) -> float {  // scaled weight of (source s, tap group c, lane l, channel ch)
    const int mrow = l & 15, kg = l >> 4, r = mrow >> 3, oc = mrow & 7;
    const int dyp = hc_tap_row(kg), dx = c, dy = dyp - r;
    return (dy >= 0 && dy <= 2) ? w[((oc * IC + s * 8 + ch) * 3 + dy) * 3 + dx] * scale : 0.f;
  };
  for (int i = tid; i < nsrc * HC_WTAB3; i += 256) {
    const int s = i / HC_WTAB3, rem = i - s * HC_WTAB3;
    const int c = rem / HC_WC3, q = rem - c * HC_WC3;
    if (q < 768) {
      const int d = q & 3, l = (q >> 2) & 63, term = q >> 8;
      uint16_t v[2];
      for (int e = 0; e < 2; ++e) {
        const float x = weight(s, c, l, 2 * d + e);
        const _Float16 w1 = (_Float16)x;
        const float r1 = x - (float)w1;
        const _Float16 w2 = (_Float16)r1;
        const _Float16 w3 = (_Float16)(r1 - (float)w2);
        v[e] = __builtin_bit_cast(uint16_t, term == 0 ? w1 : term == 1 ? w2 : w3);
      }
      out[i] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
    } else {
      const int d = (q - 768) & 1, l = (q - 768) >> 1;
      const float k = 1.0f / HC_TSCALE;
      out[i] = bf8x4(weight(s, c, l, 4 * d) * k, weight(s, c, l, 4 * d + 1) * k, weight(s, c, l, 4 * d + 2) * k, weight(s, c, l, 4 * d + 3) * k);
    }
  }
  if (tid < 64) dst[nsrc * HC_WTAB3 + tid] = (tid & 1) ? scale : 1.0f / scale;  // [0] = 1 / scale, [1] = scale
}

}  // namespace gc
