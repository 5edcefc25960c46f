// General 2-D convolution on the f16 matrix pipe with fp32-grade products (round 5): the default kernel of conv2d_enqueue for the layers
// AROUND the hot path whenever the shape allows it -- BaseBEVBackbone blocks / deblocks, shrink convolutions, heads, the Linear layers, and
// in training their input-gradient convolutions (same kernel, flipped weights).  Included by conv_kernels.h (Conv2dArgs, epilogue).
//
// Why: conv2d_igemm_kernel (exact fp32 on v_mfma_f32_32x32x2_f32) ran the stage-1 training step's convolutions at 0.43 of the fp32 matrix
// roof on the large layers and at ~0.14 on the backbone's small maps (32 K-chunks of 4 608 matrix cycles each, one workgroup per CU:
// profiles/r5_train_leg_n1.json).  The f16 pipe does 16 x the MACs per cycle; with the six-instruction product of conv8h_kernels.h (every
// operand split exactly into fp16 hi + fp16 lo + a third term, products accurate to 2^-26: finer than the 2^-24 of an fp32 FMA chain) a
// 16-channel x 9-tap chunk costs 54 x 32 = 1 728 matrix cycles instead of 72 x 64 x 2 = 9 216.
//
// Arithmetic.  Activations x (scaled by a running power of two xs so that the largest |x| seen so far lies in [2^13, 2^14): gradients are
// scaled UP, large inputs down; when the exponent grows the accumulators are rescaled, exactly) = hi + lo + t with hi = fp16(x xs),
// lo = fp16(x xs - hi), t = the rest (a power of two, stored as bf8(t 2^20)).  Weights (scaled PER OUTPUT ROW by the power of two that puts
// the row's largest |w| into [2^13, 2^14): conv_prep_w3_kernel, undone in the epilogue) = w1 + w2 + w3 (fp16, exact) and wb = bf8(w 2^-20).
//     acc  += w1 hi + w1 lo + w2 hi + w2 lo + w3 hi      (five v_mfma_f32_32x32x16_f16: every fp16 x fp16 product exact)
//     accT += wb t                                       (one v_mfma_f32_32x32x16_bf8_bf8; its own accumulator: gfx950 forwards SrcC only
//                                                         between matrix instructions of one input type)
// Left out: lo w3, t w2, t w3 (<= 2^-33 |x w|) and the bf8 rounding of w in the last term (2^-26 |x w|).
//
// Mapping: GEMM rows = output channels (A operand = weights), columns = pixels (B operand), as conv2d_igemm_kernel: for a fixed
// accumulator register the 32 lanes of a half-wave hold 32 consecutive pixels of one output channel (128-byte NCHW store runs).
// Workgroup = 4 waves = 64 output channels x (TY rows x 16 columns); a wave owns 32 channels x NA = TY / 4 groups of (2 rows x 16).
//   * WEIGHTS never touch the vector ALU: conv_prep_w3_kernel lays every (16-channel chunk, tap, 64-row block) out as the operand registers
//     of the lanes of its two 32-row halves ([w1 | w2 | w3: 2 x 64 lanes x 16 B][wb: 2 x 64 x 8 B] = 7 KiB): an operand is one fully coalesced
//     16-byte access per lane -- by LDS-DMA into a weight buffer (conv2d_h3l_kernel, stride 1) or as a global_load_dwordx4 requested
//     NBUF - 1 tap steps ahead (conv2d_h3_kernel, kept for stride 2, whose activation patch leaves no room for weight slabs).
//   * ACTIVATIONS: a stage = KS chunks of 16 input channels x the haloed pixel patch, staged as 80-byte pixel records
//     {16 hi | 16 lo | 16 bf8 t} (odd multiple of 16 B: conflict-free ds_read_b128 phases for stride 1) in one of TWO LDS buffers.
//     While stage s is on the matrix pipe, stage s + 1 (already in registers) is split and written to the other buffer and stage s + 2
//     travels from memory; the workgroup-wide max |x| of a stage is published one barrier before it is needed: ONE barrier per stage.
#pragma once
#include "conv8h_kernels.h"   // split3_pair, bf8x4s, half8_t

namespace gc {

constexpr int H3_UNIT = 7168;   // bytes of one (chunk, tap, 64-row block) weight unit: [w1 | w2 | w3: 2 x 64 lanes x 16 B each][wb: 2 x 64 lanes x 8 B]

// eligibility of a GEMM shape for the three-term kernels (the prepared buffer carries the blob exactly when this holds)
__host__ __device__ inline bool h3_eligible(int Cin, int CoutP, int KH, int KW) {
  // (1x1 layers stay on the exact-fp32 kernel: per staged pixel a 1x1 layer has a ninth of a 3x3 layer's matrix work, the operand split does
  // not pay -- V2X-ViT's Linear layers ran 11 % slower on this kernel, the Enhancer's training Linears 4 %: profiles/r5_h3_1x1_ab.txt)
  const bool k = (KH == 3 && KW == 3) || (KH == 2 && KW == 2);
  return k && Cin >= 16 && Cin % 8 == 0 && CoutP >= 32;
}
__host__ __device__ inline int h3_chunks(int Cin) { return (((Cin + 15) / 16) + 3) & ~3; }   // 16-channel chunks, padded to a multiple of 4 (stages of up to 4 chunks)
__host__ __device__ inline int h3_blocks(int CoutP) { return (CoutP + 63) / 64; }
__host__ __device__ inline size_t h3_blob_bytes(int Cin, int CoutP, int T) { return (size_t)h3_chunks(Cin) * T * h3_blocks(CoutP) * H3_UNIT; }
// floats of the prepared buffer: [Cin KH KW Cout fp32, k-major][blob][row scales], the tail only for eligible shapes
__host__ __device__ inline long long conv2d_prepared_floats(int Cin, int Cout, int KH, int KW, int transposed) {
  const long long numel = (long long)Cin * Cout * KH * KW;
  const int M = transposed == 1 ? Cout * KH * KW : Cout, T = transposed == 1 ? 1 : KH * KW;
  const int gkh = transposed == 1 ? 1 : KH, gkw = transposed == 1 ? 1 : KW;
  if (!h3_eligible(Cin, M, gkh, gkw)) return numel;
  return numel + (long long)(h3_blob_bytes(Cin, M, T) / 4) + 64LL * h3_blocks(M);
}

struct PrepW3Args {
  const float* w; float* out32; unsigned char* blob; float* wsc;
  int Cin, Cout, KH, KW, transposed;   // as gencomm_conv2d_prepare
  int M, T, nchunk, nb;                // GEMM rows, taps, padded chunk count, 64-row blocks
};
// GEMM row m, input channel ci, tap: the three weight layouts of gencomm_conv2d_prepare
__device__ __forceinline__ float w3_at(const PrepW3Args& a, int m, int ci, int tap) {
  const int khw = a.KH * a.KW;
  if (a.transposed == 0) return a.w[((size_t)m * a.Cin + ci) * khw + tap];
  if (a.transposed == 1) return a.w[(size_t)ci * a.M + m];                       // ConvTranspose2d (kernel == stride) as a 1x1 GEMM
  return a.w[((size_t)ci * a.Cout + m) * khw + (khw - 1 - tap)];                  // input-gradient convolution: flipped, transposed
}
// first launch of a prepare: wsc[m] = 2^-s(m), s(m) the power of two that puts row m's largest |w| into [2^13, 2^14).  One wave per
// GEMM row (rows beyond M: 1), lanes over k = (ci, tap); grid = 64-row blocks x 16 workgroups of 4 rows
__global__ __launch_bounds__(256) void conv_w3_rowscale_kernel(const PrepW3Args a) {
  const int m = (blockIdx.x * 4 + (threadIdx.x >> 6)), l = threadIdx.x & 63, K = a.Cin * a.T;
  float mx = 0.f;
  if (m < a.M) {
    if (a.transposed == 0) {            // a contiguous row of K floats (K % 8 == 0: Cin % 8 == 0), 16-byte aligned
      const float4* __restrict__ row = reinterpret_cast<const float4*>(a.w + (size_t)m * K);
      for (int k = l; k < K / 4; k += 64) {
        const float4 v = row[k];
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
    } else {                            // per input channel: a run of T taps (input-gradient form) or one element (transposed convolution)
      for (int ci = l; ci < a.Cin; ci += 64)
        for (int t = 0; t < a.T; ++t) mx = fmaxf(mx, fabsf(w3_at(a, m, ci, t)));
    }
  }
  mx = wave_max_nonneg(mx);
  if (l == 0) {
    float sc = 1.0f;
    if (mx > 0.f && mx < 3.0e38f) {
      int e;
      (void)frexpf(mx, &e);                      // mx = f 2^e, f in [0.5, 1)
      sc = ldexpf(1.0f, max(min(14 - e, 126), -126));
    }
    a.wsc[m] = 1.0f / sc;
  }
}
// second launch: grid (nb, nchunk, T), 128 threads = the 2 x 64 operand lanes of one unit: the fp32 k-major form AND the three-term unit
__global__ __launch_bounds__(128) void conv_prep_w3_kernel(const PrepW3Args a) {
  const int b = blockIdx.x, c = blockIdx.y, tap = blockIdx.z;
  const int blk = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, m = 64 * b + 32 * blk + r;
  const float sc = 1.0f / a.wsc[m];           // a power of two: exact
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ci = 16 * c + 8 * h + j;
    v[j] = 0.f;
    if (m < a.M && ci < a.Cin) {
      const float w = w3_at(a, m, ci, tap);
      a.out32[((size_t)ci * a.T + tap) * a.M + m] = w;
      v[j] = w * sc;
    }
  }
  uint32_t p1[4], p2[4], p3[4], pb[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = v[2 * i], x1 = v[2 * i + 1];
    const half2_t h1 = __builtin_convertvector((float2_t){x0, x1}, half2_t);
    const float r0 = x0 - (float)h1[0], r1 = x1 - (float)h1[1];
    const half2_t h2 = __builtin_convertvector((float2_t){r0, r1}, half2_t);
    const half2_t h3 = __builtin_convertvector((float2_t){r0 - (float)h2[0], r1 - (float)h2[1]}, half2_t);
    p1[i] = __builtin_bit_cast(uint32_t, h1);
    p2[i] = __builtin_bit_cast(uint32_t, h2);
    p3[i] = __builtin_bit_cast(uint32_t, h3);
  }
  constexpr float TS = 1.0f / HC_TSCALE;
  pb[0] = bf8x4(v[0] * TS, v[1] * TS, v[2] * TS, v[3] * TS);
  pb[1] = bf8x4(v[4] * TS, v[5] * TS, v[6] * TS, v[7] * TS);
  unsigned char* u = a.blob + ((size_t)(c * a.T + tap) * a.nb + b) * H3_UNIT;
  *reinterpret_cast<uint4*>(u + blk * 1024 + lane * 16) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
  *reinterpret_cast<uint4*>(u + 2048 + blk * 1024 + lane * 16) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
  *reinterpret_cast<uint4*>(u + 4096 + blk * 1024 + lane * 16) = make_uint4(p3[0], p3[1], p3[2], p3[3]);
  *reinterpret_cast<uint2*>(u + 6144 + blk * 512 + lane * 8) = make_uint2(pb[0], pb[1]);
}

template <int KH, int KW, int STRIDE, int TY, int KS>
__global__ __launch_bounds__(256, 2) void conv2d_h3_kernel(const Conv2dArgs a) {
  constexpr int TX = 16, KHW = KH * KW, NA = TY / 4, REC = 80;
  static_assert(TY == 4 || TY == 8, "row groups of four");
  constexpr int PH = (TY - 1) * STRIDE + KH, PW = (TX - 1) * STRIDE + KW, NPX = PH * PW;
  constexpr int NIT = KS * NPX * 2, PIT = (NIT + 255) / 256;          // loader items (chunk, pixel, channel octet) per stage / per thread
  constexpr int NSTEP = KHW * KS, NBUF = NSTEP % 3 == 0 ? 3 : 2;      // matrix steps per stage; weight operand sets in registers
  static_assert(NSTEP % NBUF == 0, "the operand ring must close over a stage");
  constexpr int BUF = KS * NPX * REC;
  static_assert(2 * BUF + 64 <= 65536, "two stage buffers in the LDS");
  __shared__ __align__(16) unsigned char Ps[2 * BUF];
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
  const int b_base = ((pyl * STRIDE) * PW + pxl * STRIDE) * REC + 16 * h;     // hi; lo at + 32; t at + 64 - 8 h
  constexpr int G_STRIDE = 4 * STRIDE * PW * REC;

  // weight operand stream: unit q = (chunk, tap) in blob order, this wave's 32-row block
  const int nb = h3_blocks(a.CoutP), NQ = h3_chunks(a.Cin) * KHW;
  const int nstage = (a.Cin + 16 * KS - 1) / (16 * KS);
  const size_t ustride = (size_t)nb * H3_UNIT;
  const unsigned char* __restrict__ abase = a.w3 + (size_t)blockIdx.y * H3_UNIT + (wv & 1) * 1024 + l * 16;
  half8_t wa1[NBUF], wa2[NBUF], wa3[NBUF];
  long wab[NBUF];
  auto loadA = [&](int q, int slot) {
    const unsigned char* __restrict__ p = abase + (size_t)min(q, NQ - 1) * ustride;
    wa1[slot] = *reinterpret_cast<const half8_t*>(p);
    wa2[slot] = *reinterpret_cast<const half8_t*>(p + 2048);
    wa3[slot] = *reinterpret_cast<const half8_t*>(p + 4096);
    wab[slot] = *reinterpret_cast<const long*>(p + 6144 - (wv & 1) * 512 - 8 * l);
  };

  f32x16c acc[NA], accT[NA];
#pragma unroll
  for (int g = 0; g < NA; ++g)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[g][i] = 0.f; accT[g][i] = 0.f; }

  float r1[PIT][8], r2[PIT][8];
  // Loader item j of a thread = (chunk ks of the stage, patch pixel, channel octet g).  The loads are BRANCH-FREE: an item outside the image /
  // beyond Cin reads the nearest valid address and is zeroed when it is USED (live(): a few integer instructions) -- written as
  // `ok ? load : 0` every item became an exec-mask branch whose merge point waits for its loads: 8 serial memory round trips per stage.
  auto live = [&](int s, int j) {
    const int it = tid + 256 * j, ks = it / (NPX * 2), rem = it - ks * (NPX * 2), px = rem >> 1, g = rem & 1;
    const int py = px / PW, pxx = px - py * PW;
    const int gy = iy0 + py, gx = ix0 + pxx, c = (s * KS + ks) * 16 + 8 * g;
    return it < NIT && c < a.Cin && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;   // Cin % 8 == 0: an octet is whole or absent
  };
  auto fetch = [&](int s, float (&rp)[PIT][8]) {
#pragma unroll
    for (int j = 0; j < PIT; ++j) {
      const int it = min(tid + 256 * j, NIT - 1), ks = it / (NPX * 2), rem = it - ks * (NPX * 2), px = rem >> 1, g = rem & 1;
      const int py = px / PW, pxx = px - py * PW;
      const int gy = min(max(iy0 + py, 0), a.H - 1), gx = min(max(ix0 + pxx, 0), a.W - 1), c = min((s * KS + ks) * 16 + 8 * g, a.Cin - 8);
      const float* __restrict__ src = xn + (size_t)c * plane + ((size_t)gy * a.W + gx);
#pragma unroll
      for (int e = 0; e < 8; ++e) rp[j][e] = src[(size_t)e * plane];
    }
  };
  auto wave_max = [&](int s, const float (&rp)[PIT][8]) {
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < PIT; ++j) {
      float m = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(rp[j][e]));
      mx = fmaxf(mx, live(s, j) ? m : 0.f);
    }
    return wave_max_nonneg(mx);
  };
  auto commit = [&](int s, const float (&rp)[PIT][8], float xs0, unsigned char* __restrict__ buf) {
#pragma unroll
    for (int j = 0; j < PIT; ++j) {
      const int it = tid + 256 * j, ks = it / (NPX * 2), rem = it - ks * (NPX * 2), px = rem >> 1, g = rem & 1;
      if (it < NIT) {
        const bool lv = live(s, j);                   // zero padding / channels beyond Cin (a select, not x * 0: the value read instead may be inf)
        uint32_t hi[4], lo[4];
        float t[8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          split3_pair(lv ? rp[j][2 * i] * xs0 : 0.f, lv ? rp[j][2 * i + 1] * xs0 : 0.f, hi[i], lo[i], t[2 * i], t[2 * i + 1]);
        unsigned char* rec = buf + (ks * NPX + px) * REC;
        *reinterpret_cast<uint4*>(rec + 16 * g) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(rec + 32 + 16 * g) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        *reinterpret_cast<uint2*>(rec + 64 + 8 * g) = make_uint2(bf8x4s(t[0], t[1], t[2], t[3]), bf8x4s(t[4], t[5], t[6], t[7]));
      }
    }
  };
  // the power of two that puts the largest |x| seen so far into [2^13, 2^14)
  float run_max = 0.f;
  auto scale_for = [&](int par, float prev) {
    run_max = fmaxf(run_max, fmaxf(fmaxf(s_max[par][0], s_max[par][1]), fmaxf(s_max[par][2], s_max[par][3])));
    if (!(run_max > 0.f)) return prev;
    int e;
    (void)frexpf(run_max, &e);
    return ldexpf(1.0f, max(min(14 - e, 126), -126));
  };

  // ---- prologue: stage 0 into buffer 0, stage 1 in registers with its max published, stage 2 requested
  fetch(0, r1);
#pragma unroll
  for (int i = 0; i + 1 < NBUF; ++i) loadA(i, i);
  {
    const float m0 = wave_max(0, r1);
    if (l == 0) s_max[0][wv] = m0;
  }
  fetch(min(1, nstage - 1), r2);
  __syncthreads();
  float xs_cur = scale_for(0, 1.0f);   // scale of the stage in the buffer about to be multiplied
  float xs_acc = xs_cur;               // scale the accumulators are in
  commit(0, r1, xs_cur, Ps);
  {
    const float m1 = wave_max(min(1, nstage - 1), r2);
    if (l == 0) s_max[1][wv] = m1;
  }
#pragma unroll
  for (int j = 0; j < PIT; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) r1[j][e] = r2[j][e];
  __syncthreads();

  for (int s = 0; s < nstage; ++s) {
    const unsigned char* __restrict__ cur = Ps + (s & 1) * BUF;
    const float xs_next = scale_for((s + 1) & 1, xs_cur);      // stage s + 1's scale (its max was published before the last barrier)
    // UNCONDITIONAL (stage index clamped: the tail re-reads the last stage, harmlessly): behind `if (s + 2 < nstage)` the compiler merged
    // this block with the same-condition block after the matrix phase -- the loads were issued there and waited for at once
    fetch(min(s + 2, nstage - 1), r2);
    __builtin_amdgcn_sched_barrier(0);
    if (xs_cur != xs_acc) {                                    // the exponent grew: bring what is accumulated to the new scale (exact)
      const float ratio = xs_cur / xs_acc;
#pragma unroll
      for (int g = 0; g < NA; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[g][i] *= ratio; accT[g][i] *= ratio; }
      xs_acc = xs_cur;
    }
#pragma unroll
    for (int i = 0; i < NSTEP; ++i) {
      const int ks = i / KHW, tap = i - ks * KHW, ky = tap / KW, kx = tap - ky * KW;
      constexpr int AHEAD = NBUF - 1;
      loadA(s * NSTEP + i + AHEAD, (i + AHEAD) % NBUF);
      const int slot = i % NBUF;
      const unsigned char* bp = cur + (ks * NPX + ky * PW + kx) * REC + b_base;
      half8_t bh[NA], bl[NA];
      long bt[NA];
#pragma unroll
      for (int g = 0; g < NA; ++g) {
        bh[g] = *reinterpret_cast<const half8_t*>(bp + g * G_STRIDE);
        bl[g] = *reinterpret_cast<const half8_t*>(bp + g * G_STRIDE + 32);
        bt[g] = *reinterpret_cast<const long*>(bp + g * G_STRIDE + 64 - 8 * h);    // record + 64 + 8 h (b_base carries + 16 h)
      }
#pragma unroll
      for (int g = 0; g < NA; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa1[slot], bh[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NA; ++g) accT[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(wab[slot], bt[g], accT[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NA; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa1[slot], bl[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NA; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa2[slot], bh[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NA; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa2[slot], bl[g], acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < NA; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa3[slot], bh[g], acc[g], 0, 0, 0);
    }
    commit(min(s + 1, nstage - 1), r1, xs_next, Ps + ((s + 1) & 1) * BUF);             // (past the last stage: a buffer nobody reads)
    {
      const float m2 = wave_max(min(s + 2, nstage - 1), r2);
      if (l == 0) s_max[s & 1][wv] = m2;
#pragma unroll
      for (int j = 0; j < PIT; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) r1[j][e] = r2[j][e];
    }
    xs_cur = xs_next;
    __syncthreads();
  }

  // ---- epilogue: lane = pixel (4 g + pyl, pxl), register = output channel row (as conv2d_igemm_kernel)
  const float inv_xs = 1.0f / xs_acc;
  const int s = a.ups, s2 = s * s;
  const size_t oplane = (size_t)a.Ho * s * a.Wo * s;
  float* __restrict__ yn = a.y + ((size_t)n * a.out_ctotal + a.out_coff) * oplane;
  const float* __restrict__ rn = a.res != nullptr ? a.res + ((size_t)n * a.out_ctotal + a.out_coff) * oplane : nullptr;
#pragma unroll
  for (int g = 0; g < NA; ++g) {
    const int oy = ty0 + 4 * g + pyl, ox = tx0 + pxl;
    if (oy >= a.Ho || ox >= a.Wo) continue;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int gco = co0 + 32 * (wv & 1) + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (gco >= a.CoutP) continue;
      int co = gco;
      size_t oi;
      if (s == 1) {
        oi = (size_t)co * oplane + (size_t)oy * a.Wo + ox;
      } else {
        co = gco / s2;
        const int sub = gco - co * s2, dy = sub / s, dx = sub - dy * s;
        oi = (size_t)co * oplane + (size_t)(oy * s + dy) * (a.Wo * s) + (ox * s + dx);
      }
      float v = fmaf((acc[g][reg] + accT[g][reg]) * (a.wsc[gco] * inv_xs), a.scale[co], a.shift[co]);
      if (a.relu == 1) v = fmaxf(v, 0.f);
      else if (a.relu == 2) v = gelu_erf_f(v);
      if (rn != nullptr) v += rn[oi];
      if (a.relu == 3) v = fmaxf(v, 0.f);
      yn[oi] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Stride-1 form with the weights streamed through the LDS by LDS-DMA (global_load_lds_dwordx4: no vector registers, no vector ALU).
// The form above requests a tap step's operands two steps ahead -- 400 .. 800 cycles -- and, return counters being in order, those short
// L2 loads queue behind the stage's activation loads from HBM: 5.3 us per 16-channel stage on the backbone's small maps against 0.8 us
// of matrix time.  Here a stage's weights arrive in NSUB sub-stage slabs (3x3: one kernel row = 3 taps x 7 KiB) through a ring of TWO
// weight buffers: at the top of sub-stage u the pieces of sub-stage u + 1 are requested (lane-linear units: the DMA's wave-uniform
// base + lane x 16 B IS the layout), nothing inside a matrix phase waits for memory, one vmcnt(0) + barrier per sub-stage.
// LDS: 2 x 21 KB of weights + 2 activation buffers = 71 KB for 3x3 with 8-row tiles: TWO workgroups per CU -- one workgroup's DMA
// issue (60 .. 185 cycles per piece), operand split, prologue and store tail run under the other's matrix instructions (measured
// with a whole-stage slab pair, 155 KB, one workgroup per CU: 4.4 us per stage against 1.65 us of matrix time).
// ---------------------------------------------------------------------------------------------------------------------------------
template <int KH, int KW, int TY, int KS, int NSUB>
struct H3L {
  static constexpr int TX = 16, KHW = KH * KW, NA = TY / 4, REC = 80;
  static constexpr int PH = TY - 1 + KH, PW = TX - 1 + KW, NPX = PH * PW;
  static constexpr int NIT = KS * NPX * 2, PIT = (NIT + 255) / 256;
  static constexpr int NSTEP = KHW * KS, SSTEP = NSTEP / NSUB, WSLAB = SSTEP * H3_UNIT, BUF = KS * NPX * REC;
  static_assert(NSTEP % NSUB == 0, "sub-stages of equal length");
  static constexpr int SMEM = 2 * WSLAB + 2 * BUF + 64;
  static_assert(SMEM <= 80 * 1024, "two workgroups per CU");
};

template <int KH, int KW, int TY, int KS, int NSUB>
__global__ __launch_bounds__(256, 2) void conv2d_h3l_kernel(const Conv2dArgs a, int tiles, int nsamp) {
  using G = H3L<KH, KW, TY, KS, NSUB>;
  constexpr int TX = G::TX, KHW = G::KHW, NA = G::NA, REC = G::REC, PW = G::PW, NPX = G::NPX, NIT = G::NIT, PIT = G::PIT;
  constexpr int SSTEP = G::SSTEP, WSLAB = G::WSLAB, BUF = G::BUF, NPIECE = SSTEP * 7;
  extern __shared__ __align__(16) unsigned char h3l_smem[];
  unsigned char* const Wb = h3l_smem;
  unsigned char* const Ps = h3l_smem + 2 * WSLAB;
  float (*s_max)[4] = reinterpret_cast<float (*)[4]>(h3l_smem + 2 * WSLAB + 2 * BUF);
  fp16_ovfl_clamp();
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
  // 1-D launch, XCD-aware order (common.h): XCD k walks the k-th contiguous eighth of (tile fastest, then sample, then 64-channel block), so
  // an XCD's workgroups share ONE or two channel blocks' weight slabs (L2-resident after the first touch: the 256-channel layers' 4 MB of
  // operand units no longer compete in every L2) and neighbouring tiles' halos
  const BlockId bid = xcd_block_dims(1, blockIdx.x, (unsigned)tiles, (unsigned)nsamp, gridDim.x);
  const int cbi = bid.z, n = bid.y;
  const int tiles_x = (a.Wo + TX - 1) / TX;
  const int ty0 = (bid.x / tiles_x) * TY, tx0 = (bid.x % tiles_x) * TX;
  const int co0 = cbi * 64;
  const int iy0 = ty0 - a.pad, ix0 = tx0 - a.pad;
  const size_t plane = (size_t)a.H * a.W;
  const float* __restrict__ xn = a.x + (size_t)n * a.Cin * plane;

  const int pyl = 2 * (wv >> 1) + (r >> 4), pxl = r & 15;
  const int b_base = (pyl * PW + pxl) * REC + 16 * h;     // hi; lo at + 32; t at + 64 - 8 h
  constexpr int G_STRIDE = 4 * PW * REC;
  const int a_base = (wv & 1) * 1024 + l * 16;            // w1; w2 at + 2048; w3 at + 4096; wb at 6144 + (wv & 1) 512 + 8 l

  const int nb = h3_blocks(a.CoutP);
  const int nstage = (a.Cin + 16 * KS - 1) / (16 * KS), nsub = nstage * NSUB;
  const size_t ustride = (size_t)nb * H3_UNIT;
  const unsigned char* __restrict__ wsrc = a.w3 + (size_t)cbi * H3_UNIT + l * 16;
  // sub-stage u's slab -> weight buffer u & 1: piece p = 7 step + j, pieces dealt round-robin to the four waves (blob order: unit index =
  // matrix step index over the whole sum; past the last sub-stage the last one is re-read into a buffer nobody reads)
  auto dma_weights = [&](int u) {
    const int uu = min(u, nsub - 1);
    unsigned char* const to = Wb + (u & 1) * WSLAB;
#pragma unroll
    for (int k = 0; k < (NPIECE + 3) / 4; ++k) {
      const int p = 4 * k + wv;
      if (p < NPIECE) {
        const int step = p / 7, j = p - 7 * step;
        const unsigned char* src = wsrc + (size_t)(uu * SSTEP + step) * ustride + j * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(to + p * 1024), 16, 0, 0);
      }
    }
  };

  f32x16c acc[NA], accT[NA];
#pragma unroll
  for (int g = 0; g < NA; ++g)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[g][i] = 0.f; accT[g][i] = 0.f; }

  float r1[PIT][8], r2[PIT][8];
  // Loader item j of a thread = (chunk ks of the stage, patch pixel, channel octet g).  The loads are BRANCH-FREE: an item outside the image /
  // beyond Cin reads the nearest valid address and is zeroed when it is USED (live(): a few integer instructions) -- written as
  // `ok ? load : 0` every item became an exec-mask branch whose merge point waits for its loads: 8 serial memory round trips per stage.
  auto live = [&](int s, int j) {
    const int it = tid + 256 * j, ks = it / (NPX * 2), rem = it - ks * (NPX * 2), px = rem >> 1, g = rem & 1;
    const int py = px / PW, pxx = px - py * PW;
    const int gy = iy0 + py, gx = ix0 + pxx, c = (s * KS + ks) * 16 + 8 * g;
    return it < NIT && c < a.Cin && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;   // Cin % 8 == 0: an octet is whole or absent
  };
  auto fetch = [&](int s, float (&rp)[PIT][8]) {
#pragma unroll
    for (int j = 0; j < PIT; ++j) {
      const int it = min(tid + 256 * j, NIT - 1), ks = it / (NPX * 2), rem = it - ks * (NPX * 2), px = rem >> 1, g = rem & 1;
      const int py = px / PW, pxx = px - py * PW;
      const int gy = min(max(iy0 + py, 0), a.H - 1), gx = min(max(ix0 + pxx, 0), a.W - 1), c = min((s * KS + ks) * 16 + 8 * g, a.Cin - 8);
      const float* __restrict__ src = xn + (size_t)c * plane + ((size_t)gy * a.W + gx);
#pragma unroll
      for (int e = 0; e < 8; ++e) rp[j][e] = src[(size_t)e * plane];
    }
  };
  auto wave_max = [&](int s, const float (&rp)[PIT][8]) {
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < PIT; ++j) {
      float m = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(rp[j][e]));
      mx = fmaxf(mx, live(s, j) ? m : 0.f);
    }
    return wave_max_nonneg(mx);
  };
  auto commit = [&](int s, const float (&rp)[PIT][8], float xs0, unsigned char* __restrict__ buf) {
#pragma unroll
    for (int j = 0; j < PIT; ++j) {
      const int it = tid + 256 * j, ks = it / (NPX * 2), rem = it - ks * (NPX * 2), px = rem >> 1, g = rem & 1;
      if (it < NIT) {
        const bool lv = live(s, j);                   // zero padding / channels beyond Cin (a select, not x * 0: the value read instead may be inf)
        uint32_t hi[4], lo[4];
        float t[8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          split3_pair(lv ? rp[j][2 * i] * xs0 : 0.f, lv ? rp[j][2 * i + 1] * xs0 : 0.f, hi[i], lo[i], t[2 * i], t[2 * i + 1]);
        unsigned char* rec = buf + (ks * NPX + px) * REC;
        *reinterpret_cast<uint4*>(rec + 16 * g) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(rec + 32 + 16 * g) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        *reinterpret_cast<uint2*>(rec + 64 + 8 * g) = make_uint2(bf8x4s(t[0], t[1], t[2], t[3]), bf8x4s(t[4], t[5], t[6], t[7]));
      }
    }
  };
  float run_max = 0.f;
  auto scale_for = [&](int par, float prev) {
    run_max = fmaxf(run_max, fmaxf(fmaxf(s_max[par][0], s_max[par][1]), fmaxf(s_max[par][2], s_max[par][3])));
    if (!(run_max > 0.f)) return prev;
    int e;
    (void)frexpf(run_max, &e);
    return ldexpf(1.0f, max(min(14 - e, 126), -126));
  };

  // ---- prologue: sub-stage 0's weights and stage 0's activations in the LDS, stage 1's activations in registers with their max published
  dma_weights(0);
  fetch(0, r1);
  {
    const float m0 = wave_max(0, r1);
    if (l == 0) s_max[0][wv] = m0;
  }
  fetch(min(1, nstage - 1), r2);
  __syncthreads();
  float xs_cur = scale_for(0, 1.0f);
  float xs_acc = xs_cur;
  commit(0, r1, xs_cur, Ps);
  {
    const float m1 = wave_max(min(1, nstage - 1), r2);
    if (l == 0) s_max[1][wv] = m1;
  }
#pragma unroll
  for (int j = 0; j < PIT; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) r1[j][e] = r2[j][e];
  __syncthreads();

  for (int s = 0; s < nstage; ++s) {
    const unsigned char* __restrict__ cur = Ps + (s & 1) * BUF;
    const float xs_next = scale_for((s + 1) & 1, xs_cur);
    if (xs_cur != xs_acc) {
      const float ratio = xs_cur / xs_acc;
#pragma unroll
      for (int g = 0; g < NA; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[g][i] *= ratio; accT[g][i] *= ratio; }
      xs_acc = xs_cur;
    }
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      const int u = s * NSUB + sub;
      const unsigned char* __restrict__ wcur = Wb + (u & 1) * WSLAB;
      // UNCONDITIONAL requests with clamped indices (behind `if (.. < nstage)` the compiler merged them with the same-condition blocks after
      // the matrix phase: requested there, waited for at once).  Stage s + 2's activations are requested in the first sub-stage; return
      // counters being in order, that sub-stage's barrier waits for them too (the other resident workgroup covers it).
#ifndef H3_DIAG
#define H3_DIAG 0   // tools/probes/conv_h3_probe.hip: timing experiments with parts of the loop compiled out (results garbage)
#endif
      if (!(H3_DIAG & 1)) dma_weights(u + 1);
      if (sub == 0 && !(H3_DIAG & 2)) fetch(min(s + 2, nstage - 1), r2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ii = 0; ii < SSTEP; ++ii) {
        const int i = sub * SSTEP + ii;                      // matrix step of the stage: (chunk ks, tap)
        const int ks = i / KHW, tap = i - ks * KHW, ky = tap / KW, kx = tap - ky * KW;
        const unsigned char* ap = wcur + ((H3_DIAG & 8) ? 0 : ii) * H3_UNIT + a_base;
        const half8_t w1 = *reinterpret_cast<const half8_t*>(ap);
        const half8_t w2 = *reinterpret_cast<const half8_t*>(ap + 2048);
        const half8_t w3 = *reinterpret_cast<const half8_t*>(ap + 4096);
        const long wb = *reinterpret_cast<const long*>(ap + 6144 - (wv & 1) * 512 - 8 * l);
        const unsigned char* bp = cur + ((H3_DIAG & 8) ? 0 : (ks * NPX + ky * PW + kx) * REC) + b_base;
        half8_t bh[NA], bl[NA];
        long bt[NA];
#pragma unroll
        for (int g = 0; g < NA; ++g) {
          bh[g] = *reinterpret_cast<const half8_t*>(bp + g * G_STRIDE);
          bl[g] = *reinterpret_cast<const half8_t*>(bp + g * G_STRIDE + 32);
          bt[g] = *reinterpret_cast<const long*>(bp + g * G_STRIDE + 64 - 8 * h);
        }
        if (H3_DIAG & 4) {   // no matrix instructions: the operands still have to arrive
#pragma unroll
          for (int g = 0; g < NA; ++g) { acc[g][0] += (float)w1[0] + (float)w2[1] + (float)w3[2] + (float)wb + (float)bh[g][0] + (float)bl[g][0] + (float)bt[g]; }
          continue;
        }
#pragma unroll
        for (int g = 0; g < NA; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, bh[g], acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NA; ++g) accT[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(wb, bt[g], accT[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NA; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, bl[g], acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NA; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, bh[g], acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NA; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, bl[g], acc[g], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < NA; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3, bh[g], acc[g], 0, 0, 0);
      }
      if (sub == NSUB - 1 && !(H3_DIAG & 2)) {
        commit(min(s + 1, nstage - 1), r1, xs_next, Ps + ((s + 1) & 1) * BUF);
        const float m2 = wave_max(min(s + 2, nstage - 1), r2);
        if (l == 0) s_max[s & 1][wv] = m2;
#pragma unroll
        for (int j = 0; j < PIT; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) r1[j][e] = r2[j][e];
      }
      __syncthreads();   // (hipcc drains the sub-stage's LDS-DMA with vmcnt(0) in front of the barrier: what this pipeline wants)
    }
    xs_cur = xs_next;
  }

  // ---- epilogue (as conv2d_h3_kernel)
  const float inv_xs = 1.0f / xs_acc;
  const int s = a.ups, s2 = s * s;
  const size_t oplane = (size_t)a.Ho * s * a.Wo * s;
  float* __restrict__ yn = a.y + ((size_t)n * a.out_ctotal + a.out_coff) * oplane;
  const float* __restrict__ rn = a.res != nullptr ? a.res + ((size_t)n * a.out_ctotal + a.out_coff) * oplane : nullptr;
  if (KH == 2 && s == 2 && a.relu == 0 && rn == nullptr && (a.CoutP & 3) == 0) {
    // sub-pixel form of a transposed 3x3 stride-2 convolution (the stride-2 layers' input gradient): GEMM row = 4 c + 2 a + b, so the four
    // registers of a group (reg & 3) are the 2 x 2 output pixels of ONE channel -- two 8-byte stores per channel and lane (16 lanes = one
    // 128-byte run) instead of four 4-byte stores at stride 8 with two integer divisions each
#pragma unroll
    for (int g = 0; g < NA; ++g) {
      const int oy = ty0 + 4 * g + pyl, ox = tx0 + pxl;
      if (oy >= a.Ho || ox >= a.Wo) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int gco = co0 + 32 * (wv & 1) + 8 * q + 4 * h;       // row of reg = 4 q (its channel's sub-pixel (0, 0))
        if (gco >= a.CoutP) continue;
        const int co = gco >> 2;
        const float sc = a.scale[co], sh = a.shift[co];
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf((acc[g][4 * q + j] + accT[g][4 * q + j]) * (a.wsc[gco + j] * inv_xs), sc, sh);
        float* __restrict__ d = yn + (size_t)co * oplane + (size_t)(2 * oy) * (a.Wo * 2) + 2 * ox;
        *reinterpret_cast<float2*>(d) = make_float2(v[0], v[1]);
        *reinterpret_cast<float2*>(d + a.Wo * 2) = make_float2(v[2], v[3]);
      }
    }
    return;
  }
#pragma unroll
  for (int g = 0; g < NA; ++g) {
    const int oy = ty0 + 4 * g + pyl, ox = tx0 + pxl;
    if (oy >= a.Ho || ox >= a.Wo) continue;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int gco = co0 + 32 * (wv & 1) + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (gco >= a.CoutP) continue;
      int co = gco;
      size_t oi;
      if (s == 1) {
        oi = (size_t)co * oplane + (size_t)oy * a.Wo + ox;
      } else {
        co = gco / s2;
        const int sub = gco - co * s2, dy = sub / s, dx = sub - dy * s;
        oi = (size_t)co * oplane + (size_t)(oy * s + dy) * (a.Wo * s) + (ox * s + dx);
      }
      float v = fmaf((acc[g][reg] + accT[g][reg]) * (a.wsc[gco] * inv_xs), a.scale[co], a.shift[co]);
      if (a.relu == 1) v = fmaxf(v, 0.f);
      else if (a.relu == 2) v = gelu_erf_f(v);
      if (rn != nullptr) v += rn[oi];
      if (a.relu == 3) v = fmaxf(v, 0.f);
      yn[oi] = v;
    }
  }
}


}  // namespace gc
