// Backward kernels of the diffusion UNet (training branch: cond_diff.py:342-360 back-propagates through every UNet call,
// SURVEY.md 8a / 8e).  Exact fp32, one launch per elementary gradient; the forward tensors come from re-running the HIP
// forward with every intermediate kept (UNetPlan keep_all).  Input-gradient convolutions (dgrad) run on the general
// implicit-GEMM kernel (conv_kernels.h) with transposed, tap-flipped weights; this file holds what has no forward twin:
//   conv_wgrad_kernel      dW[oc][ic][ky][kx] += sum_{n,y,x} dY[oc](y,x) X[ic](y*s+ky-p, x*s+kx-p), dB[oc] += sum dY[oc]
//                          (3x3 stride 1 / stride 2 / nearest-x2 input, 1x1; input = concat of up to two tensors)
//   gn_silu_fwd_kernel     A = SiLU(GroupNorm(x)) materialised (the forward fuses it into its consumers' staging)
//   gn_silu_bwd_*          GroupNorm(4 groups, eps 1e-6) + SiLU backward: per-(sample, channel) sums of dz and dz*xhat in
//                          f64, then dx = rstd (gamma dz - mean_g(gamma dz) - xhat mean_g(gamma dz xhat)), d gamma, d beta
//   down_dgrad_kernel      transposed stride-2 convolution of the Downsample layer (unet.py:71-75)
//   sum2x2_add_kernel      nearest-x2 upsampling backward;  nin_dgrad_kernel  1x1 shortcut backward;  axpy_kernel
// Gradient tensors are ACCUMULATED (+=): a forward tensor may have several consumers (skip connections, residuals) -- except the FIRST
// contribution of a backward walk, which writes (the maps are not zeroed: unet_bwd_host.h, UNetBwdCall::take_first).
#pragma once
#include "unet_kernels.h"

namespace gc {

struct GnArgs {
  const float* x;        // [n][8][HW] one 8-channel source
  const double* stat;    // [n][8][2]
  const float* gamma;    // [8] of this source
  const float* beta;
  const float* da;       // [n][da_ctotal][HW]: gradient w.r.t. SiLU(GN(x)), this source's channels at da_coff (backward)
  float* out;            // fwd: A [n][out_ctotal][HW] at channel offset out_coff;  bwd apply: G[x] [n][8][HW] (+=)
  double* red;           // [n][8][2] f64: sum dz, sum dz*xhat (backward)
  double inv_cnt;        // 1 / (gs * HW)
  int gs, HW, out_ctotal, out_coff, da_ctotal, da_coff;
  int da_is_dz;          // backward apply: `da` already holds dz = dA * SiLU'(.) (written by the input-gradient convolution's epilogue)
  const float* add;      // backward apply, optional [n][8][HW]: a further gradient of the same tensor added in this pass (the identity shortcut's d out)
  int first;             // backward apply: this is the first contribution to G[x] in the walk -- `out` is written, not read (it was never zeroed)
};

__global__ __launch_bounds__(256) void gn_silu_fwd_kernel(const GnArgs a) {
  const int n = blockIdx.z, c = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  float A, B;
  gn_coeff(a.stat + (size_t)n * 16, c, a.gs, a.inv_cnt, a.gamma[c], a.beta[c], &A, &B);
  if (p < a.HW) a.out[((size_t)n * a.out_ctotal + a.out_coff + c) * a.HW + p] = silu_f(fmaf(A, a.x[((size_t)n * 8 + c) * a.HW + p], B));
}

__global__ __launch_bounds__(256) void gn_silu_bwd_reduce_kernel(const GnArgs a) {
  __shared__ double s_red[4][2];
  const int n = blockIdx.z, c = blockIdx.y, tid = threadIdx.x;
  float mean, rstd;
  gn_mean_rstd(a.stat + (size_t)n * 16, c, a.gs, a.inv_cnt, &mean, &rstd);
  const float g = a.gamma[c], b = a.beta[c];
  double s1 = 0.0, s2 = 0.0;
  for (int p = blockIdx.x * 256 + tid; p < a.HW; p += gridDim.x * 256) {
    const float xh = (a.x[((size_t)n * 8 + c) * a.HW + p] - mean) * rstd;
    const float dz = a.da[((size_t)n * a.da_ctotal + a.da_coff + c) * a.HW + p] * silu_grad_f(fmaf(g, xh, b));
    s1 += dz;
    s2 += (double)dz * xh;
  }
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if ((tid & 63) == 0) { s_red[tid >> 6][0] = s1; s_red[tid >> 6][1] = s2; }
  __syncthreads();
  if (tid < 2) atomicAdd(&a.red[((size_t)n * 8 + c) * 2 + tid], s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid]);
}

__global__ __launch_bounds__(256) void gn_silu_bwd_apply_kernel(const GnArgs a) {
  const int n = blockIdx.z, c = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  float mean, rstd;
  gn_mean_rstd(a.stat + (size_t)n * 16, c, a.gs, a.inv_cnt, &mean, &rstd);
  const int g0 = c & ~(a.gs - 1);
  double m1 = 0.0, m2 = 0.0;
  for (int j = 0; j < a.gs; ++j) {
    m1 += (double)a.gamma[g0 + j] * a.red[((size_t)n * 8 + g0 + j) * 2 + 0];
    m2 += (double)a.gamma[g0 + j] * a.red[((size_t)n * 8 + g0 + j) * 2 + 1];
  }
  const float f1 = (float)(m1 * a.inv_cnt), f2 = (float)(m2 * a.inv_cnt);
  if (p < a.HW) {
    const size_t e = ((size_t)n * 8 + c) * a.HW + p;
    const float g = a.gamma[c], xh = (a.x[e] - mean) * rstd;
    const float d0 = a.da[((size_t)n * a.da_ctotal + a.da_coff + c) * a.HW + p];
    const float dz = a.da_is_dz ? d0 : d0 * silu_grad_f(fmaf(g, xh, a.beta[c]));
    a.out[e] = (a.first ? 0.f : a.out[e]) + rstd * (g * dz - f1 - xh * f2) + (a.add != nullptr ? a.add[e] : 0.f);
  }
}
// HW % 4 == 0: four pixels per lane, 128-bit accesses (second half of round 4: 93 launches per UNet call backwards, one dword per lane before)
__global__ __launch_bounds__(256) void gn_silu_bwd_apply4_kernel(const GnArgs a) {
  const int n = blockIdx.z, c = blockIdx.y, p = 4 * (blockIdx.x * 256 + threadIdx.x);
  float mean, rstd;
  gn_mean_rstd(a.stat + (size_t)n * 16, c, a.gs, a.inv_cnt, &mean, &rstd);
  const int g0 = c & ~(a.gs - 1);
  double m1 = 0.0, m2 = 0.0;
  for (int j = 0; j < a.gs; ++j) {
    m1 += (double)a.gamma[g0 + j] * a.red[((size_t)n * 8 + g0 + j) * 2 + 0];
    m2 += (double)a.gamma[g0 + j] * a.red[((size_t)n * 8 + g0 + j) * 2 + 1];
  }
  const float f1 = (float)(m1 * a.inv_cnt), f2 = (float)(m2 * a.inv_cnt);
  if (p >= a.HW) return;
  const size_t e = ((size_t)n * 8 + c) * a.HW + p;
  const float g = a.gamma[c], bt = a.beta[c];
  const float4 x4 = *reinterpret_cast<const float4*>(a.x + e);
  const float4 d4 = *reinterpret_cast<const float4*>(a.da + ((size_t)n * a.da_ctotal + a.da_coff + c) * a.HW + p);
  const float4 o4 = a.first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(a.out + e);
  const float4 a4 = a.add != nullptr ? *reinterpret_cast<const float4*>(a.add + e) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ds[4] = {d4.x, d4.y, d4.z, d4.w}, as[4] = {a4.x, a4.y, a4.z, a4.w};
  float os[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float xh = (xs[j] - mean) * rstd;
    const float dz = a.da_is_dz ? ds[j] : ds[j] * silu_grad_f(fmaf(g, xh, bt));
    os[j] += rstd * (g * dz - f1 - xh * f2) + as[j];
  }
  *reinterpret_cast<float4*>(a.out + e) = make_float4(os[0], os[1], os[2], os[3]);
}

// d gamma[c] += sum_n red[n][c][1], d beta[c] += sum_n red[n][c][0] for EVERY GroupNorm use of a UNet call at once: the
// backward walk gives each use its own slot of the reduction table and this kernel runs after the walk (one launch
// instead of one per GroupNorm).  blockIdx.x = use.
constexpr int kMaxGnUses = 192;
struct GnParamArgs {
  const double* red;   // [uses][n][8][2]
  float* graw;
  int n, uses;
  int dgamma[kMaxGnUses], dbeta[kMaxGnUses];   // float offsets into graw
};
__global__ __launch_bounds__(64) void gn_param_grad_all_kernel(const GnParamArgs a) {
  const int u = blockIdx.x, c = threadIdx.x;
  if (c >= 8) return;
  const double* __restrict__ red = a.red + (size_t)u * a.n * 16;
  double s1 = 0.0, s2 = 0.0;
  for (int i = 0; i < a.n; ++i) { s1 += red[((size_t)i * 8 + c) * 2 + 0]; s2 += red[((size_t)i * 8 + c) * 2 + 1]; }
  atomicAdd(&a.graw[a.dgamma[u] + c], (float)s2);   // two uses never share a parameter, but the blob is "+=" by contract
  atomicAdd(&a.graw[a.dbeta[u] + c], (float)s1);
}

// Timestep path backwards (unet.py:309-312, :124): the timestep term only shifts conv1's bias, so d conv1.bias of block b
// (already in graw, summed over the samples: every sample has the same t) is the gradient of temb_proj_b(SiLU(temb)).
//   emb -> h0 = W0 emb + b0 -> a0 = SiLU(h0) -> h1 = W1 a0 + b1 -> a1 = SiLU(h1) -> proj_b = Wp_b a1 + bp_b
// One workgroup; gradients are ADDED to graw at the parameters' raw offsets.
struct TembBwdArgs {
  const float* raw;
  float* graw;
  long long d0w, d0b, d1w, d1b;
  long long tpw[kMaxResBlocks], tpb[kMaxResBlocks], c1b[kMaxResBlocks];
  int nblocks, t;
};
__global__ __launch_bounds__(64) void temb_bwd_kernel(const TembBwdArgs a) {
  __shared__ float e[8], h0[32], a0[32], h1[32], a1[32], da1[32], dh1[32], da0[32];
  const int j = threadIdx.x;
  if (j < 8) {
    const int k = j & 3;
    const float f = expf((float)k * -(9.210340371976184f / 3.0f));
    const float ang = (float)a.t * f;
    e[j] = j < 4 ? sinf(ang) : cosf(ang);
  }
  __syncthreads();
  if (j < 32) {
    float s = a.raw[a.d0b + j];
    for (int k = 0; k < 8; ++k) s = fmaf(a.raw[a.d0w + j * 8 + k], e[k], s);
    h0[j] = s; a0[j] = s / (1.0f + expf(-s));
  }
  __syncthreads();
  if (j < 32) {
    float s = a.raw[a.d1b + j];
    for (int k = 0; k < 32; ++k) s = fmaf(a.raw[a.d1w + j * 32 + k], a0[k], s);
    h1[j] = s; a1[j] = s / (1.0f + expf(-s));
  }
  __syncthreads();
  // temb_proj of every block: d bias = g, d W[o][k] = g[o] a1[k], d a1[k] = sum_b sum_o W_b[o][k] g_b[o]
  if (j < 32) {
    float acc = 0.f;
    for (int b = 0; b < a.nblocks; ++b)
      for (int o = 0; o < 8; ++o) acc = fmaf(a.raw[a.tpw[b] + o * 32 + j], a.graw[a.c1b[b] + o], acc);
    da1[j] = acc;
  }
  for (int i = j; i < a.nblocks * 8; i += 64) {
    const int b = i >> 3, o = i & 7;
    const float g = a.graw[a.c1b[b] + o];
    a.graw[a.tpb[b] + o] += g;
    for (int k = 0; k < 32; ++k) a.graw[a.tpw[b] + o * 32 + k] += g * a1[k];
  }
  __syncthreads();
  auto dsilu = [](float z) { const float s = 1.0f / (1.0f + expf(-z)); return s * fmaf(z, 1.0f - s, 1.0f); };
  if (j < 32) {
    dh1[j] = da1[j] * dsilu(h1[j]);
    a.graw[a.d1b + j] += dh1[j];
    for (int k = 0; k < 32; ++k) a.graw[a.d1w + j * 32 + k] += dh1[j] * a0[k];
  }
  __syncthreads();
  if (j < 32) {
    float acc = 0.f;
    for (int i = 0; i < 32; ++i) acc = fmaf(a.raw[a.d1w + i * 32 + j], dh1[i], acc);
    da0[j] = acc * dsilu(h0[j]);   // = d h0
    a.graw[a.d0b + j] += da0[j];
    for (int k = 0; k < 8; ++k) a.graw[a.d0w + j * 8 + k] += da0[j] * e[k];
  }
}

// ---------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* dy;   // [n][Cout][Ho][Wo]
  const float* x0;   // [n][c0][Hs][Ws]  (Hs = Hi, or Hi / 2 when up)
  const float* x1;   // [n][c1][Hs][Ws] or null: the input is cat[x0, x1]
  float* dw;         // [Cout][c0 + c1][K][K]  (+=)
  float* db;         // [Cout] (+=) or null
  int Cout, c0, c1, Ho, Wo, Hi, Wi, K, stride, pad, up;
  // optional: the input is SiLU(GroupNorm(x)) of 8-channel sources (x0 = first source, x1 = second), applied while the tile is
  // staged -- the forward never materialises that tensor and neither does the backward
  const double* gn_stat[2] = {nullptr, nullptr};   // [n][8][2] per source
  const float* gn_gamma = nullptr;                 // [c0 + c1]
  const float* gn_beta = nullptr;
  double gn_inv_cnt = 0.0;
  int gn_gs = 0;
  // optional scratch of conv_wgrad_scratch_floats(...) floats: the workgroups' partial sums are written there and summed by a second
  // kernel instead of one float atomic per (workgroup, weight) on the same few hundred addresses (31 of 80 us per 8 -> 8 layer)
  float* part = nullptr;
  int tpc = 1;   // tiles per workgroup of conv_wgrad_mfma_kernel (set by conv_wgrad_enqueue)
};

template <int K, int STRIDE>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int TP = 16, PH = (TP - 1) * STRIDE + K, PS = PH * PH, KK = K * K;
  __shared__ float sdy[8][TP * TP + 1];
  __shared__ float sx[8 * PS];
  __shared__ float s_acc[4][64 * KK];
  const int tid = threadIdx.x, n = blockIdx.z;
  const int Cin = a.c0 + a.c1, icc = (Cin + 7) / 8;
  const int occ = blockIdx.y / icc, ich = blockIdx.y - occ * icc;
  const int tiles_x = (a.Wo + TP - 1) / TP;
  const int oy0 = (blockIdx.x / tiles_x) * TP, ox0 = (blockIdx.x % tiles_x) * TP;
  const int Hs = a.up ? a.Hi / 2 : a.Hi, Ws = a.up ? a.Wi / 2 : a.Wi;
  for (int i = tid; i < 8 * TP * TP; i += 256) {
    const int o = i / (TP * TP), r = i - o * (TP * TP), py = r / TP, px = r - py * TP;
    const int oc = occ * 8 + o, oy = oy0 + py, ox = ox0 + px;
    sdy[o][r] = (oc < a.Cout && oy < a.Ho && ox < a.Wo) ? a.dy[(((size_t)n * a.Cout + oc) * a.Ho + oy) * a.Wo + ox] : 0.f;
  }
  const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;
  __shared__ float s_gn[8][2];
  const bool gn = a.gn_stat[0] != nullptr;   // 8-channel sources: chunk ich is source ich
  if (gn) {
    if (tid < 8) {
      const int ic = ich * 8 + tid;
      float A = 0.f, B = 0.f;
      if (ic < Cin) gn_coeff(a.gn_stat[ic < a.c0 ? 0 : 1] + (size_t)n * 16, tid, a.gn_gs, a.gn_inv_cnt, a.gn_gamma[ic], a.gn_beta[ic], &A, &B);
      s_gn[tid][0] = A; s_gn[tid][1] = B;
    }
    __syncthreads();
  }
  for (int i = tid; i < 8 * PS; i += 256) {
    const int c = i / PS, r = i - c * PS, py = r / PH, px = r - py * PH;
    const int ic = ich * 8 + c, iy = iy0 + py, ix = ix0 + px;
    float v = 0.f;
    if (ic < Cin && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) {
      const float* __restrict__ sp = ic < a.c0 ? a.x0 + ((size_t)n * a.c0 + ic) * Hs * Ws : a.x1 + ((size_t)n * a.c1 + (ic - a.c0)) * Hs * Ws;
      v = a.up ? sp[(size_t)(iy >> 1) * Ws + (ix >> 1)] : sp[(size_t)iy * Ws + ix];
      if (gn) v = silu_f(fmaf(s_gn[c][0], v, s_gn[c][1]));   // zero padding stays zero: only inside the image
    }
    sx[i] = v;
  }
  __syncthreads();
  const int pair = tid & 63, q = tid >> 6, ol = pair >> 3, il = pair & 7;
  float acc[KK];
#pragma unroll
  for (int t = 0; t < KK; ++t) acc[t] = 0.f;
  for (int py = 4 * q; py < 4 * q + 4; ++py)
    for (int px = 0; px < TP; ++px) {
      const float d = sdy[ol][py * TP + px];
      const float* __restrict__ xp = sx + il * PS + (py * STRIDE) * PH + px * STRIDE;
#pragma unroll
      for (int t = 0; t < KK; ++t) acc[t] = fmaf(d, xp[(t / K) * PH + (t % K)], acc[t]);
    }
#pragma unroll
  for (int t = 0; t < KK; ++t) s_acc[q][pair * KK + t] = acc[t];
  __syncthreads();
  for (int i = tid; i < 64 * KK; i += 256) {
    const int pr = i / KK, t = i - pr * KK, oc = occ * 8 + (pr >> 3), ic = ich * 8 + (pr & 7);
    if (oc < a.Cout && ic < Cin) atomicAdd(&a.dw[((size_t)oc * Cin + ic) * KK + t], s_acc[0][i] + s_acc[1][i] + s_acc[2][i] + s_acc[3][i]);
  }
  if (a.db != nullptr && ich == 0 && tid < 8 && occ * 8 + tid < a.Cout) {
    float s = 0.f;
    for (int r = 0; r < TP * TP; ++r) s += sdy[tid][r];
    atomicAdd(&a.db[occ * 8 + tid], s);
  }
}

// The same gradient on the fp32 matrix cores (round 3), stride 1 and "same" padding (every 3x3 and 1x1 layer of
// the UNet but the stride-2 Downsample): a GEMM with the PIXELS as the reduction dimension,
//     dW[oc][(ic, tap)] = sum_pixels dY[oc][pixel] X[ic][pixel + tap],
// on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation): M = 16 rows = the chunk's 8 output channels + 8 zero rows,
// N = 8 input channels x K*K taps + ONE column of ones (its row sums are the bias gradient) in tiles of 16, K = 4 consecutive
// pixels of a row per instruction.  A workgroup stages a 32x16-pixel tile of dY (8 planes + a zero plane for the padding rows) and
// the haloed tile of X (8 planes + a plane of ones; SiLU(GroupNorm(x)) and the nearest-x2 index are applied while staging, as in
// the VALU kernel above); wave w walks rows 4w..4w+3: per 4 pixels one A read, NT B reads (ds_read_b32 each) and NT matrix
// instructions.  The four waves' accumulators meet in LDS; one atomic per (oc, ic, tap) and workgroup.
// VALU kernel: 82.6 us per 8 -> 8 layer at 4 x 200 x 704 (10 LDS reads per 9 FMAs); this one: see DESIGN.md section 4 "Backward".
template <int K>
__global__ __launch_bounds__(256, 3) void conv_wgrad_mfma_kernel(const WgradArgs a) {
  constexpr int TW = 32, TH = 16, PAD = K / 2, PW = TW + 2 * PAD, PHt = TH + 2 * PAD, KK = K * K;
  // K = 3 (PACK): the 16 rows of the A operand are the 8 output channels at dY rows y AND y + 1, the columns (ic, window row 0..3, dx):
  // rows 0..7 take their taps from window rows 0..2, rows 8..15 from window rows 1..3 (the same X rows seen from one dY row lower), so one
  // instruction serves two image rows -- 6 column tiles per row PAIR instead of 5 per row (8 of the 16 A rows were zeros); the bias gradient
  // is the plain sum of the A values a lane reads.  K = 1: rows 8..15 are zeros and a column of ones carries the bias, as before.
  constexpr bool PACK = K == 3;
  constexpr int NC = 8 * KK + 1, NTO = (NC + 15) / 16;         // record layout of the partial sums: (ic, tap) pairs + the bias column
  constexpr int NT = PACK ? 6 : NTO;                           // column tiles of the matrix phase (PACK: 8 x 12 = 96 columns)
  constexpr int DRS = PACK ? TW + 4 : TW;                      // dY row stride (PACK: = 4 mod 32 words, plane = 8 mod 32: rows y / y + 1 of a channel on distinct banks)
  constexpr int DPS = PACK ? TH * DRS + 8 : TH * TW + 4;       // dY plane stride: +4 words puts the 8 channels of a pixel on 8 x 4 distinct banks
#ifdef WG_DIAG   // diagnostic builds (tools/diag/wgrad_ab.sh; results are wrong by construction): 1 = no matrix phase, 2 = no B-operand LDS reads,
                 // 4 = LDS reads without matrix instructions, 8 = no SiLU in the staging, 16 = no LDS stores of the X tile, 32 = no global loads
  constexpr int WGD = WG_DIAG;
#else
  constexpr int WGD = 0;
#endif
  // X tile rows start LEAD columns left of the output tile so that the aligned 128-bit quads of the image land on aligned 128-bit LDS
  // stores (one ds_write_b128 per quad instead of four predicated scalar stores); plane stride = 4 mod 32 words: the (channel, tap row)
  // bases of the 16 columns of an operand read then fall on distinct banks
  constexpr int LEAD = PAD ? 4 : 0;
  constexpr int XQ = (LEAD + TW + PAD + 3) / 4;                // aligned quads per haloed row: 10 for K = 3 (columns -4 .. 35), 8 for K = 1
  constexpr int PWS = 4 * XQ, XPS = PHt * PWS + (K == 3 ? 20 : 4);   // X row / plane stride
  constexpr int RED = PACK ? 4 * NT * 256 + 32 : 4 * NT * 16 * 8;   // reduction buffer of the four waves (aliases the tiles)
  constexpr int NPL = PACK ? 8 : 9;                            // planes per operand tile (K = 1: + the zero / ones plane)
  constexpr int TILE_WORDS = NPL * DPS + NPL * XPS;
  __shared__ __align__(16) float smem[TILE_WORDS > RED ? TILE_WORDS : RED];
  __shared__ float s_gn[8][2];
  float* const sdy = smem;                // [NPL][DPS]; K = 1: plane 8 = zeros (rows 8..15 of the A operand)
  float* const sx = smem + NPL * DPS;     // [9][XPS], plane 8 = ones
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = blockIdx.z;
  const int Cin = a.c0 + a.c1, icc = (Cin + 7) / 8;
  const int occ = blockIdx.y / icc, ich = blockIdx.y - occ * icc;
  const int tiles_x = (a.Wo + TW - 1) / TW, tiles = tiles_x * ((a.Ho + TH - 1) / TH);
  const int Hs = a.up ? a.Hi / 2 : a.Hi, Ws = a.up ? a.Wi / 2 : a.Wi;
  const bool gn = a.gn_stat[0] != nullptr;   // 8-channel sources: chunk ich is source ich
  // Round 4: a workgroup walks a.tpc consecutive tiles of its sample with the accumulators kept in registers (one cross-wave reduction
  // and one partial-sum record per workgroup instead of per tile) and -- fast path -- the NEXT tile's global loads in flight during
  // this tile's matrix phase: a tile cost one full load -> stage -> MFMA -> reduce chain (~14 us, three workgroup rounds per launch).
  // Fast path (image width a multiple of 4, no nearest-x2 source): 128-bit loads -- 4 per thread for dY, <= 6 for X (aligned quads
  // covering columns ox0 - 4 .. ox0 + 35 of the 34-column haloed row) -- instead of 18 + 22 scalar ones.
  const bool vec = !a.up && (a.Wo & 3) == 0 && (a.Wi & 3) == 0;
  const int tpc = a.tpc > 0 ? a.tpc : 1;
  const int t_begin = blockIdx.x * tpc, t_end = min(t_begin + tpc, tiles);
  if (gn && tid < 8) {
    const int ic = ich * 8 + tid;
    float A = 0.f, B = 0.f;
    if (ic < Cin) gn_coeff(a.gn_stat[ic < a.c0 ? 0 : 1] + (size_t)n * 16, tid, a.gn_gs, a.gn_inv_cnt, a.gn_gamma[ic], a.gn_beta[ic], &A, &B);
    s_gn[tid][0] = A; s_gn[tid][1] = B;
  }
  if constexpr (!PACK) {
    for (int i = tid; i < DPS; i += 256) sdy[8 * DPS + i] = 0.f;                 // the zero plane (rows 8..15 of the A operand)
    for (int i = tid; i < PHt * PWS; i += 256) sx[8 * XPS + i] = 1.0f;             // the plane of ones (bias column)
  }

  const int j = lane & 15, kq = lane >> 4;
  const int aoff = PACK ? (j & 7) * DPS + (j >> 3) * DRS + kq : (j < 8 ? j : 8) * DPS + kq;
  int boff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int c = 16 * t + j;
    if constexpr (PACK) {
      const int ic = c / 12, w = c - ic * 12;                   // w = window row (0..3) * 3 + dx
      boff[t] = ic * XPS + (w / 3) * PWS + (w % 3) + kq + (LEAD - PAD);
    } else {
      const int ic = c < 8 * KK ? c / KK : 8, tap = c < 8 * KK ? c - ic * KK : 0;
      boff[t] = ic * XPS + (tap / K) * PWS + (tap % K) + kq + (LEAD - PAD);
    }
  }
  float bsum = 0.f;   // PACK: sum of this lane's A values = its share of the bias gradient of channel j & 7
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int QD = 8 * TH * (TW / 4) / 256;                 // dY quads per thread: 4
  constexpr int QX = (8 * PHt * XQ + 255) / 256;              // X quads per thread: 6 (K = 3) / 4 (K = 1)
  float4 qd[QD], qx[QX];
  auto load_vec = [&](int tile) {
    const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
#pragma unroll
    for (int k = 0; k < QD; ++k) {
      const int i = tid + 256 * k;
      const int o = i / (TH * (TW / 4)), r = i - o * (TH * (TW / 4)), py = r / (TW / 4), q4 = r - py * (TW / 4);
      const int oc = occ * 8 + o, oy = oy0 + py, ox = ox0 + 4 * q4;
      qd[k] = (oc < a.Cout && oy < a.Ho && ox < a.Wo) ? *reinterpret_cast<const float4*>(a.dy + (((size_t)n * a.Cout + oc) * a.Ho + oy) * a.Wo + ox)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < QX; ++k) {
      const int i = tid + 256 * k;
      const int c = i / (PHt * XQ), r = i - c * (PHt * XQ), py = r / XQ, q4 = r - py * XQ;
      const int ic = ich * 8 + c, iy = oy0 - PAD + py, ix = ox0 - LEAD + 4 * q4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < 8 && ic < Cin && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) {
        const float* __restrict__ sp = ic < a.c0 ? a.x0 + ((size_t)n * a.c0 + ic) * Hs * Ws : a.x1 + ((size_t)n * a.c1 + (ic - a.c0)) * Hs * Ws;
        v = *reinterpret_cast<const float4*>(sp + (size_t)iy * Ws + ix);
      }
      qx[k] = v;
    }
  };
  if (WGD & 32) {
#pragma unroll
    for (int k = 0; k < QD; ++k) qd[k] = make_float4(1.f, 2.f, 3.f, 4.f);
#pragma unroll
    for (int k = 0; k < QX; ++k) qx[k] = make_float4(1.f, 2.f, 3.f, 4.f);
  }
  if (vec && t_begin < t_end && !(WGD & 32)) load_vec(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
    __syncthreads();   // s_gn / the constant planes (first tile); the previous tile's matrix instructions are done
    if (vec) {
#pragma unroll
      for (int k = 0; k < QD; ++k) {
        const int i = tid + 256 * k;
        const int o = i / (TH * (TW / 4)), r = i - o * (TH * (TW / 4)), py = r / (TW / 4), q4 = r - py * (TW / 4);
        *reinterpret_cast<float4*>(sdy + o * DPS + py * DRS + 4 * q4) = qd[k];
      }
#pragma unroll
      for (int k = 0; k < QX; ++k) {
        const int i = tid + 256 * k;
        const int c = i / (PHt * XQ), r = i - c * (PHt * XQ), py = r / XQ, q4 = r - py * XQ;
        if (c < 8) {
          const int ic = ich * 8 + c, iy = oy0 - PAD + py, ix = ox0 - LEAD + 4 * q4;
          const bool inside = ic < Cin && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;   // a quad is inside or outside as a whole (W % 4 == 0)
          float4 v = qx[k];
          if (gn && inside && !(WGD & 8)) {   // zero padding stays zero
            v.x = silu_f(fmaf(s_gn[c][0], v.x, s_gn[c][1])); v.y = silu_f(fmaf(s_gn[c][0], v.y, s_gn[c][1]));
            v.z = silu_f(fmaf(s_gn[c][0], v.z, s_gn[c][1])); v.w = silu_f(fmaf(s_gn[c][0], v.w, s_gn[c][1]));
          }
          if (!(WGD & 16) || v.x == 12345.f) *reinterpret_cast<float4*>(sx + c * XPS + py * PWS + 4 * q4) = v;
        }
      }
      __syncthreads();
      if (tile + 1 < t_end && !(WGD & 32)) load_vec(tile + 1);   // in flight during this tile's matrix phase
    } else {
      constexpr int ND = 8 * TH * TW / 256, NX = (8 * PHt * PW + 255) / 256;
#pragma unroll 1
      for (int k = 0; k < ND; ++k) {
        const int i = tid + 256 * k;
        const int o = i / (TH * TW), r = i - o * (TH * TW), py = r / TW, px = r - py * TW;
        const int oc = occ * 8 + o, oy = oy0 + py, ox = ox0 + px;
        sdy[o * DPS + py * DRS + px] = (oc < a.Cout && oy < a.Ho && ox < a.Wo) ? a.dy[(((size_t)n * a.Cout + oc) * a.Ho + oy) * a.Wo + ox] : 0.f;
      }
#pragma unroll 1
      for (int k = 0; k < NX; ++k) {
        const int i = tid + 256 * k;
        if (i < 8 * PHt * PW) {
          const int c = i / (PHt * PW), r = i - c * (PHt * PW), py = r / PW, px = r - py * PW;
          const int ic = ich * 8 + c, iy = oy0 - PAD + py, ix = ox0 - PAD + px;
          float v = 0.f;
          if (ic < Cin && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) {
            const float* __restrict__ sp = ic < a.c0 ? a.x0 + ((size_t)n * a.c0 + ic) * Hs * Ws : a.x1 + ((size_t)n * a.c1 + (ic - a.c0)) * Hs * Ws;
            v = a.up ? sp[(size_t)(iy >> 1) * Ws + (ix >> 1)] : sp[(size_t)iy * Ws + ix];
            if (gn) v = silu_f(fmaf(s_gn[c][0], v, s_gn[c][1]));   // zero padding stays zero
          }
          sx[c * XPS + py * PWS + px + (LEAD - PAD)] = v;
        }
      }
      __syncthreads();
    }
    if (!(WGD & 1)) {
#pragma unroll 1
    for (int y = 4 * wave; y < 4 * wave + 4; y += PACK ? 2 : 1) {
#pragma unroll 4
      for (int s4 = 0; s4 < TW / 4; ++s4) {
        const float av = sdy[aoff + y * DRS + 4 * s4];
        if constexpr (PACK) bsum += av;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float bv = (WGD & 2) ? av : sx[boff[t] + y * PWS + 4 * s4];
          if (WGD & 4) acc[t][0] += av * bv;
          else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
    }
  }
  __syncthreads();   // the tiles are dead: the reduction buffer aliases them
  if constexpr (PACK) {
    // all 16 rows are live: [wave][column 16 t + j][row 4 kq + reg]; the wave's bias sums go behind the four waves' records
#pragma unroll
    for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(smem + ((wave * NT + t) * 16 + j) * 16 + 4 * kq) = acc[t];
    bsum += __shfl_xor(bsum, 8, 64);
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (lane < 8) smem[4 * NT * 256 + wave * 8 + lane] = bsum;
  } else if (kq < 2) {      // rows 0..7 = output channels 4 kq + reg live in lanes 0..31
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) smem[((wave * NT + t) * 16 + j) * 8 + 4 * kq + reg] = acc[t][reg];
  }
  __syncthreads();
  for (int i = tid; i < NTO * 16 * 8; i += 256) {
    const int c = i >> 3, o = i & 7, oc = occ * 8 + o;
    float v = 0.f;
    if constexpr (PACK) {
      if (c < 8 * KK) {        // dW[o][ic][ty][tx] = row o of column (ic, ty, tx) + row 8 + o of column (ic, ty + 1, tx)
        const int icl = c / KK, tap = c - icl * KK, c0 = (icl * 12 + tap) * 16 + o, c1 = (icl * 12 + tap + 3) * 16 + 8 + o;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += smem[w * NT * 256 + c0] + smem[w * NT * 256 + c1];
      } else if (c == 8 * KK) {
#pragma unroll
        for (int w = 0; w < 4; ++w) v += smem[4 * NT * 256 + w * 8 + o];
      }
    } else {
      v = smem[i] + smem[NT * 128 + i] + smem[2 * NT * 128 + i] + smem[3 * NT * 128 + i];
    }
    if (a.part != nullptr) {   // [pair][workgroup of the pair][NTO * 128]: coalesced, summed by conv_wgrad_reduce_kernel
      const size_t wg = blockIdx.x + (size_t)gridDim.x * blockIdx.z;
      a.part[((size_t)blockIdx.y * gridDim.x * gridDim.z + wg) * (NTO * 128) + i] = v;
      continue;
    }
    if (oc >= a.Cout) continue;
    if (c < 8 * KK) {
      const int icl = c / KK, tap = c - icl * KK, ic = ich * 8 + icl;
      if (ic < Cin) atomicAdd(&a.dw[((size_t)oc * Cin + ic) * KK + tap], v);
    } else if (c == 8 * KK && a.db != nullptr && ich == 0) {
      atomicAdd(&a.db[oc], v);
    }
  }
}

// second stage: column i of pair (occ, ich) summed over the pair's workgroups in `slices` slices (blockIdx.z), one atomic per slice
template <int K>
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const WgradArgs a, int nwg, int per_slice) {
  constexpr int KK = K * K, NT = (8 * KK + 1 + 15) / 16, NW = NT * 128;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= NW) return;
  const int Cin = a.c0 + a.c1, icc = (Cin + 7) / 8;
  const int occ = blockIdx.y / icc, ich = blockIdx.y - occ * icc;
  const int w0 = blockIdx.z * per_slice, w1 = min(w0 + per_slice, nwg);
  const float* __restrict__ p = a.part + ((size_t)blockIdx.y * nwg + w0) * NW + i;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int w = w0;
  for (; w + 3 < w1; w += 4, p += 4 * (size_t)NW) { s0 += p[0]; s1 += p[NW]; s2 += p[2 * (size_t)NW]; s3 += p[3 * (size_t)NW]; }
  for (; w < w1; ++w, p += NW) s0 += p[0];
  const float v = (s0 + s1) + (s2 + s3);
  const int c = i >> 3, o = i & 7, oc = occ * 8 + o;
  if (oc >= a.Cout) return;
  if (c < 8 * KK) {
    const int icl = c / KK, tap = c - icl * KK, ic = ich * 8 + icl;
    if (ic < Cin) atomicAdd(&a.dw[((size_t)oc * Cin + ic) * KK + tap], v);
  } else if (c == 8 * KK && a.db != nullptr && ich == 0) {
    atomicAdd(&a.db[oc], v);
  }
}

constexpr int kWgradDirectMaxWg = 64;   // conv_wgrad_enqueue: at most this many workgroups per channel pair add into dw directly

// floats of WgradArgs::part for a layer (stride-1 "same" layers only; 0 otherwise)
inline size_t conv_wgrad_scratch_floats(int Cout, int Cin, int K, int H, int W, int n) {
  const size_t NT = (size_t)(8 * K * K + 1 + 15) / 16;
  return (size_t)((H + 15) / 16) * ((W + 31) / 32) * n * ((Cout + 7) / 8) * ((Cin + 7) / 8) * NT * 128;
}

inline int conv_wgrad_enqueue(const WgradArgs& a, int n, hipStream_t st) {
  const int Cin = a.c0 + a.c1;
  if (a.stride == 1 && (a.K == 3 || a.K == 1) && a.pad == a.K / 2 && a.Ho == a.Hi && a.Wo == a.Wi) {   // fp32 matrix cores
    const int tiles = ((a.Ho + 15) / 16) * ((a.Wo + 31) / 32);
    const long long pairs = (long long)((a.Cout + 7) / 8) * ((Cin + 7) / 8);
    // tiles per workgroup: the fewest that make the whole launch resident at once (3 workgroups per CU = 768), at most 8
    WgradArgs b = a;
    b.tpc = (int)std::min<long long>(8, std::max<long long>(1, ((long long)tiles * pairs * n + 767) / 768));
    const dim3 grid((tiles + b.tpc - 1) / b.tpc, (unsigned)pairs, n);
    if (grid.y > 65535 || grid.z > 65535) return fail(GC_ERR_ARG, "conv_wgrad: too many channel chunks / samples");
    // Few workgroups per channel pair (the UNet of a training step at the shipped map sizes: 64 / 16 / 4): each adds its columns straight
    // into dw -- kWgradDirectMaxWg-way contention per address at most, what the second stage's 16 slices already have -- and the second
    // launch (a third of this layer's backward launches) is not issued.  Larger maps keep the partial sums + reduce.
    const int nwg = (int)(grid.x * grid.z);
    if (nwg <= kWgradDirectMaxWg) b.part = nullptr;
    if (a.K == 3) conv_wgrad_mfma_kernel<3><<<grid, 256, 0, st>>>(b);
    else conv_wgrad_mfma_kernel<1><<<grid, 256, 0, st>>>(b);
    if (b.part != nullptr) {
      const int slices = std::min(nwg, 16), per = (nwg + slices - 1) / slices;
      const int NW = ((8 * a.K * a.K + 1 + 15) / 16) * 128;
      const dim3 rg((NW + 255) / 256, grid.y, (nwg + per - 1) / per);
      if (a.K == 3) conv_wgrad_reduce_kernel<3><<<rg, 256, 0, st>>>(a, nwg, per);
      else conv_wgrad_reduce_kernel<1><<<rg, 256, 0, st>>>(a, nwg, per);
    }
    GC_HIP(hipGetLastError());
    return GC_OK;
  }
  const dim3 grid(((a.Ho + 15) / 16) * ((a.Wo + 15) / 16), ((a.Cout + 7) / 8) * ((Cin + 7) / 8), n);
  if (grid.y > 65535 || grid.z > 65535) return fail(GC_ERR_ARG, "conv_wgrad: too many channel chunks / samples");
  if (a.K == 3 && a.stride == 1) conv_wgrad_kernel<3, 1><<<grid, 256, 0, st>>>(a);
  else if (a.K == 3 && a.stride == 2) conv_wgrad_kernel<3, 2><<<grid, 256, 0, st>>>(a);
  else if (a.K == 1 && a.stride == 1) conv_wgrad_kernel<1, 1><<<grid, 256, 0, st>>>(a);
  else return fail(GC_ERR_ARG, "conv_wgrad: unsupported kernel size / stride");
  GC_HIP(hipGetLastError());
  return GC_OK;
}

// prepared weights (conv_kernels.h layout [(ci*9 + tap)][co]) of the INPUT-gradient convolution of a 3x3 stride-1 layer
// with forward weights w [Cout][Cin][3][3]: dgrad input channels = forward oc, outputs = forward ic in [ic0, ic0 + nic),
// taps flipped:  P[(oc*9 + (8 - tap))][ic - ic0] = w[oc][ic][tap].  Every layer of a UNet call in ONE launch (blockIdx.y = layer).
constexpr int kMaxDgradLayers = 192;
struct PrepDgradArgs {
  const float* raw;
  float* P;                      // base of the table
  int layers;
  int w_off[kMaxDgradLayers], p_off[kMaxDgradLayers];
  short cout[kMaxDgradLayers], cin[kMaxDgradLayers], ic0[kMaxDgradLayers], nic[kMaxDgradLayers];
  unsigned char blocked[kMaxDgradLayers];   // 1: conv_out_kernel's layout [ceil(nic / 16)][cout = 8][9][16], zero padded (conv_in's x_t gradient)
};
__global__ __launch_bounds__(256) void prep_dgrad_all_kernel(const PrepDgradArgs a) {
  const int L = blockIdx.y;
  const int Cin = a.cin[L], ic0 = a.ic0[L], nic = a.nic[L], total = a.cout[L] * nic * 9;
  const float* __restrict__ w = a.raw + a.w_off[L];
  float* __restrict__ P = a.P + a.p_off[L];
  if (a.blocked[L]) {
    const int cout = a.cout[L], tot = ((nic + 15) / 16) * cout * 144;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < tot; i += gridDim.x * 256) {
      const int o16 = i & 15, k = i >> 4, tp = k % 9, k2 = k / 9, oc = k2 % cout, ocb = k2 / cout;
      const int icl = ocb * 16 + o16;
      P[i] = icl < nic ? w[((size_t)oc * Cin + ic0 + icl) * 9 + (8 - tp)] : 0.f;
    }
    return;
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int icl = i % nic, k = i / nic, oc = k / 9, tp = k - oc * 9;
    P[i] = w[((size_t)oc * Cin + ic0 + icl) * 9 + (8 - tp)];
  }
}

// Downsample backward (pad right/bottom by one, 3x3 stride 2, no padding): G[src][ic](iy, ix) += sum over oc and the taps
// (ky, kx) with iy - ky = 2 oy, ix - kx = 2 ox inside the output
struct DownDgradArgs {
  const float* gd;  // [n][8][Ho][Wo]
  const float* w;   // raw [8][8][3][3]
  float* gs;        // [n][8][Hi][Wi] (+=, or = when `first`)
  int Ho, Wo, Hi, Wi;
  int first;        // first contribution to gs in the walk: written, not accumulated
};
__global__ __launch_bounds__(256) void down_dgrad_kernel(const DownDgradArgs a) {
  const int n = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.Hi * a.Wi) return;
  const int iy = i / a.Wi, ix = i - iy * a.Wi;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = iy - ky;
    if (ty < 0 || (ty & 1) || (ty >> 1) >= a.Ho) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = ix - kx;
      if (tx < 0 || (tx & 1) || (tx >> 1) >= a.Wo) continue;
      for (int oc = 0; oc < 8; ++oc) {
        const float g = a.gd[(((size_t)n * 8 + oc) * a.Ho + (ty >> 1)) * a.Wo + (tx >> 1)];
#pragma unroll
        for (int ic = 0; ic < 8; ++ic) acc[ic] = fmaf(as_const(a.w)[((oc * 8 + ic) * 3 + ky) * 3 + kx], g, acc[ic]);
      }
    }
  }
#pragma unroll
  for (int ic = 0; ic < 8; ++ic) {
    float* __restrict__ q = a.gs + ((size_t)n * 8 + ic) * a.Hi * a.Wi + i;
    *q = a.first ? acc[ic] : *q + acc[ic];
  }
}

// nearest x2 backward: gs[c](y, x) += sum of the 2x2 block of da
__global__ __launch_bounds__(256) void sum2x2_add_kernel(const float* __restrict__ da, float* __restrict__ gs, int planes, int Hs, int Ws, int first) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)planes * Hs * Ws) return;
  const int x = (int)(i % Ws), y = (int)((i / Ws) % Hs);
  const long long pl = i / ((long long)Hs * Ws);
  const float* __restrict__ d = da + (pl * (2 * Hs) + 2 * y) * (2 * Ws) + 2 * x;
  const float v = d[0] + d[1] + d[2 * Ws] + d[2 * Ws + 1];
  gs[i] = first ? v : gs[i] + v;
}

// 1x1 nin_shortcut backward: g0[ic] += sum_oc w[oc][ic] go[oc], g1[ic] += sum_oc w[oc][8 + ic] go[oc]   (w raw [8][16])
// f0 / f1: first contribution to g0 / g1 in the walk (written, not accumulated)
__global__ __launch_bounds__(256) void nin_dgrad_kernel(const float* __restrict__ go, const float* __restrict__ w, float* __restrict__ g0,
                                                        float* __restrict__ g1, int HW, int f0, int f1) {
  const int n = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  float o[8];
#pragma unroll
  for (int oc = 0; oc < 8; ++oc) o[oc] = go[((size_t)n * 8 + oc) * HW + p];
#pragma unroll
  for (int ic = 0; ic < 8; ++ic) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int oc = 0; oc < 8; ++oc) { s0 = fmaf(as_const(w)[oc * 16 + ic], o[oc], s0); s1 = fmaf(as_const(w)[oc * 16 + 8 + ic], o[oc], s1); }
    float* __restrict__ q0 = g0 + ((size_t)n * 8 + ic) * HW + p;
    float* __restrict__ q1 = g1 + ((size_t)n * 8 + ic) * HW + p;
    *q0 = f0 ? s0 : *q0 + s0;
    *q1 = f1 ? s1 : *q1 + s1;
  }
}

// HW % 4 == 0: four pixels per lane
__global__ __launch_bounds__(256) void nin_dgrad4_kernel(const float* __restrict__ go, const float* __restrict__ w, float* __restrict__ g0,
                                                         float* __restrict__ g1, int HW, int f0, int f1) {
  const int n = blockIdx.y, p = 4 * (blockIdx.x * 256 + threadIdx.x);
  if (p >= HW) return;
  float o[8][4];
#pragma unroll
  for (int oc = 0; oc < 8; ++oc) {
    const float4 v = *reinterpret_cast<const float4*>(go + ((size_t)n * 8 + oc) * HW + p);
    o[oc][0] = v.x; o[oc][1] = v.y; o[oc][2] = v.z; o[oc][3] = v.w;
  }
#pragma unroll
  for (int ic = 0; ic < 8; ++ic) {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a0 = f0 ? z4 : *reinterpret_cast<const float4*>(g0 + ((size_t)n * 8 + ic) * HW + p);
    float4 a1 = f1 ? z4 : *reinterpret_cast<const float4*>(g1 + ((size_t)n * 8 + ic) * HW + p);
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int oc = 0; oc < 8; ++oc) {   // the same order of the eight products per output as nin_dgrad_kernel
      const float w0 = as_const(w)[oc * 16 + ic], w1 = as_const(w)[oc * 16 + 8 + ic];
#pragma unroll
      for (int j = 0; j < 4; ++j) { s0[j] = fmaf(w0, o[oc][j], s0[j]); s1[j] = fmaf(w1, o[oc][j], s1[j]); }
    }
    a0.x += s0[0]; a0.y += s0[1]; a0.z += s0[2]; a0.w += s0[3];
    a1.x += s1[0]; a1.y += s1[1]; a1.z += s1[2]; a1.w += s1[3];
    *reinterpret_cast<float4*>(g0 + ((size_t)n * 8 + ic) * HW + p) = a0;
    *reinterpret_cast<float4*>(g1 + ((size_t)n * 8 + ic) * HW + p) = a1;
  }
}

// schedule-row form (., ., coef1, coef2, sigma) = (0, 0, alpha, beta, 0) for conv_out_kernel's fused update used as a chain adjoint
__global__ void set_chain_row_kernel(float* __restrict__ row, float alpha, float beta) {
  if (threadIdx.x < 5) row[threadIdx.x] = threadIdx.x == 2 ? alpha : threadIdx.x == 3 ? beta : 0.f;
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float alpha, long long count, int first) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < count) y[i] = first ? alpha * x[i] : fmaf(alpha, x[i], y[i]);
}
__global__ void add_vec_kernel(float* __restrict__ y, const float* __restrict__ x, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) y[i] += x[i];
}
__global__ void fill_kernel(float* __restrict__ y, float v, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) y[i] = v;
}

}  // namespace gc
