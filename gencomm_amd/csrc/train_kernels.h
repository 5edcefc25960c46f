// Elementary backward (and the matching exact-fp32 forward-recompute) kernels of the Enhancer's training path, NCHW fp32:
// LayerNorm over channels, depthwise 3x3 convolution, erf-GELU backward.  Together with the general convolution
// (conv_kernels.h: 1x1 = Linear, 3x3 partial conv; transposed weights give the input gradients) and conv_wgrad_kernel
// (unet_bwd_kernels.h: weight gradients) they let gencomm_amd/autograd.py run the backward of
// Enhancer_block / FRFN (opencood/models/gencomm_modules/enhancer.py:346-357, :222-250) without a torch conv / norm call.
// These are training-only, correctness-first kernels (one thread per pixel or element); the inference path never uses them.
#pragma once
#include "common.h"

namespace gc {

struct LnArgs {
  const float* x;      // [n][C][HW]
  const float* gamma;  // [C]
  const float* beta;   // [C]
  const float* dy;     // [n][C][HW] (backward)
  float* out;          // fwd: y (= LN(x), or x + LN(x) when residual);  bwd: dx (+= when accumulate)
  float* mean_rstd;    // [n][HW][2] written by both passes
  float eps;
  int C, HW, residual, accumulate;
};

// one thread per pixel; channel-strided accesses are coalesced across the threads of a wave
__global__ __launch_bounds__(256) void ln_nchw_fwd_kernel(const LnArgs a) {
  const int n = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.HW) return;
  const float* __restrict__ xp = a.x + (size_t)n * a.C * a.HW + p;
  float s = 0.f;
  for (int c = 0; c < a.C; ++c) s += xp[(size_t)c * a.HW];
  const float mean = s / (float)a.C;
  float q = 0.f;
  for (int c = 0; c < a.C; ++c) { const float d = xp[(size_t)c * a.HW] - mean; q = fmaf(d, d, q); }
  const float rstd = 1.0f / sqrtf(q / (float)a.C + a.eps);
  if (a.mean_rstd) { a.mean_rstd[((size_t)n * a.HW + p) * 2] = mean; a.mean_rstd[((size_t)n * a.HW + p) * 2 + 1] = rstd; }
  float* __restrict__ op = a.out + (size_t)n * a.C * a.HW + p;
  for (int c = 0; c < a.C; ++c) {
    const float xv = xp[(size_t)c * a.HW];
    const float y = fmaf((xv - mean) * rstd, a.gamma[c], a.beta[c]);
    op[(size_t)c * a.HW] = a.residual ? xv + y : y;
  }
}

// The same LayerNorm with the channels of a pixel split over 4 waves (workgroup = 64 pixels x 4 channel quarters, channel c of
// quarter c % 4) and the thread's C / 4 values held in registers: 4x the workgroups and loads in flight of the one-thread-per-pixel
// form, one read of x instead of three (V2X-ViT / Where2comm at 2 agents x 64x128: 61.6 -> see DESIGN section 7).  C <= 4 * CPT.
template <int CPT>
__global__ __launch_bounds__(256) void ln_nchw_fwd4_kernel(const LnArgs a) {
  __shared__ float s_part[2][4][64];
  const int n = blockIdx.y, pl = threadIdx.x & 63, cq = threadIdx.x >> 6, p = blockIdx.x * 64 + pl;
  const bool ok = p < a.HW;
  const float* __restrict__ xp = a.x + (size_t)n * a.C * a.HW + (ok ? p : 0);
  float v[CPT];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int c = 4 * k + cq;
    v[k] = (c < a.C) ? xp[(size_t)c * a.HW] : 0.f;
    s += v[k];
  }
  s_part[0][cq][pl] = s;
  __syncthreads();
  const float mean = (s_part[0][0][pl] + s_part[0][1][pl] + s_part[0][2][pl] + s_part[0][3][pl]) / (float)a.C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const float d = (4 * k + cq < a.C) ? v[k] - mean : 0.f;
    q = fmaf(d, d, q);
  }
  s_part[1][cq][pl] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((s_part[1][0][pl] + s_part[1][1][pl] + s_part[1][2][pl] + s_part[1][3][pl]) / (float)a.C + a.eps);
  if (!ok) return;
  if (a.mean_rstd && cq == 0) { a.mean_rstd[((size_t)n * a.HW + p) * 2] = mean; a.mean_rstd[((size_t)n * a.HW + p) * 2 + 1] = rstd; }
  float* __restrict__ op = a.out + (size_t)n * a.C * a.HW + p;
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int c = 4 * k + cq;
    if (c < a.C) {
      const float y = fmaf((v[k] - mean) * rstd, a.gamma[c], a.beta[c]);
      op[(size_t)c * a.HW] = a.residual ? v[k] + y : y;
    }
  }
}

// dx = rstd (gamma dy - mean_c(gamma dy) - xhat mean_c(gamma dy xhat))
__global__ __launch_bounds__(256) void ln_nchw_bwd_kernel(const LnArgs a) {
  const int n = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= a.HW) return;
  const float* __restrict__ xp = a.x + (size_t)n * a.C * a.HW + p;
  const float* __restrict__ gp = a.dy + (size_t)n * a.C * a.HW + p;
  float s = 0.f;
  for (int c = 0; c < a.C; ++c) s += xp[(size_t)c * a.HW];
  const float mean = s / (float)a.C;
  float q = 0.f;
  for (int c = 0; c < a.C; ++c) { const float d = xp[(size_t)c * a.HW] - mean; q = fmaf(d, d, q); }
  const float rstd = 1.0f / sqrtf(q / (float)a.C + a.eps);
  a.mean_rstd[((size_t)n * a.HW + p) * 2] = mean;
  a.mean_rstd[((size_t)n * a.HW + p) * 2 + 1] = rstd;
  float m1 = 0.f, m2 = 0.f;
  for (int c = 0; c < a.C; ++c) {
    const float gd = a.gamma[c] * gp[(size_t)c * a.HW];
    m1 += gd;
    m2 = fmaf(gd, (xp[(size_t)c * a.HW] - mean) * rstd, m2);
  }
  m1 /= (float)a.C; m2 /= (float)a.C;
  float* __restrict__ op = a.out + (size_t)n * a.C * a.HW + p;
  for (int c = 0; c < a.C; ++c) {
    const float xh = (xp[(size_t)c * a.HW] - mean) * rstd;
    const float v = rstd * (a.gamma[c] * gp[(size_t)c * a.HW] - m1 - xh * m2);
    op[(size_t)c * a.HW] = a.accumulate ? op[(size_t)c * a.HW] + v : v;
  }
}

// Round 4: the backward in ln_nchw_fwd4_kernel's structure (workgroup = 64 pixels x 4 channel quarters, a thread's C / 4 values of x and
// dy in registers): ONE read of x and dy instead of four / two channel-strided passes through the cache (1.9 TB/s at 4 x 64 x 200 x 704).
template <int CPT>
__global__ __launch_bounds__(256) void ln_nchw_bwd4_kernel(const LnArgs a) {
  __shared__ float s_part[2][4][64];
  const int n = blockIdx.y, pl = threadIdx.x & 63, cq = threadIdx.x >> 6, p = blockIdx.x * 64 + pl;
  const bool ok = p < a.HW;
  const size_t base = (size_t)n * a.C * a.HW + (ok ? p : 0);
  const float* __restrict__ xp = a.x + base;
  const float* __restrict__ gp = a.dy + base;
  float v[CPT], g[CPT];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int c = 4 * k + cq;
    v[k] = (c < a.C) ? xp[(size_t)c * a.HW] : 0.f;
    g[k] = (c < a.C) ? a.gamma[c] * gp[(size_t)c * a.HW] : 0.f;
    s += v[k];
  }
  s_part[0][cq][pl] = s;
  __syncthreads();
  const float mean = (s_part[0][0][pl] + s_part[0][1][pl] + s_part[0][2][pl] + s_part[0][3][pl]) / (float)a.C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const float d = (4 * k + cq < a.C) ? v[k] - mean : 0.f;
    q = fmaf(d, d, q);
  }
  s_part[1][cq][pl] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((s_part[1][0][pl] + s_part[1][1][pl] + s_part[1][2][pl] + s_part[1][3][pl]) / (float)a.C + a.eps);
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    if (4 * k + cq < a.C) { m1 += g[k]; m2 = fmaf(g[k], (v[k] - mean) * rstd, m2); }
  }
  __syncthreads();   // every wave has read the variance partials
  s_part[0][cq][pl] = m1;
  s_part[1][cq][pl] = m2;
  __syncthreads();
  m1 = (s_part[0][0][pl] + s_part[0][1][pl] + s_part[0][2][pl] + s_part[0][3][pl]) / (float)a.C;
  m2 = (s_part[1][0][pl] + s_part[1][1][pl] + s_part[1][2][pl] + s_part[1][3][pl]) / (float)a.C;
  if (!ok) return;
  if (cq == 0) { a.mean_rstd[((size_t)n * a.HW + p) * 2] = mean; a.mean_rstd[((size_t)n * a.HW + p) * 2 + 1] = rstd; }
  float* __restrict__ op = a.out + (size_t)n * a.C * a.HW + p;
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int c = 4 * k + cq;
    if (c < a.C) {
      const float xh = (v[k] - mean) * rstd;
      const float r = rstd * (g[k] - m1 - xh * m2);
      op[(size_t)c * a.HW] = a.accumulate ? op[(size_t)c * a.HW] + r : r;
    }
  }
}

// d gamma[c] += sum dy xhat, d beta[c] += sum dy : grid (channel, pixel chunk); a chunk's partial sums (f64) are committed
// with one f32 atomic per output
__global__ __launch_bounds__(256) void ln_nchw_param_grad_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean_rstd,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta, int n, int C, int HW) {
  __shared__ double s_red[4][2];
  const int c = blockIdx.x, tid = threadIdx.x;
  double sg = 0.0, sb = 0.0;
  for (long long i = (long long)blockIdx.y * 256 + tid; i < (long long)n * HW; i += (long long)gridDim.y * 256) {
    const int s = (int)(i / HW), p = (int)(i - (long long)s * HW);
    const size_t e = ((size_t)s * C + c) * HW + p;
    const float d = dy[e];
    sg += (double)d * ((x[e] - mean_rstd[i * 2]) * mean_rstd[i * 2 + 1]);
    sb += d;
  }
  for (int o = 32; o > 0; o >>= 1) { sg += __shfl_xor(sg, o, 64); sb += __shfl_xor(sb, o, 64); }
  if ((tid & 63) == 0) { s_red[tid >> 6][0] = sg; s_red[tid >> 6][1] = sb; }
  __syncthreads();
  if (tid == 0) {
    atomicAdd(&dgamma[c], (float)(s_red[0][0] + s_red[1][0] + s_red[2][0] + s_red[3][0]));
    atomicAdd(&dbeta[c], (float)(s_red[0][1] + s_red[1][1] + s_red[2][1] + s_red[3][1]));
  }
}

// ---- GroupNorm (+ SiLU) over NCHW for ANY channel count / group size: the general-width DiffusionUNet (gencomm_amd/unet_generic.py;
// reference unet.py:36-37 Normalize = GroupNorm(4 groups, eps 1e-6), :31-33 swish).  A group's channels are contiguous in NCHW: one
// workgroup reduces the group's cg * HW floats of one sample in f64; the apply pass is elementwise.  Correct-first: the accelerated
// UNet (ch 8, ch_mult all ones) never calls these -- its statistics come out of the producing convolution's epilogue.
__global__ __launch_bounds__(256) void gn_nchw_stats_kernel(const float* __restrict__ x, float* __restrict__ stat /*[n][G][2] mean, rstd*/,
                                                            int C, int G, int HW, float eps) {
  __shared__ double s_red[4][2];
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, cg = C / G;
  const long long cnt = (long long)cg * HW;
  const float* __restrict__ p = x + ((size_t)n * C + (size_t)g * cg) * HW;
  double s = 0.0, q = 0.0;
  for (long long i = tid; i < cnt; i += 256) { const double v = p[i]; s += v; q += v * v; }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  if ((tid & 63) == 0) { s_red[tid >> 6][0] = s; s_red[tid >> 6][1] = q; }
  __syncthreads();
  if (tid == 0) {
    const double S = s_red[0][0] + s_red[1][0] + s_red[2][0] + s_red[3][0], Q = s_red[0][1] + s_red[1][1] + s_red[2][1] + s_red[3][1];
    const double mean = S / (double)cnt, var = fmax(Q / (double)cnt - mean * mean, 0.0);
    stat[((size_t)n * G + g) * 2] = (float)mean;
    stat[((size_t)n * G + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}
__global__ __launch_bounds__(256) void gn_nchw_apply_kernel(const float* __restrict__ x, const float* __restrict__ stat, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y, int C, int G, int HW, int silu) {
  const int c = blockIdx.y, n = blockIdx.z, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const int g = c / (C / G);
  const float mean = stat[((size_t)n * G + g) * 2], rstd = stat[((size_t)n * G + g) * 2 + 1];
  const size_t e = ((size_t)n * C + c) * HW + p;
  const float v = fmaf((x[e] - mean) * rstd, gamma[c], beta[c]);
  y[e] = silu ? silu_f(v) : v;
}

// depthwise 3x3, padding 1: y[n][c] = conv(x[n][c], w[c][3][3]) + b[c].  flip = 1 computes the input gradient
// (correlation with the flipped taps, no bias).
// Round 3: four consecutive pixels of a row per thread, every tap LOADED (rows / columns clamped into the map, the out-of-range taps
// replaced by exact zeros afterwards) -- with one branch per tap the nine loads of a pixel were issued one memory latency after the
// other (1 TB/s at 4 x 128 x 200 x 704).
template <bool ACT = false>   // ACT: the map is GELU(x), evaluated on the values read (GELU(0) = 0 keeps the zero padding)
__device__ __forceinline__ void dw_load_row6(const float* __restrict__ row, int x0, int W, bool row_ok, float v[6]) {
  // v[0..5] = row[x0 - 1 .. x0 + 4], zero outside [0, W) or when the row itself is outside the map
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int xx = x0 - 1 + j;
    const float t = row[min(max(xx, 0), W - 1)];
    v[j] = (row_ok && xx >= 0 && xx < W) ? (ACT ? gelu_erf_f(t) : t) : 0.f;
  }
}
// x_ct: channels of the tensor x points into (the layer reads its first C channels: x_ct = C for a dense input)
template <bool ACT>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                        float* __restrict__ y, int C, int H, int W, int flip, int x_ct) {
  const int nc = blockIdx.y, W4 = (W + 3) >> 2, q = blockIdx.x * 256 + threadIdx.x;
  if (q >= H * W4) return;
  const int c = nc % C, h = q / W4, x0 = 4 * (q - h * W4);
  const float* __restrict__ xp = x + ((size_t)(nc / C) * x_ct + c) * H * W;
  float wk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wk[t] = w[c * 9 + (flip ? 8 - t : t)];
  const float b0 = (b != nullptr && !flip) ? b[c] : 0.f;
  float acc[4] = {b0, b0, b0, b0};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yy = h + ky - 1;
    float v[6];
    dw_load_row6<ACT>(xp + (size_t)min(max(yy, 0), H - 1) * W, x0, W, yy >= 0 && yy < H, v);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc[j] = fmaf(wk[ky * 3 + kx], v[j + kx], acc[j]);
  }
  float* __restrict__ yp = y + (size_t)nc * H * W + (size_t)h * W + x0;
  if ((W & 3) == 0) *reinterpret_cast<float4*>(yp) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  else
#pragma unroll
    for (int j = 0; j < 4; ++j) if (x0 + j < W) yp[j] = acc[j];
}
// Round 4, W % 4 == 0: a lane owns one aligned pixel quad of a column strip and walks DW_RB output rows with a three-row window in
// registers: per input row ONE 128-bit load, the left / right neighbour pixels come from the adjacent lanes (the wave's 64 lanes = 256
// consecutive pixels; only lanes 0 / 63 load their outer neighbour), (DW_RB + 2) / DW_RB input rows per output row instead of three.
// The round-3 form above issued 18 stride-4 dword loads per quad (1.67 TB/s at 4 x 128 x 200 x 704); it stays for ragged widths.
constexpr int DW_RB = 8;
template <bool ACT = false>
__device__ __forceinline__ void dw_row6_shfl(const float* __restrict__ row /* image row or null */, int x0, int W, bool lane_ok, int lane, float v[6]) {
  float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row != nullptr && lane_ok) {
    c = *reinterpret_cast<const float4*>(row + x0);
    if (ACT) c = make_float4(gelu_erf_f(c.x), gelu_erf_f(c.y), gelu_erf_f(c.z), gelu_erf_f(c.w));
  }
  float left = __shfl_up(c.w, 1, 64), right = __shfl_down(c.x, 1, 64);
  if (lane == 0) left = (row != nullptr && lane_ok && x0 > 0) ? (ACT ? gelu_erf_f(row[x0 - 1]) : row[x0 - 1]) : 0.f;
  if (lane == 63 || !lane_ok) right = 0.f;
  if (lane == 63 && row != nullptr && lane_ok && x0 + 4 < W) right = ACT ? gelu_erf_f(row[x0 + 4]) : row[x0 + 4];
  // a lane whose right neighbour lane is beyond the row end got that lane's zero quad: correct (zero padding)
  v[0] = left; v[1] = c.x; v[2] = c.y; v[3] = c.z; v[4] = c.w; v[5] = right;
}
template <bool ACT>
__global__ __launch_bounds__(256) void dwconv3x3_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                             float* __restrict__ y, int C, int H, int W, int flip, int x_ct) {
  const int nc = blockIdx.z, lane = threadIdx.x & 63, rb = threadIdx.x >> 6;
  const int x0 = 4 * (blockIdx.x * 64 + lane), r0 = (blockIdx.y * 4 + rb) * DW_RB;
  if (r0 >= H) return;                       // whole wave
  const bool lane_ok = x0 < W;
  const int c = nc % C;
  const float* __restrict__ xp = x + ((size_t)(nc / C) * x_ct + c) * H * W;
  float* __restrict__ yp = y + (size_t)nc * H * W;
  float wk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wk[t] = w[c * 9 + (flip ? 8 - t : t)];
  const float b0 = (b != nullptr && !flip) ? b[c] : 0.f;
  float win[3][6];
  dw_row6_shfl<ACT>(r0 - 1 >= 0 ? xp + (size_t)(r0 - 1) * W : nullptr, x0, W, lane_ok, lane, win[0]);
  dw_row6_shfl<ACT>(xp + (size_t)r0 * W, x0, W, lane_ok, lane, win[1]);
#pragma unroll
  for (int r = 0; r < DW_RB; ++r) {
    const int h = r0 + r;
    if (h >= H) break;                       // wave-uniform
    dw_row6_shfl<ACT>(h + 1 < H ? xp + (size_t)(h + 1) * W : nullptr, x0, W, lane_ok, lane, win[(r + 2) % 3]);
    float acc[4] = {b0, b0, b0, b0};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[j] = fmaf(wk[ky * 3 + kx], win[(r + ky) % 3][j + kx], acc[j]);
    if (lane_ok) *reinterpret_cast<float4*>(yp + (size_t)h * W + x0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}
// the weight gradient with the same walk: a wave owns (sample, 256-pixel column strip, DW_RB rows); 10 partial sums per lane, wave
// reduction, one atomic per (workgroup, tap)
template <bool ACT>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_rows_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                                   float* __restrict__ db, int n, int C, int H, int W, int x_ct) {
  __shared__ float s_red[4][10];
  const int c = blockIdx.z, lane = threadIdx.x & 63, rb = threadIdx.x >> 6;
  const int x0 = 4 * (blockIdx.x * 64 + lane);
  const bool lane_ok = x0 < W;
  const int rblocks = (H + DW_RB - 1) / DW_RB;
  float acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = 0.f;
  // blockIdx.y walks (sample, row block) items in chunks of 4 per workgroup (one per wave)
  for (int item = blockIdx.y * 4 + rb; item < n * rblocks; item += gridDim.y * 4) {
    const int s = item / rblocks, r0 = (item - s * rblocks) * DW_RB;
    const float* __restrict__ xp = x + ((size_t)s * x_ct + c) * H * W;
    const float* __restrict__ dp = dy + ((size_t)s * C + c) * H * W;
    float win[3][6];
    dw_row6_shfl<ACT>(r0 - 1 >= 0 ? xp + (size_t)(r0 - 1) * W : nullptr, x0, W, lane_ok, lane, win[0]);
    dw_row6_shfl<ACT>(xp + (size_t)r0 * W, x0, W, lane_ok, lane, win[1]);
#pragma unroll
    for (int r = 0; r < DW_RB; ++r) {
      const int h = r0 + r;
      if (h >= H) break;
      dw_row6_shfl<ACT>(h + 1 < H ? xp + (size_t)(h + 1) * W : nullptr, x0, W, lane_ok, lane, win[(r + 2) % 3]);
      float4 d4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane_ok) d4 = *reinterpret_cast<const float4*>(dp + (size_t)h * W + x0);
      const float d[4] = {d4.x, d4.y, d4.z, d4.w};
      acc[9] += (d[0] + d[1]) + (d[2] + d[3]);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[ky * 3 + kx] = fmaf(d[j], win[(r + ky) % 3][j + kx], acc[ky * 3 + kx]);
    }
  }
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    float v = acc[t];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) s_red[rb][t] = v;
  }
  __syncthreads();
  if (threadIdx.x < 10) {
    const float v = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
    if (threadIdx.x < 9) atomicAdd(&dw[c * 9 + threadIdx.x], v);
    else if (db != nullptr) atomicAdd(&db[c], v);
  }
}
// dw[c][tap] += sum_{n,p} dy[n][c](p) x[n][c](p + tap), db[c] += sum dy: grid (channel, pixel chunk), one f32 atomic per
// output and workgroup; the same four-pixel, branch-free loads
template <bool ACT>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                              float* __restrict__ db, int n, int C, int H, int W, int x_ct) {
  __shared__ float s_red[4][10];
  const int c = blockIdx.x, tid = threadIdx.x, W4 = (W + 3) >> 2;
  float acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = 0.f;
  for (long long i = (long long)blockIdx.y * 256 + tid; i < (long long)n * H * W4; i += (long long)gridDim.y * 256) {
    const int s = (int)(i / (H * W4)), q = (int)(i - (long long)s * H * W4), h = q / W4, x0 = 4 * (q - h * W4);
    const size_t base = ((size_t)s * C + c) * H * W, xbase = ((size_t)s * x_ct + c) * H * W;
    float d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float t = dy[base + (size_t)h * W + min(x0 + j, W - 1)]; d[j] = x0 + j < W ? t : 0.f; }
    acc[9] += (d[0] + d[1]) + (d[2] + d[3]);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = h + ky - 1;
      float v[6];
      dw_load_row6<ACT>(x + xbase + (size_t)min(max(yy, 0), H - 1) * W, x0, W, yy >= 0 && yy < H, v);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[ky * 3 + kx] = fmaf(d[j], v[j + kx], acc[ky * 3 + kx]);
    }
  }
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    float v = acc[t];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((tid & 63) == 0) s_red[tid >> 6][t] = v;
  }
  __syncthreads();
  if (tid < 10) {
    const float v = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
    if (tid < 9) atomicAdd(&dw[c * 9 + tid], v);
    else if (db != nullptr) atomicAdd(&db[c], v);
  }
}

// Weight gradient of a 1x1 convolution (= Linear over NCHW pixels) as a split-K GEMM on the matrix cores (exact fp32):
//   dW[co][ci] += sum_{n,p} dY[n][co][p] X[n][ci][p],  dB[co] += sum dY.
// M = 64 output channels x N = 64 input channels per workgroup (wave = one 32 x 32 block of v_mfma_f32_32x32x2_f32), K = a
// chunk of pixels of one sample, walked in 64-pixel tiles staged through LDS (rows padded to 66 words: the MFMA operand
// reads of 32 rows land in distinct banks); the chunk's 64 x 64 partial product is committed with f32 atomics.
// grid = (pixel chunks x samples, Cout / 64, Cin / 64).
struct Wgrad1x1Args {
  const float* dy;   // [n][Cout][HW]
  const float* x;    // [n][Cin][HW]
  float* dw;         // [Cout][Cin] (+=)
  float* db;         // [Cout] (+=) or null
  int Cout, Cin, HW, chunk, chunks;
};
__global__ __launch_bounds__(256) void wgrad1x1_mfma_kernel(const Wgrad1x1Args a) {
  constexpr int LD = 66;
  __shared__ float sA[64 * LD];   // dY tile [co][px]
  __shared__ float sB[64 * LD];   // X tile  [ci][px]
  using f32x16t = __attribute__((ext_vector_type(16))) float;
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
  const int n = blockIdx.x / a.chunks, ck = blockIdx.x - n * a.chunks;
  const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
  const int p_begin = ck * a.chunk, p_end = min(p_begin + a.chunk, a.HW);
  const float* __restrict__ dyn = a.dy + (size_t)n * a.Cout * a.HW;
  const float* __restrict__ xn = a.x + (size_t)n * a.Cin * a.HW;
  const int mh = wv & 1, nh = wv >> 1;
  f32x16t acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float bsum = 0.f;   // threads 0..63: bias gradient of row tid
  // staging: thread -> (row = tid / 4 + 64 j? no: 16 floats per thread) rows of 64 pixels: 4 threads per row, 16 px each
  const int srow = tid >> 2, sq = (tid & 3) * 16;
  for (int p0 = p_begin; p0 < p_end; p0 += 64) {
    float va[16], vb[16];
    const bool full = p0 + 64 <= p_end && (a.HW & 3) == 0;
    {
      const int co = co0 + srow, ci = ci0 + srow;
      const float* __restrict__ pa = dyn + (size_t)co * a.HW + p0 + sq;
      const float* __restrict__ pb = xn + (size_t)ci * a.HW + p0 + sq;
      if (full) {
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
          const float4 u = co < a.Cout ? *reinterpret_cast<const float4*>(pa + e) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 w = ci < a.Cin ? *reinterpret_cast<const float4*>(pb + e) : make_float4(0.f, 0.f, 0.f, 0.f);
          va[e] = u.x; va[e + 1] = u.y; va[e + 2] = u.z; va[e + 3] = u.w;
          vb[e] = w.x; vb[e + 1] = w.y; vb[e + 2] = w.z; vb[e + 3] = w.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const bool ok = p0 + sq + e < p_end;
          va[e] = (ok && co < a.Cout) ? pa[e] : 0.f;
          vb[e] = (ok && ci < a.Cin) ? pb[e] : 0.f;
        }
      }
    }
    __syncthreads();   // previous tile's MFMAs are done
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      *reinterpret_cast<float2*>(&sA[srow * LD + sq + e]) = make_float2(va[e], va[e + 1]);
      *reinterpret_cast<float2*>(&sB[srow * LD + sq + e]) = make_float2(vb[e], vb[e + 1]);
    }
    __syncthreads();
    if (a.db != nullptr && blockIdx.z == 0 && tid < 64) {
      float s = 0.f;
#pragma unroll 8
      for (int k = 0; k < 64; ++k) s += sA[tid * LD + k];
      bsum += s;
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const float av = sA[(mh * 32 + r) * LD + 2 * k + h];
      const float bv = sB[(nh * 32 + r) * LD + 2 * k + h];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
  }
  // acc[reg]: row (co) = (reg & 3) + 8 (reg >> 2) + 4 h, column (ci) = r
  const int ci = ci0 + nh * 32 + r;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int co = co0 + mh * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (co < a.Cout && ci < a.Cin) atomicAdd(&a.dw[(size_t)co * a.Cin + ci], acc[reg]);
  }
  if (a.db != nullptr && blockIdx.z == 0 && tid < 64 && co0 + tid < a.Cout) atomicAdd(&a.db[co0 + tid], bsum);
}
inline int wgrad1x1_enqueue(const float* dy, const float* x, float* dw, float* db, int N, int Cin, int Cout, int HW, hipStream_t st) {
  Wgrad1x1Args a{dy, x, dw, db, Cout, Cin, HW, 0, 0};
  a.chunk = 2048;
  while (a.chunk > 256 && (long long)N * ((HW + a.chunk - 1) / a.chunk) < 256) a.chunk >>= 1;   // enough workgroups on small maps
  a.chunks = (HW + a.chunk - 1) / a.chunk;
  const dim3 grid((unsigned)(N * a.chunks), (Cout + 63) / 64, (Cin + 63) / 64);
  if (grid.y > 65535 || grid.z > 65535) return fail(GC_ERR_ARG, "wgrad 1x1: too many channel blocks");
  wgrad1x1_mfma_kernel<<<grid, 256, 0, st>>>(a);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

// Weight gradient of a WIDE 3x3 convolution (backbone layers: 64..256 channels, stride 1 or 2) as a split-K implicit GEMM on the matrix
// cores (exact fp32, v_mfma_f32_32x32x2_f32):
//   dW[co][ci][ky][kx] += sum_{n,y,x} dY[n][co][y][x] X[n][ci][S y + ky - pad][S x + kx - pad],   dB[co] += sum dY.
// A workgroup owns 64 output x 64 input channels (wave = one 32 x 32 block) and NINE accumulators per wave, one per tap (144 registers);
// it walks `chunk` pixel tiles of one sample (TR rows x 32 columns of dY; the haloed X tile beside it in LDS, rows padded to an odd
// word count so that the 32 channel rows of an operand read land in distinct banks), reads the dY operand once per pixel pair and the
// X operand once per tap at the tap's shift, and commits its partial sums with f32 atomics.  grid = (tile chunks x samples, Cout / 64, Cin / 64).
// The 8 x 8-channel-chunk kernels of unet_bwd_kernels.h (made for the UNet's 8-channel layers) re-staged every tile (Cout / 8)(Cin / 8)
// times for these layers: 10.5 ms per stage-1 training step (profiles/r4_train_leg.txt) against 176 GFLOP of useful work.
struct Wgrad3x3Args {
  const float* dy;   // [n][Cout][Ho][Wo]
  const float* x;    // [n][Cin][Hi][Wi]
  float* dw;         // [Cout][Cin][3][3] (+=)
  float* db;         // [Cout] (+=) or null
  int Cout, Cin, Ho, Wo, Hi, Wi, pad, tiles_x, tiles, chunk, chunks;
  // scratch of wgrad3x3_wide_scratch_floats(...) floats, or null.  With it every workgroup STORES its partial sums, coalesced, at
  // part[group = blockIdx.x][co][tap][ci] (channel counts padded to 64) and wgrad3x3_wide_reduce_kernel adds the groups up in a fixed order:
  // the f32 atomics of the null form (144 x 64 per wave onto Cout Cin 9 addresses shared by every group, resolved beyond the XCDs' L2s)
  // cost 13x the matrix work on the 256-channel layers (790 us per layer against 61 us of MFMA time)
  float* part;
};
inline size_t wgrad3x3_wide_groups(int N, int Cin, int Cout, int Ho, int Wo, int stride, int* chunk_out = nullptr) {
  const int TR = stride == 1 ? 2 : 1;
  const long long tiles = (long long)((Wo + 31) / 32) * ((Ho + TR - 1) / TR);
  const long long cblocks = (long long)((Cout + 63) / 64) * ((Cin + 63) / 64);
  // ~1 workgroup per CU over the whole launch (the kernel's residency), at least 4 tiles per workgroup
  long long chunk = std::max<long long>(4, (tiles * N * cblocks + 255) / 256);
  chunk = std::min(chunk, tiles);
  // the per-sample rounding can push the launch just past 256 workgroups (24 x 2 x 6 = 288 on the stage-1 leg's 384-channel stride-2 layer:
  // a second resident round of 32 workgroups doubled its 0.46 ms): between one and two rounds, lengthen the chunks until one round holds them
  auto wgs = [&](long long c) { return (long long)N * ((tiles + c - 1) / c) * cblocks; };
  while (wgs(chunk) > 256 && wgs(chunk) < 512 && chunk < tiles) ++chunk;
  if (chunk_out) *chunk_out = (int)chunk;
  return (size_t)N * (size_t)((tiles + chunk - 1) / chunk);
}
inline size_t wgrad3x3_wide_scratch_floats(int N, int Cin, int Cout, int Ho, int Wo, int stride) {
  const size_t cop = (size_t)((Cout + 63) / 64) * 64, cip = (size_t)((Cin + 63) / 64) * 64;
  return wgrad3x3_wide_groups(N, Cin, Cout, Ho, Wo, stride) * cop * (9 * cip + 1);   // + one bias partial per (group, co)
}
template <int S, int TR>
// One workgroup per CU by registers (144 accumulators + the prefetched tile + its addresses: ~250 VGPRs; capping them at two per CU spills
// 560 B per thread): the nine independent accumulator chains and the prefetch keep a single wave per SIMD on the matrix pipe.
__global__ __launch_bounds__(256) void wgrad3x3_wide_kernel(const Wgrad3x3Args a) {
  constexpr int TC = 32, PX = TR * TC;                       // output pixels per tile
  constexpr int XR = (TR - 1) * S + 3, XC = (TC - 1) * S + 3;  // haloed input tile
  constexpr int LDA = PX + 1, LDB = (XR * XC) | 1;           // odd row strides (words)
  __shared__ float sA[64 * LDA];   // dY tile [co][pixel]
  __shared__ float sB[64 * LDB];   // X tile  [ci][row][col]
  using f32x16t = __attribute__((ext_vector_type(16))) float;
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
  // staging registers: the NEXT tile's global loads are issued right after this tile's LDS image is complete, so they fly during its
  // matrix phase.  dY: thread -> (channel row tid / 4, PER consecutive pixels of one image row); X: element i = tid + 256 k of the
  // [64][XR][XC] tile, columns fastest (a wave covers ~2 tile rows: coalesced)
  constexpr int PER = PX / 4;
  constexpr int NXE = 64 * XR * XC, XPT = (NXE + 255) / 256;
  float va[PER], vx[XPT];
  const int srow = tid >> 2, sq = (tid & 3) * PER;
  auto load_tile = [&](int tile) {
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = ty * TR, ox0 = tx * TC;
    const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
    {
      const int tr = sq / TC, col = sq - tr * TC;
      const int co = co0 + srow, oy = oy0 + tr, ox = ox0 + col;
      const bool row_ok = co < a.Cout && oy < a.Ho;
      const float* __restrict__ pa = dyn + ((size_t)co * a.Ho + oy) * a.Wo + ox;
      if (row_ok && ox + PER <= a.Wo && (a.Wo & 3) == 0) {
#pragma unroll
        for (int e = 0; e < PER; e += 4) {
          const float4 u = *reinterpret_cast<const float4*>(pa + e);
          va[e] = u.x; va[e + 1] = u.y; va[e + 2] = u.z; va[e + 3] = u.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < PER; ++e) va[e] = (row_ok && ox + e < a.Wo) ? pa[e] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      const int i = tid + 256 * k;
      const int c = i / (XR * XC), rem = i - c * (XR * XC), ry = rem / XC, e = rem - ry * XC;
      const int ci = ci0 + c, iy = iy0 + ry, ix = ix0 + e;
      vx[k] = (i < NXE && ci < a.Cin && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) ? xn[((size_t)ci * a.Hi + iy) * a.Wi + ix] : 0.f;
    }
  };
  int tile = ck * a.chunk;
  if (tile < t_end) load_tile(tile);
  for (; tile < t_end; ++tile) {
    __syncthreads();   // the previous tile's matrix instructions are done
#pragma unroll
    for (int e = 0; e < PER; ++e) sA[srow * LDA + sq + e] = va[e];
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      const int i = tid + 256 * k;
      const int c = i / (XR * XC), rem = i - c * (XR * XC);
      if (i < NXE) sB[c * LDB + rem] = vx[k];
    }
    __syncthreads();
    if (tile + 1 < t_end) load_tile(tile + 1);
    if (a.db != nullptr && blockIdx.z == 0 && tid < 64) {
      float s = 0.f;
#pragma unroll 8
      for (int k = 0; k < PX; ++k) s += sA[tid * LDA + k];
      bsum += s;
    }
    const float* __restrict__ pA = sA + (mh * 32 + r) * LDA + h;
    const float* __restrict__ pB = sB + (nh * 32 + r) * LDB + h * S;
#pragma unroll
    for (int tr = 0; tr < TR; ++tr) {
#pragma unroll 4
      for (int k2 = 0; k2 < TC / 2; ++k2) {
        const float av = pA[tr * TC + 2 * k2];
        const float* __restrict__ q = pB + (tr * S) * XC + 2 * k2 * S;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, q[(t / 3) * XC + (t % 3)], acc[t], 0, 0, 0);
      }
    }
  }
  // acc[t][reg]: row (co) = (reg & 3) + 8 (reg >> 2) + 4 h, column (ci) = r
  const int ci = ci0 + nh * 32 + r;
  if (a.part != nullptr) {
    const size_t cop = (size_t)gridDim.y * 64, cip = (size_t)gridDim.z * 64;
    float* __restrict__ pg = a.part + (size_t)blockIdx.x * cop * (9 * cip + 1);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int co = co0 + mh * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
#pragma unroll
      for (int t = 0; t < 9; ++t) pg[((size_t)co * 9 + t) * cip + ci] = acc[t][reg];   // 32 lanes = 128 contiguous bytes
    }
    if (blockIdx.z == 0 && tid < 64) pg[cop * 9 * cip + co0 + tid] = bsum;
    return;
  }
  if (ci < a.Cin) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int co = co0 + mh * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
      if (co < a.Cout) {
        float* __restrict__ d = a.dw + ((size_t)co * a.Cin + ci) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) atomicAdd(d + t, acc[t][reg]);
      }
    }
  }
  if (a.db != nullptr && blockIdx.z == 0 && tid < 64 && co0 + tid < a.Cout) atomicAdd(&a.db[co0 + tid], bsum);
}
// dw[co][ci][tap] += sum_g part[g][co][tap][ci], db[co] += sum_g part_b[g][co]: thread = one (co, tap, ci) of the padded table
__global__ __launch_bounds__(256) void wgrad3x3_wide_reduce_kernel(const Wgrad3x3Args a, int groups, int cop, int cip) {
  const size_t per = (size_t)cop * (9 * (size_t)cip + 1);
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per) return;
  const float* __restrict__ p = a.part + i;
  // eight loads in flight per thread (two were: 256 groups x 2 dependent chains made the launch latency-bound at 2 TB/s); the order of the
  // additions is fixed by (group mod 8), so the result does not depend on the launch
  float sv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int g = 0;
  for (; g + 7 < groups; g += 8) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = p[(size_t)(g + k) * per];
#pragma unroll
    for (int k = 0; k < 8; ++k) sv[k] += t[k];
  }
  for (int k = 0; g < groups; ++g, ++k) sv[k] += p[(size_t)g * per];
  const float v = ((sv[0] + sv[1]) + (sv[2] + sv[3])) + ((sv[4] + sv[5]) + (sv[6] + sv[7]));
  if (i < (size_t)cop * 9 * cip) {
    const int ci = (int)(i % cip), t = (int)((i / cip) % 9), co = (int)(i / ((size_t)9 * cip));
    if (co < a.Cout && ci < a.Cin) a.dw[((size_t)co * a.Cin + ci) * 9 + t] += v;
  } else if (a.db != nullptr) {
    const int co = (int)(i - (size_t)cop * 9 * cip);
    if (co < a.Cout) a.db[co] += v;
  }
}
int wgrad3x3_h3_launch(const Wgrad3x3Args& a, dim3 grid, hipStream_t st);   // wgrad_h3_kernels.h
inline int wgrad3x3_wide_enqueue(const float* dy, const float* x, float* dw, float* db, int N, int Cin, int Hi, int Wi, int Cout, int Ho, int Wo,
                                 int stride, int pad, float* scratch, hipStream_t st) {
  Wgrad3x3Args a{dy, x, dw, db, Cout, Cin, Ho, Wo, Hi, Wi, pad, 0, 0, 0, 0, scratch};
  const int TR = stride == 1 ? 2 : 1;
  a.tiles_x = (Wo + 31) / 32;
  a.tiles = a.tiles_x * ((Ho + TR - 1) / TR);
  const size_t groups = wgrad3x3_wide_groups(N, Cin, Cout, Ho, Wo, stride, &a.chunk);
  a.chunks = (int)(groups / N);
  const dim3 grid((unsigned)groups, (Cout + 63) / 64, (Cin + 63) / 64);
  if (grid.y > 65535 || grid.z > 65535) return fail(GC_ERR_ARG, "wgrad 3x3: too many channel blocks");
  // stride 1 with scratch, default arithmetic: the three-term f16-pipe kernel (wgrad_h3_kernels.h: same tiles, same partial-sum layout)
  const Modes md = modes_snapshot();
  if (stride == 1 && scratch != nullptr && md.split() && !md.split2()) {
    if (int rc = wgrad3x3_h3_launch(a, grid, st)) return rc;
  } else if (stride == 1) wgrad3x3_wide_kernel<1, 2><<<grid, 256, 0, st>>>(a);
  else wgrad3x3_wide_kernel<2, 1><<<grid, 256, 0, st>>>(a);
  if (scratch != nullptr) {
    const int cop = (int)grid.y * 64, cip = (int)grid.z * 64;
    const size_t per = (size_t)cop * (9 * (size_t)cip + 1);
    wgrad3x3_wide_reduce_kernel<<<(unsigned)((per + 255) / 256), 256, 0, st>>>(a, (int)groups, cop, cip);
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}

// out = g * d/dv GELU(v)   (erf form, nn.GELU default):  0.5 (1 + erf(v / sqrt 2)) + v exp(-v^2 / 2) / sqrt(2 pi)
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ out, long long count) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  out[i] = g[i] * gelu_grad_fast_f(v[i]);
}

// ---- elementwise / per-(sample, channel) pieces of the Enhancer's backward (enhancer.py:222-250, :315-333, :346-357), so that the
// training step's own arithmetic runs on this library's kernels only (round 2 composed them from framework tensor operations).
// All tensors NCHW fp32; a "slice" is channels [c0, c0 + nch) of a tensor with ctotal channels.
struct SliceArgs {
  const float* a; const float* b; const float* c; const float* d;
  float* o0; float* o1;
  int n, nch, HW;
  int a_ct, a_c0, b_ct, b_c0, o0_ct, o0_c0, o1_ct, o1_c0;
};
enum EwOp : int {
  EW_COPY = 0,        // o0 = a                                   (slice -> slice: replaces cat / contiguous)
  EW_GELU_SPLIT = 1,  // o0 = GELU(a[:, :nch]), o1 = GELU(a[:, nch:2 nch])      (a has 2 nch channels)
  EW_GELU_GATE = 2,   // o0 = GELU(a) * b                         (g = GELU(u) x2)
  EW_GATE_BWD = 3,    // o0 = GELU'(a) * c * b  (du);  o1 slice = GELU'(d slice) * c * GELU(a)   (a = u, b = h2, c = dg, d = v)
  EW_GELU_BWD = 4,    // o0 slice = GELU'(a slice) * b            (dv[:, :hid] = GELU'(v1) dh1)
  // second half of round 4: the GELU outputs h1 / h2 of Linear1 are no longer materialised -- their consumers evaluate GELU on the
  // slice of v they read (the depthwise kernels take an activation flag), so the forward's split pass and two saved tensors disappear
  EW_GELU2_GATE = 5,  // o0 = GELU(a) * GELU(d slice)             (g = GELU(u) GELU(v2); d has o1_ct channels, slice at o1_c0)
  EW_GATE_BWD2 = 6,   // o0 = GELU'(a) * c * GELU(d slice) (du);  o1 slice = GELU'(d slice) * c * GELU(a)   (op 3 without its b operand)
};
// one element of an op: in[] = a, b, c, d at this element (unused ones 0), out[] = o0, o1
template <int OP>
__device__ __forceinline__ void ew_apply(float a, float b, float c, float d, float* o0, float* o1) {
  if (OP == EW_COPY) {
    *o0 = a;
  } else if (OP == EW_GELU_SPLIT) {
    *o0 = gelu_erf_f(a);
    *o1 = gelu_erf_f(b);
  } else if (OP == EW_GELU_GATE) {
    *o0 = gelu_erf_f(a) * b;
  } else if (OP == EW_GATE_BWD) {
    float gu, du, gv, dv;
    gelu_and_grad_f(a, &gu, &du);
    gelu_and_grad_f(d, &gv, &dv);
    *o0 = du * (c * b);
    *o1 = dv * (c * gu);
  } else if (OP == EW_GELU_BWD) {
    *o0 = gelu_grad_fast_f(a) * b;
  } else if (OP == EW_GELU2_GATE) {
    *o0 = gelu_erf_f(a) * gelu_erf_f(d);
  } else {
    float gu, du, gv, dv;
    gelu_and_grad_f(a, &gu, &du);
    gelu_and_grad_f(d, &gv, &dv);
    *o0 = du * (c * gv);
    *o1 = dv * (c * gu);
  }
}
// operand slices of an op at (sample n, channel c): offsets in floats.  EW_GELU_SPLIT reads its second input from a's upper half;
// ops 3 / 6 read d and write o1 at the same slice (o1_ct, o1_c0), op 5 reads d there; op 4 reads a and writes o0 at (o0_ct, o0_c0).
template <int OP>
struct EwOffsets {
  size_t a, b, c, d, o0, o1;
  __device__ __forceinline__ EwOffsets(const SliceArgs& s, int n, int c_) {
    const size_t HW = (size_t)s.HW;
    const size_t dense = ((size_t)n * s.nch + c_) * HW;
    a = b = c = d = o0 = o1 = dense;
    if (OP == EW_COPY) { a = ((size_t)n * s.a_ct + s.a_c0 + c_) * HW; o0 = ((size_t)n * s.o0_ct + s.o0_c0 + c_) * HW; }
    if (OP == EW_GELU_SPLIT) { a = ((size_t)n * 2 * s.nch + c_) * HW; b = a + (size_t)s.nch * HW; }
    if (OP == EW_GATE_BWD || OP == EW_GATE_BWD2 || OP == EW_GELU2_GATE) d = o1 = ((size_t)n * s.o1_ct + s.o1_c0 + c_) * HW;
    if (OP == EW_GELU_BWD) a = o0 = ((size_t)n * s.o0_ct + s.o0_c0 + c_) * HW;
  }
};
template <int OP>
__global__ __launch_bounds__(256) void ew_slice_kernel(const SliceArgs s) {
  const int p = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, n = blockIdx.z;
  if (p >= s.HW) return;
  const EwOffsets<OP> o(s, n, c);
  constexpr bool use_b = OP == EW_GELU_GATE || OP == EW_GATE_BWD || OP == EW_GELU_BWD, use_c = OP == EW_GATE_BWD || OP == EW_GATE_BWD2;
  constexpr bool use_d = OP == EW_GATE_BWD || OP == EW_GATE_BWD2 || OP == EW_GELU2_GATE, two = OP == EW_GELU_SPLIT || OP == EW_GATE_BWD || OP == EW_GATE_BWD2;
  const float a = s.a[o.a + p];
  const float b = OP == EW_GELU_SPLIT ? s.a[o.b + p] : use_b ? s.b[o.b + p] : 0.f;
  const float cc = use_c ? s.c[o.c + p] : 0.f, d = use_d ? s.d[o.d + p] : 0.f;
  float r0, r1 = 0.f;
  ew_apply<OP>(a, b, cc, d, &r0, &r1);
  s.o0[o.o0 + p] = r0;
  if (two) s.o1[o.o1 + p] = r1;
}
// HW % 4 == 0: four pixels per lane, 128-bit loads and stores (every slice base is then 16-byte aligned with the tensors)
template <int OP>
__global__ __launch_bounds__(256) void ew_slice4_kernel(const SliceArgs s) {
  const int p = 4 * (blockIdx.x * 256 + threadIdx.x), c = blockIdx.y, n = blockIdx.z;
  if (p >= s.HW) return;
  const EwOffsets<OP> o(s, n, c);
  constexpr bool use_b = OP == EW_GELU_GATE || OP == EW_GATE_BWD || OP == EW_GELU_BWD, use_c = OP == EW_GATE_BWD || OP == EW_GATE_BWD2;
  constexpr bool use_d = OP == EW_GATE_BWD || OP == EW_GATE_BWD2 || OP == EW_GELU2_GATE, two = OP == EW_GELU_SPLIT || OP == EW_GATE_BWD || OP == EW_GATE_BWD2;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 a4 = *reinterpret_cast<const float4*>(s.a + o.a + p);
  const float4 b4 = OP == EW_GELU_SPLIT ? *reinterpret_cast<const float4*>(s.a + o.b + p) : use_b ? *reinterpret_cast<const float4*>(s.b + o.b + p) : z4;
  const float4 c4 = use_c ? *reinterpret_cast<const float4*>(s.c + o.c + p) : z4;
  const float4 d4 = use_d ? *reinterpret_cast<const float4*>(s.d + o.d + p) : z4;
  const float a[4] = {a4.x, a4.y, a4.z, a4.w}, b[4] = {b4.x, b4.y, b4.z, b4.w}, cc[4] = {c4.x, c4.y, c4.z, c4.w}, d[4] = {d4.x, d4.y, d4.z, d4.w};
  float r0[4], r1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) ew_apply<OP>(a[j], b[j], cc[j], d[j], &r0[j], &r1[j]);
  *reinterpret_cast<float4*>(s.o0 + o.o0 + p) = make_float4(r0[0], r0[1], r0[2], r0[3]);
  if (two) *reinterpret_cast<float4*>(s.o1 + o.o1 + p) = make_float4(r1[0], r1[1], r1[2], r1[3]);
}
// out[n][c][p] = x[n][c][p] * a[n][c] + b[n][c]       (d y2 = grad_out * gate + d gap / HW, enhancer.py:325-333)
// V = 4: HW % 4 == 0 and 16-byte aligned maps, four pixels per lane
template <int V>
__global__ __launch_bounds__(256) void nc_scale_kernel(const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                                       float* __restrict__ out, int HW) {
  const int p = V * (blockIdx.x * 256 + threadIdx.x), nc = blockIdx.y;
  if (p >= HW) return;
  const float s = a[nc], t = b != nullptr ? b[nc] : 0.f;
  if (V == 4) {
    const float4 v = *reinterpret_cast<const float4*>(x + (size_t)nc * HW + p);
    *reinterpret_cast<float4*>(out + (size_t)nc * HW + p) = make_float4(fmaf(v.x, s, t), fmaf(v.y, s, t), fmaf(v.z, s, t), fmaf(v.w, s, t));
  } else {
    out[(size_t)nc * HW + p] = fmaf(x[(size_t)nc * HW + p], s, t);
  }
}
// out[n][c] = sum_p x[n][c][p] * (y ? y[n][c][p] : 1)   (global average pool and d gate), f64 accumulation; out zeroed by the caller
template <int V>
__global__ __launch_bounds__(256) void nc_dot_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int HW) {
  __shared__ double s_red[4];
  const int nc = blockIdx.y, tid = threadIdx.x;
  double acc = 0.0;
  if (V == 4) {
    for (int p = 4 * (blockIdx.x * 256 + tid); p < HW; p += gridDim.x * 1024) {
      const float4 v = *reinterpret_cast<const float4*>(x + (size_t)nc * HW + p);
      if (y != nullptr) {
        const float4 u = *reinterpret_cast<const float4*>(y + (size_t)nc * HW + p);
        acc += ((double)v.x * (double)u.x + (double)v.y * (double)u.y) + ((double)v.z * (double)u.z + (double)v.w * (double)u.w);
      } else {
        acc += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
      }
    }
  } else {
    for (int p = blockIdx.x * 256 + tid; p < HW; p += gridDim.x * 256) acc += (double)x[(size_t)nc * HW + p] * (y != nullptr ? (double)y[(size_t)nc * HW + p] : 1.0);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) atomicAdd(&out[nc], (float)(s_red[0] + s_red[1] + s_red[2] + s_red[3]));
}

// out = a x + b y + c z (y, z optional; out may alias any input): the elementwise glue of the training branch's sampler chain
// (x_{t-1} = c1 x0_hat + c2 x_t + sigma eps, cond_diff.py:272-315, and its adjoint) without framework kernels
__global__ __launch_bounds__(256) void lincomb_kernel(float* out, const float* x, const float* y, const float* z, float a, float b, float c,
                                                      long long count) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long i = q * 4;
  if (i + 3 < count) {
    float4 v = *reinterpret_cast<const float4*>(x + i);
    v.x *= a; v.y *= a; v.z *= a; v.w *= a;
    if (y != nullptr) { const float4 u = *reinterpret_cast<const float4*>(y + i); v.x = fmaf(b, u.x, v.x); v.y = fmaf(b, u.y, v.y); v.z = fmaf(b, u.z, v.z); v.w = fmaf(b, u.w, v.w); }
    if (z != nullptr) { const float4 u = *reinterpret_cast<const float4*>(z + i); v.x = fmaf(c, u.x, v.x); v.y = fmaf(c, u.y, v.y); v.z = fmaf(c, u.z, v.z); v.w = fmaf(c, u.w, v.w); }
    *reinterpret_cast<float4*>(out + i) = v;
  } else {
    for (long long k = i; k < count; ++k) {
      float v = a * x[k];
      if (y != nullptr) v = fmaf(b, y[k], v);
      if (z != nullptr) v = fmaf(c, z[k], v);
      out[k] = v;
    }
  }
}

// ---- deformable 3x3 convolution (DCNv1, one offset group, padding 1) for the training path of MessageExtractorv2 ------------
// (message_extractor_v2.py:78,:108; sampling arithmetic as in dcn_kernel, msgext_kernels.h).  The convolution is split
// into its sampling half and its GEMM half so that the backward is GEMMs on the general kernels plus one scatter kernel:
//   col[n][c * 9 + k][p] = bilinear sample of x[n][c] at (y - 1 + ky + off[2k], x - 1 + kx + off[2k + 1])
//   b1 = W[64][9 C] col + bias        (1x1 convolution over 9 C channels)
// dcn_scatter_bwd: given dcol, dx[n][c][corner] += w_corner dcol (atomics), d off[2k] = sum_c dcol d val / d py, d off[2k+1] likewise.
struct DcnTap {
  float hy, ly, hx, lx;          // bilinear fractions
  int j00, j01, j10, j11;        // corner indices (0 where the corner is outside the map)
  bool c00, c01, c10, c11;       // corner inside the map (and the sample position inside (-1, H) x (-1, W))
};
__device__ __forceinline__ DcnTap dcn_tap(const float* __restrict__ off, int n, int k, int pix, int y, int x, int H, int W) {
  const int HW = H * W;
  const float py = (float)(y - 1 + k / 3) + off[((size_t)n * 18 + 2 * k) * HW + pix];
  const float px = (float)(x - 1 + k % 3) + off[((size_t)n * 18 + 2 * k + 1) * HW + pix];
  DcnTap t;
  const bool inside = py > -1.f && py < (float)H && px > -1.f && px < (float)W;
  const float fy = floorf(py), fx = floorf(px);
  const int iy = (int)fy, ix = (int)fx;
  t.ly = py - fy; t.lx = px - fx; t.hy = 1.f - t.ly; t.hx = 1.f - t.lx;
  const bool y0 = inside && iy >= 0, y1 = inside && iy + 1 <= H - 1, x0 = ix >= 0, x1 = ix + 1 <= W - 1;
  const int i00 = iy * W + ix;
  t.c00 = y0 && x0; t.c01 = y0 && x1; t.c10 = y1 && x0; t.c11 = y1 && x1;
  t.j00 = t.c00 ? i00 : 0; t.j01 = t.c01 ? i00 + 1 : 0; t.j10 = t.c10 ? i00 + W : 0; t.j11 = t.c11 ? i00 + W + 1 : 0;
  return t;
}
__global__ __launch_bounds__(256) void dcn_sample_kernel(const float* __restrict__ x, const float* __restrict__ off, float* __restrict__ col,
                                                         int C, int H, int W) {
  const int n = blockIdx.z, k = blockIdx.y, pix = blockIdx.x * 256 + threadIdx.x, HW = H * W;
  if (pix >= HW) return;
  const int y = pix / W, xx = pix - y * W;
  const DcnTap t = dcn_tap(off, n, k, pix, y, xx, H, W);
  const float w00 = t.c00 ? t.hy * t.hx : 0.f, w01 = t.c01 ? t.hy * t.lx : 0.f, w10 = t.c10 ? t.ly * t.hx : 0.f, w11 = t.c11 ? t.ly * t.lx : 0.f;
  const float* __restrict__ xn = x + (size_t)n * C * HW;
  float* __restrict__ cn = col + (size_t)n * C * 9 * HW;
  for (int c = 0; c < C; ++c) {
    const float* __restrict__ pl = xn + (size_t)c * HW;
    cn[((size_t)c * 9 + k) * HW + pix] = w00 * pl[t.j00] + w01 * pl[t.j01] + w10 * pl[t.j10] + w11 * pl[t.j11];
  }
}
// DX = false (round 4): only the offset gradients -- the input gradient comes from dcn_scatter_dx_tile_kernel below
template <bool DX>
__global__ __launch_bounds__(256) void dcn_scatter_bwd_kernel(const float* __restrict__ x, const float* __restrict__ off, const float* __restrict__ dcol,
                                                              float* __restrict__ dx, float* __restrict__ doff, int C, int H, int W) {
  const int n = blockIdx.z, k = blockIdx.y, pix = blockIdx.x * 256 + threadIdx.x, HW = H * W;
  if (pix >= HW) return;
  const int y = pix / W, xx = pix - y * W;
  const DcnTap t = dcn_tap(off, n, k, pix, y, xx, H, W);
  const float w00 = t.c00 ? t.hy * t.hx : 0.f, w01 = t.c01 ? t.hy * t.lx : 0.f, w10 = t.c10 ? t.ly * t.hx : 0.f, w11 = t.c11 ? t.ly * t.lx : 0.f;
  const float* __restrict__ xn = x + (size_t)n * C * HW;
  float* __restrict__ dxn = dx + (size_t)n * C * HW;
  const float* __restrict__ dn = dcol + (size_t)n * C * 9 * HW;
  float gy = 0.f, gx = 0.f;
  for (int c = 0; c < C; ++c) {
    const float dv = dn[((size_t)c * 9 + k) * HW + pix];
    const float* __restrict__ pl = xn + (size_t)c * HW;
    const float v00 = t.c00 ? pl[t.j00] : 0.f, v01 = t.c01 ? pl[t.j01] : 0.f, v10 = t.c10 ? pl[t.j10] : 0.f, v11 = t.c11 ? pl[t.j11] : 0.f;
    // val = hy hx v00 + hy lx v01 + ly hx v10 + ly lx v11;  d ly / d py = 1, d hy / d py = -1 (floor is constant almost everywhere)
    gy = fmaf(dv, t.hx * (v10 - v00) + t.lx * (v11 - v01), gy);
    gx = fmaf(dv, t.hy * (v01 - v00) + t.ly * (v11 - v10), gx);
    if constexpr (DX) {
      float* __restrict__ dpl = dxn + (size_t)c * HW;
      if (t.c00) atomicAdd(&dpl[t.j00], w00 * dv);
      if (t.c01) atomicAdd(&dpl[t.j01], w01 * dv);
      if (t.c10) atomicAdd(&dpl[t.j10], w10 * dv);
      if (t.c11) atomicAdd(&dpl[t.j11], w11 * dv);
    }
  }
  doff[((size_t)n * 18 + 2 * k) * HW + pix] = gy;
  doff[((size_t)n * 18 + 2 * k + 1) * HW + pix] = gx;
}

// Input gradient of the deformable sampling through an LDS tile (round 4): the one-atomic-per-corner form above issued
// 4 x 9 x C x HW x n device-scope float atomics (151 M at 4 x 128 x 64 x 128: 1.2 ms per training step).  A workgroup = 16 x 16 sampling
// pixels x 8 channels; the corners of its 9 x 256 samples fall almost always inside the tile grown by DCN_R pixels, where they are
// added with LDS atomics (ds_add_f32); the rare corner further out goes to global memory directly; at the end every touched cell of the
// (16 + 2 DCN_R)^2 x 8 region is committed with ONE global atomic (the regions of neighbouring workgroups overlap).  dx must be zeroed.
constexpr int DCN_T = 16, DCN_R = 4, DCN_L = DCN_T + 2 * DCN_R, DCN_CH = 8;
// `part` != null: the region is STORED (coalesced, every cell) at part[((n tiles + tile) Cpad + channel)][DCN_L^2] and
// dcn_scatter_dx_gather_kernel sums, per output pixel, the <= 3 x 3 regions that contain it -- no global atomic at all for in-region
// corners (the flush with one atomic per touched cell still cost 0.76 ms: 9 M uncoalesced device-scope read-modify-writes).
__global__ __launch_bounds__(256) void dcn_scatter_dx_tile_kernel(const float* __restrict__ off, const float* __restrict__ dcol, float* __restrict__ dx,
                                                                  int C, int H, int W, float* __restrict__ part) {
  __shared__ float s_acc[DCN_CH][DCN_L * DCN_L];
  const int tid = threadIdx.x, n = blockIdx.z, c0 = blockIdx.y * DCN_CH, HW = H * W;
  const int tiles_x = (W + DCN_T - 1) / DCN_T;
  const int ty0 = (blockIdx.x / tiles_x) * DCN_T, tx0 = (blockIdx.x % tiles_x) * DCN_T;
  for (int i = tid; i < DCN_CH * DCN_L * DCN_L; i += 256) (&s_acc[0][0])[i] = 0.f;
  __syncthreads();
  const int y = ty0 + (tid >> 4), xx = tx0 + (tid & 15);
  const bool live = y < H && xx < W;
  const int pix = y * W + xx;
  const int nch = min(DCN_CH, C - c0);
  float* __restrict__ dxn = dx + ((size_t)n * C + c0) * HW;
  const float* __restrict__ dn = dcol + ((size_t)n * C + c0) * 9 * HW;
  if (live) {
    for (int k = 0; k < 9; ++k) {
      // the tap's cell exactly as dcn_tap computes it, with the corner's coordinates kept
      const float py = (float)(y - 1 + k / 3) + off[((size_t)n * 18 + 2 * k) * HW + pix];
      const float px = (float)(xx - 1 + k % 3) + off[((size_t)n * 18 + 2 * k + 1) * HW + pix];
      if (!(py > -1.f && py < (float)H && px > -1.f && px < (float)W)) continue;
      const float fy = floorf(py), fx = floorf(px);
      const int iy = (int)fy, ix = (int)fx;
      const float ly = py - fy, lx = px - fx, hy = 1.f - ly, hx = 1.f - lx;
      const float wgt[4] = {hy * hx, hy * lx, ly * hx, ly * lx};
      float dvs[DCN_CH];   // the chunk's eight gradient values of this tap: eight loads in flight, not one per memory round trip
#pragma unroll
      for (int c = 0; c < DCN_CH; ++c) dvs[c] = c < nch ? dn[((size_t)c * 9 + k) * HW + pix] : 0.f;
#pragma unroll
      for (int c = 0; c < DCN_CH; ++c) {
        if (c >= nch) break;
        const float dv = dvs[c];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cy = iy + (q >> 1), cx = ix + (q & 1);
          if (cy < 0 || cy > H - 1 || cx < 0 || cx > W - 1) continue;
          const int ry = cy - (ty0 - DCN_R), rx = cx - (tx0 - DCN_R);
          if (ry >= 0 && ry < DCN_L && rx >= 0 && rx < DCN_L) atomicAdd(&s_acc[c][ry * DCN_L + rx], wgt[q] * dv);
          else atomicAdd(&dxn[(size_t)c * HW + (size_t)cy * W + cx], wgt[q] * dv);
        }
      }
    }
  }
  __syncthreads();
  if (part != nullptr) {
    const size_t cpad = (size_t)gridDim.y * DCN_CH;
    float* __restrict__ pp = part + (((size_t)n * gridDim.x + blockIdx.x) * cpad + c0) * (DCN_L * DCN_L);
    for (int i = tid; i < DCN_CH * DCN_L * DCN_L; i += 256) pp[i] = (&s_acc[0][0])[i];
    return;
  }
  for (int i = tid; i < nch * DCN_L * DCN_L; i += 256) {
    const int c = i / (DCN_L * DCN_L), r = i - c * (DCN_L * DCN_L), ry = r / DCN_L, rx = r - ry * DCN_L;
    const int gy = ty0 - DCN_R + ry, gx = tx0 - DCN_R + rx;
    const float v = s_acc[c][r];
    if (v != 0.f && gy >= 0 && gy < H && gx >= 0 && gx < W) atomicAdd(&dxn[(size_t)c * HW + (size_t)gy * W + gx], v);
  }
}

// dx[n][c][y][x] += sum over the tiles whose grown region contains (y, x) of that region's cell (plain read-add-store: runs after the
// tile kernel on the same stream; the rare out-of-region corners were added to dx by atomics before)
__global__ __launch_bounds__(256) void dcn_scatter_dx_gather_kernel(const float* __restrict__ part, float* __restrict__ dx, int C, int H, int W, int cpad) {
  const int n = blockIdx.z, c = blockIdx.y, pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= H * W) return;
  const int y = pix / W, x = pix - y * W;
  const int tiles_x = (W + DCN_T - 1) / DCN_T, tiles_y = (H + DCN_T - 1) / DCN_T, tiles = tiles_x * tiles_y;
  // tile t covers rows [16 t - R, 16 t + 16 + R)
  const int ty_lo = max(0, (y - (DCN_T + DCN_R - 1) + DCN_T - 1) / DCN_T), ty_hi = min(tiles_y - 1, (y + DCN_R) / DCN_T);
  const int tx_lo = max(0, (x - (DCN_T + DCN_R - 1) + DCN_T - 1) / DCN_T), tx_hi = min(tiles_x - 1, (x + DCN_R) / DCN_T);
  float s = 0.f;
  for (int ty = ty_lo; ty <= ty_hi; ++ty)
    for (int tx = tx_lo; tx <= tx_hi; ++tx) {
      const int ry = y - (ty * DCN_T - DCN_R), rx = x - (tx * DCN_T - DCN_R);
      s += part[(((size_t)n * tiles + ty * tiles_x + tx) * cpad + c) * (DCN_L * DCN_L) + ry * DCN_L + rx];
    }
  dx[((size_t)n * C + c) * H * W + pix] += s;
}
inline size_t dcn_scatter_scratch_floats(int n, int C, int H, int W) {
  return (size_t)n * ((H + DCN_T - 1) / DCN_T) * ((W + DCN_T - 1) / DCN_T) * ((size_t)(C + DCN_CH - 1) / DCN_CH * DCN_CH) * (DCN_L * DCN_L);
}

// ---- BatchNorm2d in TRAINING mode (batch statistics) for the conv stacks around the hot path -------------------------------
// (base_bev_backbone.py:47-52: conv -> BatchNorm2d(eps 1e-3, momentum 0.01) -> ReLU; stage 1 of the reference trains them).
//   bn2d_stats_kernel      per-channel sum / sum of squares over (n, HW) in f64 (grid: channel x chunks)
//   (mean, rstd (biased variance) -> save[c][2] and the running statistics (UNBIASED variance) are formed inside bn2d_apply_kernel)
//   bn2d_apply_kernel      y = act(gamma (x - mean) rstd + beta)
//   bn2d_bwd_reduce_kernel sums of g and g xhat per channel (g = dy masked by y > 0 when the block has a ReLU)
//   bn2d_bwd_apply_kernel  dx = gamma rstd (g - mean(g) - xhat mean(g xhat));  d gamma = sum g xhat, d beta = sum g
// V = 4 (HW % 4 == 0, 16-byte aligned maps): four pixels per lane and load, samples in an outer loop -- the one-dword form divided the
// flat index by HW for every element (bn2d_bwd_reduce: 43 us per launch at the training leg's shapes, 23 launches per step)
template <int V>
__global__ __launch_bounds__(256) void bn2d_stats_kernel(const float* __restrict__ x, double* __restrict__ acc /*[C][2]*/, int n, int C, int HW) {
  __shared__ double s_red[4][2];
  const int c = blockIdx.x, tid = threadIdx.x;
  double s = 0.0, q = 0.0;
  if (V == 4) {
    for (int b = 0; b < n; ++b) {
      const float* __restrict__ xp = x + ((size_t)b * C + c) * HW;
      for (int p = 4 * (blockIdx.y * 256 + tid); p < HW; p += gridDim.y * 1024) {
        const float4 v = *reinterpret_cast<const float4*>(xp + p);
        s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        q += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
      }
    }
  } else {
    for (long long i = (long long)blockIdx.y * 256 + tid; i < (long long)n * HW; i += (long long)gridDim.y * 256) {
      const int b = (int)(i / HW), p = (int)(i - (long long)b * HW);
      const float v = x[((size_t)b * C + c) * HW + p];
      s += v; q += (double)v * v;
    }
  }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  if ((tid & 63) == 0) { s_red[tid >> 6][0] = s; s_red[tid >> 6][1] = q; }
  __syncthreads();
  if (tid < 2) atomicAdd(&acc[c * 2 + tid], s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid]);
}
// The statistics' last step rides in this kernel (second half of round 4: one launch per BatchNorm less, 23 per training-leg step): every
// block forms mean / rstd of its channel from the f64 sums (in double, as the separate finish launch did); the block (0, c, 0) also writes save[c] (read by
// the backward) and moves the running statistics.
template <int V>
__global__ __launch_bounds__(256) void bn2d_apply_kernel(const float* __restrict__ x, const double* __restrict__ acc, float* __restrict__ save,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
                                                         long long count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ y, int C, int HW, int relu, long long* __restrict__ nbt = nullptr) {
  const int c = blockIdx.y, b = blockIdx.z, p = V * (blockIdx.x * 256 + threadIdx.x);
  if (nbt != nullptr && blockIdx.x == 0 && c == 0 && b == 0 && threadIdx.x == 0) *nbt += 1;   // num_batches_tracked (one writer per launch)
  const double md = acc[c * 2] / (double)count;
  const double var = fmax(acc[c * 2 + 1] / (double)count - md * md, 0.0);
  const float mean = (float)md, k = (float)(1.0 / sqrt(var + (double)eps));
  if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
    save[c * 2] = mean;
    save[c * 2 + 1] = k;
    if (running_mean != nullptr) {
      const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
  if (p >= HW) return;
  const size_t e = ((size_t)b * C + c) * HW + p;
  const float g = gamma[c], bt = beta[c];
  if (V == 4) {
    const float4 v = *reinterpret_cast<const float4*>(x + e);
    float o[4] = {fmaf((v.x - mean) * k, g, bt), fmaf((v.y - mean) * k, g, bt), fmaf((v.z - mean) * k, g, bt), fmaf((v.w - mean) * k, g, bt)};
    if (relu) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
    *reinterpret_cast<float4*>(y + e) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
    const float v = fmaf((x[e] - mean) * k, g, bt);
    y[e] = relu ? fmaxf(v, 0.f) : v;
  }
}
template <int V>
__global__ __launch_bounds__(256) void bn2d_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                              const float* __restrict__ save, double* __restrict__ acc /*[C][2] sum g, sum g xhat*/,
                                                              int n, int C, int HW, int relu) {
  __shared__ double s_red[4][2];
  const int c = blockIdx.x, tid = threadIdx.x;
  const float mean = save[c * 2], rstd = save[c * 2 + 1];
  double s = 0.0, q = 0.0;
  if (V == 4) {
    for (int b = 0; b < n; ++b) {
      const size_t base = ((size_t)b * C + c) * HW;
      for (int p = 4 * (blockIdx.y * 256 + tid); p < HW; p += gridDim.y * 1024) {
        const float4 x4 = *reinterpret_cast<const float4*>(x + base + p), d4 = *reinterpret_cast<const float4*>(dy + base + p);
        float g[4] = {d4.x, d4.y, d4.z, d4.w};
        if (relu) {
          const float4 y4 = *reinterpret_cast<const float4*>(y + base + p);
          if (!(y4.x > 0.f)) g[0] = 0.f;
          if (!(y4.y > 0.f)) g[1] = 0.f;
          if (!(y4.z > 0.f)) g[2] = 0.f;
          if (!(y4.w > 0.f)) g[3] = 0.f;
        }
        const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
        s += ((double)g[0] + (double)g[1]) + ((double)g[2] + (double)g[3]);
        q += ((double)g[0] * ((xs[0] - mean) * rstd) + (double)g[1] * ((xs[1] - mean) * rstd)) +
             ((double)g[2] * ((xs[2] - mean) * rstd) + (double)g[3] * ((xs[3] - mean) * rstd));
      }
    }
  } else {
    for (long long i = (long long)blockIdx.y * 256 + tid; i < (long long)n * HW; i += (long long)gridDim.y * 256) {
      const int b = (int)(i / HW), p = (int)(i - (long long)b * HW);
      const size_t e = ((size_t)b * C + c) * HW + p;
      const float g = (relu && !(y[e] > 0.f)) ? 0.f : dy[e];
      s += g; q += (double)g * ((x[e] - mean) * rstd);
    }
  }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  if ((tid & 63) == 0) { s_red[tid >> 6][0] = s; s_red[tid >> 6][1] = q; }
  __syncthreads();
  if (tid < 2) atomicAdd(&acc[c * 2 + tid], s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid]);
}
template <int V>
__global__ __launch_bounds__(256) void bn2d_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                             const float* __restrict__ save, const float* __restrict__ gamma, const double* __restrict__ acc,
                                                             float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             long long count, int C, int HW, int relu) {
  const int c = blockIdx.y, b = blockIdx.z, p = V * (blockIdx.x * 256 + threadIdx.x);
  const float mean = save[c * 2], rstd = save[c * 2 + 1];
  const float mg = (float)(acc[c * 2] / (double)count), mgx = (float)(acc[c * 2 + 1] / (double)count);
  if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {   // relu bit 1: the parameter gradients are WRITTEN (the caller did not zero them)
    if (dgamma != nullptr) dgamma[c] = ((relu & 2) ? 0.f : dgamma[c]) + (float)acc[c * 2 + 1];
    if (dbeta != nullptr) dbeta[c] = ((relu & 2) ? 0.f : dbeta[c]) + (float)acc[c * 2];
  }
  if (p >= HW) return;
  const size_t e = ((size_t)b * C + c) * HW + p;
  const float k = gamma[c] * rstd;
  if (V == 4) {
    const float4 x4 = *reinterpret_cast<const float4*>(x + e), d4 = *reinterpret_cast<const float4*>(dy + e);
    float g[4] = {d4.x, d4.y, d4.z, d4.w};
    if (relu & 1) {
      const float4 y4 = *reinterpret_cast<const float4*>(y + e);
      if (!(y4.x > 0.f)) g[0] = 0.f;
      if (!(y4.y > 0.f)) g[1] = 0.f;
      if (!(y4.z > 0.f)) g[2] = 0.f;
      if (!(y4.w > 0.f)) g[3] = 0.f;
    }
    const float xs[4] = {x4.x, x4.y, x4.z, x4.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = k * (g[j] - mg - ((xs[j] - mean) * rstd) * mgx);
    *reinterpret_cast<float4*>(dx + e) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
    const float g = ((relu & 1) && !(y[e] > 0.f)) ? 0.f : dy[e];
    const float xh = (x[e] - mean) * rstd;
    dx[e] = k * (g - mg - xh * mgx);
  }
}

// ---- max over the point slots of a pillar (PFNLayer, pillar_vfe.py:49-52) for the training path of the PointPillars encoder -----
// x [C][M][P] (channel-major rows of the 1x1-conv layout [1, C, 1, M P]) -> out [M][C] and the arg-max slot; backward routes the
// gradient to that slot (ties: the first maximal slot, as torch.max).
// A (c, m) row of P slots is read by P / 4 adjacent lanes as float4s (one 128-byte line per row for P = 32: the earlier form, one thread
// per row, had every lane of a wave on its own line -- 1.67 ms for 64 x 48 000 x 32, 4 % of HBM), reduced with lane shuffles inside the group
// (value, then the LOWEST slot among equal maxima = torch.max's first-occurrence rule).  P % 4 != 0 or P > 64 take the scalar form.
__global__ __launch_bounds__(256) void slot_max_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, unsigned char* __restrict__ arg, int C, int M, int P) {
  const long long rows = (long long)C * M;
  if ((P & 3) == 0 && P <= 64 && ((P / 4) & (P / 4 - 1)) == 0) {
    const int G = P / 4;                                    // lanes per row: a power of two <= 16
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long row = t / G;
    const int q = (int)(t - row * G);
    const bool live = row < rows;
    float best = -INFINITY;
    int bi = 0;
    if (live) {
      const float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * P + 4 * q);
      best = v.x; bi = 4 * q;
      if (v.y > best) { best = v.y; bi = 4 * q + 1; }
      if (v.z > best) { best = v.z; bi = 4 * q + 2; }
      if (v.w > best) { best = v.w; bi = 4 * q + 3; }
    }
    for (int o = 1; o < G; o <<= 1) {                       // butterfly inside the aligned group of G lanes
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (live && q == 0) {
      const int c = (int)(row / M), m = (int)(row - (long long)c * M);
      out[(size_t)m * C + c] = best;
      arg[(size_t)m * C + c] = (unsigned char)bi;
    }
    return;
  }
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows) return;
  const int c = (int)(i / M), m = (int)(i - (long long)c * M);
  const float* __restrict__ p = x + ((size_t)c * M + m) * P;
  float best = p[0];
  int bi = 0;
  for (int k = 1; k < P; ++k) if (p[k] > best) { best = p[k]; bi = k; }
  out[(size_t)m * C + c] = best;
  arg[(size_t)m * C + c] = (unsigned char)bi;
}
__global__ __launch_bounds__(256) void slot_max_bwd_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ arg, float* __restrict__ dx, int C, int M, int P) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)C * M * P) return;
  const int k = (int)(i % P);
  const long long cm = i / P;
  const int c = (int)(cm / M), m = (int)(cm - (long long)c * M);
  dx[i] = arg[(size_t)m * C + c] == k ? dout[(size_t)m * C + c] : 0.f;
}

}  // namespace gc
