// V2X-ViT fusion (reference: opencood/models/fuse_modules/fusion_in_one.py:355-407 +
// sub_modules/{v2xvit_basic, hmsa, mswin, split_attn, base_transformer}.py) -- the kernels that have no counterpart elsewhere
// in the library.  Activations are NCHW fp32 per agent, [sumN][C][H][W]; the Linear layers run as 1x1 convolutions on the
// general implicit-GEMM kernel (conv_kernels.h), LayerNorm on ln_nchw_fwd_kernel (train_kernels.h).
//   warp_affine_kernel   warp_affine_simple (torch_transformation_utils.py:323-332): every agent into the ego frame
//   hgt_attn_kernel      HGTCavAttention's core (hmsa.py:117-150): per pixel and head, attention ACROSS the agents of a scene.
//                        GenComm passes an all-zero prior encoding, so every agent has type 0 and a single relation: the host
//                        folds relation_att into the key projection and relation_msg into the value projection, which leaves a
//                        plain masked softmax attention over at most 8 agents here.
//   win_attn_kernel      BaseWindowAttention's core (mswin.py:47-83): per agent, head and window, softmax(q k^T scale +
//                        relative position bias) v with the keys / values of the window staged in LDS.
// fp32 throughout; softmax with the running maximum; one lane per query.
#pragma once
#include "common.h"

namespace gc {

struct WarpArgs {
  const float* x;        // [n][C][H][W]
  const double* theta;   // [n][2][3] (ego <- agent, normalised: output of normalize_pairwise_tfm)
  float* out;            // [n][C][H][W]
  int C, H, W;
};
__global__ __launch_bounds__(256) void warp_affine_kernel(const WarpArgs a) {
  const int n = blockIdx.y, pix = blockIdx.x * 256 + threadIdx.x;
  const int H = a.H, W = a.W, HW = H * W;
  if (pix >= HW) return;
  const int h = pix / W, w = pix - h * W;
  const double xb = (2.0 * w + 1.0) / (double)W - 1.0, yb = (2.0 * h + 1.0) / (double)H - 1.0;  // affine_grid, align_corners=False, float64
  const double* __restrict__ th = a.theta + (size_t)n * 6;
  const float gx = (float)(th[0] * xb + th[1] * yb + th[2]);
  const float gy = (float)(th[3] * xb + th[4] * yb + th[5]);
  const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)W + 1.f), y0 = (int)fminf(fmaxf(fy, -2.f), (float)H + 1.f);
  const float tx = ix - fx, ty = iy - fy;
  const bool far = fx != (float)x0 || fy != (float)y0;
  const bool xl = x0 >= 0 && x0 < W, xr = x0 + 1 >= 0 && x0 + 1 < W, yt = y0 >= 0 && y0 < H, yb_ = y0 + 1 >= 0 && y0 + 1 < H;
  const int i0 = (xl && yt && !far) ? y0 * W + x0 : -1, i1 = (xr && yt && !far) ? y0 * W + x0 + 1 : -1;
  const int i2 = (xl && yb_ && !far) ? (y0 + 1) * W + x0 : -1, i3 = (xr && yb_ && !far) ? (y0 + 1) * W + x0 + 1 : -1;
  const float w0 = (1.f - tx) * (1.f - ty), w1 = tx * (1.f - ty), w2 = (1.f - tx) * ty, w3 = tx * ty;
  const float* __restrict__ xp = a.x + (size_t)n * a.C * HW;
  float* __restrict__ op = a.out + (size_t)n * a.C * HW + pix;
  for (int c = 0; c < a.C; ++c) {
    const float* __restrict__ pl = xp + (size_t)c * HW;
    float v = 0.f;
    v = fmaf(i0 >= 0 ? pl[i0] : 0.f, w0, v);
    v = fmaf(i1 >= 0 ? pl[i1] : 0.f, w1, v);
    v = fmaf(i2 >= 0 ? pl[i2] : 0.f, w2, v);
    v = fmaf(i3 >= 0 ? pl[i3] : 0.f, w3, v);
    op[(size_t)c * HW] = v;
  }
}

// qkv [n][3*inner][HW] (q | k' | v' blocks of `inner` = heads * DH channels, head-major) -> out [n][inner][HW]
struct HgtArgs {
  const float* qkv;
  const int* scene_off;  // [B+1]
  float* out;
  int heads, HW;
  float scale;
};
template <int DH>
__global__ __launch_bounds__(256) void hgt_attn_kernel(const HgtArgs a) {
  constexpr int MAXN = 8;
  // grid.z = scene x query agent (MAXN slots per scene, the unused ones exit): one thread per (pixel, head, QUERY AGENT), so a scene of
  // N agents puts N times the waves on the machine instead of looping over its queries in one thread (2 agents x 64x128: 43 -> see DESIGN 7)
  const int b = blockIdx.z / MAXN, qi = blockIdx.z % MAXN, m = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  const int off = a.scene_off[b], N = a.scene_off[b + 1] - off;
  if (p >= a.HW || N < 1 || N > MAXN || qi >= N) return;
  const int inner = a.heads * DH;
  const size_t agent = (size_t)3 * inner * a.HW;
  const float* __restrict__ base = a.qkv + (size_t)off * agent + p;
  for (int i = qi; i == qi; ++i) {
    float q[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) q[c] = base[i * agent + (size_t)(m * DH + c) * a.HW];
    float s[MAXN];
    float mx = -INFINITY;
    for (int j = 0; j < N; ++j) {
      const float* __restrict__ kp = base + j * agent + (size_t)(inner + m * DH) * a.HW;
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < DH; ++c) d = fmaf(q[c], kp[(size_t)c * a.HW], d);
      s[j] = d * a.scale;
      mx = fmaxf(mx, s[j]);
    }
    float den = 0.f;
    for (int j = 0; j < N; ++j) { s[j] = expf(s[j] - mx); den += s[j]; }
    const float rden = 1.0f / den;
    float o[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) o[c] = 0.f;
    for (int j = 0; j < N; ++j) {
      const float* __restrict__ vp = base + j * agent + (size_t)(2 * inner + m * DH) * a.HW;
      const float wj = s[j] * rden;
#pragma unroll
      for (int c = 0; c < DH; ++c) o[c] = fmaf(wj, vp[(size_t)c * a.HW], o[c]);
    }
    float* __restrict__ op = a.out + ((size_t)(off + i) * inner + m * DH) * a.HW + p;
#pragma unroll
    for (int c = 0; c < DH; ++c) op[(size_t)c * a.HW] = o[c];
  }
}

// qkv [n][3*inner][H][W] (q | k | v, head-major inside each) -> out [n][inner][H][W]; one workgroup = 256 query tokens of
// one (agent, head): 256 / WS^2 horizontally adjacent windows.  pos [2 WS - 1][2 WS - 1].
// The same attention for LARGE maps: one thread per (pixel, head) streams the head's channels twice -- all N x N scores from one pass
// over q and k, then the N outputs from one pass over v -- so that every q / k / v value is loaded exactly once (the form above reads the
// scene's k and v once per query agent: 4 agents x 96 x 352, 8 heads of 32: 242 us for 550 MB of algorithmic traffic).
template <int DH, int N>
__device__ __forceinline__ void hgt_stream_body(const HgtArgs& a, int off, int m, int p) {
  const int inner = a.heads * DH;
  const size_t agent = (size_t)3 * inner * a.HW;
  const float* __restrict__ base = a.qkv + (size_t)off * agent + (size_t)(m * DH) * a.HW + p;
  float s[N][N];
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < N; ++j) s[i][j] = 0.f;
#pragma unroll 4
  for (int d = 0; d < DH; ++d) {
    float q[N], k[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      q[i] = base[i * agent + (size_t)d * a.HW];
      k[i] = base[i * agent + (size_t)(inner + d) * a.HW];
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = 0; j < N; ++j) s[i][j] = fmaf(q[i], k[j], s[i][j]);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < N; ++j) { s[i][j] *= a.scale; mx = fmaxf(mx, s[i][j]); }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < N; ++j) { s[i][j] = expf(s[i][j] - mx); den += s[i][j]; }
    const float rden = 1.0f / den;
#pragma unroll
    for (int j = 0; j < N; ++j) s[i][j] *= rden;
  }
  float* __restrict__ op = a.out + ((size_t)off * inner + m * DH) * a.HW + p;
#pragma unroll 4
  for (int d = 0; d < DH; ++d) {
    float v[N];
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = base[j * agent + (size_t)(2 * inner + d) * a.HW];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < N; ++j) o = fmaf(s[i][j], v[j], o);
      op[(size_t)i * inner * a.HW + (size_t)d * a.HW] = o;
    }
  }
}
template <int DH>
__global__ __launch_bounds__(256) void hgt_attn_stream_kernel(const HgtArgs a) {
  const int b = blockIdx.z, m = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  const int off = a.scene_off[b], N = a.scene_off[b + 1] - off;   // block-uniform
  if (p >= a.HW) return;
  switch (N) {
    case 1: hgt_stream_body<DH, 1>(a, off, m, p); break;
    case 2: hgt_stream_body<DH, 2>(a, off, m, p); break;
    case 3: hgt_stream_body<DH, 3>(a, off, m, p); break;
    case 4: hgt_stream_body<DH, 4>(a, off, m, p); break;
    case 5: hgt_stream_body<DH, 5>(a, off, m, p); break;
    case 6: hgt_stream_body<DH, 6>(a, off, m, p); break;
    case 7: hgt_stream_body<DH, 7>(a, off, m, p); break;
    case 8: hgt_stream_body<DH, 8>(a, off, m, p); break;
    default: break;
  }
}

struct WinArgs {
  const float* qkv;
  const float* pos;
  float* out;
  int heads, H, W;
  float scale;
};
template <int DH, int WS>
__global__ __launch_bounds__(256) void win_attn_kernel(const WinArgs a) {
  constexpr int T = WS * WS, WPB = 256 / T;  // tokens per window, windows per block
  extern __shared__ float wa_smem[];         // K [WPB][T][DH], V [WPB][T][DH], pos [(2WS-1)^2]
  float* sK = wa_smem;
  float* sV = wa_smem + WPB * T * DH;
  float* sP = sV + WPB * T * DH;
  const int n = blockIdx.z, m = blockIdx.y, tid = threadIdx.x;
  const int nw = a.W / WS, nwin = (a.H / WS) * nw;
  const int win0 = blockIdx.x * WPB;
  const int inner = a.heads * DH, HW = a.H * a.W;
  const float* __restrict__ base = a.qkv + (size_t)n * 3 * inner * HW;
  const int wl = tid / T, tok = tid - wl * T, win = win0 + wl;
  const bool live = win < nwin;
  const int wy = live ? win / nw : 0, wx = live ? win - wy * nw : 0;
  const int ty = tok / WS, tx = tok - ty * WS;
  const int pix = (wy * WS + ty) * a.W + wx * WS + tx;
  for (int i = tid; i < (2 * WS - 1) * (2 * WS - 1); i += 256) sP[i] = a.pos[i];
  float q[DH];
#pragma unroll
  for (int c = 0; c < DH; ++c) {
    const float kq = live ? base[(size_t)(m * DH + c) * HW + pix] : 0.f;
    q[c] = kq * a.scale;
    sK[(wl * T + tok) * DH + c] = live ? base[(size_t)(inner + m * DH + c) * HW + pix] : 0.f;
    sV[(wl * T + tok) * DH + c] = live ? base[(size_t)(2 * inner + m * DH + c) * HW + pix] : 0.f;
  }
  __syncthreads();
  if (!live) return;
  float mx = -INFINITY, den = 0.f;
  float o[DH];
#pragma unroll
  for (int c = 0; c < DH; ++c) o[c] = 0.f;
  const float* __restrict__ kw = sK + wl * T * DH;
  const float* __restrict__ vw = sV + wl * T * DH;
  for (int j = 0; j < T; ++j) {
    const int jy = j / WS, jx = j - jy * WS;
    float d = sP[(jy - ty + WS - 1) * (2 * WS - 1) + (jx - tx + WS - 1)];  // relative_indices[i][j] = idx[j] - idx[i] + WS - 1 (mswin.py:12-16, :34-35)
#pragma unroll
    for (int c = 0; c < DH; ++c) d = fmaf(q[c], kw[j * DH + c], d);
    const float nm = fmaxf(mx, d);
    const float corr = expf(mx - nm), e = expf(d - nm);
    den = den * corr + e;
#pragma unroll
    for (int c = 0; c < DH; ++c) o[c] = fmaf(e, vw[j * DH + c], o[c] * corr);
    mx = nm;
  }
  const float rden = 1.0f / den;
  float* __restrict__ op = a.out + ((size_t)n * inner + m * DH) * HW + pix;
#pragma unroll
  for (int c = 0; c < DH; ++c) op[(size_t)c * HW] = o[c] * rden;
}

// The same attention on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products and sums, as the lane-per-query form)
// for windows of 64 or 256 tokens.  Workgroup = 256 tokens (one 16x16 window or four 8x8 windows) of one head; its keys and values
// sit in LDS as [token][DH + 1] rows (odd stride: conflict-free fills and operand reads).  8 waves: a wave owns one block of 32 queries (two
// waves per SIMD, so one wave's softmax arithmetic runs under the other's MFMAs) and walks its window's keys in blocks of 32, flash style:
//   S^T[key][query] = K Q^T   A = K rows from LDS, B = Q (scaled) from registers (DH / 2 MFMAs); the block comes out with a lane owning
//                             ONE query column and 16 of the 32 keys (the other 16 in lane ^ 32), so the softmax statistics are
//                             register-local plus one exchange;
//   + relative position bias, running max / denominator, P = exp(S - max);
//   O^T[d][query] += V^T P    A = V rows from LDS in the key order the P registers already have, B = P straight from the registers.
// 64 MFMAs per 32 x 32 block at DH = 64 against 8 k broadcast LDS reads + 4 k FMAs per lane in the form above.
using f32x16w = __attribute__((ext_vector_type(16))) float;
template <int DH, int WS>
__global__ __launch_bounds__(512) void win_attn_mfma_kernel(const WinArgs a) {
  constexpr int T = WS * WS, WPB = 256 / T, LD = DH + 1, NKB = T / 32, PW = 2 * WS - 1;
  static_assert(T % 64 == 0 && DH % 32 == 0, "64 / 256-token windows, head width a multiple of 32");
  extern __shared__ float wa_smem[];  // K [256][LD], V [256][LD], pos [PW * PW]
  float* sK = wa_smem;
  float* sV = wa_smem + 256 * LD;
  float* sP = sV + 256 * LD;
  const int n = blockIdx.z, m = blockIdx.y, tid = threadIdx.x;
  const int nw = a.W / WS, nwin = (a.H / WS) * nw;
  const int win0 = blockIdx.x * WPB;
  const int inner = a.heads * DH, HW = a.H * a.W;
  const float* __restrict__ base = a.qkv + (size_t)n * 3 * inner * HW;
  {  // fill: threads 0..255 = the tokens' K rows, threads 256..511 = their V rows; 32 loads in flight per thread
    const int row = tid & 255, isv = tid >> 8;
    const int wl = row / T, tok = row - wl * T, win = win0 + wl;
    const bool live = win < nwin;
    const int wy = live ? win / nw : 0, wx = live ? win - wy * nw : 0;
    const int pix = (wy * WS + tok / WS) * a.W + wx * WS + tok % WS;
    for (int i = tid; i < PW * PW; i += 512) sP[i] = a.pos[i];
    const float* __restrict__ src = base + (size_t)((1 + isv) * inner + m * DH) * HW + pix;
    float* __restrict__ dst = (isv ? sV : sK) + row * LD;
#pragma unroll
    for (int c0 = 0; c0 < DH; c0 += 32) {
      float v[32];
#pragma unroll
      for (int c = 0; c < 32; ++c) v[c] = live ? src[(size_t)(c0 + c) * HW] : 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) dst[c0 + c] = v[c];
    }
  }
  __syncthreads();
  // wave = one block of 32 queries
  const int wave = tid >> 6, l = tid & 63, qi = l & 31, h = l >> 5;
  const int wl = (wave * 32) / T, win = win0 + wl;
  if (win >= nwin) return;  // wave-uniform; no barrier follows
  const int wy = win / nw, wx = win - wy * nw;
  const float* __restrict__ kw = sK + wl * T * LD;
  const float* __restrict__ vw = sV + wl * T * LD;
  const int qtok = (wave * 32) % T + qi, qy = qtok / WS, qx = qtok % WS;
  const int qpix = (wy * WS + qy) * a.W + wx * WS + qx;
  float qv[DH / 2];
#pragma unroll
  for (int s = 0; s < DH / 2; ++s) qv[s] = base[(size_t)(m * DH + 2 * s + h) * HW + qpix] * a.scale;
  f32x16w o[DH / 32];
#pragma unroll
  for (int db = 0; db < DH / 32; ++db)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
  float mrun = -INFINITY, den = 0.f;
  const int pbase = (WS - 1 - qy) * PW + (WS - 1 - qx);
#pragma unroll 1
  for (int kb = 0; kb < NKB; ++kb) {
    f32x16w sc;
#pragma unroll
    for (int i = 0; i < 16; ++i) sc[i] = 0.f;
    const float* __restrict__ kr = kw + (32 * kb + qi) * LD + h;
#pragma unroll
    for (int s = 0; s < DH / 2; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[2 * s], qv[s], sc, 0, 0, 0);
    float bm = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * kb + 8 * (r >> 2) + 4 * h + (r & 3);
      sc[r] += sP[pbase + (key / WS) * PW + key % WS];  // relative_indices[query][key] = idx[key] - idx[query] + WS - 1 (mswin.py:12-16, :34-35)
      bm = fmaxf(bm, sc[r]);
    }
    bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
    const float nm = fmaxf(mrun, bm), corr = __expf(mrun - nm);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sc[r] = __expf(sc[r] - nm); ps += sc[r]; }
    den = den * corr + ps;
    mrun = nm;
#pragma unroll
    for (int db = 0; db < DH / 32; ++db)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[db][i] *= corr;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int key = 32 * kb + 8 * (t >> 2) + 4 * h + (t & 3);
#pragma unroll
      for (int db = 0; db < DH / 32; ++db)  // the head's 32-wide halves alternate: consecutive MFMAs never share an accumulator
        o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vw[key * LD + 32 * db + qi], sc[t], o[db], 0, 0, 0);
    }
  }
  den += __shfl_xor(den, 32, 64);
  const float rden = 1.0f / den;
  float* __restrict__ op = a.out + ((size_t)n * inner + m * DH) * HW + qpix;
#pragma unroll
  for (int db = 0; db < DH / 32; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) op[(size_t)(32 * db + 8 * (r >> 2) + 4 * h + (r & 3)) * HW] = o[db][r] * rden;
}

template <int DH, int WS>
inline int win_attn_launch(const WinArgs& a, int n, hipStream_t st) {
  if constexpr (WS * WS >= 64 && DH % 32 == 0) {
    const size_t shm = ((size_t)2 * 256 * (DH + 1) + (2 * WS - 1) * (2 * WS - 1)) * sizeof(float);
    if (shm > 48 * 1024) GC_HIP(hipFuncSetAttribute((const void*)win_attn_mfma_kernel<DH, WS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    const int nwin_ = (a.H / WS) * (a.W / WS), wpb = 256 / (WS * WS);
    win_attn_mfma_kernel<DH, WS><<<dim3((nwin_ + wpb - 1) / wpb, a.heads, n), 512, shm, st>>>(a);
    GC_HIP(hipGetLastError());
    return GC_OK;
  }
  constexpr int T = WS * WS, WPB = 256 / T;
  const size_t sh = ((size_t)2 * WPB * T * DH + (2 * WS - 1) * (2 * WS - 1)) * sizeof(float);
  if (sh > 48 * 1024) GC_HIP(hipFuncSetAttribute((const void*)win_attn_kernel<DH, WS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
  const int nwin = (a.H / WS) * (a.W / WS);
  win_attn_kernel<DH, WS><<<dim3((nwin + WPB - 1) / WPB, a.heads, n), 256, sh, st>>>(a);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

// ---- radix-3 split attention (sub_modules/split_attn.py:31-62) over the three window branches -------------------------
// gap[n][c] = mean over HW of (a + b + c): one workgroup per (channel, agent)
__global__ __launch_bounds__(256) void split3_gap_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                         float* __restrict__ gap, int C, int HW) {
  __shared__ float s_red[4];
  const int ch = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const size_t base = ((size_t)n * C + ch) * HW;
  float s = 0.f;
  for (int p = tid; p < HW; p += 256) s += a[base + p] + b[base + p] + c[base + p];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((tid & 63) == 0) s_red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) gap[(size_t)n * C + ch] = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / (float)HW;
}
// gate[n][r][c] = softmax over r of fc2(ReLU(LayerNorm(fc1(gap[n])))) [r * C + c]: fc1 [C][C], fc2 [3C][C], no biases;
// one workgroup per agent, C <= 256
struct Split3GateArgs {
  const float* gap; const float* fc1; const float* lnw; const float* lnb; const float* fc2;
  float* gate;   // [n][3][C]
  int C;
};
__global__ __launch_bounds__(256) void split3_gate_kernel(const Split3GateArgs a) {
  __shared__ float s_in[256], s_h[256], s_stat[2];
  const int n = blockIdx.x, tid = threadIdx.x, C = a.C;
  if (tid < C) s_in[tid] = a.gap[(size_t)n * C + tid];
  __syncthreads();
  float h = 0.f;
  if (tid < C) for (int k = 0; k < C; ++k) h = fmaf(a.fc1[(size_t)tid * C + k], s_in[k], h);
  s_h[tid] = tid < C ? h : 0.f;
  __syncthreads();
  if (tid == 0) {
    float m = 0.f, v = 0.f;
    for (int k = 0; k < C; ++k) m += s_h[k];
    m /= (float)C;
    for (int k = 0; k < C; ++k) v += (s_h[k] - m) * (s_h[k] - m);
    s_stat[0] = m; s_stat[1] = rsqrtf(v / (float)C + 1e-5f);
  }
  __syncthreads();
  if (tid < C) s_in[tid] = fmaxf((s_h[tid] - s_stat[0]) * s_stat[1] * a.lnw[tid] + a.lnb[tid], 0.f);
  __syncthreads();
  if (tid < C) {
    float z[3];
    for (int r = 0; r < 3; ++r) {
      float acc = 0.f;
      for (int k = 0; k < C; ++k) acc = fmaf(a.fc2[((size_t)r * C + tid) * C + k], s_in[k], acc);
      z[r] = acc;
    }
    const float mx = fmaxf(z[0], fmaxf(z[1], z[2]));
    const float e0 = expf(z[0] - mx), e1 = expf(z[1] - mx), e2 = expf(z[2] - mx), inv = 1.0f / (e0 + e1 + e2);
    a.gate[((size_t)n * 3 + 0) * C + tid] = e0 * inv;
    a.gate[((size_t)n * 3 + 1) * C + tid] = e1 * inv;
    a.gate[((size_t)n * 3 + 2) * C + tid] = e2 * inv;
  }
}
// out = a g0 + b g1 + c g2 + res   (per agent and channel gates)
__global__ __launch_bounds__(256) void split3_apply_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                           const float* __restrict__ gate, const float* __restrict__ res, float* __restrict__ out,
                                                           int C, int HW) {
  const int ch = blockIdx.y, n = blockIdx.z, p = blockIdx.x * 256 + threadIdx.x;
  if (p >= HW) return;
  const float g0 = gate[((size_t)n * 3 + 0) * C + ch], g1 = gate[((size_t)n * 3 + 1) * C + ch], g2 = gate[((size_t)n * 3 + 2) * C + ch];
  const size_t e = ((size_t)n * C + ch) * HW + p;
  out[e] = fmaf(a[e], g0, fmaf(b[e], g1, fmaf(c[e], g2, res != nullptr ? res[e] : 0.f)));
}

// =====================================================================================================================
// Backward kernels (training with fusion_method v2xvit; first-correct, one lane per query / key, fp32).
// =====================================================================================================================
// warp_affine backward: dx[i][c][corner] += bilinear weight * dout[i][c][pixel]  (the sampling grid does not depend on x)
__global__ __launch_bounds__(256) void warp_affine_bwd_kernel(const WarpArgs a /* x unused; out = dx (zeroed by the caller) */, const float* __restrict__ dout) {
  const int n = blockIdx.y, pix = blockIdx.x * 256 + threadIdx.x;
  const int H = a.H, W = a.W, HW = H * W;
  if (pix >= HW) return;
  const int h = pix / W, w = pix - h * W;
  const double xb = (2.0 * w + 1.0) / (double)W - 1.0, yb = (2.0 * h + 1.0) / (double)H - 1.0;
  const double* __restrict__ th = a.theta + (size_t)n * 6;
  const float gx = (float)(th[0] * xb + th[1] * yb + th[2]);
  const float gy = (float)(th[3] * xb + th[4] * yb + th[5]);
  const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
  const float fx = floorf(ix), fy = floorf(iy);
  if (!(fx >= -1.f && fx <= (float)W && fy >= -1.f && fy <= (float)H)) return;   // all four corners outside
  const int x0 = (int)fx, y0 = (int)fy;
  const float tx = ix - fx, ty = iy - fy;
  const bool xl = x0 >= 0 && x0 < W, xr = x0 + 1 >= 0 && x0 + 1 < W, yt = y0 >= 0 && y0 < H, yb_ = y0 + 1 >= 0 && y0 + 1 < H;
  const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
  for (int c = 0; c < a.C; ++c) {
    const float g = dout[((size_t)n * a.C + c) * HW + pix];
    float* __restrict__ d = a.out + ((size_t)n * a.C + c) * HW;
    if (xl && yt) atomicAdd(&d[y0 * W + x0], w00 * g);
    if (xr && yt) atomicAdd(&d[y0 * W + x0 + 1], w01 * g);
    if (xl && yb_) atomicAdd(&d[(y0 + 1) * W + x0], w10 * g);
    if (xr && yb_) atomicAdd(&d[(y0 + 1) * W + x0 + 1], w11 * g);
  }
}

// agent-wise attention backward: dqkv [n][3 inner][HW] from dout [n][inner][HW]; one lane per (pixel, head), all agents of the scene
struct HgtBwdArgs {
  const float* qkv; const int* scene_off; const float* dout; float* dqkv;
  int heads, HW; float scale;
};
template <int DH>
__global__ __launch_bounds__(256) void hgt_attn_bwd_kernel(const HgtBwdArgs a) {
  constexpr int MAXN = 8;
  const int b = blockIdx.z, m = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
  const int off = a.scene_off[b], N = a.scene_off[b + 1] - off;
  if (p >= a.HW || N < 1 || N > MAXN) return;
  const int inner = a.heads * DH;
  const size_t agent = (size_t)3 * inner * a.HW, HWs = (size_t)a.HW;
  const float* __restrict__ base = a.qkv + (size_t)off * agent + p;
  float* __restrict__ dbase = a.dqkv + (size_t)off * agent + p;
  const float* __restrict__ dob = a.dout + ((size_t)off * inner + m * DH) * HWs + p;
  auto Q = [&](int i, int c) { return base[i * agent + (size_t)(m * DH + c) * HWs]; };
  auto K = [&](int j, int c) { return base[j * agent + (size_t)(inner + m * DH + c) * HWs]; };
  auto V = [&](int j, int c) { return base[j * agent + (size_t)(2 * inner + m * DH + c) * HWs]; };
  auto DO = [&](int i, int c) { return dob[(size_t)i * inner * HWs + (size_t)c * HWs]; };
  // probabilities P[i][j] and dS[i][j] = P (dO_i . v_j - D_i), D_i = sum_j P_ij dO_i . v_j
  float P[MAXN][MAXN], dS[MAXN][MAXN];
  for (int i = 0; i < N; ++i) {
    float s[MAXN], t[MAXN], mx = -INFINITY;
    for (int j = 0; j < N; ++j) {
      float d = 0.f, e = 0.f;
      for (int c = 0; c < DH; ++c) { d = fmaf(Q(i, c), K(j, c), d); e = fmaf(DO(i, c), V(j, c), e); }
      s[j] = d * a.scale; t[j] = e; mx = fmaxf(mx, s[j]);
    }
    float den = 0.f;
    for (int j = 0; j < N; ++j) { s[j] = expf(s[j] - mx); den += s[j]; }
    float D = 0.f;
    for (int j = 0; j < N; ++j) { P[i][j] = s[j] / den; D = fmaf(P[i][j], t[j], D); }
    for (int j = 0; j < N; ++j) dS[i][j] = P[i][j] * (t[j] - D);
  }
  for (int c = 0; c < DH; ++c) {
    for (int i = 0; i < N; ++i) {   // dQ_i = scale sum_j dS_ij k_j
      float g = 0.f;
      for (int j = 0; j < N; ++j) g = fmaf(dS[i][j], K(j, c), g);
      dbase[i * agent + (size_t)(m * DH + c) * HWs] = g * a.scale;
    }
    for (int j = 0; j < N; ++j) {   // dK_j = scale sum_i dS_ij q_i,  dV_j = sum_i P_ij dO_i
      float gk = 0.f, gv = 0.f;
      for (int i = 0; i < N; ++i) { gk = fmaf(dS[i][j], Q(i, c), gk); gv = fmaf(P[i][j], DO(i, c), gv); }
      dbase[j * agent + (size_t)(inner + m * DH + c) * HWs] = gk * a.scale;
      dbase[j * agent + (size_t)(2 * inner + m * DH + c) * HWs] = gv;
    }
  }
}

// window attention backward, phase A (lane = query): dQ, the probability / dS matrices of every window to scratch
// (PS [n][heads][nwin][T][T] x 2), d pos (LDS table, then atomics).  Phase B (lane = key): dK, dV from the columns of PS.
struct WinBwdArgs {
  const float* qkv; const float* pos; const float* out; const float* dout;
  float* dqkv; float* dpos; float* PS;
  int heads, H, W; float scale;
};
template <int DH, int WS>
__global__ __launch_bounds__(256) void win_attn_bwd_a_kernel(const WinBwdArgs a) {
  constexpr int T = WS * WS, WPB = 256 / T, NP = (2 * WS - 1) * (2 * WS - 1);
  extern __shared__ float wb_smem[];  // K [WPB][T][DH], V [WPB][T][DH], pos [NP], dpos [NP]
  float* sK = wb_smem;
  float* sV = wb_smem + WPB * T * DH;
  float* sP = sV + WPB * T * DH;
  float* sD = sP + NP;
  const int n = blockIdx.z, m = blockIdx.y, tid = threadIdx.x;
  const int nw = a.W / WS, nwin = (a.H / WS) * nw;
  const int win0 = blockIdx.x * WPB;
  const int inner = a.heads * DH, HW = a.H * a.W;
  const float* __restrict__ base = a.qkv + (size_t)n * 3 * inner * HW;
  const int wl = tid / T, tok = tid - wl * T, win = win0 + wl;
  const bool live = win < nwin;
  const int wy = live ? win / nw : 0, wx = live ? win - wy * nw : 0;
  const int ty = tok / WS, tx = tok - ty * WS;
  const int pix = (wy * WS + ty) * a.W + wx * WS + tx;
  for (int i = tid; i < NP; i += 256) { sP[i] = a.pos[i]; sD[i] = 0.f; }
  float q[DH], go[DH];
  float D = 0.f;
#pragma unroll
  for (int c = 0; c < DH; ++c) {
    q[c] = live ? base[(size_t)(m * DH + c) * HW + pix] * a.scale : 0.f;
    go[c] = live ? a.dout[((size_t)n * inner + m * DH + c) * HW + pix] : 0.f;
    D = fmaf(go[c], live ? a.out[((size_t)n * inner + m * DH + c) * HW + pix] : 0.f, D);   // sum_j P_ij dO_i . v_j = dO_i . O_i
    sK[(wl * T + tok) * DH + c] = live ? base[(size_t)(inner + m * DH + c) * HW + pix] : 0.f;
    sV[(wl * T + tok) * DH + c] = live ? base[(size_t)(2 * inner + m * DH + c) * HW + pix] : 0.f;
  }
  __syncthreads();
  if (live) {
    const float* __restrict__ kw = sK + wl * T * DH;
    const float* __restrict__ vw = sV + wl * T * DH;
    float mx = -INFINITY, den = 0.f;
    for (int j = 0; j < T; ++j) {
      const int jy = j / WS, jx = j - jy * WS;
      float d = sP[(jy - ty + WS - 1) * (2 * WS - 1) + (jx - tx + WS - 1)];
#pragma unroll
      for (int c = 0; c < DH; ++c) d = fmaf(q[c], kw[j * DH + c], d);
      const float nm = fmaxf(mx, d);
      den = den * expf(mx - nm) + expf(d - nm);
      mx = nm;
    }
    const float rden = 1.0f / den;
    float dq[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) dq[c] = 0.f;
    float* __restrict__ prow = a.PS + ((((size_t)n * a.heads + m) * nwin + win) * 2 * T + tok) * T;   // P row, then dS row at + T * T
    for (int j = 0; j < T; ++j) {
      const int jy = j / WS, jx = j - jy * WS;
      const int pi = (jy - ty + WS - 1) * (2 * WS - 1) + (jx - tx + WS - 1);
      float d = sP[pi], t = 0.f;
#pragma unroll
      for (int c = 0; c < DH; ++c) { d = fmaf(q[c], kw[j * DH + c], d); t = fmaf(go[c], vw[j * DH + c], t); }
      const float pij = expf(d - mx) * rden;
      const float ds = pij * (t - D);
      prow[j] = pij;
      prow[(size_t)T * T + j] = ds;
      atomicAdd(&sD[pi], ds);
#pragma unroll
      for (int c = 0; c < DH; ++c) dq[c] = fmaf(ds, kw[j * DH + c], dq[c]);
    }
    float* __restrict__ dqp = a.dqkv + (size_t)n * 3 * inner * HW + (size_t)(m * DH) * HW + pix;
#pragma unroll
    for (int c = 0; c < DH; ++c) dqp[(size_t)c * HW] = dq[c] * a.scale;
  }
  __syncthreads();
  for (int i = tid; i < NP; i += 256) if (sD[i] != 0.f) atomicAdd(&a.dpos[i], sD[i]);
}
template <int DH, int WS>
__global__ __launch_bounds__(256) void win_attn_bwd_b_kernel(const WinBwdArgs a) {
  constexpr int T = WS * WS, WPB = 256 / T;
  extern __shared__ float wb_smem[];  // Q [WPB][T][DH] (scaled), dO [WPB][T][DH]
  float* sQ = wb_smem;
  float* sG = wb_smem + WPB * T * DH;
  const int n = blockIdx.z, m = blockIdx.y, tid = threadIdx.x;
  const int nw = a.W / WS, nwin = (a.H / WS) * nw;
  const int win0 = blockIdx.x * WPB;
  const int inner = a.heads * DH, HW = a.H * a.W;
  const float* __restrict__ base = a.qkv + (size_t)n * 3 * inner * HW;
  const int wl = tid / T, tok = tid - wl * T, win = win0 + wl;
  const bool live = win < nwin;
  const int wy = live ? win / nw : 0, wx = live ? win - wy * nw : 0;
  const int ty = tok / WS, tx = tok - ty * WS;
  const int pix = (wy * WS + ty) * a.W + wx * WS + tx;
#pragma unroll
  for (int c = 0; c < DH; ++c) {
    sQ[(wl * T + tok) * DH + c] = live ? base[(size_t)(m * DH + c) * HW + pix] * a.scale : 0.f;
    sG[(wl * T + tok) * DH + c] = live ? a.dout[((size_t)n * inner + m * DH + c) * HW + pix] : 0.f;
  }
  __syncthreads();
  if (!live) return;
  float dk[DH], dv[DH];
#pragma unroll
  for (int c = 0; c < DH; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
  const float* __restrict__ qw = sQ + wl * T * DH;
  const float* __restrict__ gw = sG + wl * T * DH;
  const float* __restrict__ pcol = a.PS + (((size_t)n * a.heads + m) * nwin + win) * 2 * T * T + tok;   // column `tok` of P (then of dS)
  for (int i = 0; i < T; ++i) {
    const float pij = pcol[(size_t)i * T], ds = pcol[(size_t)T * T + (size_t)i * T];
#pragma unroll
    for (int c = 0; c < DH; ++c) { dv[c] = fmaf(pij, gw[i * DH + c], dv[c]); dk[c] = fmaf(ds, qw[i * DH + c], dk[c]); }   // qw is q * scale already
  }
  float* __restrict__ dkp = a.dqkv + (size_t)n * 3 * inner * HW + (size_t)(inner + m * DH) * HW + pix;
  float* __restrict__ dvp = a.dqkv + (size_t)n * 3 * inner * HW + (size_t)(2 * inner + m * DH) * HW + pix;
#pragma unroll
  for (int c = 0; c < DH; ++c) { dkp[(size_t)c * HW] = dk[c]; dvp[(size_t)c * HW] = dv[c]; }
}
template <int DH, int WS>
inline int win_attn_bwd_launch(const WinBwdArgs& a, int n, hipStream_t st) {
  constexpr int T = WS * WS, WPB = 256 / T, NP = (2 * WS - 1) * (2 * WS - 1);
  const size_t sha = ((size_t)2 * WPB * T * DH + 2 * NP) * sizeof(float), shb = (size_t)2 * WPB * T * DH * sizeof(float);
  if (sha > 48 * 1024) GC_HIP(hipFuncSetAttribute((const void*)win_attn_bwd_a_kernel<DH, WS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sha));
  if (shb > 48 * 1024) GC_HIP(hipFuncSetAttribute((const void*)win_attn_bwd_b_kernel<DH, WS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shb));
  const int nwin = (a.H / WS) * (a.W / WS);
  const dim3 grid((nwin + WPB - 1) / WPB, a.heads, n);
  win_attn_bwd_a_kernel<DH, WS><<<grid, 256, sha, st>>>(a);
  win_attn_bwd_b_kernel<DH, WS><<<grid, 256, shb, st>>>(a);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

}  // namespace gc
