// MessageExtractorv2: the producer of the 2-channel "spatial message" that conditions
// the diffusion UNet. Reference: opencood/models/gencomm_modules/message_extractor_v2.py:70-120
//   offset = conv3x3(x, C -> 18)                                   :75, :104
//   b1     = DeformConv2d(x, offset; C -> 64, 3x3, pad 1)          :78, :108   (torchvision, DCNv1)
//   gate   = sigmoid(W3 relu(W1 mean_hw(b1)))                      :88-94, :113
//   out    = W_b relu(W_a (b1 * gate))   (1x1: 64 -> 64 -> 2)      :82-86, :116
// Deformable sampling follows torchvision's deform_conv2d: offset channel 2k / 2k+1 = vertical /
// horizontal displacement of tap k, bilinear with zero outside (restated in oracle/torch_port.py;
// torchvision is not available to pin it against).
// All contractions run on the matrix cores with the weight-broadcast 4x4x1 scheme of unet_kernels.h;
// in the deformable conv every lane gathers the bilinear samples of ITS pixel and feeds them to the
// MFMA directly as the B operand -- no im2col buffer, no LDS.
#pragma once
#include "unet_kernels.h"

namespace gc {

// ---------------------------------------------------------------------------------------------
// M1: plain 3x3 conv C -> 4*NOG channels (OC of them real), input streamed through LDS in chunks of 8.
// ---------------------------------------------------------------------------------------------
struct ConvNArgs {
  const float* x;     // [n][C][H][W]
  const float* w;     // prepared [C][9][4*NOG], zero padded
  const float* bias;  // [OC]
  float* dst;         // [n][OC][H][W]
  int C, OC, H, W;
};

template <int TW, int TH, int PPL, int NOG>
__global__ __launch_bounds__((TW / PPL) * TH) void conv3x3_cN_kernel(const ConvNArgs a) {
  constexpr int NT = (TW / PPL) * TH;
  constexpr int LH = TH + 2, LS = TW + 8;
  constexpr int WF = 8 * 9 * NOG * 4, NREG = (WF + 63) / 64;
  __shared__ __align__(16) float tile[8][LH][LS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int n = blockIdx.z;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
  const int tx = tid % (TW / PPL), ty = tid / (TW / PPL);
  const size_t plane = (size_t)a.H * a.W;
  const bool wvec = (a.W & 3) == 0;
  const int nchunk = a.C / 8;
  f32x4 acc[NOG][PPL];
#pragma unroll
  for (int g = 0; g < NOG; ++g)
#pragma unroll
    for (int p = 0; p < PPL; ++p) acc[g][p] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ xp = a.x + (size_t)n * a.C * plane;
  TileRegs<TW, TH, NT, 8> R;
  float wn[NREG];
  if (wvec) stage_load<TW, TH, NT, 8, false>(R, xp, (unsigned)plane, a.W, a.H, a.W, x0, y0, tid);
  load_wregs<NREG>(wn, a.w, WF, lane);
#pragma unroll 1
  for (int ch = 0; ch < nchunk; ++ch) {
    __syncthreads();
    if (wvec) stage_store<TW, TH, NT, 8, false, LS>(tile, R, a.H, a.W, x0, y0, nullptr, tid);
    else stage_tile_scalar<TW, TH, NT, 8, false, false, LS>(tile, xp + (size_t)ch * 8 * plane, (unsigned)plane, a.W, a.H, a.W, x0, y0, nullptr, tid);
    float wcur[NREG];
#pragma unroll
    for (int g = 0; g < NREG; ++g) wcur[g] = wn[g];
    if (ch + 1 < nchunk) {
      if (wvec) stage_load<TW, TH, NT, 8, false>(R, xp + (size_t)(ch + 1) * 8 * plane, (unsigned)plane, a.W, a.H, a.W, x0, y0, tid);
      load_wregs<NREG>(wn, a.w + (size_t)(ch + 1) * WF, WF, lane);
    }
    __syncthreads();
    conv_tile_mfma<8, NOG, PPL, LH, LS, NREG>(tile, wcur, acc, tx, ty);
  }
  const int gy = y0 + ty, gx = x0 + tx * PPL;
  if (gy >= a.H) return;
  for (int o = 0; o < a.OC; ++o) {
    const float b = a.bias[o];
    float* __restrict__ dp = a.dst + ((size_t)n * a.OC + o) * plane + (size_t)gy * a.W + gx;
#pragma unroll
    for (int p = 0; p < PPL; ++p) {
      // acc is indexed with compile-time constants only: select by comparison
      float v = 0.f;
#pragma unroll
      for (int g = 0; g < NOG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) v = (o == g * 4 + i) ? acc[g][p][i] : v;
      if (gx + p < a.W) dp[p] = v + b;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// M2: deformable 3x3 conv C -> 64 (DCNv1, one offset group), + per-channel sums for the SE gate.
// ---------------------------------------------------------------------------------------------
struct DcnArgs {
  const float* x;       // [n][C][H][W]
  const float* off;     // [n][18][H][W]
  const float* w;       // prepared [C][9][64]
  const float* bias;    // [64]
  float* b1;            // [n][64][H][W]
  float* colsum;        // [n][64]
  int C, H, W;
};

__global__ __launch_bounds__(256) void dcn_kernel(const DcnArgs a) {
  __shared__ float s_red[4][64];
  const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int HW = a.H * a.W;
  const int pix = blockIdx.x * 256 + tid;
  const bool ok = pix < HW;
  const int y = ok ? pix / a.W : 0, x = ok ? pix - y * a.W : 0;
  float off[18];
#pragma unroll
  for (int k = 0; k < 18; ++k) off[k] = ok ? a.off[((size_t)n * 18 + k) * HW + pix] : 0.f;
  f32x4 acc[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ xn = a.x + (size_t)n * a.C * HW;
  const float fH = (float)a.H, fW = (float)a.W;

#pragma unroll 1
  for (int ch = 0; ch < a.C / 8; ++ch) {
    float wr[72];
    load_wregs<72>(wr, a.w + (size_t)ch * 4608, 4608, lane);
    const float* __restrict__ xc = xn + (size_t)ch * 8 * HW;
    static_for<0, 9>([&](auto TAP) {
      constexpr int k = decltype(TAP)::value, ky = k / 3, kx = k % 3;
      // sample position and bilinear taps of this pixel for tap k (torchvision bilinear_interpolate)
      const float py = (float)(y - 1 + ky) + off[2 * k], px = (float)(x - 1 + kx) + off[2 * k + 1];
      const bool inside = ok && py > -1.f && py < fH && px > -1.f && px < fW;
      const float fy = floorf(py), fx = floorf(px);
      const int iy = (int)fy, ix = (int)fx;
      const float ly = py - fy, lx = px - fx, hy = 1.f - ly, hx = 1.f - lx;
      const bool y0ok = inside && iy >= 0, y1ok = inside && iy + 1 <= a.H - 1;
      const bool x0ok = ix >= 0, x1ok = ix + 1 <= a.W - 1;
      const int i00 = iy * a.W + ix;
      const float w00 = (y0ok && x0ok) ? hy * hx : 0.f, w01 = (y0ok && x1ok) ? hy * lx : 0.f;
      const float w10 = (y1ok && x0ok) ? ly * hx : 0.f, w11 = (y1ok && x1ok) ? ly * lx : 0.f;
      const int j00 = (y0ok && x0ok) ? i00 : 0, j01 = (y0ok && x1ok) ? i00 + 1 : 0;
      const int j10 = (y1ok && x0ok) ? i00 + a.W : 0, j11 = (y1ok && x1ok) ? i00 + a.W + 1 : 0;
      static_for<0, 8>([&](auto IC) {
        constexpr int ic = decltype(IC)::value;
        const float* __restrict__ pl = xc + (size_t)ic * HW;
        const float v = w00 * pl[j00] + w01 * pl[j01] + w10 * pl[j10] + w11 * pl[j11];
        static_for<0, 16>([&](auto OG) {
          constexpr int og = decltype(OG)::value;
          acc[og] = mfma_wbcast<og>(wr[ic * 9 + k], v, acc[og]);
        });
      });
    });
  }

  // epilogue: bias, store, per-channel sums over the pixels of this workgroup
#pragma unroll
  for (int g = 0; g < 16; ++g) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int oc = g * 4 + i;
      const float v = acc[g][i] + as_const(a.bias)[oc];
      if (ok) a.b1[((size_t)n * 64 + oc) * HW + pix] = v;
      const float t = wave_total(ok ? v : 0.f);
      if (lane == 0) s_red[tid >> 6][oc] = t;
    }
  }
  __syncthreads();
  if (tid < 64) atomicAdd(&a.colsum[(size_t)n * 64 + tid], s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid]);
}

// The same convolution for SMALL maps (the shipped 64 x 128 features: dcn_kernel would launch 32 workgroups per agent, each wave a chain
// of 4.6 k dependent gathers and 18 k MFMAs): workgroup = 64 pixels, wave q = every fourth 8-channel chunk of the INPUT (the gathers are
// what binds this kernel: splitting the output channels over the waves instead repeats them four times and was slower), all 64 output
// channels as partial sums; the four partial accumulators of a pixel meet in LDS and wave q finishes output channels 16 q .. 16 q + 15.
__global__ __launch_bounds__(256) void dcn_csplit_kernel(const DcnArgs a) {
  __shared__ float s_acc[4][64][64];  // [wave][output channel][pixel lane]
  const int n = blockIdx.y, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int HW = a.H * a.W;
  const int pix = blockIdx.x * 64 + lane;
  const bool ok = pix < HW;
  const int y = ok ? pix / a.W : 0, x = ok ? pix - y * a.W : 0;
  float off[18];
#pragma unroll
  for (int k = 0; k < 18; ++k) off[k] = ok ? a.off[((size_t)n * 18 + k) * HW + pix] : 0.f;
  f32x4 acc[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ xn = a.x + (size_t)n * a.C * HW;
  const float fH = (float)a.H, fW = (float)a.W;
#pragma unroll 1
  for (int ch = q; ch < a.C / 8; ch += 4) {
    float wr[72];
    load_wregs<72>(wr, a.w + (size_t)ch * 4608, 4608, lane);
    const float* __restrict__ xc = xn + (size_t)ch * 8 * HW;
    static_for<0, 9>([&](auto TAP) {
      constexpr int k = decltype(TAP)::value, ky = k / 3, kx = k % 3;
      const float py = (float)(y - 1 + ky) + off[2 * k], px = (float)(x - 1 + kx) + off[2 * k + 1];
      const bool inside = ok && py > -1.f && py < fH && px > -1.f && px < fW;
      const float fy = floorf(py), fx = floorf(px);
      const int iy = (int)fy, ix = (int)fx;
      const float ly = py - fy, lx = px - fx, hy = 1.f - ly, hx = 1.f - lx;
      const bool y0ok = inside && iy >= 0, y1ok = inside && iy + 1 <= a.H - 1;
      const bool x0ok = ix >= 0, x1ok = ix + 1 <= a.W - 1;
      const int i00 = iy * a.W + ix;
      const float w00 = (y0ok && x0ok) ? hy * hx : 0.f, w01 = (y0ok && x1ok) ? hy * lx : 0.f;
      const float w10 = (y1ok && x0ok) ? ly * hx : 0.f, w11 = (y1ok && x1ok) ? ly * lx : 0.f;
      const int j00 = (y0ok && x0ok) ? i00 : 0, j01 = (y0ok && x1ok) ? i00 + 1 : 0;
      const int j10 = (y1ok && x0ok) ? i00 + a.W : 0, j11 = (y1ok && x1ok) ? i00 + a.W + 1 : 0;
      static_for<0, 8>([&](auto IC) {
        constexpr int ic = decltype(IC)::value;
        const float* __restrict__ pl = xc + (size_t)ic * HW;
        const float v = w00 * pl[j00] + w01 * pl[j01] + w10 * pl[j10] + w11 * pl[j11];
        static_for<0, 16>([&](auto OG) {
          constexpr int og = decltype(OG)::value;
          acc[og] = mfma_wbcast<og>(wr[ic * 9 + k], v, acc[og]);
        });
      });
    });
  }
#pragma unroll
  for (int g = 0; g < 16; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) s_acc[q][g * 4 + i][lane] = acc[g][i];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int oc = 16 * q + j;
    const float v = (s_acc[0][oc][lane] + s_acc[1][oc][lane]) + (s_acc[2][oc][lane] + s_acc[3][oc][lane]) + as_const(a.bias)[oc];
    if (ok) a.b1[((size_t)n * 64 + oc) * HW + pix] = v;
    const float t = wave_total(ok ? v : 0.f);
    if (lane == 0) atomicAdd(&a.colsum[(size_t)n * 64 + oc], t);
  }
}

// ---------------------------------------------------------------------------------------------
// M3: SE gate per agent: mean -> 1x1 (64 -> 32) -> ReLU -> 1x1 (32 -> 64) -> sigmoid
// ---------------------------------------------------------------------------------------------
struct MsgGateArgs {
  const float* colsum; const float* w1; const float* b1; const float* w3; const float* b3; float* gate; float inv_hw;
};
__global__ __launch_bounds__(64) void msg_gate_kernel(const MsgGateArgs a) {
  __shared__ float g[64], h[32];
  const int n = blockIdx.x, t = threadIdx.x;
  g[t] = a.colsum[(size_t)n * 64 + t] * a.inv_hw;
  __syncthreads();
  if (t < 32) {
    float s = a.b1[t];
    for (int c = 0; c < 64; ++c) s = fmaf(a.w1[t * 64 + c], g[c], s);
    h[t] = fmaxf(s, 0.f);
  }
  __syncthreads();
  float s = a.b3[t];
  for (int c = 0; c < 32; ++c) s = fmaf(a.w3[t * 32 + c], h[c], s);
  a.gate[(size_t)n * 64 + t] = 1.0f / (1.0f + expf(-s));
}

// ---------------------------------------------------------------------------------------------
// M4: out = W_b relu(W_a (b1 * gate) + b_a) + b_b   per pixel (1x1 convs 64 -> 64 -> 2)
// ---------------------------------------------------------------------------------------------
struct MsgFuseArgs {
  const float* b1; const float* gate;
  const float* wa;  // prepared [64 ic][64 oc]
  const float* ba;  // [64]
  const float* wb;  // [2][64]
  const float* bb;  // [2]
  float* out;       // [n][2][HW]
  int HW;
};
__global__ __launch_bounds__(256) void msg_fuse_kernel(const MsgFuseArgs a) {
  const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int pix = blockIdx.x * 256 + tid;
  const bool ok = pix < a.HW;
  float wr[64];
  load_wregs<64>(wr, a.wa, 4096, lane);
  f32x4 acc[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  static_for<0, 64>([&](auto IC) {
    constexpr int ic = decltype(IC)::value;
    const float e = ok ? a.b1[((size_t)n * 64 + ic) * a.HW + pix] * as_const(a.gate)[n * 64 + ic] : 0.f;
    static_for<0, 16>([&](auto OG) {
      constexpr int og = decltype(OG)::value;
      acc[og] = mfma_wbcast<og>(wr[ic], e, acc[og]);
    });
  });
  float o0 = as_const(a.bb)[0], o1 = as_const(a.bb)[1];
#pragma unroll
  for (int g = 0; g < 16; ++g)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = g * 4 + i;
      const float h = fmaxf(acc[g][i] + as_const(a.ba)[c], 0.f);
      o0 = fmaf(as_const(a.wb)[c], h, o0);
      o1 = fmaf(as_const(a.wb)[64 + c], h, o1);
    }
  if (ok) {
    a.out[((size_t)n * 2 + 0) * a.HW + pix] = o0;
    a.out[((size_t)n * 2 + 1) * a.HW + pix] = o1;
  }
}

}  // namespace gc
