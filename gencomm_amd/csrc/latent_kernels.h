// "Latent" sampler step: conv_out(t) -> ancestral update -> conv_in(t-1) collapsed into ONE 8 -> 8 kernel.
//
// The loop of GenComm.p_sample_loop (opencood/models/gencomm_modules/cond_diff.py:321-329) feeds x_t
// to the UNet only through conv_in, the update x_{t-1} = c1*x0_hat + c2*x_t + s*eps (:272-315) is
// linear, and x0_hat = conv_out(A) is linear in A = SiLU(GroupNorm(h)) (unet.py:340-343). So with
//     k     = W_cond (*) cond + b_in                        (constant over the steps)
//     hs0_t = k + W_x (*) x_t                               (= conv_in's output, the UNet's first map)
// the whole loop can be carried on the 8-channel map hs0 alone:
//     hs0_{t-1} = k + c2*(hs0_t - k) + c1*[ Wc5 (*) A_t + bsum + fix ] + s*( W_x (*) eps_t )
// where Wc5 is the 5x5 composite of W_x and W_out, and `fix` removes on the 1-pixel image border
// the terms that would pass through x0_hat positions OUTSIDE the image (two zero-padded convs in
// sequence never see them). The C-channel x_t is never written or read between steps: per step
// 1600 + 72*C MACs/pixel instead of 144*C, and ~72 MB of traffic instead of ~510 MB (metric config).
// Only the last step (t = 0) runs conv_out, to emit x0_hat itself. The noise field is the same
// Philox field the direct sampler draws (counter = element index of the quad, component = pixel).
// Algebra validated in float64 by tools/latent_check.py; HIP path validated against the golden
// vectors (explicit noise) and against the direct sampler (same Philox seed).
#pragma once
#include "unet_kernels.h"

namespace gc {

// ---------------------------------------------------------------------------------------------
// one-time weight preparation (f64 accumulation):
//   wc5  [8 i][25 d][8 o] = sum_c sum_{t1+t2=d} Wx[o][c][t1] * Wo[c][i][t2]
//   wc1  [9 t1][9 t2][8 i][8 o] = sum_c Wx[o][c][t1] * Wo[c][i][t2]
//   bring[9 t1][8 o] = sum_c Wx[o][c][t1] * b_out[c] ;  bsum[8 o] = sum_t1 bring
// ---------------------------------------------------------------------------------------------
struct PrepLatentArgs {
  const float* w_in;   // raw conv_in.weight [8][C+2][3][3]
  const float* w_out;  // raw conv_out.weight [C][8][3][3]
  const float* b_out;  // raw conv_out.bias [C]
  float* wc5; float* wc1; float* bring; float* bsum;
  int C;
};
__global__ void prep_latent_kernel(const PrepLatentArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int C = a.C, CI = C + 2;
  auto Wx = [&](int o, int c, int t) { return (double)a.w_in[((size_t)o * CI + 2 + c) * 9 + t]; };
  auto Wo = [&](int c, int i, int t) { return (double)a.w_out[((size_t)c * 8 + i) * 9 + t]; };
  if (idx < 1600) {
    const int o = idx & 7, d = (idx >> 3) % 25, i = idx / 200;
    const int dy = d / 5, dx = d % 5;
    double s = 0.0;
    for (int t1y = 0; t1y < 3; ++t1y)
      for (int t1x = 0; t1x < 3; ++t1x) {
        const int t2y = dy - t1y, t2x = dx - t1x;
        if (t2y < 0 || t2y > 2 || t2x < 0 || t2x > 2) continue;
        for (int c = 0; c < C; ++c) s += Wx(o, c, t1y * 3 + t1x) * Wo(c, i, t2y * 3 + t2x);
      }
    a.wc5[idx] = (float)s;
  } else if (idx < 1600 + 5184) {
    const int j = idx - 1600;
    const int o = j & 7, i = (j >> 3) & 7, t2 = (j >> 6) % 9, t1 = j / 576;
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += Wx(o, c, t1) * Wo(c, i, t2);
    a.wc1[j] = (float)s;
  } else if (idx < 1600 + 5184 + 72) {
    const int j = idx - 1600 - 5184, o = j & 7, t1 = j >> 3;
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += Wx(o, c, t1) * (double)a.b_out[c];
    a.bring[j] = (float)s;
  } else if (idx < 1600 + 5184 + 72 + 8) {
    const int o = idx - 1600 - 5184 - 72;
    double s = 0.0;
    for (int t1 = 0; t1 < 9; ++t1)
      for (int c = 0; c < C; ++c) s += Wx(o, c, t1) * (double)a.b_out[c];
    a.bsum[o] = (float)s;
  }
}

// ---------------------------------------------------------------------------------------------
// the fused step
// ---------------------------------------------------------------------------------------------
struct LatentArgs {
  const float* a_src;    // [n][8][H][W] last block's output (input of norm_out)
  const double* a_stat;  // [n][8][2]
  const float* gamma; const float* beta;  // norm_out
  const float* kmap;     // [n][8][H][W]
  float* hs0;            // [n][8][H][W] in: hs0_t, out: hs0_{t-1} (in place)
  double* hs0_stat;      // [n][8][2], zeroed by the caller
  const float* wc5; const float* wc1; const float* bring; const float* bsum;
  const float* wx;       // prepared conv_in x-part [C][9][8]
  const float* wc5h;     // fp16 hi/lo tables of wc5 + scale (latenth_kernels.h)
  const float* wxh;      // fp16 hi/lo tables of the conv_in x-part, same scale
  const float* noise;    // [n][C][H][W] explicit eps (NOISE == 1)
  const float* sched;    // device [5] row of this timestep
  double inv_cnt;        // 1 / (2*H*W)
  unsigned long long seed;
  unsigned int stream_id;
  int C, H, W;
  const unsigned long long* seed_dev;  // optional device-resident Philox key
  int xcd;
  int a_bf16;            // bf16 denoise mode: `a_src` is a bf16 map (latent_step_h_kernel only)
};

template <int TW, int TH, int NOISE>
__global__ __launch_bounds__((TW / 4) * TH) void latent_step_kernel(const LatentArgs a) {
  constexpr int NT = (TW / 4) * TH, QPR = TW / 4;
  constexpr int LH5 = TH + 4, LH = TH + 2, LS = TW + 8;
  static_assert(NT / QPR == TH, "one staging pass must cover TH rows");
  __shared__ __align__(16) float smem[8 * LH5 * LS];
  __shared__ float s_ab[8][2];
  __shared__ float s_red[NT / 64][16];
  float (*tileA)[LH5][LS] = reinterpret_cast<float (*)[LH5][LS]>(smem);
  float (*tileE)[LH][LS] = reinterpret_cast<float (*)[LH][LS]>(smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const BlockId bid = xcd_block(a.xcd);
  const int n = bid.z;
  const int x0 = bid.x * TW, y0 = bid.y * TH;
  const int tx = tid % QPR, ty = tid / QPR;
  const int H = a.H, W = a.W;
  const unsigned plane = (unsigned)(H * W);
  const float c1 = as_const(a.sched)[2], c2 = as_const(a.sched)[3], sg = as_const(a.sched)[4];
  const unsigned long long seed = (NOISE == 2 && a.seed_dev) ? *a.seed_dev : a.seed;

  // ---------------- phase 1: A' = c1 * SiLU(GN(a)) with a 2-pixel halo -> LDS ----------------
  float wc[25];
  load_wregs<25>(wc, a.wc5, 1600, lane);
  const float* __restrict__ ap = a.a_src + (size_t)n * 8 * plane;
  float4 qm[8], qr[(4 * 8 * QPR + NT - 1) / NT];
  constexpr int NREM = (4 * 8 * QPR + NT - 1) / NT;
  constexpr int NHAL = (8 * LH5 * 4 + NT - 1) / NT;
  float hh[NHAL];
  auto load_quad = [&](int c, int r, int qx) {
    const int gy = y0 - 2 + r, gx = x0 + 4 * qx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < LH5 && gy >= 0 && gy < H && gx < W) v = *reinterpret_cast<const float4*>(ap + ((unsigned)c * plane + (unsigned)gy * (unsigned)W + (unsigned)gx));
    return v;
  };
#pragma unroll
  for (int c = 0; c < 8; ++c) qm[c] = load_quad(c, ty, tx);
#pragma unroll
  for (int j = 0; j < NREM; ++j) {
    const int idx = tid + j * NT;
    const int c = idx / (4 * QPR), rr = TH + (idx / QPR) % 4;
    qr[j] = c < 8 ? load_quad(c, rr, tx) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int j = 0; j < NHAL; ++j) {  // halo columns x0-2, x0-1, x0+TW, x0+TW+1
    const int idx = tid + j * NT;
    const int c = idx / (LH5 * 4), rem = idx - c * (LH5 * 4), r = rem >> 2, s = rem & 3;
    const int gy = y0 - 2 + r, gx = s < 2 ? x0 - 2 + s : x0 + TW + (s - 2);
    hh[j] = 0.f;
    if (c < 8 && gy >= 0 && gy < H && gx >= 0 && gx < W) hh[j] = ap[(unsigned)c * plane + (unsigned)gy * (unsigned)W + (unsigned)gx];
  }
  if (tid < 8) {
    float A, B;
    gn_coeff(a.a_stat + (size_t)n * 16, tid, 2, a.inv_cnt, a.gamma[tid], a.beta[tid], &A, &B);
    s_ab[tid][0] = A;
    s_ab[tid][1] = B;
  }
  __syncthreads();
  auto act = [&](int c, float v, bool ok) { return ok ? c1 * silu_f(fmaf(s_ab[c][0], v, s_ab[c][1])) : 0.f; };
  auto store_quad = [&](int c, int r, int qx, float4 q) {
    if (r >= LH5) return;
    const int gy = y0 - 2 + r, gx = x0 + 4 * qx;
    const bool ok = gy >= 0 && gy < H && gx < W;
    *reinterpret_cast<float4*>(&tileA[c][r][4 + 4 * qx]) = make_float4(act(c, q.x, ok), act(c, q.y, ok), act(c, q.z, ok), act(c, q.w, ok));
  };
#pragma unroll
  for (int c = 0; c < 8; ++c) store_quad(c, ty, tx, qm[c]);
#pragma unroll
  for (int j = 0; j < NREM; ++j) {
    const int idx = tid + j * NT;
    const int c = idx / (4 * QPR), rr = TH + (idx / QPR) % 4;
    if (c < 8) store_quad(c, rr, tx, qr[j]);
  }
#pragma unroll
  for (int j = 0; j < NHAL; ++j) {
    const int idx = tid + j * NT;
    const int c = idx / (LH5 * 4), rem = idx - c * (LH5 * 4), r = rem >> 2, s = rem & 3;
    if (c < 8) {
      const int gy = y0 - 2 + r, gx = s < 2 ? x0 - 2 + s : x0 + TW + (s - 2);
      tileA[c][r][s < 2 ? 2 + s : TW + 4 + (s - 2)] = act(c, hh[j], gy >= 0 && gy < H && gx >= 0 && gx < W);
    }
  }
  // eps chunk 0 is requested now so that it arrives during the composite phase
  const float* __restrict__ np_ = NOISE == 1 ? a.noise + (size_t)n * a.C * plane : nullptr;
  TileRegs<TW, TH, NT, 8> R;
  if (NOISE == 1) stage_load<TW, TH, NT, 8, false>(R, np_, plane, W, H, W, x0, y0, tid);
  float wn[9];
  load_wregs<9>(wn, a.wx, 576, lane);
  __syncthreads();

  // ---------------- phase 2: 5x5 composite on the matrix cores ----------------
  f32x4 acc[2][4];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[g][p] = f32x4{0.f, 0.f, 0.f, 0.f};
  static_for<0, 8>([&](auto IC) {
    constexpr int ic = decltype(IC)::value;
    float in[5][8];
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
      const float2 l = *reinterpret_cast<const float2*>(&tileA[ic][ty + dy][tx * 4 + 2]);
      const float4 m = *reinterpret_cast<const float4*>(&tileA[ic][ty + dy][tx * 4 + 4]);
      const float2 r = *reinterpret_cast<const float2*>(&tileA[ic][ty + dy][tx * 4 + 8]);
      in[dy][0] = l.x; in[dy][1] = l.y; in[dy][2] = m.x; in[dy][3] = m.y; in[dy][4] = m.z; in[dy][5] = m.w; in[dy][6] = r.x; in[dy][7] = r.y;
    }
    static_for<0, 25>([&](auto TAP) {
      constexpr int tap = decltype(TAP)::value, dy = tap / 5, dx = tap % 5;
      static_for<0, 2>([&](auto OG) {
        constexpr int og = decltype(OG)::value;
        constexpr int f = (ic * 25 + tap) * 2 + og;
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[og][p] = mfma_wbcast<f % 16>(wc[f / 16], in[dy][p + dx], acc[og][p]);
      });
    });
  });

  // ---------------- phase 2b: border fix (lanes whose strip touches the image border) ----------------
  const int gy = y0 + ty, gx = x0 + tx * 4;
  const bool row_ok = gy < H;
  float fix[8][4];
#pragma unroll
  for (int o = 0; o < 8; ++o)
#pragma unroll
    for (int p = 0; p < 4; ++p) fix[o][p] = 0.f;
  {
    const bool mine = row_ok && gx < W && (gy == 0 || gy == H - 1 || gx == 0 || gx + 4 >= W);
    if (__any(mine)) {
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
        for (int o = 0; o < 8; ++o) {
          const float b = as_const(a.bring)[t1 * 8 + o] * c1;
#pragma unroll
          for (int p = 0; p < 4; ++p) fix[o][p] -= outp[p] ? b : 0.f;
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
          const cfloat_p w = as_const(a.wc1) + (t1 * 9 + t2) * 64;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float av[4];
#pragma unroll
            for (int p = 0; p < 4; ++p)  // tile row = ry - (y0-2), col = rx - x0 + 4 ; always inside the halo-2 tile
              av[p] = inp[p] ? tileA[i][ty + t1y + t2y][tx * 4 + p + t1x + t2x + 2] : 0.f;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
              const float wv = w[i * 8 + o];
#pragma unroll
              for (int p = 0; p < 4; ++p) fix[o][p] = fmaf(-wv, av[p], fix[o][p]);
            }
          }
        }
      }
    }
  }

  // ---------------- phase 3: s * (W_x (*) eps), eps in chunks of 8 channels through LDS ----------------
  const int nchunk = a.C / 8;
#pragma unroll 1
  for (int ch = 0; ch < nchunk; ++ch) {
    __syncthreads();  // previous LDS contents (A tile / previous chunk) are no longer read
    if (NOISE == 1) {
      // explicit noise tensor: same staging as conv_in, scaled by s
      using TR = TileRegs<TW, TH, NT, 8>;
      const int r0 = tid / TR::QPR, qx = tid % TR::QPR;
#pragma unroll
      for (int c = 0; c < 8; ++c)
        *reinterpret_cast<float4*>(&tileE[c][r0][4 + 4 * qx]) = make_float4(sg * R.v[c].x, sg * R.v[c].y, sg * R.v[c].z, sg * R.v[c].w);
      {
        const int cr = tid / (2 * TR::QPR), rr = TR::RPP + ((tid / TR::QPR) & 1);
        if (cr < 8) *reinterpret_cast<float4*>(&tileE[cr][rr][4 + 4 * qx]) = make_float4(sg * R.vr.x, sg * R.vr.y, sg * R.vr.z, sg * R.vr.w);
      }
#pragma unroll
      for (int k = 0; k < TR::HIT; ++k) {
        const int hq = tid + k * NT;
        if (hq < TR::NHALO) {
          const int row = hq >> 1, side = hq & 1, c = row / TR::LH, r = row - c * TR::LH;
          tileE[c][r][side ? TW + 4 : 3] = sg * R.hv[k];
        }
      }
    } else {
      // canonical step-noise field (common.h noise_pair_quad): one call per (channel pair, aligned quad), already scaled
      // by sigma_t and rounded to fp16 -- the field the direct sampler's conv_out epilogue adds
      auto pair_quad = [&](int cp, int r, int gxq, float (&z)[8]) {
        const int gy2 = y0 - 1 + r;
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = 0.f;
        if (gy2 >= 0 && gy2 < H && gxq >= 0 && gxq < W)
          noise_pair_quad((uint64_t)(((size_t)n * a.C + (size_t)ch * 8 + 2 * cp) * plane + (size_t)gy2 * W + gxq), a.stream_id, seed, bm_k2(sg), z);
      };
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) {
        float z[8];
        pair_quad(cp, ty, x0 + 4 * tx, z);
        *reinterpret_cast<float4*>(&tileE[2 * cp][ty][4 + 4 * tx]) = make_float4(z[0], z[1], z[2], z[3]);
        *reinterpret_cast<float4*>(&tileE[2 * cp + 1][ty][4 + 4 * tx]) = make_float4(z[4], z[5], z[6], z[7]);
      }
      {
        const int cr = tid / (2 * QPR), rr = TH + ((tid / QPR) & 1);
        if (cr < 8) {
          float z[8];
          pair_quad(cr >> 1, rr, x0 + 4 * tx, z);
          const int o = 4 * (cr & 1);
          *reinterpret_cast<float4*>(&tileE[cr][rr][4 + 4 * tx]) = make_float4(z[o], z[o + 1], z[o + 2], z[o + 3]);
        }
      }
      constexpr int NH = (8 * LH * 2 + NT - 1) / NT;
#pragma unroll
      for (int k = 0; k < NH; ++k) {
        const int hq = tid + k * NT;
        if (hq < 8 * LH * 2) {
          const int row = hq >> 1, side = hq & 1, c = row / LH, r = row - c * LH;
          float z[8];
          pair_quad(c >> 1, r, side ? x0 + TW : x0 - 4, z);  // the aligned quad that owns the halo pixel
          tileE[c][r][side ? TW + 4 : 3] = z[4 * (c & 1) + (side ? 0 : 3)];
        }
      }
    }
    float wcur[9];
#pragma unroll
    for (int g = 0; g < 9; ++g) wcur[g] = wn[g];
    if (ch + 1 < nchunk) {
      if (NOISE == 1) stage_load<TW, TH, NT, 8, false>(R, np_ + (size_t)(ch + 1) * 8 * plane, plane, W, H, W, x0, y0, tid);
      load_wregs<9>(wn, a.wx + (size_t)(ch + 1) * 576, 576, lane);
    }
    __syncthreads();
    conv_tile_mfma<8, 2, 4, LH, LS, 9>(tileE, wcur, acc, tx, ty);
  }

  // ---------------- epilogue: combine, store in place, statistics for the next GroupNorm ----------------
  const bool vec_ok = row_ok && gx + 3 < W;
  const size_t pix = (size_t)gy * W + gx;
  float part[16];
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    const size_t e = ((size_t)n * 8 + o) * plane + pix;
    float kk[4] = {0.f, 0.f, 0.f, 0.f}, hv[4] = {0.f, 0.f, 0.f, 0.f};
    if (vec_ok) {
      const float4 k4 = *reinterpret_cast<const float4*>(a.kmap + e);
      const float4 h4 = *reinterpret_cast<const float4*>(a.hs0 + e);
      kk[0] = k4.x; kk[1] = k4.y; kk[2] = k4.z; kk[3] = k4.w;
      hv[0] = h4.x; hv[1] = h4.y; hv[2] = h4.z; hv[3] = h4.w;
    }
    const float bs = c1 * as_const(a.bsum)[o];
    float v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) v[p] = kk[p] + c2 * (hv[p] - kk[p]) + (acc[o >> 2][p][o & 3] + bs + fix[o][p]);
    float s = 0.f, q = 0.f;
    if (vec_ok) {
      *reinterpret_cast<float4*>(a.hs0 + e) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int p = 0; p < 4; ++p) { s += v[p]; q = fmaf(v[p], v[p], q); }
    }
    part[o] = s;
    part[8 + o] = q;
  }
  block_stats_commit<NT>(part, s_red, a.hs0_stat + (size_t)n * 16);
}

}  // namespace gc
