// PointPillars front half, eval mode: PillarVFE (10-feature augmentation, Linear
// 10 -> 64 without bias, BatchNorm1d(eps 1e-3) folded to scale/shift, ReLU, max over the 32 point
// slots -- padded slots take part with masked (zero) features, as in the reference) fused with
// PointPillarScatter (index = z + y*nx + x; bit-exact integer indexing).
// Reference: opencood/models/sub_modules/pillar_vfe.py:105-155, :31-54; point_pillar_scatter.py:42-76.
// HBM-bound: 512 B in, 256 B out per pillar; one wave per pillar, lane = output channel.
#pragma once
#include "common.h"

namespace gc {

struct PillarArgs {
  const float* vf;      // [M][P][4]
  const int* npts;      // [M]
  const int* coords;    // [M][4] (b, z, y, x)
  const float* w;       // [64][10]
  const float* scale;   // [64]  gamma / sqrt(running_var + eps)
  const float* shift;   // [64]  beta - running_mean * scale
  float* out;           // [B][64][ny][nx], pre-zeroed
  int M, P, B, nx, ny;
  float vx, vy, vz, xo, yo, zo;
};

__global__ __launch_bounds__(256) void pillar_vfe_scatter_kernel(const PillarArgs a) {
  __shared__ float feat[4][32][10];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + wv;
  const bool live = m < a.M;
  int b = 0, cz = 0, cy = 0, cx = 0, np = 1;
  if (live) {
    b = a.coords[m * 4 + 0]; cz = a.coords[m * 4 + 1]; cy = a.coords[m * 4 + 2]; cx = a.coords[m * 4 + 3];
    np = a.npts[m];
  }
  // lanes 0..P-1 own one point slot each (P <= 32)
  float px = 0.f, py = 0.f, pz = 0.f, pi = 0.f;
  if (live && lane < a.P) {
    const float4 v = *reinterpret_cast<const float4*>(a.vf + ((size_t)m * a.P + lane) * 4);
    px = v.x; py = v.y; pz = v.z; pi = v.w;
  }
  // points_mean = sum over ALL slots / num_points (pillar_vfe.py:120-122); lanes >= P hold zeros
  const float inv = 1.0f / (float)np;
  const float mx = wave_sum(px) * inv, my = wave_sum(py) * inv, mz = wave_sum(pz) * inv;
  if (lane < a.P && lane < 32) {
    const float msk = lane < np ? 1.f : 0.f;
    float* f = feat[wv][lane];
    f[0] = px * msk; f[1] = py * msk; f[2] = pz * msk; f[3] = pi * msk;
    f[4] = (px - mx) * msk; f[5] = (py - my) * msk; f[6] = (pz - mz) * msk;
    f[7] = (px - ((float)cx * a.vx + a.xo)) * msk;
    f[8] = (py - ((float)cy * a.vy + a.yo)) * msk;
    f[9] = (pz - ((float)cz * a.vz + a.zo)) * msk;
  }
  __syncthreads();
  float wr[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) wr[k] = a.w[lane * 10 + k];
  const float sc = a.scale[lane], sh = a.shift[lane];
  float best = -INFINITY;
  for (int p = 0; p < a.P; ++p) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 10; ++k) s = fmaf(wr[k], feat[wv][p][k], s);
    best = fmaxf(best, fmaxf(fmaf(s, sc, sh), 0.f));
  }
  if (live && b >= 0 && b < a.B && cy >= 0 && cy < a.ny) {
    const long long idx = (long long)cz + (long long)cy * a.nx + cx;  // the reference's index expression
    if (idx >= 0 && idx < (long long)a.nx * a.ny)
      a.out[((size_t)b * 64 + lane) * ((size_t)a.nx * a.ny) + (size_t)idx] = best;
  }
}

__global__ void bn_fold_kernel(const float* __restrict__ g, const float* __restrict__ bt, const float* __restrict__ mean,
                               const float* __restrict__ var, float eps, float* __restrict__ scale, float* __restrict__ shift, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float s = g[i] / sqrtf(var[i] + eps);
    scale[i] = s;
    shift[i] = bt[i] - mean[i] * s;
  }
}

}  // namespace gc
