// Probe (gfx950): HBM-side efficiency of the 8-channel-map tile access pattern, planar NCHW vs pixel-major NHWC8.
// Each workgroup (256 threads) reads the 18 x 64 (+2 halo columns) x 8-channel region of a 64x16 tile and writes the
// 16 x 64 x 8 interior, exactly the bytes conv8h_kernel moves for a plain layer, with no arithmetic in between.
//   hipcc --offload-arch=gfx950 -O3 layout_probe.hip -o layout_probe.bin && ./layout_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <bool NHWC>
__global__ __launch_bounds__(256) void tile_copy(const float* __restrict__ src, float* __restrict__ dst, int H, int W) {
  const int tid = threadIdx.x, n = blockIdx.z;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 16;
  const size_t plane = (size_t)H * W;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v[8], vr;
  const int r0 = tid >> 4, qx = tid & 15;
  const int gy = y0 - 1 + r0, gx = x0 + 4 * qx;
  const bool ok = gy >= 0 && gy < H && gx < W;
  if (!NHWC) {
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = ok ? *reinterpret_cast<const float4*>(src + ((size_t)n * 8 + c) * plane + (size_t)gy * W + gx) : acc;
    const int cr = tid >> 5, rr = 16 + ((tid >> 4) & 1), gy2 = y0 - 1 + rr;
    vr = (gy2 < H && gx < W) ? *reinterpret_cast<const float4*>(src + ((size_t)n * 8 + cr) * plane + (size_t)gy2 * W + gx) : acc;
  } else {
    // (row, pixel -1..64, channel half) items, consecutive threads = consecutive 16-B pieces of a row: 1 KB per wave load
    for (int k = 0; k < 10; ++k) {
      const int item = tid + 256 * k;
      const int row = item / 132, rem = item - row * 132, px = (rem >> 1) - 1, hf = rem & 1;
      const int gy2 = y0 - 1 + row, gx2 = x0 + px;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < 18 && gy2 >= 0 && gy2 < H && gx2 >= 0 && gx2 < W) t = *reinterpret_cast<const float4*>(src + ((size_t)n * plane + (size_t)gy2 * W + gx2) * 8 + 4 * hf);
      if (k < 8) v[k] = t; else { acc.x += t.x; acc.y += t.w; }
    }
    vr = acc;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) { acc.x += v[c].x; acc.y += v[c].y; acc.z += v[c].z; acc.w += v[c].w; }
  acc.x += vr.x;
  // output: thread (lane&15 = quad, lane>>4 -> channel half / row) as the MFMA fragment owns it
  const int lane = tid & 63, wave = tid >> 6, ln = lane & 15, g = lane >> 4, ch = g & 1, rr2 = g >> 1;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int oy = y0 + 4 * wave + 2 * p + rr2, ox = x0 + 4 * ln;
    if (oy >= H || ox >= W) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!NHWC) *reinterpret_cast<float4*>(dst + ((size_t)n * 8 + 4 * ch + i) * plane + (size_t)oy * W + ox) = acc;
      else if (x0 + 16 * i + ln < W) *reinterpret_cast<float4*>(dst + ((size_t)n * plane + (size_t)oy * W + x0 + 16 * i + ln) * 8 + 4 * ch) = acc;
    }
  }
}

int main() {
  const int n = 16, H = 200, W = 704;
  const size_t bytes = (size_t)n * 8 * H * W * 4;
  float *a, *b, *c, *d;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&d, bytes);
  hipMemset(a, 0, bytes); hipMemset(c, 0, bytes);
  const dim3 grid((W + 63) / 64, (H + 15) / 16, n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) {
        // alternate buffers so that consecutive launches do not hit the same lines (4 x 72 MB in rotation)
        float* s = (i & 1) ? c : a; float* t = (i & 1) ? d : b;
        if (mode == 0) tile_copy<false><<<grid, 256>>>(s, t, H, W);
        else tile_copy<true><<<grid, 256>>>(s, t, H, W);
      }
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      printf("%s: %.1f us per launch, %.2f TB/s (read 72 MB + halo, write 72 MB)\n", mode ? "pixel-major NHWC8" : "planar NCHW     ",
             ms * 1e3 / 20, 2.0 * bytes / (ms * 1e-3 / 20) / 1e12);
    }
  }
  return 0;
}
