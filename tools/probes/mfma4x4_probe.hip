// Probe of v_mfma_f32_4x4x1_16b_f32 operand layout and the CBSZ/ABID A-broadcast on gfx950.
// Hypothesis: lane l = 4*block + idx. A: lane holds A_block[i = idx]; B: lane holds B_block[j = idx];
// D: lane holds D_block[i = reg][j = idx]. With cbsz = 4, abid = q every block uses A of block q.
// Build & run on the GPU box: hipcc --offload-arch=gfx950 -O2 mfma4x4_probe.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ void probe(const float* a, const float* b, float* d_plain, float* d_bcast5) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  f32x4 d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
  f32x4 d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 4, 5, 0);
  for (int r = 0; r < 4; ++r) { d_plain[l * 4 + r] = d0[r]; d_bcast5[l * 4 + r] = d1[r]; }
}

int main() {
  float ha[64], hb[64], hd0[256], hd1[256];
  for (int i = 0; i < 64; ++i) { ha[i] = 1.0f + i; hb[i] = 100.0f + 3 * i; }
  float *a, *b, *d0, *d1;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d0, 1024); hipMalloc(&d1, 1024);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(a, b, d0, d1);
  hipMemcpy(hd0, d0, 1024, hipMemcpyDeviceToHost); hipMemcpy(hd1, d1, 1024, hipMemcpyDeviceToHost);
  int bad0 = 0, bad1 = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int blk = l / 4, j = l % 4;
      const float e0 = ha[4 * blk + r] * hb[4 * blk + j];   // D_blk[i=r][j] = A_blk[r] * B_blk[j]
      const float e1 = ha[4 * 5 + r] * hb[4 * blk + j];     // A broadcast from block 5
      if (hd0[l * 4 + r] != e0) ++bad0;
      if (hd1[l * 4 + r] != e1) ++bad1;
    }
  printf("plain layout mismatches: %d / 256\nbroadcast(cbsz=4,abid=5) mismatches: %d / 256\n", bad0, bad1);
  if (bad0 || bad1) {
    printf("lane 9: plain %g %g %g %g | bcast %g %g %g %g\n", hd0[36], hd0[37], hd0[38], hd0[39], hd1[36], hd1[37], hd1[38], hd1[39]);
    printf("a[8..11]=%g %g %g %g b[9]=%g a[20..23]=%g..%g\n", ha[8], ha[9], ha[10], ha[11], hb[9], ha[20], ha[23]);
  }
  return (bad0 || bad1) ? 1 : 0;
}
