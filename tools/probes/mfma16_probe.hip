// Probe for gfx950: (1) operand layout of v_mfma_f32_16x16x4_f32; (2) how much VALU work of OTHER waves (and of
// the same wave) issues beside a back-to-back stream of 16x16x4 MFMAs, compared with the 4x4x1 form.
//   hipcc --offload-arch=gfx950 -O3 mfma16_probe.hip -o mfma16_probe.bin && ./mfma16_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[l * 4 + i] = c[i];
}

#define REP 32
// kind 0: 16x16x4 MFMA stream, kind 1: 4x4x1 MFMA stream, kind 2: VALU fma stream, kind 3: VALU with 1/3 v_exp
// FILL > 0: the MFMA wave itself carries FILL v_fma per MFMA
template <int FILL>
__global__ void issue(float* out, long long* cyc, int iters, int kindA, int kindB) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float w = threadIdx.x * 0.001f, x = 1.0f + threadIdx.x * 1e-6f;
  float v0 = x, v1 = x * 2, v2 = x * 3, v3 = x * 4, v4 = x * 5, v5 = x * 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kind = ((wave >> 2) & 1) ? kindB : kindA;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  {
    if (kind == 0) {
      for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int r = 0; r < REP; r += 4) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, a0, 0, 0, 0);
        if (FILL >= 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v0) : "v"(x));
        if (FILL >= 2) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v1) : "v"(x));
        if (FILL >= 3) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v2) : "v"(x));
        if (FILL >= 4) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v3) : "v"(x));
        if (FILL >= 5) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v4) : "v"(x));
        if (FILL >= 6) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v5) : "v"(x));
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, a1, 0, 0, 0);
        if (FILL >= 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v0) : "v"(x));
        if (FILL >= 2) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v1) : "v"(x));
        if (FILL >= 3) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v2) : "v"(x));
        if (FILL >= 4) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v3) : "v"(x));
        if (FILL >= 5) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v4) : "v"(x));
        if (FILL >= 6) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v5) : "v"(x));
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, a2, 0, 0, 0);
        if (FILL >= 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v0) : "v"(x));
        if (FILL >= 2) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v1) : "v"(x));
        if (FILL >= 3) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v2) : "v"(x));
        if (FILL >= 4) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v3) : "v"(x));
        if (FILL >= 5) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v4) : "v"(x));
        if (FILL >= 6) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v5) : "v"(x));
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, a3, 0, 0, 0);
        if (FILL >= 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v0) : "v"(x));
        if (FILL >= 2) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v1) : "v"(x));
        if (FILL >= 3) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v2) : "v"(x));
        if (FILL >= 4) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v3) : "v"(x));
        if (FILL >= 5) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v4) : "v"(x));
        if (FILL >= 6) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v5) : "v"(x));
      }
    } else if (kind == 1) {
      for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int r = 0; r < REP; r += 4) {
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, a0, 4, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, a1, 4, 1, 0);
        a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, a2, 4, 2, 0);
        a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, a3, 4, 3, 0);
      }
    } else if (kind == 2) {
      for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int r = 0; r < REP; r += 4) {
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v0) : "v"(x));
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v1) : "v"(x));
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v2) : "v"(x));
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v3) : "v"(x));
      }
    } else {
      for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int r = 0; r < REP; r += 4) {
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v0) : "v"(x));
        asm volatile("v_exp_f32 %0, %0" : "+v"(v1));
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v2) : "v"(x));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(v3));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x & 1023] = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3 + v4 + v5;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

static int NBLK = 1;
template <int FILL>
void run(const char* name, int threads, int kindA, int kindB) {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 4); hipMalloc(&cyc, 16 * 8);
  const int iters = 50000;
  issue<FILL><<<NBLK, threads>>>(out, cyc, iters, kindA, kindB);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  issue<FILL><<<NBLK, threads>>>(out, cyc, iters, kindA, kindB);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long h[16];
  hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
  printf("%-58s thr %4d:", name, threads);
  for (int w = 0; w < threads / 64; w += 4) printf("  w%-2d %7.2f ns/slot", w, (double)h[w] * 10.0 / ((double)iters * REP));
  printf("  | kernel %.3f ns/slot\n", ms * 1e6 / ((double)iters * REP));
  hipFree(out); hipFree(cyc);
}

int main(int argc, char** argv) {
  if (argc > 1) NBLK = atoi(argv[1]);
  printf("blocks per launch: %d\n", NBLK);
  float ha[64], hb[64], hd[256];
  for (int i = 0; i < 64; ++i) { ha[i] = 1.0f + i; hb[i] = 100.0f + 3 * i; }
  float *a, *b, *d;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  layout<<<1, 64>>>(a, b, d);
  hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
  // hypothesis: A[m = l%16][k = l/16], B[k = l/16][n = l%16], D lane l reg i = D[m = 4*(l/16)+i][n = l%16]
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i) {
      const int m = 4 * (l / 16) + i, n = l % 16;
      float e = 0;
      for (int k = 0; k < 4; ++k) e += ha[k * 16 + m] * hb[k * 16 + n];
      if (fabsf(hd[l * 4 + i] - e) > 1e-3f * fabsf(e)) ++bad;
    }
  printf("16x16x4 layout mismatches: %d / 256\n", bad);
  // slot = one instruction of the wave's stream; the counter is the 100 MHz realtime-like clock -> ns
  run<0>("MFMA16 only, 1 wave/SIMD", 256, 0, 0);
  run<0>("MFMA4x4 only, 1 wave/SIMD", 256, 1, 1);
  run<0>("VALU only, 1 wave/SIMD", 256, 2, 2);
  run<0>("VALU+trans only, 1 wave/SIMD", 256, 3, 3);
  run<0>("MFMA16 x2 waves/SIMD", 512, 0, 0);
  run<0>("split: w0-3 MFMA16 | w4-7 VALU", 512, 0, 2);
  run<0>("split: w0-3 MFMA16 | w4-7 VALU+trans", 512, 0, 3);
  run<0>("split: w0-3 MFMA4x4 | w4-7 VALU", 512, 1, 2);
  run<0>("split 3 waves/SIMD: MFMA16 | VALU | MFMA16", 768, 0, 2);
  run<0>("split 4 waves/SIMD: MFMA16 | VALU | MFMA16 | VALU", 1024, 0, 2);
  run<2>("MFMA16 + 2 fma in-wave (slot = MFMA)", 256, 0, 0);
  run<4>("MFMA16 + 4 fma in-wave", 256, 0, 0);
  run<6>("MFMA16 + 6 fma in-wave", 256, 0, 0);
  run<4>("split: MFMA16+4fma | VALU", 512, 0, 2);
  return bad ? 1 : 0;
}
