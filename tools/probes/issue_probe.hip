// Issue-port probe for gfx950: do v_mfma_f32_4x4x1 and ordinary VALU overlap (a) inside one wave,
// (b) across waves of one SIMD?   hipcc --offload-arch=gfx950 -O3 issue_probe.hip -o issue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP 64
// MODE 0: MFMA only; 1: VALU only (same count); 2: interleaved MFMA+VALU in one wave; 3: MFMA + 2 VALU
template <int MODE>
__global__ void probe(float* out, long long* cyc, int iters, int split) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float w = threadIdx.x * 0.001f, x = 1.0f + threadIdx.x * 1e-6f;
  float v0 = x, v1 = x * 2, v2 = x * 3, v3 = x * 4;
  const int wave = threadIdx.x >> 6;
  // split: waves with (wave / 4) odd run the VALU-only body, the others the MFMA-only body (MODE ignored)
  int mode = MODE;
  if (split) mode = ((wave >> 2) & 1) ? 1 : 0;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (mode == 0 || mode == 2 || mode == 3) {
#pragma unroll
      for (int r = 0; r < REP; r += 4) {
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, a0, 4, 0, 0);
        if (mode >= 2) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v0) : "v"(x)); }
        if (mode == 3) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v2) : "v"(x)); }
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, a1, 4, 1, 0);
        if (mode >= 2) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v1) : "v"(x)); }
        if (mode == 3) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v3) : "v"(x)); }
        a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, a2, 4, 2, 0);
        if (mode >= 2) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v2) : "v"(x)); }
        if (mode == 3) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v0) : "v"(x)); }
        a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, a3, 4, 3, 0);
        if (mode >= 2) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v3) : "v"(x)); }
        if (mode == 3) { asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v1) : "v"(x)); }
      }
    } else {
#pragma unroll
      for (int r = 0; r < REP; r += 4) {
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v0) : "v"(x));
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v1) : "v"(x));
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v2) : "v"(x));
        asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v3) : "v"(x));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads, int split) {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 4); hipMalloc(&cyc, 16 * 8);
  const int iters = 100000;
  probe<MODE><<<1, threads>>>(out, cyc, iters, split);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  probe<MODE><<<1, threads>>>(out, cyc, iters, split);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long h[16];
  hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
  printf("%-52s threads %4d:", name, threads);
  for (int w = 0; w < threads / 64; w += 4) printf("  wave%-2d %6.2f clk/slot", w, (double)h[w] / ((double)iters * REP));
  printf("   | kernel %.3f ms = %.3f ns/slot\n", ms, ms * 1e6 / ((double)iters * REP));
}
int main() {
  // clk is the shader clock counter (s_memtime-like, may tick at a fixed 100 MHz): compare RATIOS
  run<0>("MFMA only (1 wave/SIMD)", 256, 0);
  run<1>("VALU only (1 wave/SIMD)", 256, 0);
  run<2>("MFMA+1 VALU interleaved, one wave", 256, 0);
  run<3>("MFMA+2 VALU interleaved, one wave", 256, 0);
  run<0>("MFMA only (2 waves/SIMD)", 512, 0);
  run<1>("VALU only (2 waves/SIMD)", 512, 0);
  run<0>("split: waves 0-3 MFMA, waves 4-7 VALU (same SIMDs)", 512, 1);
  run<0>("split, 4 waves/SIMD (2 MFMA + 2 VALU)", 1024, 1);
  return 0;
}
