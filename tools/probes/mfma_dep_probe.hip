// Probe (gfx950): are dependent v_mfma_f32_16x16x32_f16 chains correct when two waves share a SIMD's matrix pipe?
// Each wave runs R rounds over D accumulators (round-robin, so instructions that accumulate into the same registers
// are D apart); operands are small integers, so every partial sum is exact and the expected result is known.
//   hipcc --offload-arch=gfx950 -O3 mfma_dep_probe.hip -o mfma_dep_probe.bin && ./mfma_dep_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(1024) void chains(float* out, int rounds, int use_lds) {
  __shared__ __align__(16) _Float16 lds[64 * 8 * 16];
  const int lane = threadIdx.x & 63;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(float)((lane + i) % 3 - 1); b[i] = (_Float16)(float)((lane * 7 + i) % 5 - 2); }
  for (int k = 0; k < 16; ++k) *reinterpret_cast<h8*>(&lds[(k * 64 + lane) * 8]) = b;
  __syncthreads();
  f4 acc[D];
  for (int d = 0; d < D; ++d) acc[d] = f4{0, 0, 0, 0};
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      h8 bb = use_lds ? *reinterpret_cast<const h8*>(&lds[(((r + d) & 15) * 64 + lane) * 8]) : b;
      acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bb, acc[d], 0, 0, 0);
    }
  }
  float* o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4 * D;
  for (int d = 0; d < D; ++d) for (int i = 0; i < 4; ++i) o[d * 4 + i] = acc[d][i];
}

template <int D>
void run(int threads, int blocks, int rounds, int use_lds) {
  float* out;
  const size_t n = (size_t)blocks * threads * 4 * D;
  hipMalloc(&out, n * 4);
  chains<D><<<1, 64>>>(out, rounds, use_lds);  // reference: one wave alone
  float* ref = (float*)malloc(64 * 4 * D * 4);
  hipMemcpy(ref, out, 64 * 4 * D * 4, hipMemcpyDeviceToHost);
  long bad = 0, badrow[4] = {0, 0, 0, 0};
  for (int rep = 0; rep < 5; ++rep) {
    chains<D><<<blocks, threads>>>(out, rounds, use_lds);
    float* h = (float*)malloc(n * 4);
    hipMemcpy(h, out, n * 4, hipMemcpyDeviceToHost);
    for (size_t t = 0; t < (size_t)blocks * threads; ++t)
      for (int e = 0; e < 4 * D; ++e)
        if (h[t * 4 * D + e] != ref[(t & 63) * 4 * D + e]) { ++bad; ++badrow[(t & 63) >> 4]; }
    free(h);
  }
  printf("D %2d threads %4d blocks %4d lds %d: mismatching elements %ld (by lane group %ld %ld %ld %ld)\n", D, threads, blocks, use_lds, bad,
         badrow[0], badrow[1], badrow[2], badrow[3]);
  free(ref);
  hipFree(out);
}

int main() {
  for (int lds = 0; lds < 2; ++lds) {
    run<1>(256, 512, 2000, lds); run<1>(512, 512, 2000, lds); run<1>(1024, 512, 2000, lds);
    run<2>(512, 512, 2000, lds); run<4>(512, 512, 2000, lds); run<4>(1024, 512, 2000, lds);
    run<8>(512, 512, 1000, lds); run<8>(1024, 512, 1000, lds); run<16>(512, 512, 500, lds);
  }
  return 0;
}
