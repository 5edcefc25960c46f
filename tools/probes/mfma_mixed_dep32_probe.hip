// Probe (gfx950, ROCm 7.2): does the stale-accumulator hazard of tools/probes/mfma_mixed_dep_probe.hip (a matrix instruction accumulating onto
// the result of a matrix instruction of ANOTHER input type, fewer than 6 wait states later, reads a stale half of it; hipcc assumes SrcC
// forwarding) also exist for the 32x32x16 forms (8 passes instead of 4)?  enh_front_h_kernel and the round-5 kernels use them.
// One inline-asm block per sequence on fixed registers (accumulator v[100:115]); operands are ones, so every result is exact:
//   A  v_mfma_f32_32x32x16_bf8_bf8 D += 16   -> N wait states -> v_mfma_f32_32x32x16_f16 D += 32     expected c + 48
//   C  v_mfma_f32_32x32x16_f16 D += 32       -> N wait states -> v_mfma_f32_32x32x16_bf8_bf8 D += 16 expected c + 48
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_mixed_dep32_probe.hip -o tools/probes/mfma_mixed_dep32_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define INIT "v_mov_b32 v100, %1\nv_add_f32 v101, 1.0, v100\nv_add_f32 v102, 1.0, v101\nv_add_f32 v103, 1.0, v102\nv_add_f32 v104, 1.0, v103\nv_add_f32 v105, 1.0, v104\nv_add_f32 v106, 1.0, v105\nv_add_f32 v107, 1.0, v106\n" \
             "v_add_f32 v108, 1.0, v107\nv_add_f32 v109, 1.0, v108\nv_add_f32 v110, 1.0, v109\nv_add_f32 v111, 1.0, v110\nv_add_f32 v112, 1.0, v111\nv_add_f32 v113, 1.0, v112\nv_add_f32 v114, 1.0, v113\nv_add_f32 v115, 1.0, v114\ns_nop 15\n"
#define MF8 "v_mfma_f32_32x32x16_bf8_bf8 v[100:115], %2, %3, v[100:115]\n"
#define MF16 "v_mfma_f32_32x32x16_f16 v[100:115], %4, %5, v[100:115]\n"
// checksum of the 16 result registers minus their expected values (c + i + 48): 0 everywhere when nothing was lost
#define FINISH "s_nop 15\ns_nop 15\ns_nop 15\nv_mov_b32 %0, 0\n" \
  "v_sub_f32 v100, v100, %1\nv_add_f32 %0, %0, v100\nv_sub_f32 v101, v101, %1\nv_add_f32 %0, %0, v101\nv_sub_f32 v102, v102, %1\nv_add_f32 %0, %0, v102\nv_sub_f32 v103, v103, %1\nv_add_f32 %0, %0, v103\n" \
  "v_sub_f32 v104, v104, %1\nv_add_f32 %0, %0, v104\nv_sub_f32 v105, v105, %1\nv_add_f32 %0, %0, v105\nv_sub_f32 v106, v106, %1\nv_add_f32 %0, %0, v106\nv_sub_f32 v107, v107, %1\nv_add_f32 %0, %0, v107\n" \
  "v_sub_f32 v108, v108, %1\nv_add_f32 %0, %0, v108\nv_sub_f32 v109, v109, %1\nv_add_f32 %0, %0, v109\nv_sub_f32 v110, v110, %1\nv_add_f32 %0, %0, v110\nv_sub_f32 v111, v111, %1\nv_add_f32 %0, %0, v111\n" \
  "v_sub_f32 v112, v112, %1\nv_add_f32 %0, %0, v112\nv_sub_f32 v113, v113, %1\nv_add_f32 %0, %0, v113\nv_sub_f32 v114, v114, %1\nv_add_f32 %0, %0, v114\nv_sub_f32 v115, v115, %1\nv_add_f32 %0, %0, v115\n"
#define CLOB : "=&v"(sum) : "v"(c), "v"(a8), "v"(b8), "v"(a16), "v"(b16) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115"
#define SEQ_A(W) asm volatile(INIT MF8 W MF16 FINISH CLOB)
#define SEQ_C(W) asm volatile(INIT MF16 W MF8 FINISH CLOB)

// expected checksum when nothing is lost: sum_i (c + i + 48 - c) = 120 + 16 * 48 = 888
__global__ void probe(float* out, int nvar) {
  const long ones8 = 0x3C3C3C3C3C3C3C3CLL;             // bf8 (e5m2) 1.0 x 8
  const long a8 = ones8, b8 = ones8;
  h8 a16, b16;
  for (int i = 0; i < 8; ++i) { a16[i] = (_Float16)2.0f; b16[i] = (_Float16)1.0f; }
  const float c = 1000.0f;
  float sum;
  float* o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * nvar;
  SEQ_A(""); o[0] = sum;
  SEQ_A("s_nop 0\n"); o[1] = sum;
  SEQ_A("s_nop 1\n"); o[2] = sum;
  SEQ_A("s_nop 3\n"); o[3] = sum;
  SEQ_A("s_nop 7\n"); o[4] = sum;
  SEQ_A("s_nop 15\n"); o[5] = sum;
  SEQ_C(""); o[6] = sum;
  SEQ_C("s_nop 0\n"); o[7] = sum;
  SEQ_C("s_nop 1\n"); o[8] = sum;
  SEQ_C("s_nop 3\n"); o[9] = sum;
  SEQ_C("s_nop 7\n"); o[10] = sum;
  SEQ_C("s_nop 15\n"); o[11] = sum;
}

int main() {
  const int nvar = 12;
  const char* names[nvar] = {"bf8 -> f16, 0 wait states", "bf8 -> f16, 1", "bf8 -> f16, 2", "bf8 -> f16, 4", "bf8 -> f16, 8", "bf8 -> f16, 16",
                             "f16 -> bf8, 0 wait states", "f16 -> bf8, 1", "f16 -> bf8, 2", "f16 -> bf8, 4", "f16 -> bf8, 8", "f16 -> bf8, 16"};
  for (int blocks : {1, 256, 1024, 3072}) {
    const size_t n = (size_t)blocks * 256;
    float* d; hipMalloc(&d, n * nvar * 4);
    hipMemset(d, 0, n * nvar * 4);
    probe<<<blocks, 256>>>(d, nvar);
    std::vector<float> h(n * nvar);
    hipMemcpy(h.data(), d, n * nvar * 4, hipMemcpyDeviceToHost);
    printf("%d workgroups of 4 waves (expected checksum 888 in every lane):\n", blocks);
    for (int v = 0; v < nvar; ++v) {
      size_t bad = 0; float ex = 0;
      for (size_t i = 0; i < n; ++i) if (h[i * nvar + v] != 888.0f) { ++bad; ex = h[i * nvar + v]; }
      printf("  %-28s wrong lanes %zu of %zu%s\n", names[v], bad, n, bad ? (std::string("  (e.g. ") + std::to_string(ex) + ")").c_str() : "");
    }
    hipFree(d);
  }
  return 0;
}
