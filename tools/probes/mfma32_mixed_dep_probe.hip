// Probe (gfx950, ROCm 7.2), companion of mfma_mixed_dep_probe.hip for the 32x32x16 forms: how many wait states does
//   v_mfma_f32_32x32x16_bf8_bf8 D, a8, b8, D  ->  N wait states  ->  v_mfma_f32_32x32x16_f16 D, a16, b16, D
// need before the second instruction sees the first one's result (hipcc inserts none: it assumes SrcC forwarding)?  And the reverse
// order.  One inline-asm block per case on fixed registers D = v[110:125]; exact integer operands; reference = 64 wait states.
//   hipcc --offload-arch=gfx950 -O3 mfma32_mixed_dep_probe.hip -o mfma32_probe.bin && ./mfma32_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define LOAD "v_mov_b32 v110, %1\nv_mov_b32 v111, %1\nv_mov_b32 v112, %1\nv_mov_b32 v113, %1\nv_mov_b32 v114, %1\nv_mov_b32 v115, %1\nv_mov_b32 v116, %1\n" \
             "v_mov_b32 v117, %1\nv_mov_b32 v118, %1\nv_mov_b32 v119, %1\nv_mov_b32 v120, %1\nv_mov_b32 v121, %1\nv_mov_b32 v122, %1\nv_mov_b32 v123, %1\n" \
             "v_mov_b32 v124, %1\nv_mov_b32 v125, %1\ns_nop 15\ns_nop 15\n"
#define T8 "v_mfma_f32_32x32x16_bf8_bf8 v[110:125], %2, %3, v[110:125]\n"
#define F16 "v_mfma_f32_32x32x16_f16 v[110:125], %4, %5, v[110:125]\n"
// checksum of the 16 result registers (integers: exact)
#define SUM "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\nv_add_f32 v110, v110, v111\nv_add_f32 v112, v112, v113\nv_add_f32 v114, v114, v115\nv_add_f32 v116, v116, v117\n" \
            "v_add_f32 v118, v118, v119\nv_add_f32 v120, v120, v121\nv_add_f32 v122, v122, v123\nv_add_f32 v124, v124, v125\nv_add_f32 v110, v110, v112\n" \
            "v_add_f32 v114, v114, v116\nv_add_f32 v118, v118, v120\nv_add_f32 v122, v122, v124\nv_add_f32 v110, v110, v114\nv_add_f32 v118, v118, v122\n" \
            "v_add_f32 %0, v110, v118\n"
#define OPS : "=&v"(r) : "v"(c), "v"(a8), "v"(b8), "v"(a16), "v"(b16) : "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", \
            "v120", "v121", "v122", "v123", "v124", "v125"
#define W64 "s_nop 15\ns_nop 15\ns_nop 15\ns_nop 15\n"
#define CASE(id, first, wait, second) \
  if (SEQ == id) { asm volatile(LOAD first W64 second SUM OPS); q = r; asm volatile(LOAD first wait second SUM OPS); }

template <int SEQ>
__global__ __launch_bounds__(256) void probe(float* out, int rounds) {
  const int lane = threadIdx.x & 63;
  const unsigned char tab[4] = {0x3C, 0x40, 0xBC, 0x00};  // e5m2: 1, 2, -1, 0
  unsigned long u8a = 0, u8b = 0;
  for (int i = 0; i < 8; ++i) { u8a |= (unsigned long)tab[(lane * 3 + i + 1) & 3] << (8 * i); u8b |= (unsigned long)tab[(lane * 3 + i + 2) & 3] << (8 * i); }
  const long a8 = (long)u8a, b8 = (long)u8b;
  h8 a16, b16;
  for (int i = 0; i < 8; ++i) { a16[i] = (_Float16)(float)(((lane * 5 + i * 3) % 5) - 2); b16[i] = (_Float16)(float)(((lane * 5 + i * 3 + 1) % 5) - 2); }
  float bad = 0.f;
  for (int it = 0; it < rounds; ++it) {
    const float c = (float)(1000 + 7 * lane + it);
    float r = 0.f, q = 0.f;
    CASE(0, T8, "", F16) CASE(1, T8, "s_nop 3\n", F16) CASE(2, T8, "s_nop 7\n", F16) CASE(3, T8, "s_nop 11\n", F16) CASE(4, T8, "s_nop 15\n", F16)
    CASE(5, T8, "s_nop 15\ns_nop 3\n", F16) CASE(6, T8, "s_nop 15\ns_nop 7\n", F16)
    CASE(10, F16, "", T8) CASE(11, F16, "s_nop 7\n", T8) CASE(12, F16, "s_nop 15\n", T8) CASE(13, F16, "s_nop 15\ns_nop 7\n", T8)
    CASE(20, F16, "", F16)
    if (SEQ == 30) { asm volatile(LOAD T8 W64 F16 SUM OPS); q = r; asm volatile(LOAD "" W64 F16 SUM OPS); }  // control: the bf8 product missing
    bad += (r != q) ? 1.f : 0.f;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = bad;
}

template <int SEQ>
void run(const char* what, int blocks, int rounds) {
  float* out;
  const size_t n = (size_t)blocks * 256;
  (void)hipMalloc(&out, n * 4);
  probe<SEQ><<<blocks, 256>>>(out, rounds);
  (void)hipDeviceSynchronize();
  std::vector<float> h(n);
  (void)hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
  double bad = 0;
  for (size_t t = 0; t < n; ++t) bad += h[t];
  printf("%-64s wgs %5d: wrong checksums %.0f of %.0f\n", what, blocks, bad, (double)n * rounds);
  (void)hipFree(out);
}

int main() {
  for (int blocks : {1, 256, 3072}) {
    run<0>("32x32x16 bf8 -> dependent f16, back to back", blocks, 500);
    run<1>("bf8 ->  4 wait states -> f16", blocks, 500);
    run<2>("bf8 ->  8 wait states -> f16", blocks, 500);
    run<3>("bf8 -> 12 wait states -> f16", blocks, 500);
    run<4>("bf8 -> 16 wait states -> f16", blocks, 500);
    run<5>("bf8 -> 20 wait states -> f16", blocks, 500);
    run<6>("bf8 -> 24 wait states -> f16", blocks, 500);
    run<10>("32x32x16 f16 -> dependent bf8, back to back", blocks, 500);
    run<11>("f16 ->  8 wait states -> bf8", blocks, 500);
    run<12>("f16 -> 16 wait states -> bf8", blocks, 500);
    run<13>("f16 -> 24 wait states -> bf8", blocks, 500);
    run<20>("32x32x16 f16 -> dependent f16, back to back (same type)", blocks, 500);
    run<30>("control: checksum without the bf8 instruction (must differ)", blocks, 500);
  }
  return 0;
}
