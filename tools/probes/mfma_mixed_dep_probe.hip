// Probe (gfx950, ROCm 7.2): stale accumulator reads behind v_mfma_f32_16x16x32_* that hipcc's hazard model does not cover.
//
// Every sequence is ONE inline-asm block on fixed registers (accumulator v[100:103]), so nothing is scheduled in between:
//   A  v_mfma_f32_16x16x32_bf8_bf8 D, a8, b8, D   -> N wait states -> v_mfma_f32_16x16x32_f16 D, a16, b16, D   (dependent, other input type)
//   B  v_mfma_f32_16x16x32_f16 D, .., D           -> 0             -> v_mfma_f32_16x16x32_f16 D, .., D         (dependent, same type)
//   C  v_mfma_f32_16x16x32_f16 D, .., D           -> 0             -> v_mfma_f32_16x16x32_bf8_bf8 D, .., D
//   D  v_mfma_f32_16x16x32_f16 D, .., D           -> N wait states -> v_pk_fma_f32 E, D[2:3], S, E   (VALU read of the LAST result registers;
//      hipcc puts `s_nop 7` = 8 wait states here.  Round 2's conv8h_kernel epilogue lost its bias behind this pair, lanes 48..63)
//   E  the same with v_fma_f32 reading D[3]
//   F  round 2's exact reader: v_pk_fma_f32 D[0:1], D[0:1], S, E op_sel:[0,0,1] (in place, SGPR-pair source)
// Operands are small integers: every result is exact; the reference is the same sequence with 32 wait states everywhere.
// Grid: 1 workgroup alone, then 1 / 4 / 12 waves per SIMD-equivalent (256 / 1024 / 3072 workgroups of 4 waves).
//   hipcc --offload-arch=gfx950 -O3 mfma_mixed_dep_probe.hip -o mfma_mixed_dep_probe.bin && ./mfma_mixed_dep_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define LOAD_D "v_mov_b32 v100, %4\nv_mov_b32 v101, %5\nv_mov_b32 v102, %6\nv_mov_b32 v103, %7\nv_mov_b32 v104, 1.0\nv_mov_b32 v105, 1.0\ns_nop 15\n"
#define STORE_D "s_nop 15\ns_nop 15\nv_mov_b32 %0, v100\nv_mov_b32 %1, v101\nv_mov_b32 %2, v102\nv_mov_b32 %3, v103\n"
#define STORE_E "s_nop 15\ns_nop 15\nv_mov_b32 %0, v104\nv_mov_b32 %1, v105\nv_mov_b32 %2, v102\nv_mov_b32 %3, v103\n"
#define MF_BF8 "v_mfma_f32_16x16x32_bf8_bf8 v[100:103], %8, %9, v[100:103]\n"
#define MF_F16 "v_mfma_f32_16x16x32_f16 v[100:103], %10, %11, v[100:103]\n"
#define MF_F16B "v_mfma_f32_16x16x32_f16 v[100:103], %12, %13, v[100:103]\n"
#define RD_PK "v_pk_fma_f32 v[104:105], v[102:103], %14, v[104:105]\n"
#define RD_FMA "v_fma_f32 v104, v103, 2.0, v104\nv_fma_f32 v105, v102, 2.0, v105\n"
// round 2's exact form: in place on the FIRST result pair, SGPR-pair multiplier, op_sel on the addend, then copied out through v[104:105]
#define RD_PK_R2 "v_pk_fma_f32 v[100:101], v[100:101], %14, v[104:105] op_sel:[0,0,1]\nv_mov_b32 v104, v100\nv_mov_b32 v105, v101\n"
#define OPS : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(a8), "v"(b8), "v"(a16), "v"(b16), "v"(a16b), "v"(b16b), "s"(s2) \
            : "v100", "v101", "v102", "v103", "v104", "v105"
#define W0 ""
#define W1 "s_nop 0\n"
#define W2 "s_nop 1\n"
#define W4 "s_nop 3\n"
#define W6 "s_nop 5\n"
#define W8 "s_nop 7\n"
#define W10 "s_nop 9\n"
#define W12 "s_nop 11\n"
#define W16 "s_nop 15\n"
#define W32 "s_nop 15\ns_nop 15\n"

__device__ __forceinline__ long bf8_pattern(int lane, int salt) {  // e5m2 bytes: 1.0 = 0x3C, 2.0 = 0x40, -1.0 = 0xBC, 0
  const unsigned char tab[4] = {0x3C, 0x40, 0xBC, 0x00};
  unsigned long v = 0;
  for (int i = 0; i < 8; ++i) v |= (unsigned long)tab[(lane * 3 + i + salt) & 3] << (8 * i);
  return (long)v;
}

template <int SEQ>
__global__ __launch_bounds__(256) void probe(float* out, int rounds) {
  const int lane = threadIdx.x & 63;
  const long a8 = bf8_pattern(lane, 1), b8 = bf8_pattern(lane, 2);
  h8 a16, b16, a16b, b16b;
  for (int i = 0; i < 8; ++i) {
    a16[i] = (_Float16)(float)(((lane * 5 + i * 3) % 5) - 2); b16[i] = (_Float16)(float)(((lane * 5 + i * 3 + 1) % 5) - 2);
    a16b[i] = (_Float16)(float)(((lane * 5 + i * 3 + 2) % 5) - 2); b16b[i] = (_Float16)(float)(((lane * 5 + i * 3 + 3) % 5) - 2);
  }
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 s2 = {2.f, 2.f};
  float bad[4] = {0, 0, 0, 0};
  for (int it = 0; it < rounds; ++it) {
    float c[4], r[4], q[4];
    for (int i = 0; i < 4; ++i) c[i] = (float)(1000 + 7 * lane + it + i);  // the start value: what a stale read loses
#define RUN(body, store) asm volatile(LOAD_D body store OPS)
    if (SEQ < 100) {  // reference of sequences A / B / C
      if (SEQ < 10) RUN(MF_BF8 W32 MF_F16, STORE_D);
      else if (SEQ == 10) RUN(MF_F16B W32 MF_F16, STORE_D);
      else RUN(MF_F16 W32 MF_BF8, STORE_D);
    } else {
      if (SEQ < 200) RUN(MF_F16 W32 RD_PK, STORE_E);
      else if (SEQ < 300) RUN(MF_F16 W32 RD_FMA, STORE_E);
      else RUN(MF_F16 W32 RD_PK_R2, STORE_E);
    }
    for (int i = 0; i < 4; ++i) q[i] = r[i];
    if (SEQ == 0) RUN(MF_BF8 W0 MF_F16, STORE_D);
    if (SEQ == 1) RUN(MF_BF8 W1 MF_F16, STORE_D);
    if (SEQ == 2) RUN(MF_BF8 W2 MF_F16, STORE_D);
    if (SEQ == 4) RUN(MF_BF8 W4 MF_F16, STORE_D);
    if (SEQ == 6) RUN(MF_BF8 W6 MF_F16, STORE_D);
    if (SEQ == 8) RUN(MF_BF8 W8 MF_F16, STORE_D);
    if (SEQ == 10) RUN(MF_F16B W0 MF_F16, STORE_D);
    if (SEQ == 20) RUN(MF_F16 W0 MF_BF8, STORE_D);
    if (SEQ == 104) RUN(MF_F16 W4 RD_PK, STORE_E);
    if (SEQ == 106) RUN(MF_F16 W6 RD_PK, STORE_E);
    if (SEQ == 108) RUN(MF_F16 W8 RD_PK, STORE_E);
    if (SEQ == 110) RUN(MF_F16 W10 RD_PK, STORE_E);
    if (SEQ == 112) RUN(MF_F16 W12 RD_PK, STORE_E);
    if (SEQ == 204) RUN(MF_F16 W4 RD_FMA, STORE_E);
    if (SEQ == 208) RUN(MF_F16 W8 RD_FMA, STORE_E);
    if (SEQ == 304) RUN(MF_F16 W4 RD_PK_R2, STORE_E);
    if (SEQ == 306) RUN(MF_F16 W6 RD_PK_R2, STORE_E);
    if (SEQ == 308) RUN(MF_F16 W8 RD_PK_R2, STORE_E);
    if (SEQ == 310) RUN(MF_F16 W10 RD_PK_R2, STORE_E);
    for (int i = 0; i < 4; ++i) bad[i] += (r[i] != q[i]) ? 1.f : 0.f;
  }
  float* o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  for (int i = 0; i < 4; ++i) o[i] = bad[i];
}

template <int SEQ>
void run(const char* what, int blocks, int rounds) {
  float* out;
  const size_t n = (size_t)blocks * 256 * 4;
  hipMalloc(&out, n * 4);
  probe<SEQ><<<blocks, 256>>>(out, rounds);
  hipDeviceSynchronize();
  std::vector<float> h(n);
  hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
  double be[4] = {0, 0, 0, 0}, bg[4] = {0, 0, 0, 0}, total = (double)blocks * 256 * rounds;
  for (size_t t = 0; t < (size_t)blocks * 256; ++t)
    for (int i = 0; i < 4; ++i) { be[i] += h[t * 4 + i]; bg[(t & 63) >> 4] += h[t * 4 + i]; }
  printf("%-62s wgs %5d: wrong per output register [%.0f %.0f %.0f %.0f] of %.0f each; by lane group 0-15|16-31|32-47|48-63 [%.0f %.0f %.0f %.0f]\n",
         what, blocks, be[0], be[1], be[2], be[3], total, bg[0], bg[1], bg[2], bg[3]);
  hipFree(out);
}

int main() {
  const int rounds = 1000;
  for (int blocks : {1, 256, 1024, 3072}) {
    run<0>("A bf8 mfma -> dependent f16 mfma, back to back", blocks, rounds);
    run<1>("A bf8 -> 1 wait state -> f16", blocks, rounds);
    run<2>("A bf8 -> 2 wait states -> f16", blocks, rounds);
    run<4>("A bf8 -> 4 wait states -> f16", blocks, rounds);
    run<6>("A bf8 -> 6 wait states -> f16", blocks, rounds);
    run<8>("A bf8 -> 8 wait states -> f16", blocks, rounds);
    run<10>("B f16 mfma -> dependent f16 mfma, back to back", blocks, rounds);
    run<20>("C f16 mfma -> dependent bf8 mfma, back to back", blocks, rounds);
    run<104>("D f16 mfma -> 4 wait states -> v_pk_fma_f32 on D[2:3]", blocks, rounds);
    run<106>("D f16 mfma -> 6 wait states -> v_pk_fma_f32", blocks, rounds);
    run<108>("D f16 mfma -> 8 wait states (hipcc: s_nop 7) -> v_pk_fma_f32", blocks, rounds);
    run<110>("D f16 mfma -> 10 wait states -> v_pk_fma_f32", blocks, rounds);
    run<112>("D f16 mfma -> 12 wait states -> v_pk_fma_f32", blocks, rounds);
    run<204>("E f16 mfma -> 4 wait states -> v_fma_f32 on D[3], D[2]", blocks, rounds);
    run<208>("E f16 mfma -> 8 wait states -> v_fma_f32", blocks, rounds);
    run<304>("F f16 mfma -> 4 wait states -> v_pk_fma_f32 in place, sgpr pair, op_sel", blocks, rounds);
    run<306>("F f16 mfma -> 6 wait states -> the same", blocks, rounds);
    run<308>("F f16 mfma -> 8 wait states (hipcc) -> the same", blocks, rounds);
    run<310>("F f16 mfma -> 10 wait states -> the same", blocks, rounds);
  }
  return 0;
}
