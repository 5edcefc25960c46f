// Reproducer (gfx950, ROCm 7.2) of round 2's conv8h_kernel epilogue fault: v_pk_fma_f32 reading a just-written
// v_mfma_f32_16x16x32_f16 result loses the MFMA's contribution sporadically, lanes 48..63, low register of the pair, only when
// several workgroups share a SIMD -- although hipcc's `s_nop 7` (8 wait states) stands between the two instructions.
// Each case is ONE inline-asm block on fixed registers (D = v[100:103]); the reference is the same block with 32 wait states.
//   R  round 2's form: v_pk_fma_f32 D[0:1], D[0:1], s[n:n+1], v[104:105] op_sel:[0,0,1]    (in place, SGPR pair, op_sel)
//   P  the same without op_sel          Q  not in place (dst v[106:107])          V  VGPR-pair multiplier instead of the SGPR pair
//   N  form R with no MFMA in front     O  form R behind an MFMA that writes other registers
//   W X Y Z  VGPR-only forms without op_sel (pk_mul in place, pk_fma, pk_add, pk_fma in place)
//   hipcc --offload-arch=gfx950 -O3 pk_fma_after_mfma_probe.hip -o pk_fma_probe.bin && ./pk_fma_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define LOAD "v_mov_b32 v100, %4\nv_mov_b32 v101, %5\nv_mov_b32 v102, %6\nv_mov_b32 v103, %7\nv_mov_b32 v104, 1.0\nv_mov_b32 v105, 3.0\n" \
             "v_mov_b32 v108, 2.0\nv_mov_b32 v109, 2.0\ns_nop 15\n"
#define MFMA "v_mfma_f32_16x16x32_f16 v[100:103], %8, %9, v[100:103]\n"
#define MFMA_OTHER "v_mfma_f32_16x16x32_f16 v[110:113], %8, %9, v[110:113]\n"  // an MFMA the reader does not depend on
#define OUT(lo, hi) "s_nop 15\ns_nop 15\nv_mov_b32 %0, " lo "\nv_mov_b32 %1, " hi "\nv_mov_b32 %2, v102\nv_mov_b32 %3, v103\n"
#define RD_R "v_pk_fma_f32 v[100:101], v[100:101], %10, v[104:105] op_sel:[0,0,1]\n"
#define RD_P "v_pk_fma_f32 v[100:101], v[100:101], %10, v[104:105]\n"
#define RD_Q "v_pk_fma_f32 v[106:107], v[100:101], %10, v[104:105] op_sel:[0,0,1]\n"
#define RD_V "v_pk_fma_f32 v[100:101], v[100:101], v[108:109], v[104:105] op_sel:[0,0,1]\n"
// the VGPR-only forms without op_sel that hand-placed packed arithmetic would use (scale in place, sum of squares, sum)
#define RD_W "v_pk_mul_f32 v[100:101], v[100:101], v[108:109]\n"
#define RD_X "v_pk_fma_f32 v[106:107], v[100:101], v[100:101], v[104:105]\n"
#define RD_Y "v_pk_add_f32 v[106:107], v[100:101], v[104:105]\n"
#define RD_Z "v_pk_fma_f32 v[100:101], v[100:101], v[108:109], v[104:105]\n"

#define OPS : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(a), "v"(b), "s"(s2) \
            : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113"
#define NOP(n) "s_nop " #n "\n"
#define W32 "s_nop 15\ns_nop 15\n"
#define CASEM(id, mf, wait, rd, lo, hi) \
  if (SEQ == id) { asm volatile(LOAD mf W32 rd OUT(lo, hi) OPS); for (int i = 0; i < 4; ++i) q[i] = r[i]; asm volatile(LOAD mf wait rd OUT(lo, hi) OPS); }
#define CASE(id, wait, rd, lo, hi) CASEM(id, MFMA, wait, rd, lo, hi)

template <int SEQ>
__global__ __launch_bounds__(256) void probe(float* out, int rounds) {
  const int lane = threadIdx.x & 63;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(float)(((lane * 5 + i * 3) % 5) - 2); b[i] = (_Float16)(float)(((lane * 5 + i * 3 + 1) % 5) - 2); }
  const f2 s2 = {2.f, 2.f};
  float bad[2] = {0, 0};
  for (int it = 0; it < rounds; ++it) {
    float c[4], r[4], q[4];
    for (int i = 0; i < 4; ++i) c[i] = (float)(1000 + 7 * lane + it + i);  // small integers: every result is exact
    CASE(0, NOP(7), RD_R, "v100", "v101") CASE(1, NOP(11), RD_R, "v100", "v101") CASE(2, NOP(15), RD_R, "v100", "v101")
    CASE(3, "s_nop 15\ns_nop 7\n", RD_R, "v100", "v101")
    CASE(4, NOP(7), RD_P, "v100", "v101") CASE(5, NOP(7), RD_Q, "v106", "v107") CASE(6, NOP(7), RD_V, "v100", "v101")
    CASEM(7, "", NOP(7), RD_R, "v100", "v101") CASEM(8, MFMA_OTHER, NOP(7), RD_R, "v100", "v101")
    CASE(9, NOP(7), RD_W, "v100", "v101") CASE(10, NOP(7), RD_X, "v106", "v107") CASE(11, NOP(7), RD_Y, "v106", "v107") CASE(12, NOP(7), RD_Z, "v100", "v101")
    for (int i = 0; i < 2; ++i) bad[i] += (r[i] != q[i]) ? 1.f : 0.f;
  }
  float* o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  o[0] = bad[0]; o[1] = bad[1];
}

template <int SEQ>
void run(const char* what, int blocks, int rounds) {
  float* out;
  const size_t n = (size_t)blocks * 256 * 2;
  hipMalloc(&out, n * 4);
  probe<SEQ><<<blocks, 256>>>(out, rounds);
  hipDeviceSynchronize();
  std::vector<float> h(n);
  hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost);
  double reg[2] = {0, 0}, grp[4] = {0, 0, 0, 0};
  for (size_t t = 0; t < n / 2; ++t)
    for (int i = 0; i < 2; ++i) { reg[i] += h[t * 2 + i]; grp[(t & 63) >> 4] += h[t * 2 + i]; }
  printf("%-58s wgs %5d: wrong lo/hi register [%.0f %.0f] of %.0f; lanes 0-15|16-31|32-47|48-63 [%.0f %.0f %.0f %.0f]\n", what, blocks, reg[0],
         reg[1], (double)blocks * 256 * rounds, grp[0], grp[1], grp[2], grp[3]);
  hipFree(out);
}

int main() {
  for (int blocks : {1, 256, 3072}) {
    run<0>("R in place, sgpr pair, op_sel,  8 wait states (hipcc)", blocks, 1000);
    run<1>("R the same, 12 wait states", blocks, 1000);
    run<2>("R the same, 16 wait states", blocks, 1000);
    run<3>("R the same, 24 wait states", blocks, 1000);
    run<4>("P without op_sel, 8 wait states", blocks, 1000);
    run<5>("Q not in place, 8 wait states", blocks, 1000);
    run<6>("V vgpr-pair multiplier, 8 wait states", blocks, 1000);
    run<7>("N form R with NO mfma in front", blocks, 1000);
    run<8>("O form R behind an mfma on other registers", blocks, 1000);
    run<9>("W v_pk_mul_f32 in place, vgpr pair, no op_sel", blocks, 1000);
    run<10>("X v_pk_fma_f32 d = a * a + c, vgprs only", blocks, 1000);
    run<11>("Y v_pk_add_f32 vgprs only", blocks, 1000);
    run<12>("Z v_pk_fma_f32 in place, vgpr pairs, no op_sel", blocks, 1000);
  }
  return 0;
}
