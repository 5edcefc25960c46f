// Probe (gfx950): where does conv2d_h3l_kernel's time go?  Times the kernel on the stage-1 training leg's shapes with parts of the
// stage loop compiled out (H3_DIAG bits: 1 = no weight DMA inside the loop, 2 = no activation fetch / split / commit inside the loop,
// 4 = no matrix instructions, 8 = no operand reads) -- the RESULTS of those builds are garbage, only their durations mean anything.
//   for d in 0 1 2 3 4 8 12; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DH3_DIAG=$d tools/probes/conv_h3_probe.hip -o /tmp/h3p_$d; done
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../gencomm_amd/csrc/common.h"
#include "../../gencomm_amd/csrc/conv_kernels.h"
namespace gc {
Modes modes_snapshot() { Modes m{}; m.v[MODE_ARITH] = 0; return m; }
char* last_error_buf() { static char b[512]; return b; }
KernelTimer& kernel_timer() { static KernelTimer t{}; return t; }
bool klog_armed() { return false; }
void klog_note(const char*) {}
}
using namespace gc;
int main() {
  struct S { int N, Cin, Cout, H, W, K; } shapes[] = {{4, 64, 64, 128, 64, 3}, {4, 256, 256, 32, 16, 3}, {4, 384, 256, 128, 64, 3}, {4, 64, 64, 200, 704, 3}, {4, 128, 512, 64, 128, 1}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto s : shapes) {
    const int T = s.K * s.K;
    const size_t nx = (size_t)s.N * s.Cin * s.H * s.W, ny = (size_t)s.N * s.Cout * s.H * s.W;
    const long long pf = conv2d_prepared_floats(s.Cin, s.Cout, s.K, s.K, 0);
    float *x, *w, *y, *prep, *ss;
    hipMalloc(&x, nx * 4); hipMalloc(&y, ny * 4); hipMalloc(&w, (size_t)s.Cin * s.Cout * T * 4); hipMalloc(&prep, pf * 4); hipMalloc(&ss, 2 * s.Cout * 4);
    std::vector<float> hx(nx), hw((size_t)s.Cin * s.Cout * T), hs(2 * s.Cout, 0.f);
    for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    for (int i = 0; i < s.Cout; ++i) hs[i] = 1.f;
    hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); hipMemcpy(ss, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    const long long total = (long long)s.Cin * s.Cout * T;
    unsigned char* blob = reinterpret_cast<unsigned char*>(prep + total);
    PrepW3Args pa{w, prep, blob, reinterpret_cast<float*>(blob + h3_blob_bytes(s.Cin, s.Cout, T)), s.Cin, s.Cout, s.K, s.K, 0, s.Cout, T, h3_chunks(s.Cin), h3_blocks(s.Cout)};
    conv_w3_rowscale_kernel<<<16 * pa.nb, 256>>>(pa);
    conv_prep_w3_kernel<<<dim3(pa.nb, pa.nchunk, T), 128>>>(pa);
    Conv2dArgs a{x, prep, ss, ss + s.Cout, y, s.Cin, s.H, s.W, s.Cout, s.H, s.W, 1, s.K / 2, 0, 1, s.Cout, 0};
    a.w3 = blob; a.wsc = pa.wsc;
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      for (int k = 0; k < 10; ++k) if (conv2d_enqueue(a, s.N, s.K, s.K, 0) != 0) { printf("enqueue failed: %s\n", last_error_buf()); return 1; }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double fl = 2.0 * s.N * s.H * s.W * (double)s.Cout * s.Cin * T;
    printf("diag %d: N %d %3d->%3d %3dx%3d k%d: %7.1f us = %6.1f TFLOP/s fp32-equivalent\n", H3_DIAG, s.N, s.Cin, s.Cout, s.H, s.W, s.K, best * 100.f, fl / (best * 1e-4) * 1e-12);
    hipFree(x); hipFree(y); hipFree(w); hipFree(prep); hipFree(ss);
  }
  return 0;
}
