// Cost of a grid-wide barrier on MI355X (cooperative launch, cooperative_groups::grid_group::sync) against the cost of a
// dependent kernel launch: decides whether a persistent whole-UNet kernel for small maps can beat ~100 dependent launches.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gridsync_probe.hip -o tools/probes/gridsync_probe.bin
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;

__global__ void sync_kernel(float* buf, int n, int iters) {
  cg::grid_group g = cg::this_grid();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = buf[i % n];
  for (int it = 0; it < iters; ++it) {
    buf[(i + it) % n] = v + 1.0f;   // a store another workgroup reads after the barrier
    g.sync();
    v = buf[(i + 7919 * (it + 1)) % n];
  }
  buf[i % n] = v;
}
__global__ void tiny_kernel(float* buf, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  buf[i % n] += 1.0f;
}

int main() {
  const int n = 1 << 20;
  float* buf;
  hipMalloc(&buf, n * sizeof(float));
  hipMemset(buf, 0, n * sizeof(float));
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {64, 256, 512, 768}) {
    for (int iters : {1, 101}) {
      int it = iters;
      int nn = n;
      void* args[] = {&buf, &nn, &it};
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, st);
        hipError_t rc = hipLaunchCooperativeKernel((const void*)sync_kernel, dim3(blocks), dim3(256), args, 0, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        if (rc != hipSuccess) { printf("blocks %d: cooperative launch failed: %s\n", blocks, hipGetErrorString(rc)); best = -1; break; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("cooperative kernel, %d workgroups x 256, %d grid syncs: %.1f us\n", blocks, iters, best * 1e3f);
    }
  }
  // dependent launches on one stream
  for (int blocks : {64, 256}) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0, st);
      for (int k = 0; k < 100; ++k) tiny_kernel<<<blocks, 256, 0, st>>>(buf, n);
      hipEventRecord(e1, st);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("100 dependent launches of a trivial kernel, %d workgroups: %.1f us (%.2f us each)\n", blocks, best * 1e3f, best * 10.f);
  }
  return 0;
}
