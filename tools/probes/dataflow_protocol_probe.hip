// Probe: the ticket / wait / publish protocol of csrc/dataflow_kernels.h alone (no layer computation), with a watchdog on the
// host that reads the counters through a second stream while the kernel runs.
//   hipcc --offload-arch=gfx950 -O3 -I../../gencomm_amd/csrc dataflow_protocol_probe.hip -o dataflow_protocol_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <chrono>
#include <thread>
#include <unistd.h>

struct POp { int tiles, items; int item_base[9]; int ticket_base[8]; };
struct PProg { int nops, n; int queue_len[8]; POp ops[32]; };
struct PArgs { const PProg* prog; unsigned* tickets; unsigned* done; unsigned* err; };

__global__ __launch_bounds__(256, 3) void protocol_kernel(const PArgs a) {
  __shared__ int s_item[3];
  const int tid = threadIdx.x;
  const PProg* __restrict__ P = a.prog;
  const int nops = P->nops, n_agents = P->n;
  const unsigned home = blockIdx.x & 7u;
  // ONE tid-0 block per iteration (publish the finished item, then fetch the next ticket): with a tid-0 block at the end of the
  // body AND one at its top, hipcc threads lane 0 from the one into the other across the back-edge and the loop's barriers
  // are no longer executed convergently (the first version of this kernel hung in its first iteration).
  auto fetch = [&](unsigned x, int qlen) {  // tid 0 only
    const int t = (int)__hip_atomic_fetch_add(a.tickets + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int op = -1, it = 0;
    if (t < qlen) {
      op = 0;
      while (op + 1 < nops && P->ops[op + 1].ticket_base[x] <= t) ++op;
      it = P->ops[op].item_base[x] + (t - P->ops[op].ticket_base[x]);
    }
    s_item[0] = op;
    s_item[1] = it;
  };
  for (int q = 0; q < 8; ++q) {
    const unsigned x = (home + (unsigned)q) & 7u;
    const int qlen = P->queue_len[x];
    if (tid == 0) fetch(x, qlen);
    for (;;) {
      __syncthreads();
      const int op = __builtin_amdgcn_readfirstlane(s_item[0]), it = __builtin_amdgcn_readfirstlane(s_item[1]);
      if (op < 0) break;
      const POp& o = P->ops[op];
      const int agent = it / o.tiles;
      if (tid == 0 && op > 0) {
        const unsigned need = (unsigned)P->ops[op - 1].tiles;
        const unsigned* cnt = a.done + (size_t)(op - 1) * n_agents + agent;
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 18)) { atomicCAS(a.err, 0u, 1u + (unsigned)op); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      // (the tile computation goes here)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(a.done + (size_t)op * n_agents + agent, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fetch(x, qlen);
      }
    }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const int nops = argc > 1 ? atoi(argv[1]) : 26, tiles = argc > 2 ? atoi(argv[2]) : 1, n = argc > 3 ? atoi(argv[3]) : 3, grid = argc > 4 ? atoi(argv[4]) : 768;
  PProg hp{};
  hp.nops = nops; hp.n = n;
  int qlen[8] = {0};
  for (int k = 0; k < nops; ++k) {
    POp& d = hp.ops[k];
    d.tiles = tiles; d.items = tiles * n;
    const int qq = d.items >> 3, rr = d.items & 7;
    for (int x = 0; x <= 8; ++x) d.item_base[x] = x * qq + (x < rr ? x : rr);
    for (int x = 0; x < 8; ++x) { d.ticket_base[x] = qlen[x]; qlen[x] += d.item_base[x + 1] - d.item_base[x]; }
  }
  for (int x = 0; x < 8; ++x) hp.queue_len[x] = qlen[x];
  PProg* dp; unsigned* words; const int nwords = 8 + 64 * n + 4;
  hipMalloc(&dp, sizeof(PProg)); hipMalloc(&words, nwords * 4);
  hipMemcpy(dp, &hp, sizeof(PProg), hipMemcpyHostToDevice);
  hipMemset(words, 0, nwords * 4);
  hipStream_t st, side; hipStreamCreateWithFlags(&st, hipStreamNonBlocking); hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
  unsigned* hw; hipHostMalloc(&hw, nwords * 4);
  PArgs pa{dp, words, words + 8, words + 8 + 64 * n};
  auto t0 = std::chrono::steady_clock::now();
  protocol_kernel<<<grid, 256, 0, st>>>(pa);
  for (int tick = 0; tick < 40; ++tick) {
    if (hipStreamQuery(st) == hipSuccess) break;
    std::this_thread::sleep_for(std::chrono::milliseconds(250));
    hipMemcpyAsync(hw, words, nwords * 4, hipMemcpyDeviceToHost, side); hipStreamSynchronize(side);
    printf("t=%.2fs tickets [%u %u %u %u %u %u %u %u] err %u done(op0..5 agent0) %u %u %u %u %u %u\n", 0.25 * (tick + 1), hw[0], hw[1], hw[2], hw[3], hw[4], hw[5], hw[6], hw[7],
           hw[8 + 64 * n], hw[8], hw[8 + n], hw[8 + 2 * n], hw[8 + 3 * n], hw[8 + 4 * n], hw[8 + 5 * n]);
    fflush(stdout);
  }
  const bool done = hipStreamQuery(st) == hipSuccess;
  double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  printf("finished %d after %.1f ms\n", (int)done, ms);
  if (!done) { printf("STUCK: leaving without waiting\n"); fflush(stdout); _exit(3); }
  hipMemcpy(hw, words, nwords * 4, hipMemcpyDeviceToHost);
  long total = 0; for (int k = 0; k < nops; ++k) for (int a = 0; a < n; ++a) total += hw[8 + k * n + a];
  printf("tickets [%u %u %u %u %u %u %u %u] err %u, tiles done %ld of %d\n", hw[0], hw[1], hw[2], hw[3], hw[4], hw[5], hw[6], hw[7], hw[8 + 64 * n], total, nops * tiles * n);
  return 0;
}
