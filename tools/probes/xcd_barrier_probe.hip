// Probe (gfx950): what does a barrier among the workgroups of ONE XCD cost, against a chip-wide one and against a kernel boundary?
// Go / no-go number for an XCD-local persistent UNet body: a layer boundary of that design is one such barrier
// per agent (GroupNorm statistics are per agent: unet.py:36-37), so the design pays N_layers x this cost instead of N_layers launches.
//
// Protocol under test -- no agent-scope fence anywhere (MI355X_MICROARCH.md "Valid forms"): every payload store is `sc1`
// (write-through), every storing wave drains with s_waitcnt vmcnt(0), the workgroup's barrier, ONE lane adds to the group's counter
// (agent-scope relaxed atomic), ONE lane polls the counter with sc1 loads (+ s_sleep), the workgroup's barrier, payload reads are sc1
// loads.  Correct under any workgroup placement; grouping by the XCC id only decides WHICH workgroups share a counter.
//   groups = 8: workgroups that read the same HW_REG_XCC_ID share a counter (32 per XCD with one workgroup per CU)
//   groups = 1: one counter for all 256
// Payload per workgroup and barrier: `payload` 16-byte records written before the arrive, `payload` records of the NEXT workgroup of the
// group read (and checked) after the barrier -- the halo exchange of a tile with its neighbour.  Variant `stream`: every workgroup also
// streams 64 KB of plain loads from a large buffer between barriers (the memory queue is busy, as in a real layer).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/xcd_barrier_probe.hip -o tools/probes/xcd_barrier_probe.bin && tools/probes/xcd_barrier_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_sc1(const uint4* p) {
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st_sc1(uint4* p, uint4 q) {
  const u32x4 v = {q.x, q.y, q.z, q.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned ld_u32_sc1(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

struct Args {
  unsigned* census;     // [16] workgroups per XCC id (phase 0)
  unsigned* rank_of;    // [grid] rank of a workgroup inside its group
  unsigned* counter;    // [16 * 32] one counter per group, 128 bytes apart
  uint4* slab;          // [grid][payload] exchanged records
  const float4* big;    // streaming source
  unsigned* errors;
  unsigned long long* ticks;   // [grid] s_memrealtime ticks (100 MHz) spent in the timed loop
  int iters, payload, groups, stream;
  long long big_quads;
};

__global__ __launch_bounds__(256) void barrier_kernel(const Args a) {
  __shared__ unsigned s_group, s_rank, s_size;
  const int tid = threadIdx.x, b = blockIdx.x;
  if (tid == 0) {
    const unsigned g = a.groups == 1 ? 0u : xcc_id();
    s_group = g;
    s_rank = atomicAdd(&a.census[g], 1u);
  }
  __syncthreads();
  const unsigned g = s_group, rank = s_rank;
  unsigned* ctr = a.counter + g * 32;
  // ---- phase 0: the census settles (every workgroup has registered) -- a chip-wide rendezvous on counter slot 15 * 32 + 16, once
  if (tid == 0) {
    atomicAdd(&a.counter[15 * 32 + 16], 1u);
    for (int spins = 0; ld_u32_sc1(&a.counter[15 * 32 + 16]) < gridDim.x && spins < (1 << 22); ++spins) __builtin_amdgcn_s_sleep(2);   // bounded
    s_size = ld_u32_sc1(&a.census[g]);
    a.rank_of[b] = rank;
  }
  __syncthreads();
  const unsigned size = s_size;
  // group-local slab index: the workgroup's slot and its successor's (found through a small table keyed by (group, rank))
  // two record sets per workgroup, used alternately: a workgroup may run one barrier ahead of the neighbour that still reads its records
  uint4* mine = a.slab + ((size_t)g * 64 + rank) * 128;
  const uint4* next = a.slab + ((size_t)g * 64 + (rank + 1) % size) * 128;
  float acc = 0.f;
  unsigned bad = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < a.iters; ++it) {
    if (a.stream) {   // 64 KB of plain streaming loads per workgroup
      const long long base = ((long long)b * a.iters + it) * 4096 % (a.big_quads - 4096);
#pragma unroll 4
      for (int k = 0; k < 16; ++k) { const float4 v = a.big[base + k * 256 + tid]; acc += v.x + v.w; }
    }
    if (tid < a.payload) st_sc1(mine + 64 * (it & 1) + tid, make_uint4((unsigned)it, rank, g, (unsigned)tid));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = size * (unsigned)(it + 1);
      int spins = 0;
      while (ld_u32_sc1(ctr) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { atomicAdd(a.errors, 1u << 16); break; }   // bounded: the grid always drains
      }
    }
    __syncthreads();
    if (tid < a.payload) {
      const uint4 v = ld_sc1(next + 64 * (it & 1) + tid);
      bad += (v.x != (unsigned)it) | (v.y != (rank + 1) % size) | (v.z != g) | (v.w != (unsigned)tid);
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (bad) atomicAdd(a.errors, bad);
  if (tid == 0) a.ticks[b] = t1 - t0;
  if (acc == 1.2345e38f) a.errors[1] = 1;
}

__global__ void tiny_kernel(float* buf, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  buf[i % n] += 1.0f;
}

int main() {
  const int grid = 256, iters = 400;
  Args a{};
  hipMalloc(&a.census, 16 * 4); hipMalloc(&a.rank_of, grid * 4); hipMalloc(&a.counter, 16 * 32 * 4);
  hipMalloc(&a.slab, (size_t)16 * 64 * 128 * sizeof(uint4)); hipMalloc(&a.errors, 8); hipMalloc(&a.ticks, grid * 8);
  a.big_quads = (long long)(1u << 26);   // 1 GiB of float4
  hipMalloc((void**)&a.big, (size_t)a.big_quads * 16);
  hipMemset((void*)a.big, 0, (size_t)a.big_quads * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("# barrier among workgroups (256 threads each, one per CU, %d barriers per launch); us per barrier = kernel time / barriers\n", iters);
  for (int stream = 0; stream < 2; ++stream)
    for (int groups : {8, 1})
      for (int payload : {0, 16, 64}) {
        a.iters = iters; a.payload = payload; a.groups = groups; a.stream = stream;
        float best = 1e9f; unsigned err = 0; std::vector<unsigned> cen(16);
        std::vector<unsigned long long> tk(grid);
        double tick_us = 0;
        for (int rep = 0; rep < 3; ++rep) {
          hipMemset(a.census, 0, 64); hipMemset(a.counter, 0, 16 * 32 * 4); hipMemset(a.errors, 0, 8);
          hipMemset(a.slab, 0xff, (size_t)16 * 64 * 128 * sizeof(uint4));
          hipDeviceSynchronize();
          hipEventRecord(e0);
          barrier_kernel<<<grid, 256>>>(a);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) {
            best = ms;
            hipMemcpy(tk.data(), a.ticks, grid * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto t : tk) s += (double)t; tick_us = s / grid * 0.01 / iters;
          }
          unsigned e[2]; hipMemcpy(e, a.errors, 8, hipMemcpyDeviceToHost); err += e[0];
          hipMemcpy(cen.data(), a.census, 64, hipMemcpyDeviceToHost);
        }
        unsigned mx = 0, mn = 1u << 30, ng = 0;
        for (unsigned c : cen) if (c) { ++ng; if (c > mx) mx = c; if (c < mn) mn = c; }
        printf("%-10s %-22s payload %3d x 16 B: %6.2f us per barrier (in-kernel %6.2f), groups %u of %u..%u workgroups, errors %u\n",
               stream ? "streaming" : "idle", groups == 8 ? "per-XCD (8 counters)" : "chip-wide (1 counter)", payload, best * 1e3f / iters, tick_us, ng, mn, mx, err);
      }
  // the alternative: a dependent kernel boundary
  float* buf; hipMalloc(&buf, 1 << 22);
  for (int blocks : {32, 256}) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      for (int k = 0; k < 200; ++k) tiny_kernel<<<blocks, 256>>>(buf, 1 << 20);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("dependent launches of a trivial kernel, %d workgroups: %.2f us each\n", blocks, best * 1e3f / 200);
  }
  return 0;
}
