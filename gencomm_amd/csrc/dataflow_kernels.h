// Persistent DATAFLOW execution of the UNet body: one launch runs a chain of layer ops -- the 8-channel
// 3x3 layers of conv8h_kernels.h, the stride-2 conv -- for all agents, without a grid barrier and without a launch boundary
// between the layers.
//
// Why.  A layer launch of the benchmark (16 agents) is 2 288 (full resolution) or 672 (half resolution) workgroups on 768
// resident slots: every launch pays the fill and drain of one workgroup's load -> stage -> matrix -> store chain, a fixed
// ~13 us of a ~30 us launch, and ~26 such launches follow each other in a UNet call.  But GroupNorm's
// statistics are per AGENT (unet.py:36-37): layer k + 1 of agent n needs layer k of agent n only.  So:
//
//   * work item = (op k, agent n, tile); items are ordered op-major, agent-major inside an op, and dealt to the 8 XCDs in
//     contiguous eighths of every op (the same rule as xcd_block: neighbouring tiles, and an agent's successive layers,
//     stay on one XCD's L2);
//   * every XCD has a ticket counter; a workgroup takes the next ticket of that XCD's queue (then helps
//     the other queues), waits until done[k - 1][n] == tiles of op k - 1, computes the tile with the SAME device function
//     the per-layer kernels call, and adds 1 to done[k][n];
//   * the UNet body is a chain (every op consumes its predecessor's output; skips and residuals come from ancestors), so
//     "predecessor complete for this agent" implies every input complete, and a buffer slot is never rewritten while a
//     reader of its previous contents is still running (the last reader is an ancestor of the writer's predecessor).
//
// Progress.  Tickets are taken in queue order by RUNNING workgroups only, and an item waits only for items of an earlier op:
// the earliest unfinished op's items are never blocked, so the grid drains for any grid size and any placement of the
// workgroups (a queue whose XCD got no workgroup is served by the helpers).  Every spin is bounded: on timeout the
// workgroup sets the error word, skips the computation but still publishes its item, so that the grid always drains; the host
// checks the word (tests) -- MI355X_MICROARCH.md "bound every spin".
//
// Visibility (cdna_hip_programming.md section 6, Guideline 16: L1 is per CU and never refreshed by other CUs' stores, the
// L2s are per XCD).  Producer: every wave drains its stores (s_waitcnt vmcnt(0)), workgroup barrier, then ONE lane issues an
// agent-scope release fence, waits again, and adds to the counter (relaxed, agent scope).  Consumer: one lane polls the
// counter relaxed, then ONE agent-scope acquire fence + s_waitcnt vmcnt(0), workgroup barrier, then plain loads.  The
// GroupNorm sums are f64 atomics at agent scope; the consumer reads them behind the same acquire on the vector path.
// All words (tickets, done counters, error word) live in the statistics block that the per-call hipMemsetAsync zeroes.
#pragma once
#include "conv8h_kernels.h"

namespace gc {

constexpr int DF_MAX_OPS = 32;
#ifndef DF_SPIN_LIMIT
#define DF_SPIN_LIMIT (1u << 19)  // polls (~1 us each) before a workgroup gives up: about half a second
#endif
enum DfKind : int { DF_CONV1 = 0, DF_CONV16 = 1, DF_CONV2_RES1 = 2, DF_CONV2_RES2 = 3, DF_UP = 4, DF_DOWN = 5 };

struct DfOp {
  int kind;
  int tx, ty;          // tiles per agent in x / y (DF_DOWN: tx = workgroups per agent, ty = 1)
  int tiles;           // tx * ty
  int items;           // tiles * n
  int item_base[9];    // first item of XCD x in this op's (agent-major) item order; [8] = items
  int ticket_base[8];  // first ticket of this op in XCD x's queue
  int bias_stride;     // floats added to conv.bias per timestep (8 for a ResnetBlock conv1: its bias table is [T][8]; else 0)
  Conv8Args conv;
  DownArgs down;
};
struct DfProgram {
  int nops, n;
  int queue_len[8];    // tickets per XCD queue
  DfOp ops[DF_MAX_OPS];
};
struct DfArgs {
  const DfProgram* prog;  // device memory
  unsigned* tickets;      // [8]
  unsigned* done;         // [nops][n]
  unsigned* err;          // [1]: 0 ok, else 1 + index of the first op that timed out
  int t;                  // timestep of this UNet call (selects the conv1 bias rows)
  int skip_compute;       // diagnostic (GENCOMM_MODE_DATAFLOW = 2): tickets, waits and counters only
};

// The program travels to device memory through the kernel-argument segment (no host-to-device copy: capture-safe, and the
// caller's stream never waits for the host): DF_CHUNK ops per launch.
constexpr int DF_CHUNK = 8;
struct DfUploadArgs {
  DfProgram* dst;
  int first, count, nops, n;
  int queue_len[8];
  DfOp ops[DF_CHUNK];
};
static_assert(sizeof(DfProgram) < (1 << 16), "program block of the workspace");
static_assert(sizeof(DfUploadArgs) <= 4000, "kernel arguments are limited to 4 KB");
__global__ __launch_bounds__(64) void df_upload_kernel(const DfUploadArgs a) {
  const unsigned* __restrict__ src = reinterpret_cast<const unsigned*>(a.ops);
  unsigned* __restrict__ dst = reinterpret_cast<unsigned*>(a.dst->ops + a.first);
  for (int i = threadIdx.x; i < a.count * (int)(sizeof(DfOp) / 4); i += 64) dst[i] = src[i];
  if (a.first == 0 && threadIdx.x < 8) a.dst->queue_len[threadIdx.x] = a.queue_len[threadIdx.x];
  if (a.first == 0 && threadIdx.x == 0) { a.dst->nops = a.nops; a.dst->n = a.n; }
}

// Each layer kind is a real function call: inlined into one body the five variants share a register allocation and spill
// 592 bytes per lane; called, each keeps the allocation it has as a kernel of its own.
template <int NSRC, bool GN, bool UP, int RES>
__device__ __attribute__((noinline)) void df_conv(const Conv8Args* a, int bias_off, int bx, int by, int bz, unsigned char* tile,
                                                  float (*s_ab)[2], float (*s_red)[16]) {
  conv8h_tile<NSRC, GN, UP, RES>(*a, BlockId{bx, by, bz}, tile, s_ab, s_red, (size_t)bias_off);
}
__device__ __attribute__((noinline)) void df_down(const DownArgs* a, int bx, int n, float (*s_red)[16]) { down8x2_tile(*a, bx, n, s_red); }

__global__ __launch_bounds__(HC_NT, 3) void unet_dataflow_kernel(const DfArgs a) {
  __shared__ __align__(16) unsigned char tile[HC_TILE_BYTES];
  __shared__ float s_ab[16][2];
  __shared__ float s_red[HC_NT / 64][16];
  __shared__ int s_item[3];  // {op, item within op, run?} of the current ticket; op -1 = queue exhausted
  const int tid = threadIdx.x;
  const DfProgram* __restrict__ P = a.prog;
  const int nops = P->nops, n_agents = P->n;
  // the dispatcher deals workgroups to the XCDs round-robin by linear id (the fact xcd_block rests on, measured with the PMC
  // passes): the home queue needs no hardware register read.  A wrong guess costs locality, not correctness.
  const unsigned home = blockIdx.x & 7u;

  // ONE tid-0 block per iteration -- publish the finished item, then fetch the next ticket.  With a tid-0 block at the end of
  // the body and another at its top, hipcc threads lane 0 from the one into the other across the loop's back-edge; the
  // barriers inside the loop are then no longer executed convergently and the first version of this kernel hung in its first
  // iteration (tools/probes/dataflow_protocol_probe.hip).  The uniform values come through readfirstlane so that the branches
  // around the barriers are scalar branches.
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
  for (int q = 0; q < 8; ++q) {  // own queue first, then the others in ring order (helpers)
    const unsigned x = (home + (unsigned)q) & 7u;
    const int qlen = P->queue_len[x];
    if (tid == 0) fetch(x, qlen);
    for (;;) {
      __syncthreads();
      const int op = __builtin_amdgcn_readfirstlane(s_item[0]), it = __builtin_amdgcn_readfirstlane(s_item[1]);
      if (op < 0) break;
      const DfOp& o = P->ops[op];
      const int tiles = __builtin_amdgcn_readfirstlane(o.tiles);
      const int agent = it / tiles, tl = it - agent * tiles;
      if (tid == 0) {
        if (op > 0) {  // wait for the predecessor op of this agent
          const unsigned need = (unsigned)P->ops[op - 1].tiles;
          const unsigned* cnt = a.done + (size_t)(op - 1) * n_agents + agent;
          unsigned spins = 0;
          while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > DF_SPIN_LIMIT) {  // never reached unless the protocol is broken
              atomicCAS(a.err, 0u, 1u + (unsigned)op);
              break;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // a run that has timed out anywhere only has to drain: skip the computation (decided by ONE lane: the branch around
        // the tile function, which contains barriers, must be uniform)
        s_item[2] = (!a.skip_compute && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) ? 1 : 0;
      }
      __syncthreads();
      if (__builtin_amdgcn_readfirstlane(s_item[2])) {
        const int kind = __builtin_amdgcn_readfirstlane(o.kind), tx = __builtin_amdgcn_readfirstlane(o.tx);
        const int by = tl / tx, bx = tl - by * tx;
        if (kind == DF_DOWN) {
          df_down(&o.down, tl, agent, s_red);
        } else {
          const int boff = a.t * __builtin_amdgcn_readfirstlane(o.bias_stride);
          switch (kind) {
            case DF_CONV1: df_conv<1, true, false, 0>(&o.conv, boff, bx, by, agent, tile, s_ab, s_red); break;
            case DF_CONV16: df_conv<2, true, false, 0>(&o.conv, boff, bx, by, agent, tile, s_ab, s_red); break;
            case DF_CONV2_RES1: df_conv<1, true, false, 1>(&o.conv, boff, bx, by, agent, tile, s_ab, s_red); break;
            case DF_CONV2_RES2: df_conv<1, true, false, 2>(&o.conv, boff, bx, by, agent, tile, s_ab, s_red); break;
            default: df_conv<1, false, true, 0>(&o.conv, boff, bx, by, agent, tile, s_ab, s_red); break;
          }
        }
      }
      // publish: every wave's stores (and statistics atomics) drained, then one release + count; then the next ticket
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

}  // namespace gc
