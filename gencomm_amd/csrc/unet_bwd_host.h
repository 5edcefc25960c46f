// Host logic of gencomm_unet_bwd: one DiffusionUNet call backwards (reference: autograd through unet.py:307-344 inside the
// training branch cond_diff.py:342-360).  Re-runs the HIP forward with every intermediate kept, then walks the launch
// program in reverse.  Attention blocks are not differentiated here (no shipped config has one): attn_mask must be 0.
#pragma once
#include <mutex>
#include "conv_kernels.h"
#include "train_kernels.h"
#include "unet_bwd_kernels.h"
#include "unet_host.h"

namespace gc {

#ifdef GC_DIAG_SCALAR_GN_APPLY   // diagnostic build (tools/diag/train_lib_ab.sh): one dword per lane in the GroupNorm-backward apply / nin_dgrad, separate axpy
constexpr bool kBwdVec4 = false;
#else
constexpr bool kBwdVec4 = true;
#endif
#ifdef GC_DIAG_ZERO_GRAD_MAPS    // diagnostic build: every gradient map zeroed per call and every contribution accumulated (before first-write tracking)
constexpr bool kBwdFirstWrite = false;
#else
constexpr bool kBwdFirstWrite = true;
#endif

// dgrad weight table of one UNet call: every 3x3 stride-1 layer's input-gradient weights, in the order the backward walk uses
// them (conv_out, then per op from the last to the first; conv_in contributes two entries: cond channels, x channels)
struct DgradEntry { long long w_off; int cout, cin, ic0, nic; long long p_off; int blocked; long long floats; };
inline std::vector<DgradEntry> dgrad_entries(const UNetPlan& p) {
  std::vector<DgradEntry> e;
  long long off = 0;
  // blocked: conv_out_kernel's [ceil(nic / 16)][cout][9][16] layout (zero padded) instead of [(cout, flipped tap)][nic]
  auto add = [&](long long w, int cout, int cin, int ic0, int nic, int blocked = 0) {
    const long long fl = (long long)align_up(blocked ? (size_t)((nic + 15) / 16) * cout * 144 : (size_t)cout * nic * 9, 64);
    e.push_back(DgradEntry{w, cout, cin, ic0, nic, off, blocked, fl});
    off += fl;
  };
  for (int oi = (int)p.ops.size() - 1; oi >= 0; --oi) {
    const Op& o = p.ops[oi];
    switch (o.kind) {
      case OP_CONV_OUT: add(p.conv_out.w, p.C, 8, 0, 8); break;
      case OP_RES_CONV2: add(p.blocks[o.blk].c2w, 8, 8, 0, 8); break;
      case OP_RES_CONV1: for (int s0 = 0; s0 < p.blocks[o.blk].cin; s0 += 8) add(p.blocks[o.blk].c1w, 8, p.blocks[o.blk].cin, s0, 8); break;
      case OP_UP: add(p.up[o.level + 1].w, 8, 8, 0, 8); break;
      // conv_in's message-channel gradient through the 8 -> 8 kernel: input channels 0..7 (the 2 message channels + 6 of x_t, discarded)
#ifdef GC_DIAG_IGEMM_CONVIN_DGRAD   // diagnostic build: conv_in's x_t gradient through the general implicit-GEMM kernel (the round-3 path)
      case OP_CONV_IN: add(p.conv_in.w, 8, p.C + 2, 0, 8); add(p.conv_in.w, 8, p.C + 2, 2, p.C); break;
#else
      case OP_CONV_IN: add(p.conv_in.w, 8, p.C + 2, 0, 8); add(p.conv_in.w, 8, p.C + 2, 2, p.C, 1); break;
#endif
      default: break;
    }
  }
  return e;
}
inline size_t dgrad_table_floats(const UNetPlan& p) {
  const std::vector<DgradEntry> e = dgrad_entries(p);
  return e.empty() ? 64 : (size_t)(e.back().p_off + e.back().floats);
}

struct UNetBwdWs {
  size_t fwd_total, g_base, A, DA, red, wtmp, ones, zeros, wpart, chain, total;
  std::vector<size_t> g_level_base;
};
inline UNetBwdWs unet_bwd_ws(const UNetPlan& p, const UNetWorkspace& w, int n, int H, int W) {
  UNetBwdWs b{};
  size_t off = align_up(w.total, 256);
  b.fwd_total = off;
  b.g_base = off;
  b.g_level_base.assign(p.L, 0);
  for (int l = 0; l < p.L; ++l) {
    b.g_level_base[l] = off;
    off += w.slot_bytes[l] * p.slots_per_level[l];
  }
  auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
  const size_t map16 = (size_t)n * 16 * H * W * sizeof(float);
  b.A = take(map16);
  b.DA = take(map16);
  b.red = take((size_t)kMaxGnUses * n * 16 * sizeof(double));
  b.wtmp = take(dgrad_table_floats(p) * sizeof(float));
  b.ones = take((size_t)(p.C + 16) * sizeof(float));
  b.zeros = take((size_t)(p.C + 16) * sizeof(float));
  // partial sums of the weight-gradient kernels (conv_wgrad_scratch_floats): the largest layer is conv_in (C + 2 -> 8) / conv_out (8 -> C)
  b.wpart = take(std::max(conv_wgrad_scratch_floats(8, p.C + 2, 3, H, W, n), conv_wgrad_scratch_floats(p.C, 8, 3, H, W, n)) * sizeof(float));
  b.chain = take(64);   // five floats: the schedule-row form (., ., alpha, beta, 0) conv_out_kernel's fused update reads (gencomm_unet_bwd_chain)
  b.total = off;
  return b;
}

struct UNetBwdCall {
  UNetCall c;            // forward call (keep_all plan)
  const UNetBwdWs* bw;
  const float* raw;      // raw parameter blob (reference layouts)
  float* graw;           // gradient of the raw blob (+=)
  float* G(int id) const {
    const TensorPlan& t = c.plan->tensors[id];
    return reinterpret_cast<float*>(c.wsp + bw->g_level_base[t.level] + c.ws->slot_bytes[t.level] * t.slot);
  }
  float* F(size_t off) const { return reinterpret_cast<float*>(c.wsp + off); }
  // walk state: next entry of the dgrad weight table, GroupNorm uses recorded for the batched parameter-gradient kernel
  const std::vector<DgradEntry>* dg = nullptr;
  mutable size_t dg_next = 0;
  mutable GnParamArgs gp{};
  // gencomm_unet_bwd_chain: grad_xt = chain_alpha * (conv_in's x_t gradient) + chain_beta * chain_prev, in that layer's epilogue
  float chain_alpha = 1.0f, chain_beta = 0.0f;
  const float* chain_prev = nullptr;
  // Gradient maps are never zeroed: the FIRST contribution to a tensor's gradient in the walk writes it, later ones accumulate (a memset of
  // every map per call and one read per first contribution less).  take_first(id): true exactly once per tensor.
  mutable std::vector<char> g_written;
  bool take_first(int id) const { const bool f = !g_written[id]; g_written[id] = 1; return kBwdFirstWrite && f; }
  // a gradient map about to be READ: every kept tensor has a consumer that contributed before its producer is walked
  int need_written(int id) const { return g_written[id] ? GC_OK : fail(GC_ERR_ARG, "gencomm_unet_bwd: a tensor's gradient is read before anything contributed to it"); }
};

// out[n][Cout][H][W] = 3x3 stride-1 pad-1 convolution of dy[n][Cin_d][H][W] with the input-gradient weights of a forward
// layer w [Cout_f = Cin_d][Cin_f][3][3], restricted to forward input channels [ic0, ic0 + nic)
inline int dgrad3x3_enqueue(const UNetBwdCall& b, const float* dy, const float* w_fwd, int Cout_f, int Cin_f, int ic0, int nic,
                            float* out, int out_ctotal, int out_coff, int n, int H, int W) {
  const DgradEntry& e = (*b.dg)[b.dg_next++];
  if (b.raw + e.w_off != w_fwd || e.cout != Cout_f || e.cin != Cin_f || e.ic0 != ic0 || e.nic != nic)
    return fail(GC_ERR_ARG, "gencomm_unet_bwd: dgrad weight table out of step with the backward walk");
  Conv2dArgs a{dy, b.F(b.bw->wtmp) + e.p_off, b.F(b.bw->ones), b.F(b.bw->zeros), out, Cout_f, H, W, nic, H, W, 1, 1, 0, 1, out_ctotal, out_coff};
  return conv2d_enqueue(a, n, 3, 3, b.c.st);
}

// The same for an 8 -> 8 channel layer (forward output channels 8, forward input channels [ic0, ic0 + 8)) through the UNet's
// own exact-fp32 8-channel kernel (conv8_kernel on v_mfma_f32_4x4x1, unet_kernels.h): the table entry for nic = 8,
// [(oc * 9 + flipped tap)][8], IS that kernel's weight layout [ic][tap][oc].  out = [n][8][H][W].
// gn_src >= 0: the layer's input was SiLU(GroupNorm(tensor gn_src)) -- the epilogue (conv8_kernel RES = 3) turns d A into
// d z = d A SiLU'(.) and accumulates GroupNorm backward's two sums into the reduction slot gn_bwd_enqueue(..., fused) will use.
inline int dgrad8_enqueue(const UNetBwdCall& b, const float* dy, const float* w_fwd, int Cin_f, int ic0, float* out, int n, int H, int W,
                          int gn_src = -1, const float* gamma = nullptr, const float* beta = nullptr, int gs = 0) {
  const DgradEntry& e = (*b.dg)[b.dg_next++];
  if (b.raw + e.w_off != w_fwd || e.cout != 8 || e.cin != Cin_f || e.ic0 != ic0 || e.nic != 8)
    return fail(GC_ERR_ARG, "gencomm_unet_bwd: dgrad weight table out of step with the backward walk");
  Conv8Args a{};
  a.src[0] = dy; a.w = b.F(b.bw->wtmp) + e.p_off; a.wh = nullptr; a.bias = b.F(b.bw->zeros); a.dst = out; a.dstat = nullptr;
  a.H = a.Hin = H; a.W = a.Win = W;
  a.xcd = b.c.m.xcd();
  Modes mf = b.c.m;
  mf.v[MODE_ARITH] = 1;   // exact fp32: gradients span many orders of magnitude
  if (gn_src >= 0) {
    if (b.gp.uses >= kMaxGnUses) return fail(GC_ERR_ARG, "gencomm_unet_bwd: too many GroupNorm uses");
    a.res[0] = b.c.tensor_ptr(gn_src); a.sstat[1] = b.c.stat_ptr(gn_src); a.gamma = gamma; a.beta = beta; a.gn_gs = gs;
    a.inv_cnt = 1.0 / ((double)gs * H * W);
    a.dstat = reinterpret_cast<double*>(b.c.wsp + b.bw->red) + (size_t)b.gp.uses * b.c.n * 16;   // the slot the next gn_bwd_enqueue takes
    launch_conv8<1, false, false, 3>(mf, pick_tile(mf, n, H, W), a, n, b.c.st);
    return GC_OK;
  }
  launch_conv8<1, false, false, 0>(mf, pick_tile(mf, n, H, W), a, n, b.c.st);
  return GC_OK;
}

// The input gradient of conv_out (forward 8 -> C, so C -> 8 here) through the UNet's own exact-fp32 C -> 8 kernel (conv_in_kernel
// without its message chunk): the table entry [(oc * 9 + flipped tap)][8] is that kernel's [ic][tap][oc] layout over C channels.  The
// general implicit-GEMM kernel pads the 8 outputs to 64: 248 us per call at 4 x 64 x 200 x 704.
inline int dgradC8_enqueue(const UNetBwdCall& b, const float* dy, const float* w_fwd, int C, float* out, int n, int H, int W) {
  const DgradEntry& e = (*b.dg)[b.dg_next++];
  if (b.raw + e.w_off != w_fwd || e.cout != C || e.cin != 8 || e.ic0 != 0 || e.nic != 8)
    return fail(GC_ERR_ARG, "gencomm_unet_bwd: dgrad weight table out of step with the backward walk");
  ConvInArgs a{nullptr, dy, b.F(b.bw->wtmp) + e.p_off, b.F(b.bw->zeros), out, nullptr, C, H, W, b.c.m.xcd()};
  Modes mf = b.c.m;
  mf.v[MODE_ARITH] = 1;
  const TileCfg tc = pick_tile(mf, n, H, W);
  int tw, th;
  tile_dims(tc, &tw, &th);
  const dim3 grid(cdiv(W, tw), cdiv(H, th), n);
  if (tc == TILE_64x16) conv_in_kernel<64, 16, 4><<<grid, 256, 0, b.c.st>>>(a);
  else if (tc == TILE_32x16) conv_in_kernel<32, 16, 4><<<grid, 128, 0, b.c.st>>>(a);
  else conv_in_kernel<32, 8, 1><<<grid, 256, 0, b.c.st>>>(a);
  return GC_OK;
}

// conv_in's gradient with respect to x_t (forward C -> 8, so 8 -> C here) through the UNet's own exact-fp32 8 -> C kernel (conv_out_kernel
// without its GroupNorm): the general implicit-GEMM kernel stages 16-pixel row segments one float at a time and took 206 us per call at
// 4 x 64 x 200 x 704 for 162 MB of traffic.  The table entry is in that kernel's blocked layout (prep_dgrad_all_kernel).
inline int dgrad8C_enqueue(const UNetBwdCall& b, const float* dy, const float* w_fwd, int Cin_f, int ic0, int C, float* out, int n, int H, int W) {
  const DgradEntry& e = (*b.dg)[b.dg_next++];
  if (b.raw + e.w_off != w_fwd || e.cout != 8 || e.cin != Cin_f || e.ic0 != ic0 || e.nic != C || !e.blocked)
    return fail(GC_ERR_ARG, "gencomm_unet_bwd: dgrad weight table out of step with the backward walk");
  ConvOutArgs co{};
  co.src = dy; co.w = b.F(b.bw->wtmp) + e.p_off; co.bias = b.F(b.bw->zeros); co.out = out;
  co.C = C; co.H = H; co.W = W; co.xcd = b.c.m.xcd();
  Modes mf = b.c.m;
  mf.v[MODE_ARITH] = 1;
  const int nocb = (C + 15) / 16;
  const TileCfg tc = pick_tile(mf, n, H, W, nocb) == TILE_64x16 ? TILE_64x16 : TILE_32x8;
  int tw, th;
  tile_dims(tc, &tw, &th);
  const dim3 grid(cdiv(W, tw), cdiv(H, th), n * nocb);
  if (b.chain_prev != nullptr) {
    // the sampler chain's adjoint d x_t = alpha (this gradient) + beta d x_{t-1} in the epilogue (the forward's fused update with
    // coef1 = alpha, coef2 = beta, sigma = 0: the noise operand is read and multiplied by zero)
    float* row = b.F(b.bw->chain);
    set_chain_row_kernel<<<1, 64, 0, b.c.st>>>(row, b.chain_alpha, b.chain_beta);
    co.xt = b.chain_prev; co.noise = b.chain_prev; co.sched = row;
    if (tc == TILE_64x16) conv_out_kernel<64, 16, 4, 1, false><<<grid, 256, 0, b.c.st>>>(co);
    else conv_out_kernel<32, 8, 1, 1, false><<<grid, 256, 0, b.c.st>>>(co);
    return GC_OK;
  }
  if (tc == TILE_64x16) conv_out_kernel<64, 16, 4, 0, false><<<grid, 256, 0, b.c.st>>>(co);
  else conv_out_kernel<32, 8, 1, 0, false><<<grid, 256, 0, b.c.st>>>(co);
  return GC_OK;
}

// weight gradient whose input is SiLU(GroupNorm(.)) of one or two 8-channel forward tensors, applied in the staging
inline void wgrad_gn_sources(const UNetBwdCall& b, WgradArgs& wa, int src0, int src1, const float* gamma, const float* beta, int gs, int HW) {
  wa.x0 = b.c.tensor_ptr(src0);
  wa.gn_stat[0] = b.c.stat_ptr(src0);
  if (src1 >= 0) { wa.x1 = b.c.tensor_ptr(src1); wa.gn_stat[1] = b.c.stat_ptr(src1); }
  wa.gn_gamma = gamma; wa.gn_beta = beta; wa.gn_gs = gs; wa.gn_inv_cnt = 1.0 / ((double)gs * HW);
}

// SiLU(GN(x)) of one 8-channel source into A at channel offset coff (ctotal channels)
inline void gn_fwd_enqueue(const UNetBwdCall& b, int src_id, const float* gamma, const float* beta, int gs, int HW, float* A, int ctotal, int coff) {
  GnArgs g{};
  g.x = b.c.tensor_ptr(src_id); g.stat = b.c.stat_ptr(src_id); g.gamma = gamma; g.beta = beta; g.out = A;
  g.inv_cnt = 1.0 / ((double)gs * HW); g.gs = gs; g.HW = HW; g.out_ctotal = ctotal; g.out_coff = coff;
  gn_silu_fwd_kernel<<<dim3(cdiv(HW, 256), 8, b.c.n), 256, 0, b.c.st>>>(g);
}
// G[src] += backward of SiLU(GN(src)) given dA (channels coff.. of a ctotal-channel tensor); d gamma / d beta accumulated
// add: optional further gradient of the same tensor, summed in the same pass (the identity shortcut's d out: saves the axpy launch)
inline int gn_bwd_enqueue(const UNetBwdCall& b, int src_id, const float* gamma, const float* beta, int gs, int HW, const float* dA, int ctotal,
                          int coff, long long dgamma_off, long long dbeta_off, bool fused = false, const float* add = nullptr) {
  hipStream_t st = b.c.st;
  const int u = b.gp.uses;
  if (u >= kMaxGnUses) return fail(GC_ERR_ARG, "gencomm_unet_bwd: too many GroupNorm uses");
  double* red = reinterpret_cast<double*>(b.c.wsp + b.bw->red) + (size_t)u * b.c.n * 16;   // zeroed once per call
  b.gp.dgamma[u] = (int)dgamma_off; b.gp.dbeta[u] = (int)dbeta_off; b.gp.uses = u + 1;
  GnArgs g{};
  g.x = b.c.tensor_ptr(src_id); g.stat = b.c.stat_ptr(src_id); g.gamma = gamma; g.beta = beta; g.da = dA; g.out = b.G(src_id); g.red = red;
  g.inv_cnt = 1.0 / ((double)gs * HW); g.gs = gs; g.HW = HW; g.da_ctotal = ctotal; g.da_coff = coff;
  g.da_is_dz = fused ? 1 : 0;   // fused: dgrad8_enqueue's epilogue wrote d z and this slot's sums
  g.add = add;
  g.first = b.take_first(src_id) ? 1 : 0;
  if (!fused) gn_silu_bwd_reduce_kernel<<<dim3(std::min(cdiv(HW, 256), 64), 8, b.c.n), 256, 0, st>>>(g);
  if (kBwdVec4 && (HW & 3) == 0) gn_silu_bwd_apply4_kernel<<<dim3(cdiv(HW / 4, 256), 8, b.c.n), 256, 0, st>>>(g);   // workspace maps are 256-byte aligned
  else gn_silu_bwd_apply_kernel<<<dim3(cdiv(HW, 256), 8, b.c.n), 256, 0, st>>>(g);
  return GC_OK;
}

// The library's side stream of a device (GENCOMM_MODE_BWD_STREAMS): created on first use, kept for the life of the process.  A call forks
// work onto it with `fork` (recorded on the caller's stream, waited for by the side stream) and joins with `join` (the reverse).
// Two host threads may run gencomm_unet_bwd on one device (two autograd graphs, multi-stream training): the stream and the event pair
// are shared, so every record + wait PAIR is made under `mu` -- hipStreamWaitEvent binds to the most recent record of the event at
// the time of the call, and with the pair atomic that record is the caller's own (the two callers' side work then simply queues on
// the one side stream in call order).
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  std::mutex mu;
  int fork_from(hipStream_t st) {   // side stream waits for everything enqueued on st so far
    std::lock_guard<std::mutex> lock(mu);
    GC_HIP(hipEventRecord(fork, st));
    GC_HIP(hipStreamWaitEvent(s, fork, 0));
    return GC_OK;
  }
  int join_into(hipStream_t st) {   // st waits for everything enqueued on the side stream so far
    std::lock_guard<std::mutex> lock(mu);
    GC_HIP(hipEventRecord(join, s));
    GC_HIP(hipStreamWaitEvent(st, join, 0));
    return GC_OK;
  }
};
inline int side_stream(SideStream** out) {
  static std::mutex mu;
  static SideStream tab[64];
  int dev = 0;
  GC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return fail(GC_ERR_ARG, "side_stream: device index out of range");
  std::lock_guard<std::mutex> lock(mu);
  SideStream& e = tab[dev];
  if (e.s == nullptr) {
    hipStream_t s = nullptr;
    hipEvent_t f = nullptr, j = nullptr;
    GC_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    GC_HIP(hipEventCreateWithFlags(&f, hipEventDisableTiming));
    GC_HIP(hipEventCreateWithFlags(&j, hipEventDisableTiming));
    e.fork = f; e.join = j; e.s = s;
  }
  *out = &e;
  return GC_OK;
}

inline int unet_bwd_walk(const UNetBwdCall& b, const float* x_t, const float* cond, const float* grad_x0, float* grad_xt, float* grad_cond,
                         SideStream* ss);

inline int unet_bwd_enqueue(const UNetBwdCall& b, const float* x_t, const float* cond, int t, const float* grad_x0,
                            float* grad_xt, float* grad_cond, bool forward_done) {
  const UNetCall& c = b.c;
  const UNetPlan& p = *c.plan;
  hipStream_t st = c.st;
  const int n = c.n, C = p.C;
  // ---- forward with every intermediate kept (x0_hat itself is not needed: it lands in grad_xt, overwritten below)
  if (!forward_done) {   // otherwise gencomm_unet_fwd_train left them in this workspace
    GC_HIP(hipMemsetAsync(c.amax(), 0, 256, st));
    amax_kernel<<<256, 256, 0, st>>>(cond, (long long)n * 2 * c.H * c.W, c.amax());
    amax_kernel<<<1024, 256, 0, st>>>(x_t, (long long)n * C * c.H * c.W, c.amax() + 1);
    ConvOutArgs co{};
    co.out = grad_xt;
    if (int rc = unet_enqueue(c, x_t, cond, t, 0, co)) return rc;
  }
  // ---- gradient buffers, constants
  b.g_written.assign(p.tensors.size(), 0);   // no memset of the gradient maps: first contributions write (UNetBwdCall::take_first)
  if (!kBwdFirstWrite) {
    size_t gbytes = 0;
    for (int l = 0; l < p.L; ++l) gbytes += c.ws->slot_bytes[l] * p.slots_per_level[l];
    GC_HIP(hipMemsetAsync(c.wsp + b.bw->g_base, 0, gbytes, st));
  }
  GC_HIP(hipMemsetAsync(c.wsp + b.bw->red, 0, (size_t)kMaxGnUses * n * 16 * sizeof(double), st));
  {
    const std::vector<DgradEntry>& e = *b.dg;
    if ((int)e.size() > kMaxDgradLayers) return fail(GC_ERR_ARG, "gencomm_unet_bwd: too many layers");
    PrepDgradArgs pa{};
    pa.raw = b.raw; pa.P = b.F(b.bw->wtmp); pa.layers = (int)e.size();
    int most = 0;
    for (size_t i = 0; i < e.size(); ++i) {
      pa.w_off[i] = (int)e[i].w_off; pa.p_off[i] = (int)e[i].p_off;
      pa.cout[i] = (short)e[i].cout; pa.cin[i] = (short)e[i].cin; pa.ic0[i] = (short)e[i].ic0; pa.nic[i] = (short)e[i].nic;
      pa.blocked[i] = (unsigned char)e[i].blocked;
      most = std::max(most, (int)e[i].floats);
    }
    prep_dgrad_all_kernel<<<dim3(std::min(cdiv(most, 256), 8), (unsigned)e.size()), 256, 0, st>>>(pa);
  }
  b.dg_next = 0;
  b.gp = GnParamArgs{};
  b.gp.red = reinterpret_cast<const double*>(c.wsp + b.bw->red); b.gp.graw = b.graw; b.gp.n = n;
  fill_kernel<<<cdiv(C + 16, 256), 256, 0, st>>>(b.F(b.bw->ones), 1.0f, C + 16);
  GC_HIP(hipMemsetAsync(b.F(b.bw->zeros), 0, (size_t)(C + 16) * sizeof(float), st));
  // the weight gradients depend on nothing the walk produces after their own layer and nothing depends on them before the timestep MLP
  // below: they go to the side stream, one after the other (they share the partial-sum scratch), beside the input-gradient chain
  SideStream* ss = nullptr;
  // automatic: only where the step is GPU-bound -- on small maps (the shipped 64 x 128) the two event calls per layer cost the host more
  // than the overlap returns (7.1 -> 8.0 ms per step measured), on 4 x 200 x 704 the step goes from 15.2 to 13.7 ms
  const long long bs = c.m.v[MODE_BWD_STREAMS];
  if (bs == 2 || (bs == 1 && (long long)n * c.H * c.W >= (1ll << 17)))
    if (int rc = side_stream(&ss)) return rc;
  const int walk_rc = unet_bwd_walk(b, x_t, cond, grad_x0, grad_xt, grad_cond, ss);
  if (ss != nullptr) {   // joined on every path: the caller's buffers are ordered on its own stream when this call returns
    if (int rc = ss->join_into(st)) return rc;
  }
  if (walk_rc) return walk_rc;
  if (b.gp.uses > 0) gn_param_grad_all_kernel<<<b.gp.uses, 64, 0, st>>>(b.gp);
  {
    // timestep MLP: needs every block's d conv1.bias (the wgrad launches above), adds the temb.dense / temb_proj gradients
    TembBwdArgs ta{};
    ta.raw = b.raw; ta.graw = b.graw; ta.d0w = p.d0w; ta.d0b = p.d0b; ta.d1w = p.d1w; ta.d1b = p.d1b;
    ta.nblocks = (int)p.blocks.size(); ta.t = t;
    for (size_t i = 0; i < p.blocks.size(); ++i) { ta.tpw[i] = p.blocks[i].tpw; ta.tpb[i] = p.blocks[i].tpb; ta.c1b[i] = p.blocks[i].c1b; }
    temb_bwd_kernel<<<1, 64, 0, st>>>(ta);
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}

inline int unet_bwd_walk(const UNetBwdCall& b, const float* x_t, const float* cond, const float* grad_x0, float* grad_xt, float* grad_cond,
                         SideStream* ss) {
  const UNetCall& c = b.c;
  const UNetPlan& p = *c.plan;
  hipStream_t st = c.st;
  const int n = c.n, C = p.C;
  float* DA = b.F(b.bw->DA);
  auto conv_wgrad_enqueue = [&](const WgradArgs& wa, int nn, hipStream_t) -> int {   // shadows the free function for the launches below
    if (ss == nullptr) return gc::conv_wgrad_enqueue(wa, nn, st);
    if (int rc = ss->fork_from(st)) return rc;
    return gc::conv_wgrad_enqueue(wa, nn, ss->s);
  };

  for (int oi = (int)p.ops.size() - 1; oi >= 0; --oi) {
    const Op& o = p.ops[oi];
    const int Hl = c.ws->Hl[o.level], Wl = c.ws->Wl[o.level], HW = Hl * Wl;
    switch (o.kind) {
      case OP_CONV_OUT: {
        // y = conv_out(SiLU(GN_out(h))) + b
        WgradArgs wa{grad_x0, nullptr, nullptr, b.graw + p.conv_out.w, b.graw + p.conv_out.b, C, 8, 0, Hl, Wl, Hl, Wl, 3, 1, 1, 0};
        wgrad_gn_sources(b, wa, o.src[0], -1, b.raw + p.nout_w, b.raw + p.nout_b, 2, HW);
        wa.part = b.F(b.bw->wpart);
        if (int rc = conv_wgrad_enqueue(wa, n, st)) return rc;
        if (int rc = dgradC8_enqueue(b, grad_x0, b.raw + p.conv_out.w, C, DA, n, Hl, Wl)) return rc;
        if (int rc = gn_bwd_enqueue(b, o.src[0], b.raw + p.nout_w, b.raw + p.nout_b, 2, HW, DA, 8, 0, p.nout_w, p.nout_b)) return rc;
        break;
      }
      case OP_RES_CONV2: {
        // out = shortcut(in) + conv2(SiLU(GN2(tmp))) + b2 (+ nin bias)
        const ResBlockPlan& rb = p.blocks[o.blk];
        if (int rc = b.need_written(o.dst)) return rc;
        const float* go = b.G(o.dst);
        WgradArgs wa{go, nullptr, nullptr, b.graw + rb.c2w, b.graw + rb.c2b, 8, 8, 0, Hl, Wl, Hl, Wl, 3, 1, 1, 0};
        wgrad_gn_sources(b, wa, o.src[0], -1, b.raw + rb.n2w, b.raw + rb.n2b, 2, HW);
        wa.part = b.F(b.bw->wpart);
        if (int rc = conv_wgrad_enqueue(wa, n, st)) return rc;
        if (int rc = dgrad8_enqueue(b, go, b.raw + rb.c2w, 8, 0, DA, n, Hl, Wl, o.src[0], b.raw + rb.n2w, b.raw + rb.n2b, 2)) return rc;
        if (int rc = gn_bwd_enqueue(b, o.src[0], b.raw + rb.n2w, b.raw + rb.n2b, 2, HW, DA, 8, 0, rb.n2w, rb.n2b, true)) return rc;
        if (rb.cin == 8) {
          // identity shortcut: d in += d out rides in conv1's GroupNorm-backward apply below (OP_RES_CONV1 of this block, the next op of the
          // walk: both add to G[in]); only a block whose conv1 does not follow directly keeps the separate pass
          const bool rides = kBwdVec4 && oi > 0 && p.ops[oi - 1].kind == OP_RES_CONV1 && p.ops[oi - 1].blk == o.blk && p.ops[oi - 1].src[0] == o.res[0];
          if (!rides) axpy_kernel<<<cdiv(n * 8 * HW, 256), 256, 0, st>>>(b.G(o.res[0]), go, 1.0f, (long long)n * 8 * HW, b.take_first(o.res[0]) ? 1 : 0);
        } else {
          WgradArgs wn{go, c.tensor_ptr(o.res[0]), c.tensor_ptr(o.res[1]), b.graw + rb.ninw, b.graw + rb.ninb, 8, 8, 8, Hl, Wl, Hl, Wl, 1, 1, 0, 0};
          wn.part = b.F(b.bw->wpart);
          if (int rc = conv_wgrad_enqueue(wn, n, st)) return rc;
          const int f0 = b.take_first(o.res[0]) ? 1 : 0, f1 = b.take_first(o.res[1]) ? 1 : 0;
          if (kBwdVec4 && (HW & 3) == 0) nin_dgrad4_kernel<<<dim3(cdiv(HW / 4, 256), n), 256, 0, st>>>(go, b.raw + rb.ninw, b.G(o.res[0]), b.G(o.res[1]), HW, f0, f1);
          else nin_dgrad_kernel<<<dim3(cdiv(HW, 256), n), 256, 0, st>>>(go, b.raw + rb.ninw, b.G(o.res[0]), b.G(o.res[1]), HW, f0, f1);
        }
        break;
      }
      case OP_RES_CONV1: {
        // tmp = conv1(SiLU(GN1(cat in))) + b1 + temb_proj(...)  (the timestep term only shifts the bias: its gradient is d b1)
        const ResBlockPlan& rb = p.blocks[o.blk];
        if (int rc = b.need_written(o.dst)) return rc;
        const float* gt = b.G(o.dst);
        const int nsrc = rb.cin / 8, gs = rb.cin == 8 ? 2 : 4;
        WgradArgs wa{gt, nullptr, nullptr, b.graw + rb.c1w, b.graw + rb.c1b, 8, 8, nsrc == 2 ? 8 : 0, Hl, Wl, Hl, Wl, 3, 1, 1, 0};
        wgrad_gn_sources(b, wa, o.src[0], nsrc == 2 ? o.src[1] : -1, b.raw + rb.n1w, b.raw + rb.n1b, gs, HW);
        wa.part = b.F(b.bw->wpart);
        if (int rc = conv_wgrad_enqueue(wa, n, st)) return rc;
        // identity shortcut of this block (cin == 8): d in += d out, skipped by OP_RES_CONV2 above when this op follows it directly
        const float* shortcut = nullptr;
        if (kBwdVec4 && rb.cin == 8 && oi + 1 < (int)p.ops.size() && p.ops[oi + 1].kind == OP_RES_CONV2 && p.ops[oi + 1].blk == o.blk && p.ops[oi + 1].res[0] == o.src[0])
          shortcut = b.G(p.ops[oi + 1].dst);
        for (int s = 0; s < nsrc; ++s) {   // one 8-channel input gradient per source (its epilogue = GroupNorm backward's reductions), then the apply
          if (int rc = dgrad8_enqueue(b, gt, b.raw + rb.c1w, rb.cin, 8 * s, DA + (size_t)s * n * 8 * HW, n, Hl, Wl, o.src[s], b.raw + rb.n1w + 8 * s,
                                      b.raw + rb.n1b + 8 * s, gs)) return rc;
          if (int rc = gn_bwd_enqueue(b, o.src[s], b.raw + rb.n1w + 8 * s, b.raw + rb.n1b + 8 * s, gs, HW, DA + (size_t)s * n * 8 * HW, 8, 0,
                                      rb.n1w + 8 * s, rb.n1b + 8 * s, true, s == 0 ? shortcut : nullptr)) return rc;
        }
        break;
      }
      case OP_DOWN: {
        const int lin = o.level - 1, Hi = c.ws->Hl[lin], Wi = c.ws->Wl[lin];
        if (int rc = b.need_written(o.dst)) return rc;
        const float* gd = b.G(o.dst);
        WgradArgs wa{gd, c.tensor_ptr(o.src[0]), nullptr, b.graw + p.down[lin].w, b.graw + p.down[lin].b, 8, 8, 0, Hl, Wl, Hi, Wi, 3, 2, 0, 0};
        wa.part = b.F(b.bw->wpart);
        if (int rc = conv_wgrad_enqueue(wa, n, st)) return rc;
        DownDgradArgs da{gd, b.raw + p.down[lin].w, b.G(o.src[0]), Hl, Wl, Hi, Wi, b.take_first(o.src[0]) ? 1 : 0};
        down_dgrad_kernel<<<dim3(cdiv(Hi * Wi, 256), 1, n), 256, 0, st>>>(da);
        break;
      }
      case OP_UP: {
        const int lin = o.level + 1, Hs = c.ws->Hl[lin], Ws = c.ws->Wl[lin];
        if (int rc = b.need_written(o.dst)) return rc;
        const float* gu = b.G(o.dst);
        WgradArgs wa{gu, c.tensor_ptr(o.src[0]), nullptr, b.graw + p.up[lin].w, b.graw + p.up[lin].b, 8, 8, 0, Hl, Wl, Hl, Wl, 3, 1, 1, 1};
        wa.part = b.F(b.bw->wpart);
        if (int rc = conv_wgrad_enqueue(wa, n, st)) return rc;
        if (int rc = dgrad8_enqueue(b, gu, b.raw + p.up[lin].w, 8, 0, DA, n, Hl, Wl)) return rc;
        sum2x2_add_kernel<<<cdiv(n * 8 * Hs * Ws, 256), 256, 0, st>>>(DA, b.G(o.src[0]), n * 8, Hs, Ws, b.take_first(o.src[0]) ? 1 : 0);
        break;
      }
      case OP_CONV_IN: {
        if (int rc = b.need_written(o.dst)) return rc;
        const float* gh = b.G(o.dst);
        WgradArgs wa{gh, cond, x_t, b.graw + p.conv_in.w, b.graw + p.conv_in.b, 8, 2, C, Hl, Wl, Hl, Wl, 3, 1, 1, 0};
        wa.part = b.F(b.bw->wpart);
        if (int rc = conv_wgrad_enqueue(wa, n, st)) return rc;
        {  // d cond: the general kernel pads 2 output channels to 64 (124 us at 4 x 200 x 704); the 8 -> 8 kernel computes channels 0..7, two are kept
          if (int rc = dgrad8_enqueue(b, gh, b.raw + p.conv_in.w, C + 2, 0, DA, n, Hl, Wl)) return rc;
          SliceArgs sa{DA, nullptr, nullptr, nullptr, grad_cond, nullptr, n, 2, Hl * Wl, 8, 0, 0, 0, 2, 0, 0, 0};
          ew_slice_kernel<EW_COPY><<<dim3(cdiv(Hl * Wl, 256), 2, n), 256, 0, st>>>(sa);
        }
#ifdef GC_DIAG_IGEMM_CONVIN_DGRAD
        if (int rc = dgrad3x3_enqueue(b, gh, b.raw + p.conv_in.w, 8, C + 2, 2, C, grad_xt, C, 0, n, Hl, Wl)) return rc;
#else
        if (int rc = dgrad8C_enqueue(b, gh, b.raw + p.conv_in.w, C + 2, 2, C, grad_xt, n, Hl, Wl)) return rc;
#endif
        break;
      }
      case OP_ATTN:
        return fail(GC_ERR_ARG, "gencomm_unet_bwd: AttnBlock backward is not implemented (attn_mask must be 0)");
    }
  }
  return GC_OK;
}

}  // namespace gc
