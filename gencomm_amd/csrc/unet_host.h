// Host launch logic of the UNet / sampler (stream-ordered, allocation-free, capture-safe).
#pragma once
#include <stdlib.h>

#include "attn_kernels.h"
#include "conv8b_kernels.h"
#include "conv8h_kernels.h"
#include "conv8h8_kernels.h"
#include "dataflow_kernels.h"
#include "latent_kernels.h"
#include "latenth_kernels.h"
#include "unet_kernels.h"
#include "unet_plan.h"

namespace gc {

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Tile choice. 64x16 px / 256 threads / 4 px per lane (least LDS traffic per FMA) when that
// still gives every one of the 256 CUs ~2 workgroups; else 32x16 / 128 threads; maps that are
// smaller still run 32x8 px / 256 threads / ONE pixel per lane, which has 4x the waves and a
// quarter of the serial work per wave (these layers are latency-bound, not throughput-bound).
enum TileCfg { TILE_64x16 = 0, TILE_32x16 = 1, TILE_32x8 = 2 };
// Arithmetic mode (Modes, common.h): split (default) = 64x16 tiles of the 8-channel convolutions run conv8h_kernel (fp16 hi/lo
// split on the f16 matrix pipe, fp32-grade products); exact = the fp32 conv8_kernel everywhere.
inline long long tile_want(const Modes& m) {
  // minimum number of 64x16 workgroups before the 64x16-tile kernels are chosen.  512 for the fp32 kernels (two
  // workgroups per CU); 160 for the f16-pipe kernels, whose workgroups are short enough that a partly filled chip beats
  // the smaller fp32 tiles (1 scene x 1 stream 80.9 -> 86.4 scenes/s, 2 x 2 131.6 -> 141.3, 4 x 3 unchanged).
  // MODE_TILE_WANT overrides (tuning / tests: 1 forces the 64x16 kernels onto small maps).
  return m.v[MODE_TILE_WANT] > 0 ? m.v[MODE_TILE_WANT] : (m.split() ? 160LL : 512LL);
}
inline TileCfg pick_tile(const Modes& m, int n, int H, int W, int zmul = 1) {
  if (m.bf16()) return TILE_64x16;  // the bf16 mode has one kernel family
  const long long want = tile_want(m);
  if ((long long)cdiv(W, 64) * cdiv(H, 16) * n * zmul >= want) return TILE_64x16;
  if ((long long)cdiv(W, 32) * cdiv(H, 16) * n * zmul >= want) return TILE_32x16;
  return TILE_32x8;
}
inline void tile_dims(TileCfg t, int* tw, int* th) {
  *tw = t == TILE_64x16 ? 64 : 32;
  *th = t == TILE_32x8 ? 8 : 16;
}

// CUs of the current device (cached per device index): the persistent kernels are launched with 3 workgroups per CU
inline int device_cu_count() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cus[dev] = v;
  }
  return cus[dev];
}

template <int NSRC, bool GN, bool UP, int RES>
inline void launch_conv8(const Modes& m, TileCfg t, const Conv8Args& a, int n, hipStream_t st) {
  // algorithmic bytes: the source map(s), the residual source(s) and the destination, once each, fp32
  const double abytes = 4.0 * n * ((double)NSRC * 8 * a.Hin * a.Win + (RES == 1 ? 8.0 : RES == 2 ? 16.0 : 0.0) * a.H * a.W + 8.0 * a.H * a.W);
  TimedLaunch tl(UP ? KF_UP : (NSRC == 2 ? KF_CONV16 : (RES == 1 ? KF_CONV8_RES1 : RES == 2 ? KF_CONV8_RES2 : KF_CONV8)), st, abytes);
  int tw, th;
  tile_dims(t, &tw, &th);
  const dim3 grid(cdiv(a.W, tw), cdiv(a.H, th), n);
  const int variant = UP ? 16 : (RES == 2 ? 8 : (RES == 1 ? 4 : (NSRC == 2 ? 2 : 1)));  // MODE_CONV8H_MASK: diagnostic
  if constexpr (RES != 3)   // RES 3 = the backward's GroupNorm epilogue: exact-fp32 kernel only
  if (t == TILE_64x16 && a.wh != nullptr && m.split() && (m.v[MODE_CONV8H_MASK] & variant)) {
    if constexpr (GN && !UP) {   // 64 x 8 tiles for launches with few workgroups (MODE_TILE8: the half-resolution level)
      // automatic (-1): launches below a third of a resident round (256 workgroups of 64 x 16: the half-resolution level of ONE four-agent
      // scene, 168) take the 64 x 8 tiles -- single-scene latency 10.07 -> 9.78 ms, throughput at 16 agents per launch unchanged (its
      // half-resolution launches have 672); not when the caller forces a tile size (MODE_TILE_WANT != 0: tests of the 64 x 16 kernels)
      const long long t8 = m.v[MODE_TILE8] >= 0 ? m.v[MODE_TILE8] : (m.v[MODE_TILE_WANT] == 0 ? 256 : 0);
      if (t8 > 0 && (long long)grid.x * grid.y * grid.z < t8 && (a.W & 3) == 0 && a.W >= 4) {
        GC_KLOG(NSRC == 2 ? "conv8h8_kernel<2,GN,0> (64x8 tiles)" : RES == 2 ? "conv8h8_kernel<1,GN,2> (64x8 tiles)" : RES == 1 ? "conv8h8_kernel<1,GN,1> (64x8 tiles)" : "conv8h8_kernel<1,GN,0> (64x8 tiles)");
        conv8h8_kernel<NSRC, GN, RES><<<dim3(cdiv(a.W, 64), cdiv(a.H, 8), n), 256, 0, st>>>(a);
        return;
      }
    }
    if constexpr (GN || UP) {   // persistent form (MODE_PERSIST): more tiles than resident slots, vector widths
      const long long tiles = (long long)grid.x * grid.y * grid.z;
      const int slots = 3 * device_cu_count();
      if ((m.v[MODE_PERSIST] & variant) && tiles > slots && (UP ? (a.Win & 1) == 0 : (a.W & 3) == 0)) {
        GC_KLOG(UP ? "conv8hp_kernel<1,0,UP,0> (persistent)" : NSRC == 2 ? "conv8hp_kernel<2,GN,0,0> (persistent)" : RES == 2 ? "conv8hp_kernel<1,GN,0,2> (persistent)"
                   : RES == 1 ? "conv8hp_kernel<1,GN,0,1> (persistent)" : "conv8hp_kernel<1,GN,0,0> (persistent)");
        conv8hp_kernel<NSRC, GN, UP, RES><<<dim3((unsigned)slots), 256, 0, st>>>(a, (int)grid.x, (int)grid.y, (int)grid.z);
        return;
      }
    }
    GC_KLOG(UP ? "conv8h_kernel<1,0,UP,0>" : NSRC == 2 ? "conv8h_kernel<2,GN,0,0>" : RES == 2 ? "conv8h_kernel<1,GN,0,2>" : RES == 1 ? "conv8h_kernel<1,GN,0,1>" : GN ? "conv8h_kernel<1,GN,0,0>" : "conv8h_kernel<1,0,0,0>");
    conv8h_kernel<NSRC, GN, UP, RES><<<grid, 256, 0, st>>>(a);
    return;
  }
  GC_KLOG(t == TILE_64x16 ? "conv8_kernel<64,16,4,...> (exact fp32)" : t == TILE_32x16 ? "conv8_kernel<32,16,4,...> (exact fp32)" : "conv8_kernel<32,8,1,...> (exact fp32)");
  switch (t) {
    case TILE_64x16: conv8_kernel<64, 16, 4, NSRC, GN, UP, RES><<<grid, 256, 0, st>>>(a); break;
    case TILE_32x16: conv8_kernel<32, 16, 4, NSRC, GN, UP, RES><<<grid, 128, 0, st>>>(a); break;
    default: conv8_kernel<32, 8, 1, NSRC, GN, UP, RES><<<grid, 256, 0, st>>>(a); break;
  }
}

template <int NSRC, bool GN, bool UP, int RES>
inline void launch_conv8b(const Conv8BArgs& a, int n, hipStream_t st) {
  const double elt = 2.0;  // bf16 maps (the fp32 hs0 map appears in two launches per call: counted as bf16, a lower bound)
  const double abytes = elt * n * ((double)NSRC * 8 * a.Hin * a.Win + (RES == 1 ? 8.0 : RES == 2 ? 16.0 : 0.0) * a.H * a.W + 8.0 * a.H * a.W);
  TimedLaunch tl(UP ? KF_UP : (NSRC == 2 ? KF_CONV16 : (RES == 1 ? KF_CONV8_RES1 : RES == 2 ? KF_CONV8_RES2 : KF_CONV8)), st, abytes);
  GC_KLOG("conv8b_kernel<...> (bf16 storage)");
  conv8b_kernel<NSRC, GN, UP, RES><<<dim3(cdiv(a.W, 64), cdiv(a.H, 16), n), 256, 0, st>>>(a);
}

struct UNetCall {
  const UNetPlan* plan;
  const UNetWorkspace* ws;
  const float* prepared;
  char* wsp;  // workspace base
  int n, H, W;
  hipStream_t st;
  Modes m;
  float* amax() const { return reinterpret_cast<float*>(wsp + ws->amax_off); }  // {max|cond|, max|x_t|}, set by the caller
  // bf16 denoise mode: every 8-channel map is bf16 except the sampler's carried state hs0 (tensor 0)
  int is_f32(int id) const { return (!m.bf16() || id == plan->hs0_tensor) ? 1 : 0; }

  float* tensor_ptr(int id) const {
    const TensorPlan& t = plan->tensors[id];
    return reinterpret_cast<float*>(wsp + ws->level_base[t.level] + ws->slot_bytes[t.level] * t.slot);
  }
  double* stat_ptr(int id) const { return reinterpret_cast<double*>(wsp) + (size_t)id * n * 16; }
};

// Launch arguments of the 8-channel layers (shared by the per-layer launches and the dataflow program)
inline Conv8Args conv1_args(const UNetCall& c, const Op& o, int t) {
  const ResBlockPlan& b = c.plan->blocks[o.blk];
  const float* P = c.prepared;
  const int Hl = c.ws->Hl[o.level], Wl = c.ws->Wl[o.level];
  Conv8Args a{};
  a.src[0] = c.tensor_ptr(o.src[0]); a.sstat[0] = c.stat_ptr(o.src[0]);
  if (o.src[1] >= 0) { a.src[1] = c.tensor_ptr(o.src[1]); a.sstat[1] = c.stat_ptr(o.src[1]); }
  a.gamma = P + b.n1w; a.beta = P + b.n1b;
  a.w = P + b.p_c1w; a.wh = P + b.p_c1wh; a.bias = P + b.p_bias1 + (size_t)t * 8;
  a.dst = c.tensor_ptr(o.dst); a.dstat = c.stat_ptr(o.dst);
  a.H = a.Hin = Hl; a.W = a.Win = Wl;
  a.inv_cnt = 1.0 / ((b.cin == 8 ? 2.0 : 4.0) * Hl * Wl);
  a.xcd = c.m.xcd();
  return a;
}
inline Conv8Args conv2_args(const UNetCall& c, const Op& o) {
  const ResBlockPlan& b = c.plan->blocks[o.blk];
  const float* P = c.prepared;
  const int Hl = c.ws->Hl[o.level], Wl = c.ws->Wl[o.level];
  Conv8Args a{};
  a.src[0] = c.tensor_ptr(o.src[0]); a.sstat[0] = c.stat_ptr(o.src[0]);
  a.gamma = P + b.n2w; a.beta = P + b.n2b;
  a.w = P + b.p_c2w; a.wh = P + b.p_c2wh; a.bias = P + b.p_bias2;
  a.res[0] = c.tensor_ptr(o.res[0]);
  if (o.res[1] >= 0) { a.res[1] = c.tensor_ptr(o.res[1]); a.ninw = P + b.p_ninw; }
  a.dst = c.tensor_ptr(o.dst); a.dstat = c.stat_ptr(o.dst);
  a.H = a.Hin = Hl; a.W = a.Win = Wl;
  a.inv_cnt = 1.0 / (2.0 * Hl * Wl);
  a.xcd = c.m.xcd();
  return a;
}
inline Conv8Args up_args(const UNetCall& c, const Op& o) {
  const UNetPlan& p = *c.plan;
  const float* P = c.prepared;
  const int lin = o.level + 1;
  Conv8Args a{};
  a.src[0] = c.tensor_ptr(o.src[0]);
  a.sstat[0] = c.stat_ptr(o.src[0]);  // raw input: range bound from its sum of squares (conv8h_kernel)
  a.w = P + p.up[lin].p_w; a.wh = P + p.up[lin].p_wh; a.bias = P + p.up[lin].b;
  a.dst = c.tensor_ptr(o.dst); a.dstat = c.stat_ptr(o.dst);
  a.H = c.ws->Hl[o.level]; a.W = c.ws->Wl[o.level]; a.Hin = c.ws->Hl[lin]; a.Win = c.ws->Wl[lin];
  a.xcd = c.m.xcd();
  return a;
}
inline DownArgs down_args(const UNetCall& c, const Op& o) {
  const UNetPlan& p = *c.plan;
  const float* P = c.prepared;
  const int lin = o.level - 1;
  return DownArgs{c.tensor_ptr(o.src[0]), P + p.down[lin].p_w, P + p.down[lin].b, c.tensor_ptr(o.dst), c.stat_ptr(o.dst),
                  c.ws->Hl[o.level], c.ws->Wl[o.level], c.ws->Hl[lin], c.ws->Wl[lin], !c.is_f32(o.src[0]), !c.is_f32(o.dst)};
}

// ---- ResnetBlock fusion, emulated (GENCOMM_MODE_RESFUSE_EMU = 1).  TIMING EXPERIMENT, never a product path: the
// outputs are not the UNet's.  A fused 8 -> 8 ResnetBlock (unet.py:119-138) would be (A) a statistics-only pass of conv1 -- GroupNorm2
// needs the sums over the whole map before conv2 can start -- and (B) one kernel that stages the block input, recomputes conv1 on the
// tile + ring, applies GroupNorm2 + SiLU in registers and runs conv2 with the residual epilogue: 72 + 144 MB instead of 144 + 216 MB per
// 16-agent launch pair.  The emulation launches exactly that traffic and at least that matrix work with the existing tile function:
// (A) = conv1 without its store; (B) = the conv2 variant reading the block input as its source (its residual is the same tensor) with
// the matrix phase run twice.  It leaves out what the real kernel adds on top (a 20 x 68 instead of 18 x 66 input tile, the 1.3x
// ring, one LDS round trip of the intermediate tile and two barriers), so it is an UPPER bound of the fusion's gain.
inline bool resfuse_emulated(const UNetCall& c, const ResBlockPlan& b, TileCfg tc, bool wvec) {
  return c.m.v[MODE_RESFUSE_EMU] != 0 && b.cin == 8 && tc == TILE_64x16 && c.m.split() && !c.m.bf16() && wvec;
}

// ---- dataflow execution of ops [f, l) (dataflow_kernels.h).  Eligible: GENCOMM_MODE_DATAFLOW on, f16-pipe arithmetic (not the
// bf16 storage mode), every op of the range one of {ResnetBlock conv1 / conv2, Downsample, Upsample} on 64x16 tiles.
inline bool df_eligible(const UNetCall& c, int f, int l) {
  if (c.m.v[MODE_DATAFLOW] == 0 || !c.m.split() || c.m.bf16() || l - f < 2 || l - f > DF_MAX_OPS || c.n > 64 * 1024) return false;
  if (c.m.v[MODE_CONV8H_MASK] != -1) return false;
  for (int oi = f; oi < l; ++oi) {
    const Op& o = c.plan->ops[oi];
    if (o.kind != OP_RES_CONV1 && o.kind != OP_RES_CONV2 && o.kind != OP_DOWN && o.kind != OP_UP) return false;
    if (o.kind == OP_DOWN) {
      if ((c.ws->Wl[o.level - 1] & 3) != 0) return false;
    } else if (pick_tile(c.m, c.n, c.ws->Hl[o.level], c.ws->Wl[o.level]) != TILE_64x16) return false;
  }
  return true;
}
inline int df_enqueue(const UNetCall& c, int t, int f, int l, bool upload) {
  const UNetPlan& p = *c.plan;
  DfProgram* dprog = reinterpret_cast<DfProgram*>(c.wsp + c.ws->df_prog_off);
  unsigned* words = reinterpret_cast<unsigned*>(c.wsp + c.ws->df_words_off);
  const int nops = l - f;
  if (upload) {
    static thread_local DfProgram hp;  // host staging (3 KB per 8 ops travel as kernel arguments)
    hp.nops = nops; hp.n = c.n;
    int qlen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < nops; ++k) {
      const Op& o = p.ops[f + k];
      DfOp& d = hp.ops[k];
      d = DfOp{};
      const int Hl = c.ws->Hl[o.level], Wl = c.ws->Wl[o.level];
      if (o.kind == OP_DOWN) {
        d.kind = DF_DOWN; d.tx = cdiv(Hl * ((Wl + 1) / 2), 256); d.ty = 1; d.down = down_args(c, o);
      } else {
        d.tx = cdiv(Wl, 64); d.ty = cdiv(Hl, 16);
        if (o.kind == OP_RES_CONV1) {
          d.kind = p.blocks[o.blk].cin == 8 ? DF_CONV1 : DF_CONV16; d.conv = conv1_args(c, o, 0); d.bias_stride = 8;
        } else if (o.kind == OP_RES_CONV2) {
          d.kind = p.blocks[o.blk].cin == 8 ? DF_CONV2_RES1 : DF_CONV2_RES2; d.conv = conv2_args(c, o);
        } else {
          d.kind = DF_UP; d.conv = up_args(c, o);
        }
      }
      d.tiles = d.tx * d.ty;
      d.items = d.tiles * c.n;
      // contiguous eighths of the op's agent-major item order (the rule of xcd_block)
      const int qq = d.items >> 3, rr = d.items & 7;
      for (int x = 0; x <= 8; ++x) d.item_base[x] = x * qq + (x < rr ? x : rr);
      for (int x = 0; x < 8; ++x) { d.ticket_base[x] = qlen[x]; qlen[x] += d.item_base[x + 1] - d.item_base[x]; }
    }
    for (int x = 0; x < 8; ++x) hp.queue_len[x] = qlen[x];
    for (int k0 = 0; k0 < nops; k0 += DF_CHUNK) {
      DfUploadArgs ua{};
      ua.dst = dprog; ua.first = k0; ua.count = std::min(DF_CHUNK, nops - k0); ua.nops = nops; ua.n = c.n;
      for (int x = 0; x < 8; ++x) ua.queue_len[x] = qlen[x];
      for (int k = 0; k < ua.count; ++k) ua.ops[k] = hp.ops[k0 + k];
      df_upload_kernel<<<1, 64, 0, c.st>>>(ua);
    }
  }
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  DfArgs da{dprog, words, words + 8, words + 8 + 64 * (size_t)c.n, t, c.m.v[MODE_DATAFLOW] == 2 ? 1 : 0};
  GC_KLOG("unet_dataflow_kernel (persistent, per-agent layer dependencies)");
  TimedLaunch tl(KF_DATAFLOW, c.st);
  unet_dataflow_kernel<<<dim3(3 * cus), HC_NT, 0, c.st>>>(da);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

// Enqueue one UNet evaluation for integer timestep t.  The last op (conv_out) is launched with
// `post` (0 plain x0, 1 explicit noise, 2 Philox) and the pointers in `co` (xt/noise/sched/out/seed).
// ops [first, last) of the launch program; `keep_hs0_stats` leaves tensor 0's statistics alone
// (they were produced by the previous latent step).
inline int unet_enqueue_range(const UNetCall& c, const float* x_t, const float* cond, int t, int post,
                              ConvOutArgs co, int first, int last, bool keep_hs0_stats, bool df_upload = true) {
  const UNetPlan& p = *c.plan;
  const float* P = c.prepared;
  {
    const size_t skip = keep_hs0_stats ? (size_t)c.n * 16 * sizeof(double) : 0;  // hs0 is tensor 0
    GC_HIP(hipMemsetAsync(c.wsp + skip, 0, c.ws->stats_bytes - skip, c.st));
  }
  // the body (everything between conv_in and conv_out) as ONE persistent dataflow launch where eligible
  const int df_f = std::max(first, 1), df_l = std::min(last, (int)p.ops.size() - 1);
  const bool dataflow = df_eligible(c, df_f, df_l);
  for (int oi = first; oi < last; ++oi) {
    if (dataflow && oi == df_f) {
      if (int rc = df_enqueue(c, t, df_f, df_l, df_upload)) return rc;
      oi = df_l - 1;
      continue;
    }
    const Op& o = p.ops[oi];
    const int Hl = c.ws->Hl[o.level], Wl = c.ws->Wl[o.level];
    switch (o.kind) {
      case OP_CONV_IN: {
        ConvInArgs a{cond, x_t, P + p.conv_in.p_w, P + p.conv_in.b, c.tensor_ptr(o.dst), c.stat_ptr(o.dst), p.C, Hl, Wl, c.m.xcd()};
        const TileCfg tc = pick_tile(c.m, c.n, Hl, Wl);
        int tw, th;
        tile_dims(tc, &tw, &th);
        const dim3 grid(cdiv(Wl, tw), cdiv(Hl, th), c.n);
        TimedLaunch tl(KF_CONV_IN, c.st);
        if (tc == TILE_64x16 && c.m.split() && (Wl & 3) == 0) {
          ConvInHArgs ah{cond, x_t, P + p.p_wch, P + p.p_wxh, P + p.p_wc5h + HL_W5TAB3, P + p.conv_in.b, c.tensor_ptr(o.dst),
                         c.stat_ptr(o.dst), p.C, Hl, Wl, c.m.xcd(), c.amax()};
          GC_KLOG("conv_in_h_kernel");
          conv_in_h_kernel<<<grid, 256, 0, c.st>>>(ah);
        } else {
          GC_KLOG("conv_in_kernel<tile> (exact fp32)");
          if (tc == TILE_64x16) conv_in_kernel<64, 16, 4><<<grid, 256, 0, c.st>>>(a);
          else if (tc == TILE_32x16) conv_in_kernel<32, 16, 4><<<grid, 128, 0, c.st>>>(a);
          else conv_in_kernel<32, 8, 1><<<grid, 256, 0, c.st>>>(a);
        }
        break;
      }
      case OP_RES_CONV1: {
        const ResBlockPlan& b = p.blocks[o.blk];
        if (c.m.bf16()) {
          Conv8BArgs a{};
          a.src[0] = c.tensor_ptr(o.src[0]); a.src_f32[0] = c.is_f32(o.src[0]); a.sstat[0] = c.stat_ptr(o.src[0]);
          if (o.src[1] >= 0) { a.src[1] = c.tensor_ptr(o.src[1]); a.src_f32[1] = c.is_f32(o.src[1]); a.sstat[1] = c.stat_ptr(o.src[1]); }
          a.gamma = P + b.n1w; a.beta = P + b.n1b; a.wb = P + b.p_c1wb; a.bias = P + b.p_bias1 + (size_t)t * 8;
          a.dst = c.tensor_ptr(o.dst); a.dst_f32 = c.is_f32(o.dst); a.dstat = c.stat_ptr(o.dst);
          a.H = a.Hin = Hl; a.W = a.Win = Wl; a.inv_cnt = 1.0 / ((b.cin == 8 ? 2.0 : 4.0) * Hl * Wl); a.xcd = c.m.xcd();
          if (b.cin == 8) launch_conv8b<1, true, false, 0>(a, c.n, c.st);
          else launch_conv8b<2, true, false, 0>(a, c.n, c.st);
          break;
        }
        const Conv8Args a = conv1_args(c, o, t);
        const TileCfg tc = pick_tile(c.m, c.n, Hl, Wl);
        if (resfuse_emulated(c, b, tc, (Wl & 3) == 0)) {   // "ResnetBlock fusion, emulated" (above): the statistics-only pass of conv1
          Conv8Args e = a;
          e.term_mask = 63 | 64;
          TimedLaunch tl(KF_CONV8, c.st, 4.0 * c.n * 8.0 * Hl * Wl);
          GC_KLOG("conv8h_kernel<1,GN,0,0,DIAG> (MODE_RESFUSE_EMU: statistics only)");
          conv8h_kernel<1, true, false, 0, true><<<dim3(cdiv(Wl, 64), cdiv(Hl, 16), c.n), 256, 0, c.st>>>(e);
          break;
        }
        if (b.cin == 8) launch_conv8<1, true, false, 0>(c.m, tc, a, c.n, c.st);
        else launch_conv8<2, true, false, 0>(c.m, tc, a, c.n, c.st);
        break;
      }
      case OP_RES_CONV2: {
        const ResBlockPlan& b = p.blocks[o.blk];
        if (c.m.bf16()) {
          Conv8BArgs a{};
          a.src[0] = c.tensor_ptr(o.src[0]); a.src_f32[0] = c.is_f32(o.src[0]); a.sstat[0] = c.stat_ptr(o.src[0]);
          a.gamma = P + b.n2w; a.beta = P + b.n2b; a.wb = P + b.p_c2wb; a.bias = P + b.p_bias2;
          a.res[0] = c.tensor_ptr(o.res[0]); a.res_f32[0] = c.is_f32(o.res[0]);
          if (o.res[1] >= 0) { a.res[1] = c.tensor_ptr(o.res[1]); a.res_f32[1] = c.is_f32(o.res[1]); a.ninw = P + b.p_ninw; }
          a.dst = c.tensor_ptr(o.dst); a.dst_f32 = c.is_f32(o.dst); a.dstat = c.stat_ptr(o.dst);
          a.H = a.Hin = Hl; a.W = a.Win = Wl; a.inv_cnt = 1.0 / (2.0 * Hl * Wl); a.xcd = c.m.xcd();
          if (b.cin == 8) launch_conv8b<1, true, false, 1>(a, c.n, c.st);
          else launch_conv8b<1, true, false, 2>(a, c.n, c.st);
          break;
        }
        const Conv8Args a = conv2_args(c, o);
        const TileCfg tc = pick_tile(c.m, c.n, Hl, Wl);
        if (resfuse_emulated(c, b, tc, (Wl & 3) == 0)) {   // the fused pass: reads the BLOCK INPUT (not conv1's output), two convolutions of matrix work
          Conv8Args e = a;
          e.src[0] = a.res[0]; e.sstat[0] = c.stat_ptr(o.res[0]); e.gamma = P + b.n1w; e.beta = P + b.n1b;
          e.term_mask = 63 | 128;
          TimedLaunch tl(KF_CONV8_RES1, c.st, 4.0 * c.n * 16.0 * Hl * Wl);
          GC_KLOG("conv8h_kernel<1,GN,0,1,DIAG> (MODE_RESFUSE_EMU: block input, double matrix work)");
          conv8h_kernel<1, true, false, 1, true><<<dim3(cdiv(Wl, 64), cdiv(Hl, 16), c.n), 256, 0, c.st>>>(e);
          break;
        }
        if (b.cin == 8) launch_conv8<1, true, false, 1>(c.m, tc, a, c.n, c.st);
        else launch_conv8<1, true, false, 2>(c.m, tc, a, c.n, c.st);
        break;
      }
      case OP_DOWN: {
        const DownArgs a = down_args(c, o);
        TimedLaunch tl(KF_DOWN, c.st, 4.0 * c.n * 8.0 * ((double)a.Hin * a.Win + (double)Hl * Wl));
        GC_KLOG((a.Win & 3) == 0 ? "down8x2_kernel" : "down8_kernel");
        if ((a.Win & 3) == 0) down8x2_kernel<<<dim3(cdiv(Hl * ((Wl + 1) / 2), 256), 1, c.n), 256, 0, c.st>>>(a);
        else down8_kernel<<<dim3(cdiv(Hl * Wl, 256), 1, c.n), 256, 0, c.st>>>(a);
        break;
      }
      case OP_UP: {
        const int lin = o.level + 1;
        if (c.m.bf16()) {
          Conv8BArgs a{};
          a.src[0] = c.tensor_ptr(o.src[0]); a.src_f32[0] = c.is_f32(o.src[0]);
          a.wb = P + p.up[lin].p_wb; a.bias = P + p.up[lin].b;
          a.dst = c.tensor_ptr(o.dst); a.dst_f32 = c.is_f32(o.dst); a.dstat = c.stat_ptr(o.dst);
          a.H = Hl; a.W = Wl; a.Hin = c.ws->Hl[lin]; a.Win = c.ws->Wl[lin]; a.xcd = c.m.xcd();
          launch_conv8b<1, false, true, 0>(a, c.n, c.st);
          break;
        }
        const Conv8Args a = up_args(c, o);
        launch_conv8<1, false, true, 0>(c.m, pick_tile(c.m, c.n, Hl, Wl), a, c.n, c.st);
        break;
      }
      case OP_ATTN: {
        const AttnPlan& at = p.attns[o.blk];
        const int HW = Hl * Wl;
        float* Q = c.tensor_ptr(o.src[1]); float* K = c.tensor_ptr(o.res[0]); float* V = c.tensor_ptr(o.res[1]);
        AttnQkvArgs qa{c.tensor_ptr(o.src[0]), c.stat_ptr(o.src[0]), P + at.nw, P + at.nb, P + at.qw, P + at.qb,
                       P + at.kw, P + at.kb, P + at.vw, P + at.vb, Q, K, V, 1.0 / (2.0 * HW), HW};
        attn_qkv_kernel<<<dim3(cdiv(HW, 256), c.n), 256, 0, c.st>>>(qa);
        AttnFlashArgs fa{Q, K, V, c.tensor_ptr(o.src[0]), P + at.pw, P + at.pb, c.tensor_ptr(o.dst), c.stat_ptr(o.dst), HW};
        if (HW >= 128) attn_flash_mfma_kernel<<<dim3(cdiv(HW, 64), c.n), 512, 0, c.st>>>(fa);   // fp32 matrix cores
        else attn_flash_kernel<<<dim3(cdiv(HW, 256), c.n), 256, 0, c.st>>>(fa);
        break;
      }
      case OP_CONV_OUT: {
        co.src = c.tensor_ptr(o.src[0]); co.sstat = c.stat_ptr(o.src[0]);
        co.gamma = P + p.nout_w; co.beta = P + p.nout_b;
        co.w = P + p.conv_out.p_w; co.wh = P + p.conv_out.p_wh; co.bias = P + p.conv_out.b;
        co.C = p.C; co.H = Hl; co.W = Wl;
        co.inv_cnt = 1.0 / (2.0 * Hl * Wl);
        co.xcd = c.m.xcd();
        co.amax_out = post != 0 ? c.amax() + 1 : nullptr;  // literal sampler: x_{t-1} feeds the next step's conv_in
        co.src_bf16 = !c.is_f32(o.src[0]);
        const int nocb = (p.C + 15) / 16;
        const TileCfg tc = pick_tile(c.m, c.n, Hl, Wl, nocb) == TILE_64x16 ? TILE_64x16 : TILE_32x8;
        int tw, th;
        tile_dims(tc, &tw, &th);
        const dim3 grid(cdiv(Wl, tw), cdiv(Hl, th), c.n * nocb);
#define GC_LAUNCH_CO(POST)                                                             \
  do {                                                                                 \
    if (tc == TILE_64x16) conv_out_kernel<64, 16, 4, POST><<<grid, 256, 0, c.st>>>(co); \
    else conv_out_kernel<32, 8, 1, POST><<<grid, 256, 0, c.st>>>(co);                  \
  } while (0)
        TimedLaunch tl(KF_CONV_OUT, c.st);
        if (c.m.split() && (Wl & 3) == 0 && pick_tile(c.m, c.n, Hl, Wl) == TILE_64x16) {
          const dim3 gh(cdiv(Wl, 64), cdiv(Hl, 16), c.n);
          GC_KLOG(post == 0 ? "conv_out_h_kernel<0> (x0_hat)" : post == 1 ? "conv_out_h_kernel<1> (explicit noise)" : "conv_out_h_kernel<2> (in-kernel Philox)");
          if (post == 0) conv_out_h_kernel<0><<<gh, 256, 0, c.st>>>(co);
          else if (post == 1) conv_out_h_kernel<1><<<gh, 256, 0, c.st>>>(co);
          else conv_out_h_kernel<2><<<gh, 256, 0, c.st>>>(co);
        } else {
          GC_KLOG(post == 0 ? "conv_out_kernel<tile,0> (exact fp32, x0_hat)" : post == 1 ? "conv_out_kernel<tile,1> (exact fp32, explicit noise)" : "conv_out_kernel<tile,2> (exact fp32, in-kernel Philox)");
          if (post == 0) GC_LAUNCH_CO(0);
          else if (post == 1) GC_LAUNCH_CO(1);
          else GC_LAUNCH_CO(2);
        }
#undef GC_LAUNCH_CO
        break;
      }
    }
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}

inline int unet_enqueue(const UNetCall& c, const float* x_t, const float* cond, int t, int post, ConvOutArgs co) {
  return unet_enqueue_range(c, x_t, cond, t, post, co, 0, (int)c.plan->ops.size(), false);
}

// k = W_cond (*) cond + b_in : conv_in restricted to the two message channels (C = 0 mode)
inline void kmap_enqueue(const UNetCall& c, const float* cond) {
  const UNetPlan& p = *c.plan;
  ConvInArgs a{cond, nullptr, c.prepared + p.conv_in.p_w, c.prepared + p.conv_in.b,
               reinterpret_cast<float*>(c.wsp + c.ws->kmap_off), nullptr, 0, c.H, c.W, c.m.xcd()};
  const TileCfg tc = pick_tile(c.m, c.n, c.H, c.W);
  int tw, th;
  tile_dims(tc, &tw, &th);
  const dim3 grid(cdiv(c.W, tw), cdiv(c.H, th), c.n);
  if (tc == TILE_64x16 && c.m.split() && (c.W & 3) == 0) {
    ConvInHArgs ah{cond, nullptr, c.prepared + p.p_wch, c.prepared + p.p_wxh, c.prepared + p.p_wc5h + HL_W5TAB3, c.prepared + p.conv_in.b,
                   reinterpret_cast<float*>(c.wsp + c.ws->kmap_off), nullptr, 0, c.H, c.W, c.m.xcd(), c.amax()};
    conv_in_h_kernel<<<grid, 256, 0, c.st>>>(ah);
  } else if (tc == TILE_64x16) conv_in_kernel<64, 16, 4><<<grid, 256, 0, c.st>>>(a);
  else if (tc == TILE_32x16) conv_in_kernel<32, 16, 4><<<grid, 128, 0, c.st>>>(a);
  else conv_in_kernel<32, 8, 1><<<grid, 256, 0, c.st>>>(a);
}

// one latent step: hs0 <- k + c2 (hs0 - k) + c1 [Wc5 (*) A + bsum + fix] + s (W_x (*) eps)
inline int latent_step_enqueue(const UNetCall& c, const float* sched_row, const float* noise, unsigned long long seed,
                               unsigned stream_id, const unsigned long long* seed_dev = nullptr) {
  const UNetPlan& p = *c.plan;
  const float* P = c.prepared;
  GC_HIP(hipMemsetAsync(c.stat_ptr(p.hs0_tensor), 0, (size_t)c.n * 16 * sizeof(double), c.st));
  LatentArgs a{};
  a.a_src = c.tensor_ptr(p.last_body_tensor); a.a_stat = c.stat_ptr(p.last_body_tensor);
  a.gamma = P + p.nout_w; a.beta = P + p.nout_b;
  a.kmap = reinterpret_cast<const float*>(c.wsp + c.ws->kmap_off);
  a.hs0 = c.tensor_ptr(p.hs0_tensor); a.hs0_stat = c.stat_ptr(p.hs0_tensor);
  a.wc5 = P + p.p_wc5; a.wc1 = P + p.p_wc1; a.bring = P + p.p_bring; a.bsum = P + p.p_bsum;
  a.wx = P + p.conv_in.p_w + 144;
  a.wc5h = P + p.p_wc5h; a.wxh = P + p.p_wxh;
  a.noise = noise; a.sched = sched_row; a.inv_cnt = 1.0 / (2.0 * c.H * c.W);
  a.seed = seed; a.seed_dev = seed_dev; a.stream_id = stream_id; a.C = p.C; a.H = c.H; a.W = c.W;
  a.xcd = c.m.xcd();
  a.a_bf16 = !c.is_f32(p.last_body_tensor);
  // algorithmic bytes: last block's output (read), k map (read), hs0 (read + write): 4 eight-channel maps
  TimedLaunch tl(KF_LATENT_STEP, c.st, 4.0 * c.n * 32.0 * c.H * c.W);
  // small maps too: per pixel this kernel is 1600 + 72 C MACs, and a 64 x 128 map gives the fp32 kernel only 32 workgroups of
  // two waves (122 us per launch at C = 128 against 40 us for the f16-pipe kernel on 16 workgroups of four waves)
  const bool small_h = c.m.split() && (c.W & 3) == 0 && c.W >= 64 && c.H >= 16 && c.m.v[MODE_TILE_WANT] == 0;
  if ((pick_tile(c.m, c.n, c.H, c.W) == TILE_64x16 || small_h) && c.m.split()) {
    const dim3 grid(cdiv(c.W, 64), cdiv(c.H, 16), c.n);
    GC_KLOG(noise ? "latent_step_h_kernel<1> (explicit noise)" : "latent_step_h_kernel<2> (in-kernel Philox)");
    if (noise) latent_step_h_kernel<1><<<grid, 256, 0, c.st>>>(a);
    else latent_step_h_kernel<2><<<grid, 256, 0, c.st>>>(a);
  } else if (pick_tile(c.m, c.n, c.H, c.W) == TILE_64x16) {
    const dim3 grid(cdiv(c.W, 64), cdiv(c.H, 16), c.n);
    GC_KLOG(noise ? "latent_step_kernel<64,16,1> (exact fp32, explicit noise)" : "latent_step_kernel<64,16,2> (exact fp32, in-kernel Philox)");
    if (noise) latent_step_kernel<64, 16, 1><<<grid, 256, 0, c.st>>>(a);
    else latent_step_kernel<64, 16, 2><<<grid, 256, 0, c.st>>>(a);
  } else {
    const dim3 grid(cdiv(c.W, 32), cdiv(c.H, 16), c.n);
    GC_KLOG(noise ? "latent_step_kernel<32,16,1> (exact fp32, explicit noise)" : "latent_step_kernel<32,16,2> (exact fp32, in-kernel Philox)");
    if (noise) latent_step_kernel<32, 16, 1><<<grid, 128, 0, c.st>>>(a);
    else latent_step_kernel<32, 16, 2><<<grid, 128, 0, c.st>>>(a);
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}

inline int unet_prepare_enqueue(const UNetPlan& p, const float* raw, float* prepared, hipStream_t st) {
  GC_HIP(hipMemcpyAsync(prepared, raw, (size_t)p.raw_floats * sizeof(float), hipMemcpyDeviceToDevice, st));
  auto conv_w = [&](long long src, long long dst, int OC, int IC, int OCB) {
    const int total = OC * IC * 9;
    prep_conv_w_kernel<<<cdiv(total, 256), 256, 0, st>>>(raw + src, prepared + dst, OC, IC, OCB);
  };
  conv_w(p.conv_in.w, p.conv_in.p_w, 8, p.C + 2, 8);
  GC_HIP(hipMemsetAsync(prepared + p.conv_out.p_w, 0, (size_t)((p.C + 15) / 16 * 16) * 72 * sizeof(float), st));
  conv_w(p.conv_out.w, p.conv_out.p_w, p.C, 8, 16);
  prep_conv_out_h_kernel<<<1, 256, 0, st>>>(raw + p.conv_out.w, prepared + p.conv_out.p_wh, p.C);
  for (int l = 0; l < p.L; ++l) {
    if (p.down[l].w >= 0) conv_w(p.down[l].w, p.down[l].p_w, 8, 8, 8);
    if (p.up[l].w >= 0) {
      conv_w(p.up[l].w, p.up[l].p_w, 8, 8, 8);
      prep_conv8h_kernel<<<1, 256, 0, st>>>(raw + p.up[l].w, prepared + p.up[l].p_wh, 8);
      prep_conv8b_kernel<<<1, 256, 0, st>>>(raw + p.up[l].w, prepared + p.up[l].p_wb, 8);
    }
  }
  TembArgs ta{};
  ta.raw = raw; ta.prepared = prepared;
  ta.d0w = p.d0w; ta.d0b = p.d0b; ta.d1w = p.d1w; ta.d1b = p.d1b;
  ta.nblocks = (int)p.blocks.size(); ta.T = p.T;
  for (size_t i = 0; i < p.blocks.size(); ++i) {
    const ResBlockPlan& b = p.blocks[i];
    conv_w(b.c1w, b.p_c1w, 8, b.cin, 8);
    conv_w(b.c2w, b.p_c2w, 8, 8, 8);
    prep_conv8h_kernel<<<1, 256, 0, st>>>(raw + b.c1w, prepared + b.p_c1wh, b.cin);
    prep_conv8h_kernel<<<1, 256, 0, st>>>(raw + b.c2w, prepared + b.p_c2wh, 8);
    prep_conv8b_kernel<<<1, 256, 0, st>>>(raw + b.c1w, prepared + b.p_c1wb, b.cin);
    prep_conv8b_kernel<<<1, 256, 0, st>>>(raw + b.c2w, prepared + b.p_c2wb, 8);
    if (b.cin != 8) prep_nin_w_kernel<<<1, 256, 0, st>>>(raw + b.ninw, prepared + b.p_ninw, 8, b.cin);
    prep_add_kernel<<<1, 64, 0, st>>>(raw + b.c2b, b.cin != 8 ? raw + b.ninb : nullptr, prepared + b.p_bias2, 8);
    ta.tpw[i] = b.tpw; ta.tpb[i] = b.tpb; ta.c1b[i] = b.c1b; ta.dst[i] = b.p_bias1;
  }
  prep_temb_kernel<<<p.T, 64, 0, st>>>(ta);
  {
    PrepLatentArgs la{raw + p.conv_in.w, raw + p.conv_out.w, raw + p.conv_out.b, prepared + p.p_wc5, prepared + p.p_wc1,
                      prepared + p.p_bring, prepared + p.p_bsum, p.C};
    prep_latent_kernel<<<cdiv(1600 + 5184 + 72 + 8, 256), 256, 0, st>>>(la);
    prep_latent_h_kernel<<<1, 256, 0, st>>>(prepared + p.p_wc5, raw + p.conv_in.w, prepared + p.p_wc5h, prepared + p.p_wxh,
                                           prepared + p.p_wch, p.C);
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}

}  // namespace gc
