// Host-side description of one DiffusionUNet instance: parameter enumeration (the "raw" blob the
// caller packs), the prepared-blob layout, and the launch program with liveness-based placement
// of the 8-channel intermediates in the caller's workspace.
//
// Mirrors the construction order of DiffusionUNet.__init__ / forward in the reference
// (opencood/models/gencomm_modules/unet.py:198-344) for ch = 8, ch_mult = (1,)*levels.
#pragma once
#include <string>
#include <vector>

#include "common.h"

namespace gc {

constexpr int kWtab3 = 3 * (3 * 64 * 4 + 64 * 2);  // = HC_WTAB3 (conv8h_kernels.h, static_assert there)

struct ParamEntry {
  std::string name;
  long long numel;
  long long off;  // float offset in the raw blob
};

struct ResBlockPlan {
  int cin;  // 8 or 16
  // raw offsets
  long long n1w, n1b, c1w, c1b, tpw, tpb, n2w, n2b, c2w, c2b, ninw, ninb;
  // prepared offsets
  long long p_c1w, p_c2w, p_ninw, p_bias1 /*[T][8]*/, p_bias2 /*[8]*/;
  long long p_c1wh, p_c2wh;  // three-term A-operand tables (conv8h_kernels.h): cin/8 * kWtab3 dwords + 64
  long long p_c1wb, p_c2wb;  // bf16 A-operand tables (conv8b_kernels.h): cin/8 * 768 dwords
};

struct AttnPlan {  // AttnBlock parameters (raw offsets; 1x1 conv weights are used as stored, [oc][ic])
  long long nw, nb, qw, qb, kw, kb, vw, vb, pw, pb;
};

struct ConvPlan {  // plain 3x3 conv (conv_in, downsample, upsample, conv_out)
  long long w, b;  // raw
  long long p_w;   // prepared
  long long p_wh = -1;  // fp16 hi/lo table (upsample conv only)
  long long p_wb = -1;  // bf16 table (upsample conv only)
};

enum OpKind { OP_CONV_IN, OP_RES_CONV1, OP_RES_CONV2, OP_DOWN, OP_UP, OP_CONV_OUT, OP_ATTN };

struct Op {
  OpKind kind;
  int blk;      // index into resblocks (OP_RES_*) or convs
  int src[2];   // tensor ids (-1 = none)
  int res[2];   // residual tensor ids (conv2)
  int dst;      // tensor id (-1 for conv_out)
  int level;    // resolution level of the OUTPUT
};

struct TensorPlan {
  int level;
  int slot;        // buffer slot within its level
  int last_use;    // op index
};

struct UNetPlan {
  int C, L, R, T, attn_mask;
  std::vector<AttnPlan> attns;
  std::vector<ParamEntry> params;
  long long raw_floats = 0;
  long long prepared_floats = 0;
  long long d0w, d0b, d1w, d1b;
  ConvPlan conv_in, conv_out;
  long long nout_w, nout_b;
  long long p_wc5, p_wc1, p_bring, p_bsum;  // latent sampler tables (latent_kernels.h)
  long long p_wc5h, p_wxh, p_wch;           // their fp16 hi/lo A-operand forms + conv_in's message chunk (latenth_kernels.h)
  int hs0_tensor = 0, last_body_tensor = 0;
  std::vector<ConvPlan> down, up;  // indexed by level (down[l] valid for l < L-1, up[l] for l > 0)
  std::vector<ResBlockPlan> blocks;
  std::vector<Op> ops;
  std::vector<TensorPlan> tensors;
  std::vector<int> slots_per_level;

  long long add(const std::string& name, long long numel) {
    params.push_back({name, numel, raw_floats});
    const long long o = raw_floats;
    raw_floats += numel;
    return o;
  }
  long long padd(long long numel) {
    const long long o = prepared_floats;
    prepared_floats += (numel + 63) / 64 * 64;  // keep every table 256-B aligned
    return o;
  }

  int add_resblock(const std::string& p, int cin) {
    ResBlockPlan b{};
    b.cin = cin;
    b.n1w = add(p + ".norm1.weight", cin);
    b.n1b = add(p + ".norm1.bias", cin);
    b.c1w = add(p + ".conv1.weight", 8LL * cin * 9);
    b.c1b = add(p + ".conv1.bias", 8);
    b.tpw = add(p + ".temb_proj.weight", 8 * 32);
    b.tpb = add(p + ".temb_proj.bias", 8);
    b.n2w = add(p + ".norm2.weight", 8);
    b.n2b = add(p + ".norm2.bias", 8);
    b.c2w = add(p + ".conv2.weight", 8 * 8 * 9);
    b.c2b = add(p + ".conv2.bias", 8);
    b.ninw = b.ninb = -1;
    if (cin != 8) {
      b.ninw = add(p + ".nin_shortcut.weight", 8LL * cin);
      b.ninb = add(p + ".nin_shortcut.bias", 8);
    }
    blocks.push_back(b);
    return (int)blocks.size() - 1;
  }

  int add_attn(const std::string& p) {
    AttnPlan a{};
    a.nw = add(p + ".norm.weight", 8); a.nb = add(p + ".norm.bias", 8);
    a.qw = add(p + ".q.weight", 64); a.qb = add(p + ".q.bias", 8);
    a.kw = add(p + ".k.weight", 64); a.kb = add(p + ".k.bias", 8);
    a.vw = add(p + ".v.weight", 64); a.vb = add(p + ".v.bias", 8);
    a.pw = add(p + ".proj_out.weight", 64); a.pb = add(p + ".proj_out.bias", 8);
    attns.push_back(a);
    return (int)attns.size() - 1;
  }
  // AttnBlock = one op (two launches); Q, K, V are temporaries with the footprint of an 8-ch map
  int emit_attn(int attn, int in, int level) {
    const int q = new_tensor(level), k = new_tensor(level), v = new_tensor(level), out = new_tensor(level);
    ops.push_back({OP_ATTN, attn, {in, q}, {k, v}, out, level});
    return out;
  }

  int new_tensor(int level) {
    tensors.push_back({level, -1, -1});
    return (int)tensors.size() - 1;
  }

  // res-block = two ops
  int emit_resblock(int blk, int in0, int in1, int level) {
    const int tmp = new_tensor(level), out = new_tensor(level);
    ops.push_back({OP_RES_CONV1, blk, {in0, in1}, {-1, -1}, tmp, level});
    ops.push_back({OP_RES_CONV2, blk, {tmp, -1}, {in0, in1}, out, level});
    return out;
  }

  // Returns nullptr on success, else an error string.
  // keep_all: every intermediate gets a slot of its own (the backward pass reads all of them back)
  const char* build(int C_, int L_, int R_, int attn_mask_, int T_, bool keep_all = false) {
    C = C_; L = L_; R = R_; T = T_; attn_mask = attn_mask_;
    if (attn_mask < 0 || attn_mask >= (1 << L_)) return "attn_mask has bits beyond the number of levels";
    if (C < 8 || C % 8 != 0) return "C must be a positive multiple of 8";
    if (L < 1 || L > 4) return "levels must be in 1..4";
    if (R < 1 || R > 4) return "res_blocks must be in 1..4";
    if (T < 1) return "T must be >= 1";
    params.clear(); blocks.clear(); ops.clear(); tensors.clear(); attns.clear();
    raw_floats = prepared_floats = 0;
    down.assign(L, ConvPlan{-1, -1, -1});
    up.assign(L, ConvPlan{-1, -1, -1});

    // ---- parameters, in the order DiffusionUNet executes them ----
    d0w = add("temb.dense.0.weight", 32 * 8);
    d0b = add("temb.dense.0.bias", 32);
    d1w = add("temb.dense.1.weight", 32 * 32);
    d1b = add("temb.dense.1.bias", 32);
    conv_in.w = add("conv_in.weight", 8LL * (C + 2) * 9);
    conv_in.b = add("conv_in.bias", 8);
    std::vector<std::vector<int>> down_blk(L), up_blk(L), down_att(L), up_att(L);
    for (int l = 0; l < L; ++l) {
      for (int b = 0; b < R; ++b) {
        down_blk[l].push_back(add_resblock("down." + std::to_string(l) + ".block." + std::to_string(b), 8));
        if (attn_mask >> l & 1) down_att[l].push_back(add_attn("down." + std::to_string(l) + ".attn." + std::to_string(b)));
      }
      if (l != L - 1) {
        down[l].w = add("down." + std::to_string(l) + ".downsample.conv.weight", 8 * 8 * 9);
        down[l].b = add("down." + std::to_string(l) + ".downsample.conv.bias", 8);
      }
    }
    const int mid1 = add_resblock("mid.block_1", 8), mid2 = add_resblock("mid.block_2", 8);
    for (int l = L - 1; l >= 0; --l) {
      for (int b = 0; b <= R; ++b) {
        up_blk[l].push_back(add_resblock("up." + std::to_string(l) + ".block." + std::to_string(b), 16));
        if (attn_mask >> l & 1) up_att[l].push_back(add_attn("up." + std::to_string(l) + ".attn." + std::to_string(b)));
      }
      if (l != 0) {
        up[l].w = add("up." + std::to_string(l) + ".upsample.conv.weight", 8 * 8 * 9);
        up[l].b = add("up." + std::to_string(l) + ".upsample.conv.bias", 8);
      }
    }
    nout_w = add("norm_out.weight", 8);
    nout_b = add("norm_out.bias", 8);
    conv_out.w = add("conv_out.weight", (long long)C * 8 * 9);
    conv_out.b = add("conv_out.bias", C);
    if ((int)blocks.size() > kMaxResBlocks) return "too many res-blocks";

    // ---- prepared blob: verbatim copy of raw, then re-laid-out tensors and tables ----
    prepared_floats = (raw_floats + 63) / 64 * 64;
    conv_in.p_w = padd(8LL * (C + 2) * 9);
    conv_out.p_w = padd((long long)((C + 15) / 16 * 16) * 8 * 9);
    conv_out.p_wh = padd((long long)((C + 15) / 16) * kWtab3 + 64);
    for (int l = 0; l < L; ++l) {
      if (down[l].w >= 0) down[l].p_w = padd(8 * 8 * 9);
      if (up[l].w >= 0) { up[l].p_w = padd(8 * 8 * 9); up[l].p_wh = padd(kWtab3 + 64); up[l].p_wb = padd(768); }
    }
    p_wc5 = padd(1600); p_wc1 = padd(5184); p_bring = padd(72); p_bsum = padd(8);
    p_wc5h = padd(8 * (kWtab3 / 3) + 64); p_wxh = padd((long long)(C / 8) * kWtab3); p_wch = padd(kWtab3);
    for (auto& b : blocks) {
      b.p_c1w = padd(8LL * b.cin * 9);
      b.p_c2w = padd(8 * 8 * 9);
      b.p_ninw = b.cin != 8 ? padd(8LL * b.cin) : -1;
      b.p_bias1 = padd((long long)T * 8);
      b.p_bias2 = padd(8);
      b.p_c1wh = padd((long long)(b.cin / 8) * kWtab3 + 64);
      b.p_c2wh = padd(kWtab3 + 64);
      b.p_c1wb = padd((long long)(b.cin / 8) * 768);
      b.p_c2wb = padd(768);
    }

    // ---- launch program (unet.py:307-344) ----
    std::vector<int> hs;
    int h = new_tensor(0);
    ops.push_back({OP_CONV_IN, 0, {-1, -1}, {-1, -1}, h, 0});
    hs.push_back(h);
    for (int l = 0; l < L; ++l) {
      for (int b = 0; b < R; ++b) {
        int t = emit_resblock(down_blk[l][b], hs.back(), -1, l);
        if (attn_mask >> l & 1) t = emit_attn(down_att[l][b], t, l);
        hs.push_back(t);
      }
      if (l != L - 1) {
        const int d = new_tensor(l + 1);
        ops.push_back({OP_DOWN, l, {hs.back(), -1}, {-1, -1}, d, l + 1});
        hs.push_back(d);
      }
    }
    h = hs.back();
    h = emit_resblock(mid1, h, -1, L - 1);
    h = emit_resblock(mid2, h, -1, L - 1);
    for (int l = L - 1; l >= 0; --l) {
      for (int b = 0; b <= R; ++b) {
        const int skip = hs.back();
        hs.pop_back();
        h = emit_resblock(up_blk[l][b], h, skip, l);
        if (attn_mask >> l & 1) h = emit_attn(up_att[l][b], h, l);
      }
      if (l != 0) {
        const int u = new_tensor(l - 1);
        ops.push_back({OP_UP, l, {h, -1}, {-1, -1}, u, l - 1});
        h = u;
      }
    }
    ops.push_back({OP_CONV_OUT, 0, {h, -1}, {-1, -1}, -1, 0});
    hs0_tensor = ops[0].dst;
    last_body_tensor = h;

    // ---- liveness -> slots ----
    for (int i = 0; i < (int)ops.size(); ++i) {
      const Op& o = ops[i];
      for (int k = 0; k < 2; ++k) {
        if (o.src[k] >= 0) tensors[o.src[k]].last_use = i;
        if (o.res[k] >= 0) tensors[o.res[k]].last_use = i;
      }
    }
    tensors[hs0_tensor].last_use = (int)ops.size();  // never released: the latent step rewrites it in place
    slots_per_level.assign(L, 0);
    std::vector<std::vector<int>> free_slots(L);
    auto take_slot = [&](int id) {
      TensorPlan& t = tensors[id];
      if (!free_slots[t.level].empty()) {
        t.slot = free_slots[t.level].back();
        free_slots[t.level].pop_back();
      } else {
        t.slot = slots_per_level[t.level]++;
      }
    };
    for (int i = 0; i < (int)ops.size(); ++i) {
      const Op& o = ops[i];
      if (o.kind == OP_ATTN) { take_slot(o.src[1]); take_slot(o.res[0]); take_slot(o.res[1]); }
      if (o.dst >= 0) take_slot(o.dst);
      for (int k = 0; k < 4; ++k) {
        const int id = k < 2 ? o.src[k] : o.res[k - 2];
        if (id >= 0 && tensors[id].last_use == i && tensors[id].slot >= 0) {
          // a tensor may appear twice in one op (src and res): release once
          bool dup = false;
          for (int k2 = 0; k2 < k; ++k2) dup |= ((k2 < 2 ? o.src[k2] : o.res[k2 - 2]) == id);
          if (!dup && !keep_all) free_slots[tensors[id].level].push_back(tensors[id].slot);
        }
      }
    }
    return nullptr;
  }
};

// Spatial size of level l and the workspace carve-up for (n, H, W).
struct UNetWorkspace {
  std::vector<int> Hl, Wl;
  std::vector<size_t> level_base, slot_bytes;
  size_t stats_bytes = 0, total = 0, kmap_off = 0, amax_off = 0;
  size_t df_words_off = 0, df_prog_off = 0;  // dataflow kernel: tickets / done counters / error word (inside the zeroed statistics block), program

  const char* build(const UNetPlan& p, int n, int H, int W) {
    Hl.assign(p.L, 0); Wl.assign(p.L, 0);
    Hl[0] = H; Wl[0] = W;
    for (int l = 1; l < p.L; ++l) {
      // skip concat after nearest-x2 upsampling requires even sizes at every level above the last
      if ((Hl[l - 1] & 1) || (Wl[l - 1] & 1)) return "H and W must be even at every level that is downsampled";
      Hl[l] = Hl[l - 1] / 2; Wl[l] = Wl[l - 1] / 2;
    }
    if (Hl[p.L - 1] < 1 || Wl[p.L - 1] < 1) return "feature map too small for the number of levels";
    // statistics of every tensor, then the dataflow kernel's words: [8] tickets, [64 ops][n] done counters, [1] error word
    df_words_off = align_up(p.tensors.size() * (size_t)n * 16 * sizeof(double), 256);
    stats_bytes = align_up(df_words_off + (8 + 64 * (size_t)n + 4) * sizeof(unsigned), 256);
    size_t off = stats_bytes;
    level_base.assign(p.L, 0); slot_bytes.assign(p.L, 0);
    for (int l = 0; l < p.L; ++l) {
      slot_bytes[l] = align_up((size_t)n * 8 * Hl[l] * Wl[l] * sizeof(float), 256);
      level_base[l] = off;
      off += slot_bytes[l] * p.slots_per_level[l];
    }
    kmap_off = off;  // persistent k = W_cond (*) cond + b_in of the latent sampler
    off += slot_bytes[0];
    amax_off = off;  // two floats: device bounds on max|cond|, max|x_t| (range guard of conv_in on the f16 pipe)
    off += 256;
    df_prog_off = off;  // program of the dataflow kernel (dataflow_kernels.h DfProgram)
    off += 1 << 16;
    total = off;
    return nullptr;
  }
};

}  // namespace gc
