// Host side of MessageExtractorv2: parameter enumeration, workspace, launch sequence.
#pragma once
#include <string>
#include <vector>

#include "msgext_kernels.h"
#include "unet_host.h"

namespace gc {

struct MsgExtPlan {
  int C = 0;
  std::vector<ParamEntry> params;
  long long raw_floats = 0;
  long long ow, ob, dw, db, a1w, a1b, a3w, a3b, f0w, f0b, f2w, f2b;
  long long add(const std::string& name, long long numel) {
    params.push_back({name, numel, raw_floats});
    const long long o = raw_floats;
    raw_floats += numel;
    return o;
  }
  const char* build(int C_) {
    if (C_ < 8 || C_ % 8 != 0) return "message extractor: in_channels must be a positive multiple of 8";
    C = C_; params.clear(); raw_floats = 0;
    const std::string p = "bev_extractor.";
    ow = add(p + "offset1.weight", 18LL * C * 9); ob = add(p + "offset1.bias", 18);
    dw = add(p + "dcn1.weight", 64LL * C * 9);    db = add(p + "dcn1.bias", 64);
    f0w = add(p + "fuse.0.weight", 64 * 64);      f0b = add(p + "fuse.0.bias", 64);
    f2w = add(p + "fuse.2.weight", 2 * 64);       f2b = add(p + "fuse.2.bias", 2);
    a1w = add(p + "attn.1.weight", 32 * 64);      a1b = add(p + "attn.1.bias", 32);
    a3w = add(p + "attn.3.weight", 64 * 32);      a3b = add(p + "attn.3.bias", 64);
    return nullptr;
  }
};

struct MsgExtWs { size_t wOff, wDcn, wFuse, colsum, gate, off, b1, total; };
inline MsgExtWs msgext_ws(const MsgExtPlan& p, int n, int H, int W) {
  MsgExtWs w{};
  size_t o = 0;
  auto take = [&](size_t floats) { const size_t r = o; o += align_up(floats * sizeof(float), 256); return r; };
  w.wOff = take((size_t)p.C * 9 * 20);
  w.wDcn = take((size_t)p.C * 9 * 64);
  w.wFuse = take(4096);
  w.colsum = take((size_t)n * 64);
  w.gate = take((size_t)n * 64);
  w.off = take((size_t)n * 18 * H * W);
  w.b1 = take((size_t)n * 64 * H * W);
  w.total = o;
  return w;
}

inline int msgext_enqueue(const MsgExtPlan& p, const float* raw, const float* x, float* out, int n, int H, int W,
                          char* wsp, hipStream_t st) {
  const MsgExtWs w = msgext_ws(p, n, H, W);
  const int HW = H * W, C = p.C;
  auto F = [&](size_t off) { return reinterpret_cast<float*>(wsp + off); };
  GC_HIP(hipMemsetAsync(F(w.wOff), 0, (size_t)C * 9 * 20 * sizeof(float), st));
  GC_HIP(hipMemsetAsync(F(w.colsum), 0, (size_t)n * 64 * sizeof(float), st));
  prep_conv_w_kernel<<<cdiv(18 * C * 9, 256), 256, 0, st>>>(raw + p.ow, F(w.wOff), 18, C, 20);
  prep_conv_w_kernel<<<cdiv(64 * C * 9, 256), 256, 0, st>>>(raw + p.dw, F(w.wDcn), 64, C, 64);
  prep_nin_w_kernel<<<16, 256, 0, st>>>(raw + p.f0w, F(w.wFuse), 64, 64);
  {
    ConvNArgs a{x, F(w.wOff), raw + p.ob, F(w.off), C, 18, H, W};
    const TileCfg tc = pick_tile(modes_snapshot(), n, H, W);
    int tw, th;
    tile_dims(tc, &tw, &th);
    const dim3 grid(cdiv(W, tw), cdiv(H, th), n);
    if (tc == TILE_64x16) conv3x3_cN_kernel<64, 16, 4, 5><<<grid, 256, 0, st>>>(a);
    else if (tc == TILE_32x16) conv3x3_cN_kernel<32, 16, 4, 5><<<grid, 128, 0, st>>>(a);
    else conv3x3_cN_kernel<32, 8, 1, 5><<<grid, 256, 0, st>>>(a);
  }
  {
    DcnArgs a{x, F(w.off), F(w.wDcn), raw + p.db, F(w.b1), F(w.colsum), C, H, W};
    if ((long long)cdiv(HW, 256) * n < 512) dcn_csplit_kernel<<<dim3(cdiv(HW, 64), n), 256, 0, st>>>(a);   // small maps: 4x the workgroups, a quarter of the chain per wave
    else dcn_kernel<<<dim3(cdiv(HW, 256), n), 256, 0, st>>>(a);
  }
  {
    MsgGateArgs a{F(w.colsum), raw + p.a1w, raw + p.a1b, raw + p.a3w, raw + p.a3b, F(w.gate), 1.0f / (float)HW};
    msg_gate_kernel<<<n, 64, 0, st>>>(a);
  }
  {
    MsgFuseArgs a{F(w.b1), F(w.gate), F(w.wFuse), raw + p.f0b, raw + p.f2w, raw + p.f2b, out, HW};
    msg_fuse_kernel<<<dim3(cdiv(HW, 256), n), 256, 0, st>>>(a);
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}

}  // namespace gc
