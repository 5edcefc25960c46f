// Host side of the Enhancer: live-parameter enumeration, workspace carve-up, launch sequence.
#pragma once
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "enhancer_kernels.h"
#include "unet_plan.h"  // ParamEntry

namespace gc {

struct EnhancerPlan {
  int C = 0, dc = 0, dcp = 0, hid = 0;
  std::vector<ParamEntry> params;
  long long raw_floats = 0;
  long long n1w, n1b, n2w, n2b, pcw, l1w, l1b, dww, dwb, l2w, l2b, fc1, bnw, bnb, fc2;

  long long add(const std::string& name, long long numel) {
    params.push_back({name, numel, raw_floats});
    const long long o = raw_floats;
    raw_floats += numel;
    return o;
  }
  // Only the parameters the live code touches (block_1.{norm1,norm2,mlp.*}, split_attn.*); the
  // other 39 keys of the reference's state_dict (block_2/3, *.attn.*) never reach a kernel.
  const char* build(int C_) {
    if (C_ < 8 || C_ % 8 != 0) return "enhancer: C must be a positive multiple of 8";
    C = C_; dc = C / 4; dcp = (dc + 15) / 16 * 16; hid = 2 * C;
    params.clear(); raw_floats = 0;
    n1w = add("block_1.norm1.weight", C);
    n1b = add("block_1.norm1.bias", C);
    n2w = add("block_1.norm2.weight", C);
    n2b = add("block_1.norm2.bias", C);
    pcw = add("block_1.mlp.partial_conv3.weight", (long long)dc * dc * 9);
    l1w = add("block_1.mlp.linear1.0.weight", (long long)2 * hid * C);
    l1b = add("block_1.mlp.linear1.0.bias", 2 * hid);
    dww = add("block_1.mlp.dwconv.0.weight", (long long)hid * 9);
    dwb = add("block_1.mlp.dwconv.0.bias", hid);
    l2w = add("block_1.mlp.linear2.0.weight", (long long)C * hid);
    l2b = add("block_1.mlp.linear2.0.bias", C);
    fc1 = add("split_attn.fc1.weight", (long long)C * C);
    bnw = add("split_attn.bn1.weight", C);
    bnb = add("split_attn.bn1.bias", C);
    fc2 = add("split_attn.fc2.weight", (long long)C * C);
    return nullptr;
  }
};

struct EnhancerWs {
  size_t Y, Z, Zc, Hd, G, O, colsum, gate, wT, tab1, tab2, wsc, total;
};

inline EnhancerWs enhancer_ws(const EnhancerPlan& p, int n, int H, int W) {
  const size_t M = (size_t)n * H * W;
  EnhancerWs w{};
  size_t off = 0;
  auto take = [&](size_t floats) { const size_t o = off; off += align_up(floats * sizeof(float), 256); return o; };
  w.colsum = take((size_t)n * p.C);
  w.gate = take((size_t)n * p.C);
  // pconv weights: fp32 [9][dc][dcp] or the three-term A-operand table (oc blocks x k slices x 896 dwords + 64)
  w.wT = take(std::max((size_t)9 * p.dc * p.dcp, (size_t)(p.dc / 16) * ((9 * p.dc + 31) / 32) * 896 + 64));
  w.Y = take(M * p.C);
  w.Z = take(M * p.C);
  w.Zc = take(M * p.dc);
  w.Hd = take(M * 2 * p.hid);
  w.G = take(M * p.hid);
  // the token-major result of every launch structure lives in the hidden tensor's slot: dead by the time Linear2 runs in the separate /
  // half-fused structures, never used by the fully fused one -- and never Z, whose halos neighbouring workgroups of the fused kernel
  // still read.  One place for all modes: gencomm_warp_attfuse_tok_fwd does not have to know which structure ran.
  w.O = w.Hd;
  w.tab1 = take((size_t)(p.hid / 16) * (p.C / 16) * 896);  // fused front kernel: Linear1 / Linear2 operand tables
  w.tab2 = take((size_t)(p.hid / 16) * (p.C / 32 > 0 ? p.C / 32 : 1) * 896);
  w.wsc = take(64);   // per-tensor weight scales of the two tables
  w.total = off;
  return w;
}
// 0: separate launches, 1: Linear1 + depthwise stage fused, 2: + Linear2
inline int enh_fuse_level(const Modes& m, int C) { return (m.split() && C == 64) ? (int)m.v[MODE_ENH_FUSE] : 0; }
inline size_t enhancer_workspace_bytes(const EnhancerPlan& p, int n, int H, int W) { return enhancer_ws(p, n, H, W).total; }

inline int enhancer_enqueue(const EnhancerPlan& p, const float* raw, const float* x, float* out,
                            int n, int H, int W, char* wsp, hipStream_t st, const Modes& m) {
  const EnhancerWs w = enhancer_ws(p, n, H, W);
  const int HW = H * W, C = p.C;
  auto F = [&](size_t off) { return reinterpret_cast<float*>(wsp + off); };
  GC_HIP(hipMemsetAsync(F(w.colsum), 0, (size_t)n * C * sizeof(float), st));

  {  // K1
    EnhLnArgs a{x, raw + p.n1w, raw + p.n1b, raw + p.n2w, raw + p.n2b, F(w.Y), F(w.Z), F(w.Zc), C, p.dc, HW};
    const size_t sh = ((size_t)C * 65 + 256 + 128) * sizeof(float);
    if (sh > 48 * 1024) GC_HIP(hipFuncSetAttribute((const void*)enh_ln_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
    TimedLaunch tl(KF_ENH_LN, st);
    if (C == 64) enh_ln64_kernel<<<dim3((HW + 255) / 256, n), 256, 0, st>>>(a);   // registers + per-wave transpose, no workgroup barriers
    else enh_ln_kernel<<<dim3((HW + 63) / 64, n), 256, sh, st>>>(a);
  }
  if (m.split() && (p.dc == 16 || p.dc == 32 || p.dc == 64) && (C & 3) == 0) {  // K2 on the f16 matrix pipe
    const int nslice = (9 * p.dc + 31) / 32;
    enh_prep_pconv_h_kernel<<<(p.dc / 16) * nslice * 4, 256, 0, st>>>(raw + p.pcw, F(w.wT), p.dc, nslice);   // <= 256 table entries per workgroup
    TimedLaunch tl(KF_ENH_PCONV, st);
    EnhPconvHArgs a{F(w.Zc), F(w.wT), F(w.Z), C, p.dc, H, W, nslice};
    // 32-pixel-wide tiles when they fit the LDS (dc <= 32) and still give every CU a workgroup; 16-pixel-wide ones otherwise
    const bool wide = p.dc <= 32 && (long long)((W + 31) / 32) * ((H + 15) / 16) * n >= 256;
    const size_t sh = (size_t)18 * (wide ? 34 : 18) * p.dc * 5;   // fp16 hi + lo planes, bf8 third-term plane
    if (wide) {
      if (sh > 48 * 1024) GC_HIP(hipFuncSetAttribute((const void*)enh_pconv_h_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
      enh_pconv_h_kernel<32><<<dim3((W + 31) / 32, (H + 15) / 16, n), 256, sh, st>>>(a);
    } else {
      if (sh > 48 * 1024) GC_HIP(hipFuncSetAttribute((const void*)enh_pconv_h_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
      enh_pconv_h_kernel<16><<<dim3((W + 15) / 16, (H + 15) / 16, n), 256, sh, st>>>(a);
    }
  } else
  {  // K2
    enh_prep_pconv_kernel<<<(9 * p.dc * p.dcp + 255) / 256, 256, 0, st>>>(raw + p.pcw, F(w.wT), p.dc, p.dcp);
    TimedLaunch tl(KF_ENH_PCONV, st);
    EnhPconvArgs a{F(w.Zc), F(w.wT), F(w.Z), C, p.dc, p.dcp, H, W};
    if (p.dc <= 32) {
      const size_t sh = (size_t)18 * 18 * (p.dc + 1) * sizeof(float);
      enh_pconv_kernel<16, 16><<<dim3((W + 15) / 16, (H + 15) / 16, n), 256, sh, st>>>(a);
    } else {
      const size_t sh = (size_t)10 * 18 * (p.dc + 1) * sizeof(float);
      if (sh > 48 * 1024) GC_HIP(hipFuncSetAttribute((const void*)enh_pconv_kernel<16, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
      enh_pconv_kernel<16, 8><<<dim3((W + 15) / 16, (H + 7) / 8, n), 128, sh, st>>>(a);
    }
  }
  const int fuse = enh_fuse_level(m, C);
  if (fuse != 0) {
    // K3 + K4 (+ K5) fused: the hidden tensors never reach HBM
    enh_wscale_kernel<<<2, 256, 0, st>>>(raw + p.l1w, (long long)2 * p.hid * C, raw + p.l2w, (long long)C * p.hid, F(w.wsc));
    enh_prep_front_kernel<<<32, 256, 0, st>>>(raw + p.l1w, F(w.tab1), C, p.hid, F(w.wsc));
    if (fuse >= 2) enh_prep_back_kernel<<<16, 256, 0, st>>>(raw + p.l2w, F(w.tab2), C, p.hid, F(w.wsc));
    EnhFrontArgs a{F(w.Z), F(w.tab1), raw + p.l1b, raw + p.dww, raw + p.dwb, F(w.G), C, p.hid, H, W,
                   F(w.tab2), raw + p.l2b, F(w.Y), F(w.O), F(w.colsum), F(w.wsc)};
    TimedLaunch tl(KF_ENH_GEMM1, st);
    if (fuse >= 2) enh_front_h_kernel<true><<<dim3((W + 7) / 8, (H + 7) / 8, n), 256, 0, st>>>(a);
    else enh_front_h_kernel<false><<<dim3((W + 7) / 8, (H + 7) / 8, n), 256, 0, st>>>(a);
  } else {
  {  // K3: linear1 + GELU
    GemmArgs a{F(w.Z), raw + p.l1w, raw + p.l1b, nullptr, F(w.Hd), nullptr, HW, 2 * p.hid, C};
    TimedLaunch tl(KF_ENH_GEMM1, st);
    if (m.split() && (C & 3) == 0) gemm_f16s_mfma_kernel<0><<<dim3((HW + 127) / 128, (2 * p.hid + 63) / 64, n), 256, 0, st>>>(a);
    else gemm_f32_mfma_kernel<0><<<dim3((HW + 127) / 128, (2 * p.hid + 63) / 64, n), 256, 0, st>>>(a);
  }
  {  // K4: dwconv + GELU, gate
    constexpr int SL = 8;
    EnhDwArgs a{F(w.Hd), raw + p.dww, raw + p.dwb, F(w.G), p.hid, H, W};
    const long long total = (long long)H * ((W + SL - 1) / SL) * p.hid;
    TimedLaunch tl(KF_ENH_DWGATE, st);
    enh_dwgate_kernel<SL><<<dim3((unsigned)((total + 255) / 256), n), 256, 0, st>>>(a);
  }
  }
  float* const Otok = F(w.O);
  if (fuse < 2) {  // K5: linear2 + residual, column sums for the global average pool
    GemmArgs a{F(w.G), raw + p.l2w, raw + p.l2b, F(w.Y), F(w.O), F(w.colsum), HW, C, p.hid};
    TimedLaunch tl(KF_ENH_GEMM2, st);
    if (m.split() && (p.hid & 3) == 0) gemm_f16s_mfma_kernel<1><<<dim3((HW + 127) / 128, (C + 63) / 64, n), 256, 0, st>>>(a);
    else gemm_f32_mfma_kernel<1><<<dim3((HW + 127) / 128, (C + 63) / 64, n), 256, 0, st>>>(a);
  }
  {  // K6
    EnhGateArgs a{F(w.colsum), raw + p.fc1, raw + p.bnw, raw + p.bnb, raw + p.fc2, F(w.gate), C, 1.0f / (float)HW};
    TimedLaunch tl(KF_ENH_GATE, st);
    enh_gate_kernel<<<n, 256, (2 * C + 8) * sizeof(float), st>>>(a);
  }
  if (out != nullptr) {  // K7 (skipped when the caller consumes the token-major result + gate directly)
    EnhOutArgs a{Otok, F(w.gate), out, C, HW};
    const size_t sh = (size_t)32 * (C + 1) * sizeof(float);
    TimedLaunch tl(KF_ENH_OUT, st);
    enh_scale_transpose_kernel<<<dim3((HW + 31) / 32, n), 256, sh, st>>>(a);
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}

}  // namespace gc
