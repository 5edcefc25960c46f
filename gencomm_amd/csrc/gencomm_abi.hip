// C ABI of libgencomm_hip.so (declared in include/gencomm_hip.h).
#include "../../include/gencomm_hip.h"

#include "common.h"
#include "conv_kernels.h"
#include "detect_kernels.h"
#include "enhancer_host.h"
#include "fusion_kernels.h"
#include "msgext_host.h"
#include "noise_kernels.h"
#include "pillar_kernels.h"
#include "pfn_kernels.h"
#include "train_kernels.h"
#include "wgrad_h3_kernels.h"
#include "unet_bwd_host.h"
#include "unet_host.h"

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <stdlib.h>
#include <string.h>

namespace gc {
char* last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}
KernelTimer& kernel_timer() {
  static KernelTimer t;
  return t;
}
static std::atomic<long long> g_modes[MODE_COUNT] = {{0}, {0}, {0}, {2}, {-1}, {1}, {0}, {0}, {-1}, {1}, {16}};  // defaults, see ModeKey
static std::atomic<bool> g_klog_armed{false};
static std::mutex g_klog_mu;
static std::map<std::string, int> g_klog;
bool klog_armed() { return g_klog_armed.load(std::memory_order_relaxed); }
void klog_note(const char* name) {
  std::lock_guard<std::mutex> lk(g_klog_mu);
  ++g_klog[name];
}
Modes modes_snapshot() {
  Modes m;
  for (int i = 0; i < MODE_COUNT; ++i) m.v[i] = g_modes[i].load(std::memory_order_relaxed);
  return m;
}
}  // namespace gc

using namespace gc;

extern "C" {

int gencomm_abi_version(void) { return GENCOMM_ABI_VERSION; }
#ifndef GENCOMM_BUILD_FLAGS
#define GENCOMM_BUILD_FLAGS "unknown"
#endif
const char* gencomm_build_info(void) { return GENCOMM_BUILD_FLAGS; }
const char* gencomm_last_error(void) { return last_error_buf(); }

// ------------------------------------------------------------------------------------ modes
int gencomm_set_mode(int key, long long value) {
  GC_CHECK_ARG(key >= 0 && key < MODE_COUNT, "unknown mode key");
  GC_CHECK_ARG(key != MODE_ARITH || (value >= 0 && value <= 3), "GENCOMM_MODE_ARITH: 0 (three-term f16-pipe path), 1 (exact fp32), 2 (bf16 denoise) or 3 (0 + two-term general convolutions)");
  GC_CHECK_ARG(key != MODE_SAMPLER || (value >= 0 && value <= 2), "GENCOMM_MODE_SAMPLER: 0 (automatic), 1 (direct) or 2 (latent)");
  GC_CHECK_ARG(key != MODE_TILE_WANT || value >= 0, "GENCOMM_MODE_TILE_WANT: 0 (automatic) or a positive workgroup count");
  GC_CHECK_ARG(key != MODE_TILE8 || value >= -1, "GENCOMM_MODE_TILE8: -1 (automatic), 0 (off) or a positive workgroup count");
  GC_CHECK_ARG(key != MODE_BWD_STREAMS || (value >= 0 && value <= 2), "GENCOMM_MODE_BWD_STREAMS: 0 (off), 1 (automatic) or 2 (always)");
  GC_CHECK_ARG(key != MODE_PERSIST || (value >= 0 && value <= 31), "GENCOMM_MODE_PERSIST: a mask of the five layer variants (0 .. 31)");
  g_modes[key].store(value, std::memory_order_relaxed);
  return GC_OK;
}
long long gencomm_get_mode(int key) {
  if (key < 0 || key >= MODE_COUNT) { fail(GC_ERR_ARG, "unknown mode key"); return -1; }
  return g_modes[key].load(std::memory_order_relaxed);
}

// ------------------------------------------------------------------------------------ timer
int gencomm_timer_num_kernels(void) { return KF_COUNT; }
const char* gencomm_timer_kernel_name(int family) { return kernel_family_name(family); }

int gencomm_timer_start_mask(unsigned long long mask, int capacity) {
  KernelTimer& t = kernel_timer();
  GC_CHECK_ARG(t.ev == nullptr, "timer already armed");
  GC_CHECK_ARG(mask != 0 && (mask >> KF_COUNT) == 0 && capacity >= 1 && capacity <= (1 << 20), "bad family mask / capacity");
  t.ev = new hipEvent_t[2 * (size_t)capacity];
  for (int i = 0; i < 2 * capacity; ++i) {
    if (hipEventCreate(&t.ev[i]) != hipSuccess) {
      for (int j = 0; j < i; ++j) (void)hipEventDestroy(t.ev[j]);
      delete[] t.ev;
      t.ev = nullptr;
      return fail(GC_ERR_HIP, "hipEventCreate failed");
    }
  }
  t.fam = new int[capacity];
  t.bytes = new double[capacity];
  t.mask = mask; t.cap = capacity; t.count = 0;
  return GC_OK;
}
int gencomm_timer_start(int family, int capacity) {
  GC_CHECK_ARG(family >= 0 && family < KF_COUNT, "bad family");
  return gencomm_timer_start_mask(1ull << family, capacity);
}

int gencomm_timer_stop_families(double* ms, int* launches, double* algorithmic_bytes, int n_families) {
  KernelTimer& t = kernel_timer();
  GC_CHECK_ARG(t.ev != nullptr, "timer not armed");
  GC_CHECK_ARG(n_families >= 0 && n_families <= KF_COUNT, "bad n_families");
  for (int f = 0; f < n_families; ++f) {
    if (ms) ms[f] = 0.0;
    if (launches) launches[f] = 0;
    if (algorithmic_bytes) algorithmic_bytes[f] = 0.0;
  }
  int rc = GC_OK;
  for (int i = 0; i < t.count; ++i) {
    float e = 0.f;
    if (hipEventSynchronize(t.ev[2 * i + 1]) != hipSuccess || hipEventElapsedTime(&e, t.ev[2 * i], t.ev[2 * i + 1]) != hipSuccess)
      rc = fail(GC_ERR_HIP, "event timing failed");
    const int f = t.fam[i];
    if (f < n_families) {
      if (ms) ms[f] += e;
      if (launches) launches[f] += 1;
      if (algorithmic_bytes) algorithmic_bytes[f] += t.bytes[i];
    }
  }
  for (int i = 0; i < 2 * t.cap; ++i) (void)hipEventDestroy(t.ev[i]);
  delete[] t.ev;
  delete[] t.fam;
  delete[] t.bytes;
  t = KernelTimer{};
  return rc;
}

int gencomm_timer_stop(double* total_ms, int* launches) {
  double ms[KF_COUNT];
  int n[KF_COUNT];
  const int rc = gencomm_timer_stop_families(ms, n, nullptr, KF_COUNT);
  double sum = 0.0;
  int cnt = 0;
  for (int f = 0; f < KF_COUNT; ++f) { sum += ms[f]; cnt += n[f]; }
  if (total_ms) *total_ms = sum;
  if (launches) *launches = cnt;
  return rc;
}

// ------------------------------------------------------------------------------------ kernel log
int gencomm_klog_start(void) {
  std::lock_guard<std::mutex> lk(g_klog_mu);
  g_klog.clear();
  g_klog_armed.store(true, std::memory_order_relaxed);
  return GC_OK;
}
__global__ void clock_probe_kernel(unsigned long long* out, long long spin_ticks) {
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  unsigned long long w1 = w0;
  while ((long long)(w1 - w0) < spin_ticks) w1 = wall_clock64();     // bounded by construction: the 100 MHz counter always advances
  const unsigned long long c1 = clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}
int gencomm_clock_probe(unsigned long long* out_dev, int spin_us, void* stream) {
  GC_CHECK_ARG(out_dev && spin_us >= 1 && spin_us <= 100000, "bad arguments");
  clock_probe_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out_dev, (long long)spin_us * 100);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_klog_stop(char* buf, int cap) {
  g_klog_armed.store(false, std::memory_order_relaxed);
  std::lock_guard<std::mutex> lk(g_klog_mu);
  std::string out;
  for (const auto& kv : g_klog) out += kv.first + "\t" + std::to_string(kv.second) + "\n";
  g_klog.clear();
  GC_CHECK_ARG(buf != nullptr && cap > 0, "null buffer");
  GC_CHECK_ARG((int)out.size() < cap, "buffer too small for the kernel log");
  memcpy(buf, out.c_str(), out.size() + 1);
  return GC_OK;
}

#ifdef GC_STAMPS
// diagnostic build only: copy the stamp buffer to the host (synchronises the device)
int gencomm_diag_read_stamps(unsigned long long* host, int nblocks) {
  GC_HIP(hipDeviceSynchronize());
  GC_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(gc::g_stamps), (size_t)nblocks * 8 * sizeof(unsigned long long)));
  return GC_OK;
}
#endif

// ------------------------------------------------------------------------------------ UNet
int gencomm_unet_num_params(int C, int levels, int res_blocks, int attn_mask) {
  UNetPlan p;
  if (const char* e = p.build(C, levels, res_blocks, attn_mask, 1)) { fail(GC_ERR_ARG, e); return -1; }
  return (int)p.params.size();
}

int gencomm_unet_param_info(int C, int levels, int res_blocks, int attn_mask, int index, char* name, int name_cap,
                            long long* numel, long long* offset) {
  UNetPlan p;
  if (const char* e = p.build(C, levels, res_blocks, attn_mask, 1)) return fail(GC_ERR_ARG, e);
  GC_CHECK_ARG(index >= 0 && index < (int)p.params.size(), "param index out of range");
  GC_CHECK_ARG(name && name_cap > 0 && numel && offset, "null output pointer");
  snprintf(name, (size_t)name_cap, "%s", p.params[index].name.c_str());
  *numel = p.params[index].numel;
  *offset = p.params[index].off;
  return GC_OK;
}

long long gencomm_unet_raw_floats(int C, int levels, int res_blocks, int attn_mask) {
  UNetPlan p;
  if (const char* e = p.build(C, levels, res_blocks, attn_mask, 1)) { fail(GC_ERR_ARG, e); return -1; }
  return p.raw_floats;
}

long long gencomm_unet_prepared_floats(int C, int levels, int res_blocks, int attn_mask, int T) {
  UNetPlan p;
  if (const char* e = p.build(C, levels, res_blocks, attn_mask, T)) { fail(GC_ERR_ARG, e); return -1; }
  return p.prepared_floats;
}

int gencomm_unet_prepare(const float* raw, float* prepared, int C, int levels, int res_blocks, int attn_mask, int T, void* stream) {
  UNetPlan p;
  if (const char* e = p.build(C, levels, res_blocks, attn_mask, T)) return fail(GC_ERR_ARG, e);
  GC_CHECK_ARG(raw && prepared, "null pointer");
  return unet_prepare_enqueue(p, raw, prepared, (hipStream_t)stream);
}

long long gencomm_denoise_workspace_bytes(int n, int C, int H, int W, int levels, int res_blocks, int attn_mask) {
  UNetPlan p;
  if (const char* e = p.build(C, levels, res_blocks, attn_mask, 1)) { fail(GC_ERR_ARG, e); return -1; }
  if (n < 1 || H < 1 || W < 1) { fail(GC_ERR_ARG, "n, H, W must be positive"); return -1; }
  UNetWorkspace w;
  if (const char* e = w.build(p, n, H, W)) { fail(GC_ERR_ARG, e); return -1; }
  return (long long)w.total;
}

// Diagnostic: the dataflow kernel's error word of the LAST UNet call on this workspace (0 = every dependency wait was satisfied;
// k + 1 = a workgroup gave up waiting for the predecessor of body op k).  Synchronises `stream`.
int gencomm_dataflow_error(const void* workspace, int n, int C, int H, int W, int levels, int res_blocks, int attn_mask, void* stream) {
  UNetPlan p;
  if (const char* e = p.build(C, levels, res_blocks, attn_mask, 1)) { fail(GC_ERR_ARG, e); return -1; }
  UNetWorkspace w;
  if (workspace == nullptr || n < 1) { fail(GC_ERR_ARG, "null workspace / bad n"); return -1; }
  if (const char* e = w.build(p, n, H, W)) { fail(GC_ERR_ARG, e); return -1; }
  unsigned v = 0;
  const char* src = (const char*)workspace + w.df_words_off + (8 + 64 * (size_t)n) * sizeof(unsigned);
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || hipMemcpy(&v, src, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) {
    fail(GC_ERR_HIP, "reading the dataflow error word failed");
    return -1;
  }
  return (int)v;
}
// Diagnostic: the first `count` words of the dataflow kernel's block ([8] tickets, [64][n] done counters, error word) to host memory.
int gencomm_dataflow_words(const void* workspace, int n, int C, int H, int W, int levels, int res_blocks, int attn_mask,
                           unsigned int* host_out, int count, void* stream) {
  UNetPlan p;
  if (const char* e = p.build(C, levels, res_blocks, attn_mask, 1)) return fail(GC_ERR_ARG, e);
  UNetWorkspace w;
  GC_CHECK_ARG(workspace && host_out && n >= 1 && count >= 1 && count <= 8 + 64 * n + 1, "bad arguments");
  if (const char* e = w.build(p, n, H, W)) return fail(GC_ERR_ARG, e);
  GC_HIP(hipStreamSynchronize((hipStream_t)stream));
  GC_HIP(hipMemcpy(host_out, (const char*)workspace + w.df_words_off, (size_t)count * sizeof(unsigned), hipMemcpyDeviceToHost));
  return GC_OK;
}

// bf16 denoise mode: one kernel family (64x16 tiles, vector loads), no AttnBlock
static int check_bf16_mode(const Modes& m, const UNetPlan& p, int W) {
  if (!m.bf16()) return GC_OK;
  GC_CHECK_ARG(p.attn_mask == 0, "bf16 denoise mode does not cover AttnBlocks");
  GC_CHECK_ARG(W % (4 << (p.L - 1)) == 0, "bf16 denoise mode needs W divisible by 4 at every resolution level");
  return GC_OK;
}

static int check_dims(int n, int C, int H, int W) {
  GC_CHECK_ARG(n >= 1 && n <= 65535, "n (agents) must be in 1..65535");
  GC_CHECK_ARG(H >= 1 && W >= 1 && (long long)H * W * (C + 8) < (1LL << 31), "C*H*W per agent must stay below 2^31 elements");
  GC_CHECK_ARG((long long)n * ((C + 15) / 16) <= 65535, "n * ceil(C/16) exceeds the grid z limit");
  return GC_OK;
}

int gencomm_unet_fwd(const float* prepared, const float* x_t, const float* cond, float* x0_out, int t,
                     int n, int C, int H, int W, int levels, int res_blocks, int attn_mask, int T,
                     void* workspace, long long workspace_bytes, void* stream) {
  UNetPlan p;
  if (const char* e = p.build(C, levels, res_blocks, attn_mask, T)) return fail(GC_ERR_ARG, e);
  if (int rc = check_dims(n, C, H, W)) return rc;
  GC_CHECK_ARG(prepared && x_t && cond && x0_out && workspace, "null pointer");
  GC_CHECK_ARG(t >= 0 && t < T, "timestep out of range");
  UNetWorkspace w;
  if (const char* e = w.build(p, n, H, W)) return fail(GC_ERR_ARG, e);
  if ((long long)w.total > workspace_bytes) return fail(GC_ERR_WORKSPACE, "workspace too small (see gencomm_denoise_workspace_bytes)");
  UNetCall c{&p, &w, prepared, (char*)workspace, n, H, W, (hipStream_t)stream, modes_snapshot()};
  if (int rc = check_bf16_mode(c.m, p, W)) return rc;
  GC_HIP(hipMemsetAsync(c.amax(), 0, 256, c.st));
  amax_kernel<<<256, 256, 0, c.st>>>(cond, (long long)n * 2 * H * W, c.amax());
  amax_kernel<<<1024, 256, 0, c.st>>>(x_t, (long long)n * C * H * W, c.amax() + 1);
  ConvOutArgs co{};
  co.out = x0_out;
  return unet_enqueue(c, x_t, cond, t, 0, co);
}

long long gencomm_unet_bwd_workspace_bytes(int n, int C, int H, int W, int levels, int res_blocks, int attn_mask) {
  UNetPlan p;
  if (const char* e = p.build(C, levels, res_blocks, attn_mask, 1, true)) { fail(GC_ERR_ARG, e); return -1; }
  if (n < 1 || H < 1 || W < 1) { fail(GC_ERR_ARG, "n, H, W must be positive"); return -1; }
  UNetWorkspace w;
  if (const char* e = w.build(p, n, H, W)) { fail(GC_ERR_ARG, e); return -1; }
  return (long long)unet_bwd_ws(p, w, n, H, W).total;
}

// The forward of a call that WILL be differentiated: every intermediate and its GroupNorm statistics are kept in `workspace`
// (gencomm_unet_bwd_workspace_bytes), which the caller hands to gencomm_unet_bwd(forward_done = 1) later -- no recomputation.
int gencomm_unet_fwd_train(const float* prepared, const float* x_t, const float* cond, float* x0_out, int t,
                           int n, int C, int H, int W, int levels, int res_blocks, int attn_mask, int T,
                           void* workspace, long long workspace_bytes, void* stream) {
  UNetPlan p;
  if (const char* e = p.build(C, levels, res_blocks, attn_mask, T, true)) return fail(GC_ERR_ARG, e);
  if (int rc = check_dims(n, C, H, W)) return rc;
  GC_CHECK_ARG(attn_mask == 0, "gencomm_unet_fwd_train: AttnBlock backward is not implemented (attn_mask must be 0)");
  GC_CHECK_ARG(prepared && x_t && cond && x0_out && workspace, "null pointer");
  GC_CHECK_ARG(t >= 0 && t < T, "timestep out of range");
  UNetWorkspace w;
  if (const char* e = w.build(p, n, H, W)) return fail(GC_ERR_ARG, e);
  const UNetBwdWs bw = unet_bwd_ws(p, w, n, H, W);
  if ((long long)bw.total > workspace_bytes) return fail(GC_ERR_WORKSPACE, "workspace too small (see gencomm_unet_bwd_workspace_bytes)");
  UNetCall c{&p, &w, prepared, (char*)workspace, n, H, W, (hipStream_t)stream, modes_snapshot()};
  GC_CHECK_ARG(!c.m.bf16(), "gencomm_unet_fwd_train keeps fp32 intermediates: not available in bf16 denoise mode (GENCOMM_MODE_ARITH = 2)");
  GC_HIP(hipMemsetAsync(c.amax(), 0, 256, c.st));
  amax_kernel<<<256, 256, 0, c.st>>>(cond, (long long)n * 2 * H * W, c.amax());
  amax_kernel<<<1024, 256, 0, c.st>>>(x_t, (long long)n * C * H * W, c.amax() + 1);
  ConvOutArgs co{};
  co.out = x0_out;
  return unet_enqueue(c, x_t, cond, t, 0, co);
}

// The same with the sampler's update of step t > 0 fused into conv_out's epilogue, as the inference loop runs it (cond_diff.py:272-315):
//   x_prev = coef1_t x0_hat + coef2_t x_t + sigma_t eps      (sched_row = the timestep's five schedule constants on the device)
// eps = step_noise [n][C][H][W] when given, else the sampler's Philox field of (seed, stream t) -- exactly the field gencomm_denoise_fwd
// adds.  x0_hat itself is not stored (the backward does not need it); x_t is only read.
int gencomm_unet_fwd_train_step(const float* prepared, const float* x_t, const float* cond, float* x_prev, int t, const float* sched_row,
                                const float* step_noise, unsigned long long seed, int n, int C, int H, int W, int levels, int res_blocks,
                                int attn_mask, int T, void* workspace, long long workspace_bytes, void* stream) {
  UNetPlan p;
  if (const char* e = p.build(C, levels, res_blocks, attn_mask, T, true)) return fail(GC_ERR_ARG, e);
  if (int rc = check_dims(n, C, H, W)) return rc;
  GC_CHECK_ARG(attn_mask == 0, "gencomm_unet_fwd_train_step: AttnBlock backward is not implemented (attn_mask must be 0)");
  GC_CHECK_ARG(prepared && x_t && cond && x_prev && sched_row && workspace && x_prev != x_t, "null pointer / in place (x_t is kept for the backward)");
  GC_CHECK_ARG(t >= 1 && t < T, "timestep out of range (step 0 has no update: gencomm_unet_fwd_train)");
  UNetWorkspace w;
  if (const char* e = w.build(p, n, H, W)) return fail(GC_ERR_ARG, e);
  const UNetBwdWs bw = unet_bwd_ws(p, w, n, H, W);
  if ((long long)bw.total > workspace_bytes) return fail(GC_ERR_WORKSPACE, "workspace too small (see gencomm_unet_bwd_workspace_bytes)");
  UNetCall c{&p, &w, prepared, (char*)workspace, n, H, W, (hipStream_t)stream, modes_snapshot()};
  GC_CHECK_ARG(!c.m.bf16(), "gencomm_unet_fwd_train_step keeps fp32 intermediates: not available in bf16 denoise mode (GENCOMM_MODE_ARITH = 2)");
  GC_HIP(hipMemsetAsync(c.amax(), 0, 256, c.st));
  amax_kernel<<<256, 256, 0, c.st>>>(cond, (long long)n * 2 * H * W, c.amax());
  amax_kernel<<<1024, 256, 0, c.st>>>(x_t, (long long)n * C * H * W, c.amax() + 1);
  ConvOutArgs co{};
  co.out = x_prev;
  co.xt = x_t;
  co.sched = sched_row;
  co.noise = step_noise;
  co.seed = seed;
  co.stream_id = (unsigned)t;
  const int rc = unet_enqueue(c, x_t, cond, t, step_noise != nullptr ? 1 : 2, co);
  return rc;
}

// grad_xt = alpha * (the call's gradient with respect to x_t) + beta * d_prev, formed in the epilogue of conv_in's input-gradient layer:
// the adjoint of the sampler chain's x_{t-1} = coef1 x0_hat(x_t) + coef2 x_t + ... with grad_x0 = d_prev = d x_{t-1} UNSCALED
// (alpha = coef1_t, beta = coef2_t).  grad_cond / grad_raw are the gradients for grad_x0 as given: the caller scales them by alpha when it
// accumulates them (everything in this call is linear in grad_x0).  d_prev == NULL: alpha, beta ignored (gencomm_unet_bwd).
int gencomm_unet_bwd_chain(const float* prepared, const float* raw, const float* x_t, const float* cond, int t, const float* grad_x0,
                           float alpha, float beta, const float* d_prev, float* grad_xt, float* grad_cond, float* grad_raw, int n, int C,
                           int H, int W, int levels, int res_blocks, int attn_mask, int T, int forward_done, void* workspace,
                           long long workspace_bytes, void* stream) {
  UNetPlan p;
  if (const char* e = p.build(C, levels, res_blocks, attn_mask, T, true)) return fail(GC_ERR_ARG, e);
  if (int rc = check_dims(n, C, H, W)) return rc;
  GC_CHECK_ARG(attn_mask == 0, "gencomm_unet_bwd: AttnBlock backward is not implemented (attn_mask must be 0)");
  GC_CHECK_ARG(prepared && raw && x_t && cond && grad_x0 && grad_xt && grad_cond && grad_raw && workspace, "null pointer");
  GC_CHECK_ARG(d_prev != grad_xt, "grad_xt must not alias d_prev (tiles read d_prev while others write grad_xt)");
  GC_CHECK_ARG(t >= 0 && t < T, "timestep out of range");
  UNetWorkspace w;
  if (const char* e = w.build(p, n, H, W)) return fail(GC_ERR_ARG, e);
  const UNetBwdWs bw = unet_bwd_ws(p, w, n, H, W);
  if ((long long)bw.total > workspace_bytes) return fail(GC_ERR_WORKSPACE, "workspace too small (see gencomm_unet_bwd_workspace_bytes)");
  const std::vector<DgradEntry> dg = dgrad_entries(p);
  UNetBwdCall b{UNetCall{&p, &w, prepared, (char*)workspace, n, H, W, (hipStream_t)stream, modes_snapshot()}, &bw, raw, grad_raw};
  b.dg = &dg;
  if (d_prev != nullptr) { b.chain_alpha = alpha; b.chain_beta = beta; b.chain_prev = d_prev; }
  GC_CHECK_ARG(!b.c.m.bf16(), "gencomm_unet_bwd reads fp32 intermediates: not available in bf16 denoise mode (GENCOMM_MODE_ARITH = 2)");
  return unet_bwd_enqueue(b, x_t, cond, t, grad_x0, grad_xt, grad_cond, forward_done != 0);
}
int gencomm_unet_bwd(const float* prepared, const float* raw, const float* x_t, const float* cond, int t, const float* grad_x0,
                     float* grad_xt, float* grad_cond, float* grad_raw, int n, int C, int H, int W, int levels, int res_blocks,
                     int attn_mask, int T, int forward_done, void* workspace, long long workspace_bytes, void* stream) {
  return gencomm_unet_bwd_chain(prepared, raw, x_t, cond, t, grad_x0, 1.0f, 0.0f, nullptr, grad_xt, grad_cond, grad_raw, n, C, H, W, levels,
                                res_blocks, attn_mask, T, forward_done, workspace, workspace_bytes, stream);
}

// One plain 8 -> 8 channel 3x3 convolution (stride 1, zero padding 1, bias, no norm, no residual) through the same
// kernels the UNet layers use -- the unit under test of tests/test_gpu_conv8.py.  `scratch` >= 4096 floats.
int gencomm_conv8_fwd(const float* src, const float* w_oihw, const float* bias, float* dst, double* dstat,
                      float* scratch, int n, int H, int W, int split, void* stream) {
  GC_CHECK_ARG(src && w_oihw && bias && dst && scratch, "null pointer");
  GC_CHECK_ARG(n >= 1 && n <= 65535 && H >= 1 && W >= 1, "bad n/H/W");
  hipStream_t st = (hipStream_t)stream;
  float* p_w = scratch;            // [8][9][8]
  float* p_wh = scratch + 1024;    // three-term tables + scale (HC_WTAB3 + 64 = 2752 floats)
  float* p_b = scratch + 3840;     // bias copy (16-B aligned)
  prep_conv_w_kernel<<<cdiv(576, 256), 256, 0, st>>>(w_oihw, p_w, 8, 8, 8);
  prep_conv8h_kernel<<<1, 256, 0, st>>>(w_oihw, p_wh, 8);
  GC_HIP(hipMemcpyAsync(p_b, bias, 8 * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (dstat) GC_HIP(hipMemsetAsync(dstat, 0, (size_t)n * 16 * sizeof(double), st));
  float* p_amax = scratch + 3856;  // device bound on max|src|: the f16-pipe kernel's range guard
  GC_HIP(hipMemsetAsync(p_amax, 0, sizeof(float), st));
  amax_kernel<<<256, 256, 0, st>>>(src, (long long)n * 8 * H * W, p_amax);
  const Modes m = modes_snapshot();
  Conv8Args a{};
  a.src[0] = src; a.w = p_w; a.wh = split ? p_wh : nullptr; a.bias = p_b; a.dst = dst; a.dstat = dstat;
  a.H = a.Hin = H; a.W = a.Win = W;
  a.xcd = m.xcd();
  a.amax = p_amax;
  if (split & 0x100) {  // diagnostic: only the selected terms of the three-term product (64x16 tiles, any map size)
    a.term_mask = split & 63;
    conv8h_kernel<1, false, false, 0, true><<<dim3(cdiv(W, 64), cdiv(H, 16), n), 256, 0, st>>>(a);
    GC_HIP(hipGetLastError());
    return GC_OK;
  }
  launch_conv8<1, false, false, 0>(m, pick_tile(m, n, H, W), a, n, st);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

// 3x3 stride-1 pad-1 convolution 16 -> 16 channels without bias on the UNet's exact-fp32 two-source kernel (conv8_kernel<.., NSRC = 2>),
// one launch per 8-channel output group.  transposed = 1: the input gradient of such a layer (w is the FORWARD weight [16][16][3][3]:
// dx[c] = sum_{o, tap} dy[o](. - tap) w[o][c][tap], i.e. channels swapped and taps flipped).
__global__ void prep_c16_kernel(const float* __restrict__ w, float* __restrict__ dst /*[2 groups][16 ic][9][8 oc] + 8 zeros*/, int transposed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 8) dst[2 * 1152 + i] = 0.f;
  if (i >= 2 * 1152) return;
  const int o = i & 7, tap = (i >> 3) % 9, ic = (i / 72) & 15, g = i / 1152;
  const int oc = 8 * g + o;
  dst[i] = transposed ? w[((size_t)ic * 16 + oc) * 9 + (8 - tap)] : w[((size_t)oc * 16 + ic) * 9 + tap];
}
long long gencomm_conv3x3_c16_scratch_floats(void) { return 2 * 1152 + 64; }
int gencomm_conv3x3_c16_fwd(const float* x, int x_ct, const float* w, int transposed, float* y, int y_ct, float* scratch, int n, int H, int W,
                            void* stream) {
  GC_CHECK_ARG(x && w && y && scratch && n >= 1 && H >= 1 && W >= 1 && x_ct >= 16 && y_ct >= 16 && (transposed == 0 || transposed == 1), "bad arguments");
  GC_CHECK_ARG(x != y, "in place is not supported (a tile reads its neighbours' pixels)");
  hipStream_t st = (hipStream_t)stream;
  prep_c16_kernel<<<cdiv(2 * 1152, 256), 256, 0, st>>>(w, scratch, transposed);
  Modes m = modes_snapshot();
  m.v[MODE_ARITH] = 1;   // exact fp32 (gradients pass through this layer)
  const size_t plane = (size_t)H * W;
  for (int g = 0; g < 2; ++g) {
    Conv8Args a{};
    a.src[0] = x; a.src[1] = x + 8 * plane; a.src_ct = x_ct;
    a.w = scratch + g * 1152; a.bias = scratch + 2 * 1152;
    a.dst = y + (size_t)8 * g * plane; a.dst_ct = y_ct;
    a.H = a.Hin = H; a.W = a.Win = W;
    a.xcd = m.xcd();
    launch_conv8<2, false, false, 0>(m, pick_tile(m, n, H, W), a, n, st);
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}

static void launch_q_sample(const QSampleArgs& q, int n, bool philox, hipStream_t st) {
  TimedLaunch tl(KF_Q_SAMPLE, st);
  // 8 octets (64 elements) per thread: few enough workgroups that the max|x| commit (one atomic each) stays cheap on small maps
  const dim3 qgrid((unsigned)std::max<long long>(1, std::min<long long>((q.per_agent / 64 + 255) / 256, 2048)), n);
  GC_KLOG(philox ? "q_sample_kernel<true> (in-kernel Philox)" : "q_sample_kernel<false> (explicit noise)");
  if (philox) q_sample_kernel<true><<<qgrid, 256, 0, st>>>(q);
  else q_sample_kernel<false><<<qgrid, 256, 0, st>>>(q);
}

int gencomm_q_sample_fwd(const float* sched_row, const float* feat, int n_feat_rows, const int* src_row,
                         const float* noise, unsigned long long seed, unsigned int stream_id,
                         float* out, int n, int C, int H, int W, void* stream) {
  GC_CHECK_ARG(sched_row && feat && src_row && out, "null pointer");
  GC_CHECK_ARG(n >= 1 && n <= 65535 && n_feat_rows >= 1 && C >= 1 && H >= 1 && W >= 1, "bad n/C/H/W");
  QSampleArgs q{feat, src_row, noise, sched_row, out, seed, stream_id, (long long)C * H * W, nullptr, nullptr};
  launch_q_sample(q, n, noise == nullptr, (hipStream_t)stream);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_step_noise_fwd(const float* sched_row, unsigned long long seed, unsigned int stream_id, float* out,
                           int n, int C, int H, int W, int unrounded, void* stream) {
  GC_CHECK_ARG(sched_row && out, "null pointer");
  GC_CHECK_ARG(n >= 1 && n <= 65535 && C >= 2 && (C & 1) == 0 && H >= 1 && W >= 1, "bad n/C/H/W (C must be even)");
  StepNoiseArgs a{out, sched_row, seed, stream_id, C, H, W, unrounded ? 1 : 0, (long long)n * (C / 2) * H * ((W + 3) / 4)};
  const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((a.items + 255) / 256, 65536));
  step_noise_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_denoise_fwd(const float* prepared, const float* sched,
                        const float* feat, int n_feat_rows, const int* src_row, const float* cond,
                        float* out, const float* noise0, const float* step_noise, unsigned long long seed,
                        int n, int C, int H, int W, int levels, int res_blocks, int attn_mask, int T,
                        void* workspace, long long workspace_bytes, void* stream) {
  return gencomm_denoise_fwd_dseed(prepared, sched, feat, n_feat_rows, src_row, cond, out, noise0, step_noise, seed, nullptr,
                                   n, C, H, W, levels, res_blocks, attn_mask, T, workspace, workspace_bytes, stream);
}

int gencomm_denoise_fwd_dseed(const float* prepared, const float* sched,
                              const float* feat, int n_feat_rows, const int* src_row, const float* cond,
                              float* out, const float* noise0, const float* step_noise, unsigned long long seed,
                              const unsigned long long* seed_dev,
                              int n, int C, int H, int W, int levels, int res_blocks, int attn_mask, int T,
                              void* workspace, long long workspace_bytes, void* stream) {
  UNetPlan p;
  if (const char* e = p.build(C, levels, res_blocks, attn_mask, T)) return fail(GC_ERR_ARG, e);
  if (int rc = check_dims(n, C, H, W)) return rc;
  GC_CHECK_ARG(prepared && sched && feat && src_row && cond && out && workspace, "null pointer");
  GC_CHECK_ARG(n_feat_rows >= 1, "n_feat_rows must be positive");
  GC_CHECK_ARG((noise0 == nullptr) == (step_noise == nullptr), "noise0 and step_noise must both be given or both be null");
  UNetWorkspace w;
  if (const char* e = w.build(p, n, H, W)) return fail(GC_ERR_ARG, e);
  if ((long long)w.total > workspace_bytes) return fail(GC_ERR_WORKSPACE, "workspace too small (see gencomm_denoise_workspace_bytes)");
  hipStream_t st = (hipStream_t)stream;
  const bool philox = noise0 == nullptr;
  const long long per_agent = (long long)C * H * W;

  UNetCall c{&p, &w, prepared, (char*)workspace, n, H, W, st, modes_snapshot()};
  if (int rc = check_bf16_mode(c.m, p, W)) return rc;
  // range guard of conv_in on the f16 pipe: device bounds {max|cond|, max|x_T|} (common.h act_scale)
  GC_HIP(hipMemsetAsync(c.amax(), 0, 256, st));
  amax_kernel<<<256, 256, 0, st>>>(cond, (long long)n * 2 * H * W, c.amax());
  QSampleArgs q{feat, src_row, noise0, sched + (size_t)(T - 1) * 5, out, seed, (unsigned)T, per_agent, seed_dev, c.amax() + 1};
  launch_q_sample(q, n, philox, st);

  // Sampler structure: "latent" (default) carries the loop on the 8-channel map hs0 = conv_in(x_t)
  // (latent_kernels.h); "direct" is the literal conv_in ... conv_out + update per step.
  // Automatic choice: the latent structure exists to keep the C-channel x_t out of HBM -- on maps large enough for the
  // 64x16-tile kernels; on small maps (everything is cache-resident, the step's serial chain per workgroup dominates) the
  // literal structure is faster: shipped shape, one scene in flight, 0.95 -> 0.84 ms.  Same noise field either way.
  const long long smode = c.m.v[MODE_SAMPLER];  // 1 / 2 force one structure (tests compare both)
  const bool want_latent = smode == 2 || (smode == 0 && pick_tile(c.m, n, H, W) == TILE_64x16);
  const bool latent = want_latent && T >= 2 && (W % 4) == 0;
  if (!latent) {
    for (int i = 0; i < T; ++i) {
      const int t = T - 1 - i;
      ConvOutArgs co{};
      co.out = out;
      co.xt = out;
      co.sched = sched + (size_t)t * 5;
      co.noise = philox ? nullptr : step_noise + (size_t)i * n * per_agent;
      co.seed = seed;
      co.seed_dev = seed_dev;
      co.stream_id = (unsigned)t;
      const int post = t == 0 ? 0 : (philox ? 2 : 1);
      if (int rc = unet_enqueue(c, out, cond, t, post, co)) return rc;
    }
    return GC_OK;
  }
  const int nops = (int)p.ops.size();
  kmap_enqueue(c, cond);
  for (int i = 0; i < T; ++i) {
    const int t = T - 1 - i;
    ConvOutArgs co{};
    co.out = out;
    // step 0 of the loop starts from x_{T-1} through conv_in (op 0); later steps start from hs0
    if (int rc = unet_enqueue_range(c, out, cond, t, 0, co, i == 0 ? 0 : 1, t == 0 ? nops : nops - 1, i != 0, /*df_upload=*/i == 0)) return rc;
    if (t > 0) {
      const float* nz = philox ? nullptr : step_noise + (size_t)i * n * per_agent;
      if (int rc = latent_step_enqueue(c, sched + (size_t)t * 5, nz, seed, (unsigned)t, seed_dev)) return rc;
    }
  }
  return GC_OK;
}

// ------------------------------------------------------------------------------------ Enhancer
int gencomm_enhancer_num_params(int C) {
  EnhancerPlan p;
  if (const char* e = p.build(C)) { fail(GC_ERR_ARG, e); return -1; }
  return (int)p.params.size();
}
int gencomm_enhancer_param_info(int C, int index, char* name, int name_cap, long long* numel, long long* offset) {
  EnhancerPlan p;
  if (const char* e = p.build(C)) return fail(GC_ERR_ARG, e);
  GC_CHECK_ARG(index >= 0 && index < (int)p.params.size(), "param index out of range");
  GC_CHECK_ARG(name && name_cap > 0 && numel && offset, "null output pointer");
  snprintf(name, (size_t)name_cap, "%s", p.params[index].name.c_str());
  *numel = p.params[index].numel;
  *offset = p.params[index].off;
  return GC_OK;
}
long long gencomm_enhancer_raw_floats(int C) {
  EnhancerPlan p;
  if (const char* e = p.build(C)) { fail(GC_ERR_ARG, e); return -1; }
  return p.raw_floats;
}
long long gencomm_enhancer_workspace_bytes(int n, int C, int H, int W) {
  EnhancerPlan p;
  if (const char* e = p.build(C)) { fail(GC_ERR_ARG, e); return -1; }
  if (n < 1 || H < 1 || W < 1) { fail(GC_ERR_ARG, "n, H, W must be positive"); return -1; }
  return (long long)enhancer_workspace_bytes(p, n, H, W);
}
int gencomm_enhancer_fwd(const float* raw, const float* x, float* out, int n, int C, int H, int W,
                         void* workspace, long long workspace_bytes, void* stream) {
  EnhancerPlan p;
  if (const char* e = p.build(C)) return fail(GC_ERR_ARG, e);
  GC_CHECK_ARG(raw && x && workspace, "null pointer");  // out may be NULL: token-major result stays in the workspace
  GC_CHECK_ARG(n >= 1 && n <= 65535 && H >= 1 && W >= 1, "bad n/H/W");
  if ((long long)enhancer_workspace_bytes(p, n, H, W) > workspace_bytes)
    return fail(GC_ERR_WORKSPACE, "workspace too small (see gencomm_enhancer_workspace_bytes)");
  return enhancer_enqueue(p, raw, x, out, n, H, W, (char*)workspace, (hipStream_t)stream, modes_snapshot());
}

// ------------------------------------------------------------------------------------ message extractor
int gencomm_msgext_num_params(int C) {
  MsgExtPlan p;
  if (const char* e = p.build(C)) { fail(GC_ERR_ARG, e); return -1; }
  return (int)p.params.size();
}
int gencomm_msgext_param_info(int C, int index, char* name, int name_cap, long long* numel, long long* offset) {
  MsgExtPlan p;
  if (const char* e = p.build(C)) return fail(GC_ERR_ARG, e);
  GC_CHECK_ARG(index >= 0 && index < (int)p.params.size(), "param index out of range");
  GC_CHECK_ARG(name && name_cap > 0 && numel && offset, "null output pointer");
  snprintf(name, (size_t)name_cap, "%s", p.params[index].name.c_str());
  *numel = p.params[index].numel;
  *offset = p.params[index].off;
  return GC_OK;
}
long long gencomm_msgext_raw_floats(int C) {
  MsgExtPlan p;
  if (const char* e = p.build(C)) { fail(GC_ERR_ARG, e); return -1; }
  return p.raw_floats;
}
long long gencomm_msgext_workspace_bytes(int n, int C, int H, int W) {
  MsgExtPlan p;
  if (const char* e = p.build(C)) { fail(GC_ERR_ARG, e); return -1; }
  if (n < 1 || H < 1 || W < 1) { fail(GC_ERR_ARG, "n, H, W must be positive"); return -1; }
  return (long long)msgext_ws(p, n, H, W).total;
}
int gencomm_msgext_fwd(const float* raw, const float* x, float* out, int n, int C, int H, int W,
                       void* workspace, long long workspace_bytes, void* stream) {
  MsgExtPlan p;
  if (const char* e = p.build(C)) return fail(GC_ERR_ARG, e);
  GC_CHECK_ARG(raw && x && out && workspace, "null pointer");
  GC_CHECK_ARG(n >= 1 && n <= 65535 && H >= 1 && W >= 1 && (long long)H * W * (C + 64) < (1LL << 31), "bad n/H/W");
  if ((long long)msgext_ws(p, n, H, W).total > workspace_bytes)
    return fail(GC_ERR_WORKSPACE, "workspace too small (see gencomm_msgext_workspace_bytes)");
  return msgext_enqueue(p, raw, x, out, n, H, W, (char*)workspace, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------ PointPillars front half
int gencomm_pillar_encode_fwd(const float* voxel_features, const int* voxel_num_points, const int* voxel_coords,
                              const float* linear_w, const float* bn_weight, const float* bn_bias,
                              const float* bn_running_mean, const float* bn_running_var,
                              float* out, float* scratch128, int M, int P, int B, int nx, int ny,
                              const float* voxel_size3, const float* pc_range6, void* stream) {
  GC_CHECK_ARG(voxel_features && voxel_num_points && voxel_coords && linear_w && bn_weight && bn_bias &&
               bn_running_mean && bn_running_var && out && scratch128 && voxel_size3 && pc_range6, "null pointer");
  GC_CHECK_ARG(M >= 0 && P >= 1 && P <= 32 && B >= 1 && nx >= 1 && ny >= 1, "bad M/P/B/nx/ny (P <= 32 point slots)");
  hipStream_t st = (hipStream_t)stream;
  GC_HIP(hipMemsetAsync(out, 0, (size_t)B * 64 * nx * ny * sizeof(float), st));
  bn_fold_kernel<<<1, 64, 0, st>>>(bn_weight, bn_bias, bn_running_mean, bn_running_var, 1e-3f, scratch128, scratch128 + 64, 64);
  if (M > 0) {
    PillarArgs a{voxel_features, voxel_num_points, voxel_coords, linear_w, scratch128, scratch128 + 64, out, M, P, B, nx, ny,
                 voxel_size3[0], voxel_size3[1], voxel_size3[2],
                 voxel_size3[0] / 2 + pc_range6[0], voxel_size3[1] / 2 + pc_range6[1], voxel_size3[2] / 2 + pc_range6[2]};
    pillar_vfe_scatter_kernel<<<(M + 3) / 4, 256, 0, st>>>(a);
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}

// ------------------------------------------------------------------------------------ general conv (backbone / heads)
int gencomm_conv2d_prepare(const float* weight, float* prepared, int Cin, int Cout, int KH, int KW, int transposed, void* stream) {
  GC_CHECK_ARG(weight && prepared, "null pointer");
  GC_CHECK_ARG(Cin >= 1 && Cout >= 1 && KH >= 1 && KW >= 1, "bad Cin/Cout/KH/KW");
  GC_CHECK_ARG(transposed >= 0 && transposed <= 2, "transposed: 0 (Conv2d), 1 (ConvTranspose2d, kernel == stride) or 2 (input-gradient convolution)");
  const long long total = (long long)Cin * Cout * KH * KW;
  const int M = transposed == 1 ? Cout * KH * KW : Cout, T = transposed == 1 ? 1 : KH * KW;
  if (h3_eligible(Cin, M, transposed == 1 ? 1 : KH, transposed == 1 ? 1 : KW)) {   // row scales, then both forms in one launch: fp32 k-major + three-term operand units
    unsigned char* blob = reinterpret_cast<unsigned char*>(prepared + total);
    PrepW3Args a{weight, prepared, blob, reinterpret_cast<float*>(blob + h3_blob_bytes(Cin, M, T)), Cin, Cout, KH, KW, transposed,
                 M, T, h3_chunks(Cin), h3_blocks(M)};
    GC_CHECK_ARG(a.nchunk <= 65535 && T <= 65535, "too many input-channel chunks");
    conv_w3_rowscale_kernel<<<16 * a.nb, 256, 0, (hipStream_t)stream>>>(a);          // one wave per GEMM row
    conv_prep_w3_kernel<<<dim3(a.nb, a.nchunk, T), 128, 0, (hipStream_t)stream>>>(a);
    GC_HIP(hipGetLastError());
    return GC_OK;
  }
  PrepWArgs a{weight, prepared, Cin, Cout, KH, KW, transposed};
  conv_prep_w_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(a);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
long long gencomm_conv2d_prepared_floats(int Cin, int Cout, int KH, int KW, int transposed) {
  if (Cin < 1 || Cout < 1 || KH < 1 || KW < 1 || transposed < 0 || transposed > 2) {
    fail(GC_ERR_ARG, "gencomm_conv2d_prepared_floats: bad dims");
    return -1;
  }
  return conv2d_prepared_floats(Cin, Cout, KH, KW, transposed);
}
// the three-term operand form behind the fp32 form of a prepared buffer (present exactly for eligible GEMM shapes)
static void conv2d_attach_w3(Conv2dArgs& a, const float* prepared, int KH, int KW) {
  if (!h3_eligible(a.Cin, a.CoutP, KH, KW)) return;
  const unsigned char* blob = reinterpret_cast<const unsigned char*>(prepared + (size_t)a.Cin * a.CoutP * KH * KW);
  a.w3 = blob;
  a.wsc = reinterpret_cast<const float*>(blob + h3_blob_bytes(a.Cin, a.CoutP, KH * KW));
}

int gencomm_conv2d_fold(const float* bn_weight, const float* bn_bias, const float* bn_running_mean, const float* bn_running_var,
                        const float* conv_bias, float eps, int C, float* scale, float* shift, void* stream) {
  GC_CHECK_ARG(scale && shift && C >= 1, "null pointer / bad C");
  GC_CHECK_ARG((bn_weight == nullptr) == (bn_bias == nullptr) && (bn_weight == nullptr) == (bn_running_mean == nullptr) &&
               (bn_weight == nullptr) == (bn_running_var == nullptr), "BatchNorm tensors must be all present or all null");
  FoldArgs a{bn_weight, bn_bias, bn_running_mean, bn_running_var, conv_bias, scale, shift, eps, C};
  conv_fold_kernel<<<(C + 63) / 64, 64, 0, (hipStream_t)stream>>>(a);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_conv2d_fwd(const float* x, const float* prepared, const float* scale, const float* shift, float* y,
                       int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad, int relu,
                       int ups, int out_ctotal, int out_coff, void* stream) {
  GC_CHECK_ARG(x && prepared && scale && shift && y, "null pointer");
  GC_CHECK_ARG(N >= 1 && Cin >= 1 && H >= 1 && W >= 1 && Cout >= 1 && stride >= 1 && pad >= 0 && ups >= 1, "bad dims");
  // KH = KW = 2 with ups = 2: the sub-pixel form of a transposed 3x3 stride-2 pad-1 convolution (the input gradient of the backbone's
  // stride-2 layers): output pixel (2u + a, 2v + b) of channel c is GEMM row c * 4 + a * 2 + b of a 2x2 window over input rows u, u + 1
  // and columns v, v + 1 (zero beyond the last row / column) -- 16 instead of the 36 tap-products per output quad of a zero-stuffed input
  const bool subpixel = KH == 2 && KW == 2 && ups == 2 && stride == 1 && pad == 0;
  GC_CHECK_ARG(ups == 1 || subpixel || (KH == 1 && KW == 1 && stride == 1 && pad == 0),
               "ups > 1 runs as a 1x1 GEMM (ConvTranspose2d, kernel == stride) or as the 2x2 sub-pixel form of a transposed 3x3 stride-2 convolution");
  GC_CHECK_ARG(out_coff >= 0 && out_coff + Cout <= out_ctotal, "output channel slice out of range");
  const int Ho = subpixel ? H : (H + 2 * pad - KH) / stride + 1, Wo = subpixel ? W : (W + 2 * pad - KW) / stride + 1;
  GC_CHECK_ARG(Ho >= 1 && Wo >= 1, "empty output");
  Conv2dArgs a{x, prepared, scale, shift, y, Cin, H, W, Cout * ups * ups, Ho, Wo, stride, pad, relu, ups, out_ctotal, out_coff};
  conv2d_attach_w3(a, prepared, KH, KW);
  return conv2d_enqueue(a, N, KH, KW, (hipStream_t)stream);
}
// the same with act in {0 none, 1 ReLU, 2 erf-GELU, 3 ReLU applied AFTER the residual add} and an optional residual (layout of y)
int gencomm_conv2d_act_res_fwd(const float* x, const float* prepared, const float* scale, const float* shift, const float* residual, float* y,
                               int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad, int act, void* stream) {
  GC_CHECK_ARG(x && prepared && scale && shift && y, "null pointer");
  GC_CHECK_ARG(N >= 1 && Cin >= 1 && H >= 1 && W >= 1 && Cout >= 1 && stride >= 1 && pad >= 0 && act >= 0 && act <= 3, "bad dims");
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  GC_CHECK_ARG(Ho >= 1 && Wo >= 1, "empty output");
  Conv2dArgs a{x, prepared, scale, shift, y, Cin, H, W, Cout, Ho, Wo, stride, pad, act, 1, Cout, 0};
  a.res = residual;
  conv2d_attach_w3(a, prepared, KH, KW);
  return conv2d_enqueue(a, N, KH, KW, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------ training building blocks (NCHW fp32)
long long gencomm_conv2d_wgrad_scratch_floats(int N, int Cin, int Hi, int Wi, int Cout, int K, int stride, int pad) {
  if (!(N >= 1 && Cin >= 1 && Cout >= 1 && Hi >= 1 && Wi >= 1 && (K == 1 || K == 3) && (stride == 1 || stride == 2) && pad >= 0)) {
    fail(GC_ERR_ARG, "gencomm_conv2d_wgrad_scratch_floats: bad dims");
    return -1;
  }
  const int Ho = (Hi + 2 * pad - K) / stride + 1, Wo = (Wi + 2 * pad - K) / stride + 1;
  if (K == 3 && Cin >= 32 && Cout >= 32 && Ho >= 1 && Wo >= 1) return (long long)wgrad3x3_wide_scratch_floats(N, Cin, Cout, Ho, Wo, stride);
  return 0;
}
int gencomm_conv2d_wgrad_ws(const float* dy, const float* x, float* dw, float* db, int N, int Cin, int Hi, int Wi, int Cout,
                            int K, int stride, int pad, float* scratch, long long scratch_floats, void* stream) {
  GC_CHECK_ARG(dy && x && dw, "null pointer");
  GC_CHECK_ARG(N >= 1 && Cin >= 1 && Cout >= 1 && Hi >= 1 && Wi >= 1 && (K == 1 || K == 3) && (stride == 1 || stride == 2) && pad >= 0, "bad dims");
  const int Ho = (Hi + 2 * pad - K) / stride + 1, Wo = (Wi + 2 * pad - K) / stride + 1;
  GC_CHECK_ARG(Ho >= 1 && Wo >= 1, "empty output");
  // Linear layers: split-K GEMM on the matrix cores -- also narrow ones over many pixels (PillarVFE's Linear 10 -> 64 over 1.5 M points: the
  // 8 x 8-channel-chunk kernel below spent 8.2 ms there, 131 k workgroups each committing 640 atomics to the same 640 weights)
  if (K == 1 && stride == 1 && pad == 0 && ((Cin >= 32 && Cout >= 32) || ((long long)N * Hi * Wi >= (1 << 16) && std::max(Cin, Cout) >= 32)))
    return wgrad1x1_enqueue(dy, x, dw, db, N, Cin, Cout, Hi * Wi, (hipStream_t)stream);
  if (K == 3 && Cin >= 32 && Cout >= 32) {   // wide layers (BEV backbone): 64 x 64-channel implicit GEMM, nine tap accumulators per wave
    const long long need = (long long)wgrad3x3_wide_scratch_floats(N, Cin, Cout, Ho, Wo, stride);
    GC_CHECK_ARG(scratch == nullptr || scratch_floats >= need, "scratch smaller than gencomm_conv2d_wgrad_scratch_floats");
    return wgrad3x3_wide_enqueue(dy, x, dw, db, N, Cin, Hi, Wi, Cout, Ho, Wo, stride, pad, scratch, (hipStream_t)stream);
  }
  WgradArgs a{dy, x, nullptr, dw, db, Cout, Cin, 0, Ho, Wo, Hi, Wi, K, stride, pad, 0};
  return conv_wgrad_enqueue(a, N, (hipStream_t)stream);
}
int gencomm_conv2d_wgrad(const float* dy, const float* x, float* dw, float* db, int N, int Cin, int Hi, int Wi, int Cout,
                         int K, int stride, int pad, void* stream) {
  return gencomm_conv2d_wgrad_ws(dy, x, dw, db, N, Cin, Hi, Wi, Cout, K, stride, pad, nullptr, 0, stream);
}

// BatchNorm2d with batch statistics (training mode) around the HIP convolutions: scratch >= 2 C doubles (zeroed here)
int gencomm_bn2d_train_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* y, float* save,
                           double* scratch, float momentum, float eps, int relu, int n, int C, int HW, long long* num_batches_tracked,
                           void* stream) {
  GC_CHECK_ARG(x && gamma && beta && y && save && scratch && n >= 1 && n <= 65535 && C >= 1 && C <= 65535 && HW >= 1, "bad arguments");
  GC_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running statistics: both or neither");
  hipStream_t st = (hipStream_t)stream;
  if (!(relu & 4)) GC_HIP(hipMemsetAsync(scratch, 0, (size_t)C * 2 * sizeof(double), st));   // bit 2: the caller's scratch is zero already
  relu &= 1;
  const bool v4 = (HW & 3) == 0 && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0);   // four pixels per lane
  if (v4) bn2d_stats_kernel<4><<<dim3(C, (unsigned)std::min((HW + 1023) / 1024, 64)), 256, 0, st>>>(x, scratch, n, C, HW);
  else bn2d_stats_kernel<1><<<dim3(C, (unsigned)std::min<long long>(((long long)n * HW + 4095) / 4096, 64)), 256, 0, st>>>(x, scratch, n, C, HW);
  // mean / rstd, save[] and the running statistics are formed inside the apply kernel: one launch less
  if (v4) bn2d_apply_kernel<4><<<dim3((HW / 4 + 255) / 256, C, n), 256, 0, st>>>(x, scratch, save, running_mean, running_var, momentum, eps, (long long)n * HW, gamma, beta, y, C, HW, relu, num_batches_tracked);
  else bn2d_apply_kernel<1><<<dim3((HW + 255) / 256, C, n), 256, 0, st>>>(x, scratch, save, running_mean, running_var, momentum, eps, (long long)n * HW, gamma, beta, y, C, HW, relu, num_batches_tracked);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
// dx overwritten; dgamma / dbeta ACCUMULATED (+=, may be null); y = the forward's output (ReLU mask when relu != 0)
int gencomm_bn2d_train_bwd(const float* x, const float* y, const float* dy, const float* save, const float* gamma, float* dx, float* dgamma,
                           float* dbeta, double* scratch, int relu, int n, int C, int HW, void* stream) {
  GC_CHECK_ARG(x && y && dy && save && gamma && dx && scratch && n >= 1 && n <= 65535 && C >= 1 && C <= 65535 && HW >= 1, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (!(relu & 4)) GC_HIP(hipMemsetAsync(scratch, 0, (size_t)C * 2 * sizeof(double), st));   // bit 2: the caller's scratch is zero already
  relu &= 3;
  const bool v4 = (HW & 3) == 0 && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0);   // four pixels per lane
  if (v4) {
    bn2d_bwd_reduce_kernel<4><<<dim3(C, (unsigned)std::min((HW + 1023) / 1024, 64)), 256, 0, st>>>(x, y, dy, save, scratch, n, C, HW, relu & 1);
    bn2d_bwd_apply_kernel<4><<<dim3((HW / 4 + 255) / 256, C, n), 256, 0, st>>>(x, y, dy, save, gamma, scratch, dx, dgamma, dbeta, (long long)n * HW, C, HW, relu);
  } else {
    bn2d_bwd_reduce_kernel<1><<<dim3(C, (unsigned)std::min<long long>(((long long)n * HW + 4095) / 4096, 64)), 256, 0, st>>>(x, y, dy, save, scratch, n, C, HW, relu & 1);
    bn2d_bwd_apply_kernel<1><<<dim3((HW + 255) / 256, C, n), 256, 0, st>>>(x, y, dy, save, gamma, scratch, dx, dgamma, dbeta, (long long)n * HW, C, HW, relu);
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}

// max over the P point slots of every pillar (training path of the PointPillars encoder): x [C][M][P] -> out [M][C], arg [M][C] (uint8)
// One conv -> BatchNorm2d(batch statistics) -> ReLU layer of the backbones in training mode per call (base_bev_backbone.py:40-83): the
// forward is prepare + convolution + statistics + normalisation, the backward (stride 1) BatchNorm backward + weight gradient + input
// gradient -- compositions of the entries above, so that a layer costs the host ONE foreign call per direction instead of three / four
// (the stage-1 step is ~1 400 launches; on a slow host its Python side, not the device, sets the step time).
int gencomm_convbn_train_fwd(const float* x, const float* weight, const float* bias, const float* unit_scale, const float* zero_shift,
                             const float* gamma, const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked,
                             float momentum, float eps, int relu, float* prepared, float* pre, float* y, float* save, double* stat_scratch,
                             int N, int Cin, int H, int W, int Cout, int K, int stride, int pad, void* stream) {
  GC_CHECK_ARG(x && weight && unit_scale && zero_shift && gamma && beta && prepared && pre && y && save && stat_scratch, "null pointer");
  GC_CHECK_ARG((K == 1 || K == 3) && (stride == 1 || stride == 2) && pad >= 0, "gencomm_convbn_train_fwd: 1x1 / 3x3, stride 1 / 2");
  const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  GC_CHECK_ARG(Ho >= 1 && Wo >= 1, "empty output");
  if (int rc = gencomm_conv2d_prepare(weight, prepared, Cin, Cout, K, K, 0, stream)) return rc;
  if (int rc = gencomm_conv2d_fwd(x, prepared, unit_scale, bias != nullptr ? bias : zero_shift, pre, N, Cin, H, W, Cout, K, K, stride, pad, 0, 1, Cout, 0, stream)) return rc;
  return gencomm_bn2d_train_fwd(pre, gamma, beta, running_mean, running_var, y, save, stat_scratch, momentum, eps, (relu & 1) | 4, N, Cout, Ho * Wo,
                                num_batches_tracked, stream);   // stat_scratch arrives ZEROED
}
// Backward of the same layer for stride 1.  dpre / prepared / wgrad_scratch: caller-owned scratch ([N][Cout][Ho][Wo] floats;
// gencomm_conv2d_prepared_floats(Cout, Cin, K, K, 2); gencomm_conv2d_wgrad_scratch_floats(...)); dw and dbias (may be null) arrive ZEROED and are
// accumulated into, dgamma / dbeta are written, stat_scratch (2 Cout doubles) arrives ZEROED; dx null = no input gradient wanted.
int gencomm_convbn_train_bwd(const float* x, const float* weight, const float* pre, const float* y, const float* gy, const float* save, const float* gamma,
                             const float* unit_scale, const float* zero_shift, int relu, float* dpre, float* dx, float* dw, float* dbias, float* dgamma,
                             float* dbeta, double* stat_scratch, float* prepared, float* wgrad_scratch, long long wgrad_scratch_floats,
                             int N, int Cin, int H, int W, int Cout, int K, int pad, void* stream) {
  GC_CHECK_ARG(x && weight && pre && y && gy && save && gamma && unit_scale && zero_shift && dpre && dw && dgamma && dbeta && stat_scratch, "null pointer");
  GC_CHECK_ARG((K == 1 || K == 3) && pad >= 0 && pad <= K - 1, "gencomm_convbn_train_bwd: 1x1 / 3x3, stride 1");
  const int Ho = H + 2 * pad - K + 1, Wo = W + 2 * pad - K + 1;
  GC_CHECK_ARG(Ho >= 1 && Wo >= 1, "empty output");
  if (int rc = gencomm_bn2d_train_bwd(pre, y, gy, save, gamma, dpre, dgamma, dbeta, stat_scratch, (relu & 1) | 2 | 4, N, Cout, Ho * Wo, stream)) return rc;
  if (int rc = gencomm_conv2d_wgrad_ws(dpre, x, dw, dbias, N, Cin, H, W, Cout, K, 1, pad, wgrad_scratch, wgrad_scratch_floats, stream)) return rc;
  if (dx == nullptr) return GC_OK;
  GC_CHECK_ARG(prepared != nullptr, "gencomm_convbn_train_bwd: the input gradient needs the prepared-weight scratch");
  if (int rc = gencomm_conv2d_prepare(weight, prepared, Cout, Cin, K, K, 2, stream)) return rc;
  return gencomm_conv2d_fwd(dpre, prepared, unit_scale, zero_shift, dx, N, Cout, Ho, Wo, Cin, K, K, 1, K - 1 - pad, 0, 1, Cin, 0, stream);
}

// PFNLayer in training mode without its [M P, C] intermediates (pfn_kernels.h).  F = 10 (the shipped encoder: 4 raw + 3 cluster + 3 centre
// features) or 9 / 11 (no intensity / with distance); C in {32, 64, 128, 256}; P <= 255 (the arg slot is a byte)
static int pfn_check(const char* who, int M, int P, int F, int C) {
  if (!(M >= 1 && P >= 1 && P <= 255 && (F == 9 || F == 10 || F == 11) && (C == 32 || C == 64 || C == 128 || C == 256))) {
    char b[160];
    snprintf(b, sizeof b, "%s: supported: F in {9, 10, 11}, C in {32, 64, 128, 256}, P <= 255", who);
    return fail(GC_ERR_ARG, b);
  }
  if ((size_t)(256 / C) * P * F * sizeof(float) > 64 * 1024) return fail(GC_ERR_ARG, "pfn: a block's pillars do not fit the LDS");
  return GC_OK;
}
int gencomm_pfn_train_fwd(const float* feats, const float* weight, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          long long* num_batches_tracked, float momentum, float eps, float* out, unsigned char* arg, float* save, double* moments,
                          int M, int P, int F, int C, void* stream) {
  GC_CHECK_ARG(feats && weight && gamma && beta && out && arg && save && moments, "null pointer");
  GC_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running statistics: both or neither");
  if (int rc = pfn_check("gencomm_pfn_train_fwd", M, P, F, C)) return rc;
  PfnArgs a{feats, weight, gamma, beta, moments, save, M, P, F, C};
  hipStream_t st = (hipStream_t)stream;
  if (F == 9) return pfn_train_fwd_t<9>(a, out, arg, running_mean, running_var, num_batches_tracked, momentum, eps, st);
  if (F == 10) return pfn_train_fwd_t<10>(a, out, arg, running_mean, running_var, num_batches_tracked, momentum, eps, st);
  return pfn_train_fwd_t<11>(a, out, arg, running_mean, running_var, num_batches_tracked, momentum, eps, st);
}
int gencomm_pfn_train_bwd(const float* feats, const float* weight, const float* gamma, const float* beta, const float* save, const double* moments,
                          const float* gout, const unsigned char* arg, float* dweight, float* dgamma, float* dbeta, double* scratch,
                          int M, int P, int F, int C, void* stream) {
  GC_CHECK_ARG(feats && weight && gamma && beta && save && moments && gout && arg && scratch, "null pointer");
  if (int rc = pfn_check("gencomm_pfn_train_bwd", M, P, F, C)) return rc;
  PfnArgs a{feats, weight, gamma, beta, const_cast<double*>(moments), const_cast<float*>(save), M, P, F, C};
  hipStream_t st = (hipStream_t)stream;
  if (F == 9) return pfn_train_bwd_t<9>(a, gout, arg, dweight, dgamma, dbeta, scratch, st);
  if (F == 10) return pfn_train_bwd_t<10>(a, gout, arg, dweight, dgamma, dbeta, scratch, st);
  return pfn_train_bwd_t<11>(a, gout, arg, dweight, dgamma, dbeta, scratch, st);
}
long long gencomm_pfn_moment_doubles(int F) { return F >= 1 && F <= PFN_MAXF ? (long long)pfn_moment_doubles(F) : -1; }
long long gencomm_pfn_bwd_scratch_doubles(int F, int C) { return F >= 1 && F <= PFN_MAXF && C >= 1 ? (long long)PFN_BWD_BLOCKS * C * (F + 2) : -1; }

int gencomm_slot_max_fwd(const float* x, float* out, unsigned char* arg, int C, int M, int P, void* stream) {
  GC_CHECK_ARG(x && out && arg && C >= 1 && M >= 0 && P >= 1 && P <= 255, "bad arguments");
  if (M == 0) return GC_OK;
  const bool grouped = (P & 3) == 0 && P <= 64 && ((P / 4) & (P / 4 - 1)) == 0;   // P / 4 lanes per row (train_kernels.h)
  const long long total = (long long)C * M * (grouped ? P / 4 : 1);
  GC_CHECK_ARG((total + 255) / 256 < (1ll << 31), "too many pillars");
  slot_max_fwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, out, arg, C, M, P);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
int gencomm_slot_max_bwd(const float* dout, const unsigned char* arg, float* dx, int C, int M, int P, void* stream) {
  GC_CHECK_ARG(dout && arg && dx && C >= 1 && M >= 0 && P >= 1 && P <= 255, "bad arguments");
  if (M == 0) return GC_OK;
  const long long total = (long long)C * M * P;
  slot_max_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(dout, arg, dx, C, M, P);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

// deformable 3x3 convolution split into sampling + GEMM for the training path of MessageExtractorv2 (train_kernels.h)
int gencomm_dcn_sample_fwd(const float* x, const float* offset, float* col, int n, int C, int H, int W, void* stream) {
  GC_CHECK_ARG(x && offset && col && n >= 1 && n <= 65535 && C >= 1 && H >= 1 && W >= 1, "bad arguments");
  dcn_sample_kernel<<<dim3((H * W + 255) / 256, 9, n), 256, 0, (hipStream_t)stream>>>(x, offset, col, C, H, W);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
long long gencomm_dcn_scatter_scratch_floats(int n, int C, int H, int W) {
  if (!(n >= 1 && C >= 1 && H >= 1 && W >= 1)) { fail(GC_ERR_ARG, "gencomm_dcn_scatter_scratch_floats: bad dims"); return -1; }
  return (long long)dcn_scatter_scratch_floats(n, C, H, W);
}
int gencomm_dcn_scatter_bwd_ws(const float* x, const float* offset, const float* dcol, float* dx, float* doffset, int n, int C, int H, int W,
                               float* scratch, long long scratch_floats, void* stream) {
  GC_CHECK_ARG(x && offset && dcol && dx && doffset && n >= 1 && n <= 65535 && C >= 1 && H >= 1 && W >= 1, "bad arguments");
  GC_CHECK_ARG((C + DCN_CH - 1) / DCN_CH <= 65535 && C <= 65535, "too many channels");
  GC_CHECK_ARG(scratch == nullptr || scratch_floats >= (long long)dcn_scatter_scratch_floats(n, C, H, W), "scratch smaller than gencomm_dcn_scatter_scratch_floats");
  hipStream_t st = (hipStream_t)stream;
  // offset gradients per (pixel, tap); the input gradient through LDS tiles (dx zeroed by the caller)
  dcn_scatter_bwd_kernel<false><<<dim3((H * W + 255) / 256, 9, n), 256, 0, st>>>(x, offset, dcol, dx, doffset, C, H, W);
  const unsigned tiles = ((H + DCN_T - 1) / DCN_T) * ((W + DCN_T - 1) / DCN_T), chunks = (C + DCN_CH - 1) / DCN_CH;
  dcn_scatter_dx_tile_kernel<<<dim3(tiles, chunks, n), 256, 0, st>>>(offset, dcol, dx, C, H, W, scratch);
  if (scratch != nullptr) dcn_scatter_dx_gather_kernel<<<dim3((H * W + 255) / 256, C, n), 256, 0, st>>>(scratch, dx, C, H, W, (int)chunks * DCN_CH);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
int gencomm_dcn_scatter_bwd(const float* x, const float* offset, const float* dcol, float* dx, float* doffset, int n, int C, int H, int W,
                            void* stream) {
  return gencomm_dcn_scatter_bwd_ws(x, offset, dcol, dx, doffset, n, C, H, W, nullptr, 0, stream);
}

// GroupNorm (+ SiLU when silu != 0) over NCHW for any channel count; stat: n * groups * 2 floats of scratch (mean, rstd)
int gencomm_gn_nchw_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stat, float eps, int silu,
                        int n, int C, int groups, int HW, void* stream) {
  GC_CHECK_ARG(x && gamma && beta && y && stat && n >= 1 && n <= 65535 && C >= 1 && C <= 65535 && groups >= 1 && C % groups == 0 && HW >= 1, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  gn_nchw_stats_kernel<<<dim3(groups, n), 256, 0, st>>>(x, stat, C, groups, HW, eps);
  gn_nchw_apply_kernel<<<dim3((HW + 255) / 256, C, n), 256, 0, st>>>(x, stat, gamma, beta, y, C, groups, HW, silu);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_ln_nchw_fwd(const float* x, const float* gamma, const float* beta, float* out, float eps, int residual, int n, int C, int HW, void* stream) {
  GC_CHECK_ARG(x && gamma && beta && out && n >= 1 && n <= 65535 && C >= 1 && HW >= 1, "bad arguments");
  LnArgs a{x, gamma, beta, nullptr, out, nullptr, eps, C, HW, residual, 0};
  const dim3 g4((HW + 63) / 64, n);
  if (C <= 64) ln_nchw_fwd4_kernel<16><<<g4, 256, 0, (hipStream_t)stream>>>(a);
  else if (C <= 128) ln_nchw_fwd4_kernel<32><<<g4, 256, 0, (hipStream_t)stream>>>(a);
  else if (C <= 256) ln_nchw_fwd4_kernel<64><<<g4, 256, 0, (hipStream_t)stream>>>(a);
  else ln_nchw_fwd_kernel<<<dim3((HW + 255) / 256, n), 256, 0, (hipStream_t)stream>>>(a);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_ln_nchw_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, float* scratch,
                        float eps, int accumulate, int n, int C, int HW, void* stream) {
  GC_CHECK_ARG(x && gamma && dy && dx && dgamma && dbeta && scratch && n >= 1 && n <= 65535 && C >= 1 && C <= 65535 && HW >= 1, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  LnArgs a{x, gamma, nullptr, dy, dx, scratch, eps, C, HW, 0, accumulate};
  const dim3 g4((HW + 63) / 64, n);
  if (C <= 64) ln_nchw_bwd4_kernel<16><<<g4, 256, 0, st>>>(a);
  else if (C <= 128) ln_nchw_bwd4_kernel<32><<<g4, 256, 0, st>>>(a);
  else if (C <= 256) ln_nchw_bwd4_kernel<64><<<g4, 256, 0, st>>>(a);
  else ln_nchw_bwd_kernel<<<dim3((HW + 255) / 256, n), 256, 0, st>>>(a);
  ln_nchw_param_grad_kernel<<<dim3(C, (unsigned)std::min<long long>(((long long)n * HW + 4095) / 4096, 128)), 256, 0, st>>>(x, dy, scratch, dgamma, dbeta, n, C, HW);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_dwconv3x3_act_fwd(const float* x, int x_ct, int act, const float* w, const float* b, float* y, int n, int C, int H, int W, int flip, void* stream) {
  GC_CHECK_ARG(x && w && y && n >= 1 && C >= 1 && x_ct >= C && (long long)n * C <= 65535 && H >= 1 && W >= 1 && (act == 0 || act == 1), "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if ((W & 3) == 0) {   // sliding three-row window, 128-bit loads, neighbours by lane shuffle
    const dim3 grid((W / 4 + 63) / 64, (H + 4 * DW_RB - 1) / (4 * DW_RB), n * C);
    if (act) dwconv3x3_rows_kernel<true><<<grid, 256, 0, st>>>(x, w, b, y, C, H, W, flip, x_ct);
    else dwconv3x3_rows_kernel<false><<<grid, 256, 0, st>>>(x, w, b, y, C, H, W, flip, x_ct);
  } else {
    const dim3 grid((H * ((W + 3) / 4) + 255) / 256, n * C);
    if (act) dwconv3x3_kernel<true><<<grid, 256, 0, st>>>(x, w, b, y, C, H, W, flip, x_ct);
    else dwconv3x3_kernel<false><<<grid, 256, 0, st>>>(x, w, b, y, C, H, W, flip, x_ct);
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}
int gencomm_dwconv3x3_fwd(const float* x, const float* w, const float* b, float* y, int n, int C, int H, int W, int flip, void* stream) {
  return gencomm_dwconv3x3_act_fwd(x, C, 0, w, b, y, n, C, H, W, flip, stream);
}

int gencomm_dwconv3x3_act_wgrad(const float* x, int x_ct, int act, const float* dy, float* dw, float* db, int n, int C, int H, int W, void* stream) {
  GC_CHECK_ARG(x && dy && dw && n >= 1 && C >= 1 && x_ct >= C && C <= 65535 && H >= 1 && W >= 1 && (act == 0 || act == 1), "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if ((W & 3) == 0) {
    const long long items = (long long)n * ((H + DW_RB - 1) / DW_RB);
    const unsigned gy = (unsigned)std::min<long long>((items + 3) / 4, 32);   // <= 32 workgroups (atomics) per channel and column strip
    const dim3 grid((W / 4 + 63) / 64, gy, C);
    if (act) dwconv3x3_wgrad_rows_kernel<true><<<grid, 256, 0, st>>>(x, dy, dw, db, n, C, H, W, x_ct);
    else dwconv3x3_wgrad_rows_kernel<false><<<grid, 256, 0, st>>>(x, dy, dw, db, n, C, H, W, x_ct);
  } else {
    const dim3 grid(C, (unsigned)std::min<long long>(((long long)n * H * ((W + 3) / 4) + 1023) / 1024, 128));
    if (act) dwconv3x3_wgrad_kernel<true><<<grid, 256, 0, st>>>(x, dy, dw, db, n, C, H, W, x_ct);
    else dwconv3x3_wgrad_kernel<false><<<grid, 256, 0, st>>>(x, dy, dw, db, n, C, H, W, x_ct);
  }
  GC_HIP(hipGetLastError());
  return GC_OK;
}
int gencomm_dwconv3x3_wgrad(const float* x, const float* dy, float* dw, float* db, int n, int C, int H, int W, void* stream) {
  return gencomm_dwconv3x3_act_wgrad(x, C, 0, dy, dw, db, n, C, H, W, stream);
}

int gencomm_gelu_bwd(const float* v, const float* g, float* out, long long count, void* stream) {
  GC_CHECK_ARG(v && g && out && count >= 0 && (count + 255) / 256 < (1LL << 31), "bad arguments");
  if (count > 0) gelu_bwd_kernel<<<(unsigned)((count + 255) / 256), 256, 0, (hipStream_t)stream>>>(v, g, out, count);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_ew_slice_fwd(int op, const float* a, const float* b, const float* c, const float* d, float* o0, float* o1, int n, int nch, int HW,
                         int a_ct, int a_c0, int o0_ct, int o0_c0, int o1_ct, int o1_c0, void* stream) {
  GC_CHECK_ARG(a && o0 && n >= 1 && n <= 65535 && nch >= 1 && nch <= 65535 && HW >= 1 && op >= 0 && op <= 6, "bad arguments");
  SliceArgs s{a, b, c, d, o0, o1, n, nch, HW, a_ct, a_c0, 0, 0, o0_ct, o0_c0, o1_ct, o1_c0};
  hipStream_t st = (hipStream_t)stream;
  // HW % 4 == 0 and 16-byte aligned tensors: every slice base is aligned too -> four pixels per lane
  const bool v4 = (HW & 3) == 0 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)o0 | (uintptr_t)o1) & 15) == 0);
  const dim3 grid(v4 ? (HW / 4 + 255) / 256 : (HW + 255) / 256, nch, n);
#define GC_EW_LAUNCH(OP) do { if (v4) ew_slice4_kernel<OP><<<grid, 256, 0, st>>>(s); else ew_slice_kernel<OP><<<grid, 256, 0, st>>>(s); } while (0)
  switch (op) {
    case EW_COPY: GC_EW_LAUNCH(EW_COPY); break;
    case EW_GELU_SPLIT: GC_CHECK_ARG(o1, "null pointer"); GC_EW_LAUNCH(EW_GELU_SPLIT); break;
    case EW_GELU_GATE: GC_CHECK_ARG(b, "null pointer"); GC_EW_LAUNCH(EW_GELU_GATE); break;
    case EW_GATE_BWD: GC_CHECK_ARG(b && c && d && o1, "null pointer"); GC_EW_LAUNCH(EW_GATE_BWD); break;
    case EW_GELU_BWD: GC_CHECK_ARG(b, "null pointer"); GC_EW_LAUNCH(EW_GELU_BWD); break;
    case EW_GELU2_GATE: GC_CHECK_ARG(d, "null pointer"); GC_EW_LAUNCH(EW_GELU2_GATE); break;
    default: GC_CHECK_ARG(c && d && o1, "null pointer"); GC_EW_LAUNCH(EW_GATE_BWD2); break;
  }
#undef GC_EW_LAUNCH
  GC_HIP(hipGetLastError());
  return GC_OK;
}
int gencomm_nc_scale_fwd(const float* x, const float* a, const float* b, float* out, int n, int C, int HW, void* stream) {
  GC_CHECK_ARG(x && a && out && n >= 1 && C >= 1 && (long long)n * C <= 65535 && HW >= 1, "bad arguments");
  if ((HW & 3) == 0 && ((((uintptr_t)x | (uintptr_t)out) & 15) == 0)) nc_scale_kernel<4><<<dim3((HW / 4 + 255) / 256, n * C), 256, 0, (hipStream_t)stream>>>(x, a, b, out, HW);
  else nc_scale_kernel<1><<<dim3((HW + 255) / 256, n * C), 256, 0, (hipStream_t)stream>>>(x, a, b, out, HW);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
int gencomm_nc_dot_fwd(const float* x, const float* y, float* out, int n, int C, int HW, void* stream) {
  GC_CHECK_ARG(x && out && n >= 1 && C >= 1 && (long long)n * C <= 65535 && HW >= 1, "bad arguments");
  GC_HIP(hipMemsetAsync(out, 0, (size_t)n * C * sizeof(float), (hipStream_t)stream));
  if ((HW & 3) == 0 && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0)) nc_dot_kernel<4><<<dim3(std::min((HW / 4 + 255) / 256, 64), n * C), 256, 0, (hipStream_t)stream>>>(x, y, out, HW);
  else nc_dot_kernel<1><<<dim3(std::min((HW + 255) / 256, 64), n * C), 256, 0, (hipStream_t)stream>>>(x, y, out, HW);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_lincomb_fwd(float* out, const float* x, const float* y, const float* z, float a, float b, float c, long long count, void* stream) {
  GC_CHECK_ARG(out && x && count >= 0 && (count + 1023) / 1024 < (1LL << 31), "bad arguments");
  GC_CHECK_ARG((((uintptr_t)out | (uintptr_t)x | (uintptr_t)y | (uintptr_t)z) & 15) == 0, "pointers must be 16-byte aligned");
  if (count > 0) lincomb_kernel<<<(unsigned)((count + 1023) / 1024), 256, 0, (hipStream_t)stream>>>(out, x, y, z, a, b, c, count);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

// ------------------------------------------------------------------------------------ detection tail
long long gencomm_det_workspace_bytes(int H, int W, int A) {
  if (H < 1 || W < 1 || A < 1) { fail(GC_ERR_ARG, "bad H/W/A"); return -1; }
  return det_workspace_bytes(H, W, A);
}
long long gencomm_nms_workspace_bytes(void) { return nms_workspace_bytes(); }

int gencomm_det_decode_fwd(const float* cls_preds, const float* reg_preds, const float* dir_preds, const float* anchors,
                           const float* transformation_matrix, int H, int W, int A, int num_bins, float score_threshold,
                           float dir_offset, int order_hwl, float* corners, float* scores, int* anchor_index, int* count,
                           int capacity, void* workspace, long long workspace_bytes, void* stream) {
  GC_CHECK_ARG(cls_preds && reg_preds && anchors && transformation_matrix && corners && scores && anchor_index && count && workspace, "null pointer");
  GC_CHECK_ARG(H >= 1 && W >= 1 && A >= 1 && capacity >= 1 && (dir_preds == nullptr || num_bins >= 1), "bad H/W/A/capacity/num_bins");
  GC_CHECK_ARG((long long)H * W * A < (1LL << 31), "too many anchors");
  if (det_workspace_bytes(H, W, A) > workspace_bytes) return fail(GC_ERR_ARG, "workspace too small (gencomm_det_workspace_bytes)");
  DetArgs a{};
  a.cls = cls_preds; a.reg = reg_preds; a.dir = dir_preds; a.anchors = anchors; a.T = transformation_matrix;
  a.corners = corners; a.scores = scores; a.anchor_idx = anchor_index; a.count = count;
  a.H = H; a.W = W; a.A = A; a.nb = num_bins; a.cap = capacity; a.hwl = order_hwl;
  a.thr = score_threshold; a.dir_offset = dir_offset;
  return det_decode_enqueue(a, workspace, (hipStream_t)stream);
}

int gencomm_nms_rotated_fwd(const float* corners, const float* scores, const int* n_candidates, float iou_threshold, int top,
                            const float* keep_range6, float* out_boxes, float* out_scores, int* out_index, int* out_count,
                            void* workspace, long long workspace_bytes, void* stream) {
  GC_CHECK_ARG(corners && scores && n_candidates && out_boxes && out_scores && out_index && out_count && workspace, "null pointer");
  GC_CHECK_ARG(top >= 1 && top <= kNmsMaxTop, "top must be in 1..1024");
  if (nms_workspace_bytes() > workspace_bytes) return fail(GC_ERR_ARG, "workspace too small (gencomm_nms_workspace_bytes)");
  return nms_rotated_enqueue(corners, scores, n_candidates, iou_threshold, top, keep_range6, out_boxes, out_scores, out_index,
                             out_count, workspace, (hipStream_t)stream);
}

int gencomm_nms_max_candidates(void) { return kNmsMaxN; }

int gencomm_bbox_overlaps_fwd(const float* boxes, const float* query_boxes, float* overlaps, int N, int K, void* stream) {
  GC_CHECK_ARG(N >= 0 && K >= 0, "bad N/K");
  if (N == 0 || K == 0) return GC_OK;
  GC_CHECK_ARG(boxes && query_boxes && overlaps, "null pointer");
  const long long total = (long long)N * K;
  GC_CHECK_ARG((total + 255) / 256 < (1LL << 31), "N*K too large");
  bbox_overlaps_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(boxes, query_boxes, overlaps, N, K);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

// ------------------------------------------------------------------------------------ fusion
int gencomm_warp_attfuse_fwd(const float* x, const double* theta, const int* scene_off, float* out,
                             int B, int n, int C, int H, int W, void* stream) {
  GC_CHECK_ARG(x && theta && scene_off && out, "null pointer");
  GC_CHECK_ARG(B >= 1 && B <= 65535 && n >= B && C >= 1 && H >= 1 && W >= 1, "bad B/n/C/H/W");
  return warp_attfuse_enqueue(x, theta, scene_off, out, B, n, C, H, W, (hipStream_t)stream);
}

long long gencomm_warp_attfuse_bwd_scratch_floats(int n, int H, int W) {
  if (n < 1 || H < 1 || W < 1) { fail(GC_ERR_ARG, "bad n/H/W"); return -1; }
  return (long long)warp_attfuse_bwd_scratch_floats(n, H, W);
}
int gencomm_warp_attfuse_bwd(const float* x, const double* theta, const int* scene_off, const float* grad_out, float* grad_x, float* scratch,
                             int B, int n, int C, int H, int W, void* stream) {
  GC_CHECK_ARG(x && theta && scene_off && grad_out && grad_x && scratch, "null pointer");
  GC_CHECK_ARG(B >= 1 && B <= 65535 && n >= B && n <= 65535 && C >= 1 && H >= 1 && W >= 1, "bad B/n/C/H/W");
  return warp_attfuse_bwd_enqueue(x, theta, scene_off, grad_out, grad_x, scratch, B, n, C, H, W, (hipStream_t)stream);
}

int gencomm_warp_maxfuse_fwd(const float* x, const double* theta, const int* scene_off, float* out,
                             int B, int n, int C, int H, int W, void* stream) {
  GC_CHECK_ARG(x && theta && scene_off && out, "null pointer");
  GC_CHECK_ARG(B >= 1 && B <= 65535 && n >= B && C >= 1 && H >= 1 && W >= 1, "bad B/n/C/H/W");
  return warp_attfuse_enqueue(x, theta, scene_off, out, B, n, C, H, W, (hipStream_t)stream, 1);
}

int gencomm_warp_attfuse_tok_fwd(const void* enhancer_workspace, const double* theta, const int* scene_off, float* out,
                                 int B, int n, int C, int H, int W, void* stream) {
  EnhancerPlan p;
  if (const char* e = p.build(C)) return fail(GC_ERR_ARG, e);
  GC_CHECK_ARG(enhancer_workspace && theta && scene_off && out, "null pointer");
  GC_CHECK_ARG(B >= 1 && B <= 65535 && n >= B && H >= 1 && W >= 1, "bad B/n/H/W");
  const EnhancerWs w = enhancer_ws(p, n, H, W);
  const char* base = (const char*)enhancer_workspace;
  return warp_attfuse_tok_enqueue(reinterpret_cast<const float*>(base + w.O), reinterpret_cast<const float*>(base + w.gate),
                                  theta, scene_off, out, B, C, H, W, modes_snapshot().xcd(), (hipStream_t)stream);
}

}  // extern "C"
