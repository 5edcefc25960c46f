// Shared device/host helpers for the gfx950 (CDNA4, wave64) kernels of the GenComm hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace gc {

constexpr int kWave = 64;  // CDNA wavefront width (hard-coded: warpSize folds to 64 on gfx950)
constexpr int kMaxResBlocks = 48;  // res-blocks one UNet may have (sizes the timestep-table kernel args)

// ---------------------------------------------------------------------------------------------
// error reporting: entry points return an int status and never exit()/abort()
// ---------------------------------------------------------------------------------------------
enum Status : int {
  GC_OK = 0,
  GC_ERR_ARG = 1,        // bad shape / unsupported configuration / null pointer
  GC_ERR_WORKSPACE = 2,  // caller-provided workspace too small
  GC_ERR_HIP = 3,        // a HIP runtime call failed
};

char* last_error_buf();  // thread-local, defined in gencomm_abi.hip
inline int fail(int code, const char* msg) {
  snprintf(last_error_buf(), 512, "%s", msg);
  return code;
}
#define GC_CHECK_ARG(cond, msg) \
  do {                          \
    if (!(cond)) return ::gc::fail(::gc::GC_ERR_ARG, msg); \
  } while (0)
#define GC_HIP(expr)                                                             \
  do {                                                                           \
    hipError_t e__ = (expr);                                                     \
    if (e__ != hipSuccess) {                                                     \
      snprintf(::gc::last_error_buf(), 512, "%s failed: %s", #expr, hipGetErrorString(e__)); \
      return ::gc::GC_ERR_HIP;                                                   \
    }                                                                            \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): set it once per device ordinal this process launches
// the kernel on, not once per process (a single static flag covers only the device of the first launch).
// One static instance per launch helper; a race sets the attribute twice, which is harmless.
struct LdsAttrOnce {
  unsigned long long done = 0;   // bit d: set for device ordinal d (ordinals >= 64 set it on every launch)
  template <class K>
  int set(K kernel, int bytes) {
    int d = 0;
    GC_HIP(hipGetDevice(&d));
    const unsigned long long bit = d < 64 ? 1ull << d : 0ull;
    if (done & bit) return 0;
    GC_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    __atomic_fetch_or(&done, bit, __ATOMIC_RELAXED);
    return 0;
  }
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// Library modes (gencomm_set_mode / gencomm_get_mode in the C ABI): explicit, atomic, read ONCE per entry-point call
// into a Modes snapshot that travels with the call -- never an environment variable, never re-read per launch.
// ---------------------------------------------------------------------------------------------
enum ModeKey : int {
  MODE_ARITH = 0,        // 0: products on the f16 matrix pipe from exact fp16 hi/lo splits (default); 1: exact-fp32 kernels;
                         // 2: bf16 denoise mode (bf16 storage of the UNet's 8-channel maps, single bf16 products: conv8b_kernels.h)
  MODE_SAMPLER = 1,      // 0: automatic (latent structure on large maps, literal on small); 1: literal conv_in .. conv_out + update per step; 2: latent
  MODE_TILE_WANT = 2,    // 0: automatic; > 0: minimum number of 64x16 workgroups before the 64x16-tile kernels are chosen
  MODE_ENH_FUSE = 3,     // Enhancer at C = 64 -- 2 (default): Linear1 + depthwise stage + Linear2 in one kernel; 1: Linear1 + depthwise; 0: separate launches
  MODE_CONV8H_MASK = 4,  // diagnostic: bit mask of conv8h variants allowed on the f16 pipe (-1: all)
  MODE_XCD_REMAP = 5,    // 1: workgroup -> tile mapping keeps neighbouring tiles on one XCD (default); 0: plain grid order
  MODE_DATAFLOW = 6,     // 1: the UNet body of a call runs as ONE persistent dataflow launch (dataflow_kernels.h); 0: one launch per layer
  MODE_RESFUSE_EMU = 7,  // TIMING EXPERIMENT ONLY, off by default -- the results are NOT the UNet's: unet_host.h "ResnetBlock fusion, emulated"
  MODE_TILE8 = 8,        // n > 0: GroupNorm'd 8-channel layers on the f16 pipe whose launch has fewer than n workgroups of 64 x 16 pixels run
                         // 64 x 8 tiles (conv8h8_kernels.h; VERDICT r3 item 3b: the half-resolution level); 0: 64 x 16 everywhere; -1 (default):
                         // automatic = 256 unless MODE_TILE_WANT forces a tile size (single-scene launches: latency 10.07 -> 9.78 ms)
  MODE_BWD_STREAMS = 9,  // 1 (default): gencomm_unet_bwd enqueues its weight-gradient launches on a library-owned side stream (forked from and
                         // joined back to the caller's stream inside the call), so that they overlap the input-gradient chain, when the call
                         // has at least 2^17 pixels (n H W); 2: always; 0: one stream
  MODE_PERSIST = 10,     // bit mask of 8-channel f16-pipe layer variants (1: conv1 8->8, 2: conv1 16->8, 4: conv2 + identity, 8: conv2 + shortcut,
                         // 16: Upsample) that run as the PERSISTENT kernel (conv8hp_kernel: at most 3 workgroups per CU walk the tiles, the next
                         // tile's loads requested one tile ahead) whenever the launch has more tiles than resident slots; 0: one tile per workgroup.
                         // Default 16: measured per variant (profiles/r5_persist_ab.txt) only the Upsample convolution gains (38.7 -> 35.0 us)
  MODE_COUNT = 11
};
struct Modes {
  long long v[MODE_COUNT];
  bool split() const { return v[MODE_ARITH] != 1; }  // f16-pipe kernel family of the hot path (three-term fp16 split, or bf16 mode)
  bool split2() const { return v[MODE_ARITH] == 3; } // opt-in: the general convolutions AROUND the path on the two-term split (22-bit products)
  bool bf16() const { return v[MODE_ARITH] == 2; }
  int xcd() const { return v[MODE_XCD_REMAP] != 0 ? 1 : 0; }
};
Modes modes_snapshot();  // defined in gencomm_abi.hip

// ---------------------------------------------------------------------------------------------
// Diagnostic kernel timer (gencomm_timer_start / gencomm_timer_stop): while armed for one kernel
// family, every launch of that family is bracketed by a pair of HIP events recorded on the launch
// stream. Off by default; the only process-global state in the library.
// ---------------------------------------------------------------------------------------------
enum KernelFamily : int {
  KF_CONV_IN = 0, KF_CONV8 = 1, KF_CONV16 = 2, KF_DOWN = 3, KF_UP = 4, KF_CONV_OUT = 5, KF_Q_SAMPLE = 6,
  KF_ENH_LN = 7, KF_ENH_PCONV = 8, KF_ENH_GEMM1 = 9, KF_ENH_DWGATE = 10, KF_ENH_GEMM2 = 11,
  KF_ENH_GATE = 12, KF_ENH_OUT = 13, KF_WARP_ATTFUSE = 14, KF_LATENT_STEP = 15, KF_CONV8_RES1 = 16, KF_CONV8_RES2 = 17,
  KF_DATAFLOW = 18,
  KF_CONV2D = 19,   // the general convolution around the path (conv2d_h3l / conv2d_h3 / conv2d_igemm kernels: BEV backbone, shrink conv, heads,
                    // Linear layers): its `bytes` slot carries algorithmic FLOPs (2 N Ho Wo Cout Cin KH KW)
  KF_COUNT = 20
};
inline const char* kernel_family_name(int id) {
  static const char* names[KF_COUNT] = {
      "conv_in_h_kernel | conv_in_kernel", "conv8h_kernel<1,GN,0> | conv8_kernel (ResnetBlock conv1, 8 -> 8)",
      "conv8h_kernel<2,GN,0> | conv8_kernel (ResnetBlock conv1, 16 -> 8)", "down8x2_kernel | down8_kernel",
      "conv8h_kernel<UP> | conv8_kernel<UP>", "conv_out_h_kernel | conv_out_kernel", "q_sample_kernel", "enh_ln_kernel",
      "enh_pconv_h_kernel | enh_pconv_kernel", "enh_front_h_kernel | gemm_*_mfma_kernel<0>", "enh_dwgate_kernel",
      "gemm_*_mfma_kernel<1>", "enh_gate_kernel", "enh_scale_transpose_kernel", "warp_attfuse_kernel | warp_attfuse_tok_kernel",
      "latent_step_h_kernel | latent_step_kernel", "conv8h_kernel<1,GN,RES=1> | conv8_kernel (ResnetBlock conv2 + identity)",
      "conv8h_kernel<1,GN,RES=2> | conv8_kernel (ResnetBlock conv2 + nin_shortcut)",
      "unet_dataflow_kernel (UNet body of one call, persistent)",
      "conv2d_h3l_kernel | conv2d_h3_kernel | conv2d_igemm_kernel (general 3x3 / 1x1 / 2x2 convolution; slot = FLOPs)"};
  return (id >= 0 && id < KF_COUNT) ? names[id] : "?";
}
struct KernelTimer {
  unsigned long long mask = 0;  // families armed (bit per KernelFamily)
  int cap = 0, count = 0;
  hipEvent_t* ev = nullptr;  // 2 * cap
  int* fam = nullptr;        // family of each recorded launch
  double* bytes = nullptr;   // ALGORITHMIC bytes of each recorded launch (tensor bytes the layer must read + write)
};
KernelTimer& kernel_timer();  // defined in gencomm_abi.hip
struct TimedLaunch {
  bool armed;
  hipStream_t st;
  TimedLaunch(int family, hipStream_t s, double algorithmic_bytes = 0.0) : st(s) {
    KernelTimer& t = kernel_timer();
    armed = ((t.mask >> family) & 1ull) && t.count < t.cap;
    if (armed) {
      t.fam[t.count] = family;
      t.bytes[t.count] = algorithmic_bytes;
      (void)hipEventRecord(t.ev[2 * t.count], st);
    }
  }
  ~TimedLaunch() {
    if (armed) {
      KernelTimer& t = kernel_timer();
      (void)hipEventRecord(t.ev[2 * t.count + 1], st);
      ++t.count;
    }
  }
};

// ---------------------------------------------------------------------------------------------
// Diagnostic kernel log (gencomm_klog_start / gencomm_klog_stop): while armed, every launch site of the hot path notes
// the exact kernel instantiation it chose, so a benchmark / test can state WHICH kernels ran (the timed Philox
// instantiations are different templates from the explicit-noise ones the golden tests inject noise into).
// ---------------------------------------------------------------------------------------------
bool klog_armed();                  // defined in gencomm_abi.hip
void klog_note(const char* name);
#define GC_KLOG(name)                                   \
  do {                                                  \
    if (::gc::klog_armed()) ::gc::klog_note(name);      \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Wave-uniform read-only tables (conv weights, biases, schedule rows) are read through the
// constant address space so that they always come in via the scalar cache (s_load_dwordxN into
// SGPRs) whatever the compiler can or cannot prove about aliasing; they are written only by
// EARLIER kernels, never by the kernel that reads them.
// ---------------------------------------------------------------------------------------------
typedef const float __attribute__((address_space(4)))* cfloat_p;
__device__ __forceinline__ cfloat_p as_const(const float* p) { return (cfloat_p)(p); }

// ---------------------------------------------------------------------------------------------
// XCD-aware workgroup -> tile mapping.  The dispatcher deals workgroups to the 8 XCDs round-robin by linear workgroup
// id, and every XCD has its own 4 MiB L2 (not coherent with the others): with the plain mapping the tiles left/right of
// a tile run on OTHER XCDs, so every halo row/column a tile shares with its neighbours is fetched across the fabric
// once per XCD that touches it.  Remapped, XCD k walks the k-th contiguous eighth of the (x fastest, then y, then
// sample) tile order, so horizontally and vertically adjacent tiles are resident on the same XCD at about the same
// time and their shared lines are L2 hits.  Bijective for any grid size (XCD k gets total/8 tiles, +1 for k < total%8).
// ---------------------------------------------------------------------------------------------
struct BlockId { int x, y, z; };
// the tile with linear index `lin` of a (gx, gy, total / (gx gy)) tile grid under the same rule (persistent kernels: the launch grid is 1-D)
__device__ __forceinline__ BlockId xcd_block_dims(int remap, unsigned lin, unsigned gx, unsigned gy, unsigned total) {
  unsigned nl = lin;
  if (remap) {
    const unsigned xcd = lin & 7u, j = lin >> 3;
    const unsigned q = total >> 3, r = total & 7u;
    nl = xcd * q + (xcd < r ? xcd : r) + j;
  }
  BlockId b;
  const unsigned row = nl / gx;
  b.x = (int)(nl - row * gx);
  b.z = (int)(row / gy);
  b.y = (int)(row - (unsigned)b.z * gy);
  return b;
}
// the tile of the workgroup with linear id `lin` (blockIdx.x fastest) under the same rule
__device__ __forceinline__ BlockId xcd_block_lin(int remap, unsigned lin) {
  const unsigned gx = gridDim.x, gy = gridDim.y;
  BlockId b;
  {
    const unsigned row = lin / gx;
    b.x = (int)(lin - row * gx);
    b.z = (int)(row / gy);
    b.y = (int)(row - (unsigned)b.z * gy);
  }
  if (remap) {
    const unsigned total = gx * gy * gridDim.z;
    const unsigned xcd = lin & 7u, j = lin >> 3;
    const unsigned q = total >> 3, r = total & 7u;
    const unsigned nl = xcd * q + (xcd < r ? xcd : r) + j;
    const unsigned row = nl / gx;
    b.x = (int)(nl - row * gx);
    b.z = (int)(row / gy);
    b.y = (int)(row - (unsigned)b.z * gy);
  }
  return b;
}
__device__ __forceinline__ BlockId xcd_block(int remap) {
  BlockId b{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  if (remap) {
    const unsigned gx = gridDim.x, gy = gridDim.y;
    const unsigned total = gx * gy * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = lin & 7u, j = lin >> 3;
    const unsigned q = total >> 3, r = total & 7u;
    const unsigned nl = xcd * q + (xcd < r ? xcd : r) + j;
    const unsigned row = nl / gx;
    b.x = (int)(nl - row * gx);
    b.z = (int)(row / gy);
    b.y = (int)(row - (unsigned)b.z * gy);
  }
  return b;
}

// ---------------------------------------------------------------------------------------------
// device math
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_f(float x) {
  // 1 / (1 + 2^(-x*log2e)): one v_exp_f32 + one v_rcp_f32 (~1 ulp each), no IEEE divide sequence
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float gelu_erf_f(float x) {
  // erf-GELU (nn.GELU() default) with erf(t) = 1 - 2^(-t Q(t)), t = |x| / sqrt(2) clamped to 4.3 (erf(4.3) = 1 - 1.2e-9),
  // Q a degree-6 polynomial fitted to -log2(erfc(t)) / t on [0, 4.3] (reweighted least squares on Chebyshev nodes,
  // float32 Horner evaluation checked on a 600k-point grid against float64 erf): |erf error| <= 1.7e-7,
  // |gelu error| <= 2.2e-7 over the whole line (2 % of the 1e-5 + 1e-4|y| parity budget).  Branch-free: 13 VALU + ONE
  // transcendental (v_exp_f32); the Abramowitz-Stegun 7.1.26 form it replaces needed v_rcp_f32 as well and was bounded
  // by 4.7e-7; the library erff takes two divergent branches.
  const float t = fminf(fabsf(x) * 0.70710678118654752440f, 4.3f);
  float q = fmaf(-1.0021018e-4f, t, 4.6151079e-4f);
  q = fmaf(q, t, 2.3023714e-3f);
  q = fmaf(q, t, -2.9452650e-2f);
  q = fmaf(q, t, 1.4896373e-1f);
  q = fmaf(q, t, 9.1832864e-1f);
  q = fmaf(q, t, 1.6279137e+0f);
  const float erf_abs = 1.0f - __builtin_amdgcn_exp2f(-t * q);
  const float h = 0.5f * x;
  return fmaf(h, copysignf(erf_abs, x), h);
}
// GELU(x) and d/dx GELU(x) = Phi(x) + x phi(x) from ONE evaluation of the erf form above (Phi = 0.5 + 0.5 erf(x / sqrt 2), error
// <= 0.85e-7) and one more v_exp_f32 for the density phi(x) = exp(-x^2 / 2) / sqrt(2 pi): the Enhancer's backward kernels (train_kernels.h)
// evaluated the library's erff + expf here (two divergent branches and a range reduction per element: 2.3 TB/s on a pass that moves
// 864 MB); branch-free, 2 transcendentals + ~20 VALU for both values.
__device__ __forceinline__ void gelu_and_grad_f(float x, float* gelu, float* grad) {
  const float t = fminf(fabsf(x) * 0.70710678118654752440f, 4.3f);
  float q = fmaf(-1.0021018e-4f, t, 4.6151079e-4f);
  q = fmaf(q, t, 2.3023714e-3f);
  q = fmaf(q, t, -2.9452650e-2f);
  q = fmaf(q, t, 1.4896373e-1f);
  q = fmaf(q, t, 9.1832864e-1f);
  q = fmaf(q, t, 1.6279137e+0f);
  const float erf_abs = 1.0f - __builtin_amdgcn_exp2f(-t * q);
  const float Phi = fmaf(0.5f, copysignf(erf_abs, x), 0.5f);
  const float phi = 0.3989422804014327f * __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);   // exp(-x^2 / 2)
  *gelu = x * Phi;
  *grad = fmaf(x, phi, Phi);
}
__device__ __forceinline__ float gelu_grad_fast_f(float x) {
  float g, d;
  gelu_and_grad_f(x, &g, &d);
  return d;
}

// ---------------------------------------------------------------------------------------------
// Range guard of the fp16 hi/lo operand splits (conv8h_kernels.h).  A two-term split x = hi + lo needs |x| < 65504:
//  * kernels whose operands are RAW tensors (conv_in's x_t / message channels, the Upsample conv's input, the unit-test
//    entry) scale them by an exact power of two 2^-k chosen from a device-side bound on max|x| (act_scale: k = 0, i.e. no
//    change at all, while the bound is below 2^14) and undo it in the epilogue -- the result is the fp32 one at any
//    input magnitude a float can hold;
//  * operands that are GroupNorm / LayerNorm outputs are bounded by |gamma| sqrt(group size) + |beta| whatever the
//    input; every split kernel additionally runs with MODE.FP16_OVFL = 1, under which an overflowing f32 -> f16
//    conversion saturates to +-65504 instead of producing inf: a pathological gamma degrades accuracy, it cannot
//    create inf / NaN out of finite inputs.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void fp16_ovfl_clamp() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }
__device__ __forceinline__ float act_scale(float bound) {
  if (!(bound >= 16384.f)) return 1.0f;
  int ex = 0;
  (void)frexpf(bound, &ex);  // bound = f * 2^ex, f in [0.5, 1)
  return ldexpf(1.0f, 14 - ex);
}
// non-negative floats order like their bit patterns: max|x| over many workgroups with one integer atomic per wave
__device__ __forceinline__ void wave_amax_commit(float v, float* __restrict__ dst) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  // the maximum saturates after the first few waves: look before adding (a stale smaller value read here only costs a
  // redundant atomic; 131 k unconditional atomics on one address took 1.3 ms in q_sample_kernel)
  if ((threadIdx.x & 63) == 0) {
    unsigned int* __restrict__ p = reinterpret_cast<unsigned int*>(dst);
    const unsigned int bits = __float_as_uint(v);
    if (bits > __builtin_nontemporal_load(p)) atomicMax(p, bits);
  }
}
// the same with ONE atomic per workgroup (256 threads): kernels whose workgroups all start together would otherwise all read
// the initial 0 and queue one atomic per wave on the same address (8 200 of them took 150 us in q_sample_kernel on small maps)
__device__ __forceinline__ void block_amax_commit(float v, float* __restrict__ dst) {
  __shared__ float s_amax[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0) s_amax[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float m = fmaxf(fmaxf(s_amax[0], s_amax[1]), fmaxf(s_amax[2], s_amax[3]));
    unsigned int* __restrict__ p = reinterpret_cast<unsigned int*>(dst);
    const unsigned int bits = __float_as_uint(m);
    if (bits > __builtin_nontemporal_load(p)) atomicMax(p, bits);
  }
}
static __global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long long count, float* __restrict__ dst) {
  float m = 0.f;
  const long long nvec = (count & 3) ? 0 : (count >> 2);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (long long i = (nvec << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
  block_amax_commit(m, dst);
}

// Butterfly sum, result in every lane (ds_bpermute based).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over the 64 lanes with DPP only (no LDS traffic): row_shr 1/2/4/8 builds a prefix sum inside
// each 16-lane row, row_bcast:15 / row_bcast:31 carry the row totals upward; lane 63 ends up with
// the wave total, which is returned as a wave-uniform value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float wave_total(float v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row holds the row sum
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Maximum of NON-NEGATIVE per-lane values over the wave, the same DPP ladder (lanes without a source contribute 0).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max0(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return fmaxf(v, __builtin_bit_cast(float, moved));
}
__device__ __forceinline__ float wave_max_nonneg(float v) {
  v = dpp_max0<0x111, 0xf>(v);
  v = dpp_max0<0x112, 0xf>(v);
  v = dpp_max0<0x114, 0xf>(v);
  v = dpp_max0<0x118, 0xf>(v);
  v = dpp_max0<0x142, 0xa>(v);
  v = dpp_max0<0x143, 0xc>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// 16 per-lane partials -> wave totals with a recursive-halving exchange instead of 16 full ladders:
// lanes trade halves of the vector across lane bits 0 and 1 (quad_perm DPP), the surviving 4 values
// are summed down each 16-lane row (row_shr 4/8) and across the 4 rows (two ds_bpermute steps).
// On return lanes with (lane & 15) >= 12 hold, in tot[i], the WAVE total of
// part[8 * (lane & 1) + 4 * ((lane >> 1) & 1) + i]; other lanes hold partial sums.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ void wave_reduce16(const float (&part)[16], float (&tot)[4]) {
  const int lane = threadIdx.x & 63;
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
  float h[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float keep = b0 ? part[8 + i] : part[i], send = b0 ? part[i] : part[8 + i];
    h[i] = keep + dpp_mov<0xB1>(send);  // quad_perm [1,0,3,2]: partner lane ^ 1
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float keep = b1 ? h[4 + i] : h[i], send = b1 ? h[i] : h[4 + i];
    float g = keep + dpp_mov<0x4E>(send);  // quad_perm [2,3,0,1]: partner lane ^ 2
    g = dpp_add<0x114, 0xf>(g);            // row_shr:4
    g = dpp_add<0x118, 0xf>(g);            // row_shr:8 -> lanes 12..15 of a row: row total of their class
    g += __shfl_xor(g, 16, 64);
    g += __shfl_xor(g, 32, 64);
    tot[i] = g;
  }
}

// ---------------------------------------------------------------------------------------------
// Noise: Philox4x32-7 counter RNG (Salmon et al., SC'11: 7 rounds is the fewest that passes BigCrush; the customary
// 10 are a safety margin a sampler's noise field does not need) + Box-Muller.  One call yields four 32-bit words =
// four Box-Muller pairs = EIGHT normals: word j -> angle from its high 16 bits (in revolutions, the unit of
// v_sin_f32 / v_cos_f32), radius uniform from its low 16 bits, u = (k + 1/2) / 65536 (midpoint rule: masses exact to
// O(2^-32)) -- REFINED where the 16-bit grid is too coarse for the tail: a word with k < 16 (u < 2.4e-4, radius > 4.08)
// takes 32 further bits from a second Philox block of the same counter (c3 = 1), u = (k 2^32 + f + 1/2) 2^-48.  The
// radius is therefore continuous to 2^-49 (|z| up to 8.24; round 2 stopped at 4.85 and gave the whole k = 0 bin one
// radius).  The second block is needed by 1 call in 1 000 per lane, 6 % per wave: one v_min3/v_min/v_cmp/branch in
// the common path.
// Round 1 used Philox4x32-10 with 32-bit uniforms, one call per FOUR normals: 20 quarter-rate v_mad_u64_u32 + 8
// transcendentals per 4 normals made the sampler step VALU-bound (56 k of 70 k issue cycles per tile-wave); this
// form spends 14 multiplies + 14 three-way xors (v_bitop3_b32) + 16 transcendentals per EIGHT normals.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);  // v_bitop3_b32, truth table of a ^ b ^ c: one instruction instead of two v_xor
}
// Diagnostic builds only (-DGC_NOISE_DIAG=bits, tools/diag/noise_ab.sh: the VALU budget of the in-kernel noise by subtraction; the field
// such a build produces is NOT the canonical one): 1 = Philox rounds replaced by one xor, 2 = Box-Muller without transcendentals,
// 4 = no tail refinement, 8 = no generator at all (words = counter)
#ifdef GC_NOISE_DIAG
constexpr int kNoiseDiag = GC_NOISE_DIAG;
#else
constexpr int kNoiseDiag = 0;
#endif
__device__ __forceinline__ void philox4x32_7(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                             uint32_t k0, uint32_t k1, uint32_t out[4]) {
  if (kNoiseDiag & 1) {
    out[0] = c0 ^ k0; out[1] = c1 + 0x9E3779B9u; out[2] = c2 ^ k1; out[3] = (c3 + c0) ^ 0xBB67AE85u;
    return;
  }
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    // one v_mad_u64_u32 per 32x32->64 product (integer multiplies are quarter-rate on CDNA:
    // a separate mul_hi + mul_lo pair would cost twice as much)
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = xor3(hi1, c1, k0), n2 = xor3(hi0, c3, k1);
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// The four words of one counter with their radius uniforms (see above): w[j] >> 16 is the angle of pair j.
constexpr uint32_t kNoiseRefineBelow = 16u;
struct NoiseWords {
  uint32_t w[4];
  float ur[4];
};
__device__ __forceinline__ void noise_words(uint64_t ctr, uint32_t stream, uint64_t seed, NoiseWords& q) {
  philox4x32_7((uint32_t)ctr, (uint32_t)(ctr >> 32), stream, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), q.w);
  const float k16 = 1.52587890625e-5f;  // 2^-16
  uint32_t k[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    k[j] = q.w[j] & 0xffffu;
    q.ur[j] = fmaf((float)k[j], k16, 0.5f * k16);
  }
  if (!(kNoiseDiag & 4) && min(min(k[0], k[1]), min(k[2], k[3])) < kNoiseRefineBelow) {
    uint32_t f[4];
    philox4x32_7((uint32_t)ctr, (uint32_t)(ctr >> 32), stream, 1u, (uint32_t)seed, (uint32_t)(seed >> 32), f);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k[j] < kNoiseRefineBelow) q.ur[j] = fmaf((float)k[j], k16, fmaf((float)f[j], 3.5527136788005009e-15f /* 2^-48 */, 1.7763568394002505e-15f /* 2^-49 */));
  }
}

// Box-Muller pair with standard deviation sg, on the raw hardware transcendentals: radius sqrt(k2 * log2 u), k2 =
// -2 ln 2 * sg^2 (bm_k2; u in [2^-49, 1): never 0, never denormal), times (cos, sin) of the angle.  The scale rides inside
// the square root (one multiply per pair less than sg * sqrt(..)); k2 = 0 gives (+-0, +-0), which is how callers blank
// the lanes whose pixels lie outside the image without a branch.
__device__ __forceinline__ float bm_k2(float sg) { return -1.3862943611198906f * sg * sg; }
__device__ __forceinline__ void bm_pair(float ur, uint32_t w, float k2, float& zc, float& zs) {
  const float ut = (float)(w >> 16) * 1.52587890625e-5f;
  if (kNoiseDiag & 2) {   // same non-transcendental instructions, no v_log / v_sqrt / v_cos / v_sin
    const float m = k2 * ur;
    zc = m * ut;
    zs = m * (ut - 0.5f);
    asm("" : "+v"(zc), "+v"(zs));
    return;
  }
  const float m = __builtin_amdgcn_sqrtf(k2 * __builtin_amdgcn_logf(ur));
  zc = m * __builtin_amdgcn_cosf(ut);
  zs = m * __builtin_amdgcn_sinf(ut);
  // The products are opaque to the optimiser from here on: otherwise it may fuse `m * cos` with the CALLER's fp16 conversion
  // into one v_fma_mixlo_f16 (a single rounding of the exact product) in some kernels and not in others -- it did so in
  // conv_out_kernel<64,16,4,2> only, `#pragma clang fp contract(off)` does not stop it -- and the "canonical" field then
  // differs by an fp16 ulp between kernels wherever the fp32 product lands on an fp16 tie (2^-13 of the values; found by
  // tests/test_gpu_philox_replay.py).  Rounded to fp32 first, everywhere; costs no instruction.
  asm("" : "+v"(zc), "+v"(zs));
}
// the pair as one packed fp16x2 dword (cosine branch in the low half)
typedef _Float16 nz_half2_t __attribute__((ext_vector_type(2)));
typedef float nz_float2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bm_pair_h(float ur, uint32_t w, float k2) {
  float zc, zs;
  bm_pair(ur, w, k2, zc, zs);
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((nz_float2_t){zc, zs}, nz_half2_t));
}

// 8 N(0,1) floats of one counter: z[2j], z[2j+1] = the pair of word j (q_sample's initial noise: element index / 8)
__device__ __forceinline__ void normal8(uint64_t ctr, uint32_t stream, uint64_t seed, float z[8]) {
  NoiseWords q;
  noise_words(ctr, stream, seed, q);
#pragma unroll
  for (int j = 0; j < 4; ++j) bm_pair(q.ur[j], q.w[j], -1.3862943611198906f, z[2 * j], z[2 * j + 1]);
}

// The sampler's step noise, CANONICAL FIELD (identical in the latent and the literal sampler structure, independent of
// tiling): nu_t(n, c, y, x) = fp16( sigma_t * z ), z ~ N(0,1).  One call serves the channel pair (c & ~1, c | 1) at the
// aligned pixel quad x & ~3: counter = element index of (n, c & ~1, y, x & ~3) in the [n][C][H][W] tensor, stream = t;
// word j belongs to pixel (x & ~3) + j, its cosine branch to the even channel, its sine branch to the odd one.
// Rounding sigma*z to fp16 (relative 2^-11, unbiased) is what lets the noise convolution run on the f16 matrix pipe
// with ONE operand term; both sampler structures add exactly this value.  gencomm_step_noise_fwd (noise_kernels.h)
// writes the field out with these same functions: the parity tests replay it through the oracle.
// bm_pair_h(ur[j], w[j], k2) = packed (even channel, odd channel) fp16 pair of pixel j -- one dword of the LDS record
// of that pixel.  k2 = bm_k2(sigma_t) everywhere (0 for lanes outside the image).
__device__ __forceinline__ void noise_pair_quad_h(uint64_t elem, uint32_t stream, uint64_t seed, float k2, uint32_t (&h)[4]) {
  NoiseWords q;
  noise_words(elem, stream, seed, q);
#pragma unroll
  for (int j = 0; j < 4; ++j) h[j] = bm_pair_h(q.ur[j], q.w[j], k2);
}
// the same values as floats: z[j] = even channel, pixel j; z[4 + j] = odd channel, pixel j
__device__ __forceinline__ void noise_pair_quad(uint64_t elem, uint32_t stream, uint64_t seed, float k2, float (&z)[8]) {
  uint32_t h[4];
  noise_pair_quad_h(elem, stream, seed, k2, h);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const nz_half2_t v = __builtin_bit_cast(nz_half2_t, h[j]);
    z[j] = (float)v[0];
    z[4 + j] = (float)v[1];
  }
}

}  // namespace gc
