// Sparse 3-D convolutions of the SECOND encoder without spconv (reference call sites:
// opencood/models/heter_encoders.py:52-81, sub_modules/sparse_backbone_3d.py:33-152, mean_vfe.py:14-33,
// height_compression.py:10-30).  spconv itself is not under /root/reference: the semantics below are its published ones
// (PARITY UNPINNED, oracle/second_port.py restates them on dense volumes):
//   SubMConv3d     output sites = input sites; out[p] = sum over the kernel offsets whose input site is ACTIVE
//   SparseConv3d   output sites = every position whose receptive field holds at least one active input site;
//                  spatial shape floor((D + 2 pad - k) / stride) + 1; same sum
//   both without bias, followed by BatchNorm1d(eps 1e-3) on the active rows and ReLU (sparse_backbone_3d.py:12-31)
//   .dense()       zeros at inactive sites
//
// MI355X design: a sparse tensor is (keys int64 [n] ASCENDING, features fp32 [n][C] row-major).  key = ((b D + z) H + y) W + x.
// Keeping every level sorted makes neighbour lookup a binary search (17 steps for 100 k sites; no hash table, no
// insertion races, deterministic) and makes the output-site set of a strided layer a radix sort + unique of the
// candidate keys (rocPRIM).  A layer = rulebook nbr[offset][site] (int32, -1 = inactive; shared by the SubM layers of one
// level exactly like spconv's indice_key) + ONE gather-GEMM kernel on v_mfma_f32_32x32x2_f32 (exact fp32):
// rows = 64 output sites (A operand, gathered feature rows staged in LDS), columns = 64 output channels (B operand,
// weight slab of the current kernel offset), K = input channels, looping over the kernel offsets and skipping offsets
// for which no site of the tile has a neighbour.  Epilogue: folded BatchNorm, ReLU, 128-byte row stores.
#pragma once
#include <rocprim/rocprim.hpp>

#include "common.h"

namespace gc {

constexpr long long kSpNoKey = 0x7FFFFFFFFFFFFFFFLL;
using f32x16s = __attribute__((ext_vector_type(16))) float;

struct SpGrid {
  int B, D, H, W;
};
__device__ __forceinline__ long long sp_encode(const SpGrid g, int b, int z, int y, int x) {
  return (((long long)b * g.D + z) * g.H + y) * g.W + x;
}
__device__ __forceinline__ void sp_decode(const SpGrid g, long long k, int& b, int& z, int& y, int& x) {
  x = (int)(k % g.W); k /= g.W;
  y = (int)(k % g.H); k /= g.H;
  z = (int)(k % g.D);
  b = (int)(k / g.D);
}
// index of `key` in the ascending array keys[0..n), or -1
__device__ __forceinline__ int sp_find(const long long* __restrict__ keys, int n, long long key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < key) lo = mid + 1; else hi = mid;
  }
  return (lo < n && keys[lo] == key) ? lo : -1;
}

// coords [n][4] = (b, z, y, x) as the dataloader collates them (sparse_backbone_3d.py:108-114) -> key, value = row
__global__ __launch_bounds__(256) void sp_key_kernel(const int* __restrict__ coords, int n, const SpGrid g, long long* __restrict__ key, int* __restrict__ val) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int b = coords[4 * i], z = coords[4 * i + 1], y = coords[4 * i + 2], x = coords[4 * i + 3];
  const bool ok = b >= 0 && b < g.B && z >= 0 && z < g.D && y >= 0 && y < g.H && x >= 0 && x < g.W;
  key[i] = ok ? sp_encode(g, b, z, y, x) : kSpNoKey;
  val[i] = i;
}

struct SpConvGeom {
  SpGrid in, out;
  int k[3], stride[3], pad[3];  // (z, y, x)
};

// rulebook: nbr[o][j] = index of the input site  out_coord * stride - pad + offset_o  in in_keys, or -1
__global__ __launch_bounds__(256) void sp_rules_kernel(const long long* __restrict__ out_keys, int n_out, const long long* __restrict__ in_keys, int n_in,
                                                       const SpConvGeom g, int* __restrict__ nbr) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int o = blockIdx.y;
  if (j >= n_out) return;
  const long long key = out_keys[j];
  int res = -1;
  if (key != kSpNoKey) {
    int b, z, y, x;
    sp_decode(g.out, key, b, z, y, x);
    const int kx = o % g.k[2], ky = (o / g.k[2]) % g.k[1], kz = o / (g.k[2] * g.k[1]);
    const int iz = z * g.stride[0] - g.pad[0] + kz, iy = y * g.stride[1] - g.pad[1] + ky, ix = x * g.stride[2] - g.pad[2] + kx;
    if (iz >= 0 && iz < g.in.D && iy >= 0 && iy < g.in.H && ix >= 0 && ix < g.in.W) res = sp_find(in_keys, n_in, sp_encode(g.in, b, iz, iy, ix));
  }
  nbr[(size_t)o * n_out + j] = res;
}

// candidate output sites of a strided SparseConv3d.  Along an axis only the kernel offsets congruent to (coord + pad)
// modulo the stride reach an output position: ceil(k / stride) slots per axis instead of k (8 instead of 27 for k 3,
// stride 2).  Invalid slots get `nokey` = number of cells of the output grid (sorts last, needs no extra key bits).
struct SpSlots {
  int n[3];   // slots per axis
};
__host__ __device__ inline int sp_slot_count(const SpConvGeom& g, int j) { return (g.k[j] + g.stride[j] - 1) / g.stride[j]; }
__global__ __launch_bounds__(256) void sp_candidates_kernel(const long long* __restrict__ in_keys, int n_in, const SpConvGeom g, const SpSlots sl,
                                                            long long nokey, long long* __restrict__ cand) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int slot = blockIdx.y;
  if (i >= n_in) return;
  const long long key = in_keys[i];
  long long res = nokey;
  if (key != kSpNoKey) {
    int b, z, y, x;
    sp_decode(g.in, key, b, z, y, x);
    const int sx = slot % sl.n[2], sy = (slot / sl.n[2]) % sl.n[1], sz = slot / (sl.n[2] * sl.n[1]);
    const int kz = (z + g.pad[0]) % g.stride[0] + sz * g.stride[0], ky = (y + g.pad[1]) % g.stride[1] + sy * g.stride[1],
              kx = (x + g.pad[2]) % g.stride[2] + sx * g.stride[2];
    const int tz = z + g.pad[0] - kz, ty = y + g.pad[1] - ky, tx = x + g.pad[2] - kx;   // multiples of the stride by construction
    if (kz < g.k[0] && ky < g.k[1] && kx < g.k[2] && tz >= 0 && ty >= 0 && tx >= 0) {
      const int oz = tz / g.stride[0], oy = ty / g.stride[1], ox = tx / g.stride[2];
      if (oz < g.out.D && oy < g.out.H && ox < g.out.W) res = sp_encode(g.out, b, oz, oy, ox);
    }
  }
  cand[(size_t)slot * n_in + i] = res;
}
// the unique pass keeps the "no key" value as one last entry: drop it from the count
__global__ void sp_fix_count_kernel(const long long* __restrict__ keys, long long nokey, int* __restrict__ count) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int c = *count;
    if (c > 0 && keys[c - 1] >= nokey) *count = c - 1;
  }
}

struct SpSitesWs {
  size_t cand, sorted, temp, temp_bytes, total;
};
inline SpSitesWs sp_sites_ws(long long n_cand) {
  SpSitesWs w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
  const size_t nn = (size_t)(n_cand > 0 ? n_cand : 1);
  w.cand = take(nn * 8); w.sorted = take(nn * 8);
  size_t t1 = 0, t2 = 0;
  (void)rocprim::radix_sort_keys(nullptr, t1, (long long*)nullptr, (long long*)nullptr, nn, 0, 64, (hipStream_t)0);
  (void)rocprim::unique(nullptr, t2, (long long*)nullptr, (long long*)nullptr, (int*)nullptr, nn, rocprim::equal_to<long long>(), (hipStream_t)0);
  w.temp_bytes = std::max(t1, t2) + 256;
  w.temp = take(w.temp_bytes);
  w.total = off;
  return w;
}
struct SpSortWs {
  size_t key, val, temp, temp_bytes, total;
};
inline SpSortWs sp_sort_ws(int n) {
  SpSortWs w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += align_up(bytes, 256); return o; };
  const size_t nn = (size_t)(n > 0 ? n : 1);
  w.key = take(nn * 8); w.val = take(nn * 4);
  size_t t1 = 0;
  (void)rocprim::radix_sort_pairs(nullptr, t1, (long long*)nullptr, (long long*)nullptr, (int*)nullptr, (int*)nullptr, nn, 0, 64, (hipStream_t)0);
  w.temp_bytes = t1 + 256;
  w.temp = take(w.temp_bytes);
  w.total = off;
  return w;
}

// MeanVFE (mean_vfe.py:24-32): sum over ALL point slots / max(num_points, 1); row j of the output = voxel perm[j]
__global__ __launch_bounds__(256) void mean_vfe_kernel(const float* __restrict__ voxels, const int* __restrict__ num_points, const int* __restrict__ perm,
                                                       float* __restrict__ out, int n, int P, int F) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * F) return;
  const int j = i / F, f = i - j * F;
  const int v = perm != nullptr ? perm[j] : j;
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += voxels[((size_t)v * P + p) * F + f];
  out[i] = s / fmaxf((float)num_points[v], 1.0f);
}

// spconv weights -> kernel layout [K][CinP / 2][CoutP][2] (channel pairs interleaved for the two k-lanes of the MFMA);
// layout 0: spconv 2.x [Cout][K][Cin], layout 1: spconv 1.x [K][Cin][Cout]
__global__ __launch_bounds__(256) void sp_prep_w_kernel(const float* __restrict__ w, float* __restrict__ out, int K, int Cin, int Cout, int CinP, int CoutP, int layout) {
  const long long total = (long long)K * CinP * CoutP;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 1);
  const int co = (int)((i >> 1) % CoutP);
  const long long r = (i >> 1) / CoutP;
  const int cp = (int)(r % (CinP / 2)), o = (int)(r / (CinP / 2));
  const int ci = 2 * cp + lane;
  float v = 0.f;
  // layout 2: the input-gradient convolution of a layout-0 weight [Cout_f = Cin][K][Cin_f = Cout]: in / out channels swapped,
  // offsets mirrored when `mirror` (SubM layers: dx[i] = sum_o W_o^T dy[site at coord_i - delta_o])
  if (ci < Cin && co < Cout) {
    if (layout == 0) v = w[((size_t)co * K + o) * Cin + ci];
    else if (layout == 1) v = w[((size_t)o * Cin + ci) * Cout + co];
    else v = w[((size_t)ci * K + (layout == 3 ? K - 1 - o : o)) * Cout + co];
  }
  out[i] = v;
}

struct SpConvArgs {
  const float* x;      // [n_in][Cin]
  const int* nbr;      // [K][n_out]
  const float* w;      // prepared [K][CIN / 2][CoutP][2]
  const float* scale;  // [Cout]
  const float* shift;  // [Cout]
  float* y;            // [n_out][Cout]
  int n_out, K, Cin, Cout, CoutP, relu;
};

template <int CIN>
__global__ __launch_bounds__(256) void sp_conv_kernel(const SpConvArgs a) {
  static_assert(CIN % 4 == 0 && CIN <= 64, "padded input channels");
  constexpr int EPT = CIN / 4;                 // floats per thread and staging pass
  __shared__ float As[(CIN / 2) * 64 * 2];     // [cpair][site][2]
  __shared__ float Bs[(CIN / 2) * 64 * 2];     // [cpair][co][2]
  __shared__ unsigned int s_mask;
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
  const int site0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
  const int sh = wv >> 1, ch = wv & 1;
  const bool co_live = co0 + ch * 32 < a.CoutP;   // wave-uniform
  // staging: lane = site (consecutive lanes -> consecutive LDS words), wave = channel quarter
  const int gsite = site0 + l;
  const int c0 = wv * EPT;

  // kernel offsets at which at least one site of this tile has a neighbour (K <= 32)
  if (tid == 0) s_mask = 0u;
  __syncthreads();
  {
    unsigned int m = 0u;
    if (gsite < a.n_out)
      for (int o = wv; o < a.K; o += 4) m |= (a.nbr[(size_t)o * a.n_out + gsite] >= 0) ? (1u << o) : 0u;
    for (int d = 32; d >= 1; d >>= 1) m |= __shfl_xor(m, d);
    if (l == 0 && m) atomicOr(&s_mask, m);
  }
  __syncthreads();
  unsigned int mask = s_mask;

  f32x16s acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  float va[EPT], vb[EPT];
  auto fetch = [&](int o) {
    const int idx = gsite < a.n_out ? a.nbr[(size_t)o * a.n_out + gsite] : -1;
#pragma unroll
    for (int e = 0; e < EPT; ++e) va[e] = 0.f;
    if (idx >= 0) {
      const float* __restrict__ src = a.x + (size_t)idx * a.Cin + c0;
      if constexpr (EPT % 4 == 0) {
        if (a.Cin == CIN) {
#pragma unroll
          for (int e = 0; e < EPT; e += 4) {
            const float4 v = *reinterpret_cast<const float4*>(src + e);
            va[e] = v.x; va[e + 1] = v.y; va[e + 2] = v.z; va[e + 3] = v.w;
          }
        } else {
#pragma unroll
          for (int e = 0; e < EPT; ++e) if (c0 + e < a.Cin) va[e] = src[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) if (c0 + e < a.Cin) va[e] = src[e];
      }
    }
    // weight slab of this offset: (CIN / 2) rows of 128 floats (64 co x 2 k-lanes), consecutive threads = consecutive words
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = e * 256 + tid;
      const int cp = i >> 7, rem = i & 127;
      vb[e] = (co0 + (rem >> 1) < a.CoutP) ? a.w[(((size_t)o * (CIN / 2) + cp) * a.CoutP + co0) * 2 + rem] : 0.f;
    }
  };
  auto commit = [&]() {
    if constexpr (EPT >= 2) {
#pragma unroll
      for (int e = 0; e < EPT; e += 2)
        *reinterpret_cast<float2*>(&As[(((c0 + e) >> 1) * 64 + l) * 2]) = make_float2(va[e], va[e + 1]);
    } else {
      As[((c0 >> 1) * 64 + l) * 2 + (c0 & 1)] = va[0];
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) Bs[e * 256 + tid] = vb[e];
  };

  if (mask != 0u) {
    fetch(__builtin_ctz(mask));
    while (true) {
      __syncthreads();   // every wave is done with the previous offset's tiles
      commit();
      __syncthreads();
      mask &= mask - 1u;
      if (mask != 0u) fetch(__builtin_ctz(mask));   // next offset's rows and weights travel while the matrix cores run
      if (co_live) {
#pragma unroll
        for (int cp = 0; cp < CIN / 2; ++cp) {
          const float av = As[(cp * 64 + sh * 32 + r) * 2 + h];
          const float bv = Bs[(cp * 64 + ch * 32 + r) * 2 + h];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
      }
      if (mask == 0u) break;
    }
  }
  if (!co_live) return;
  const int co = co0 + ch * 32 + r;
  if (co >= a.Cout) return;
  const float sc = a.scale[co], sf = a.shift[co];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int site = site0 + sh * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (site >= a.n_out) continue;
    float v = fmaf(acc[reg], sc, sf);
    if (a.relu) v = fmaxf(v, 0.f);
    a.y[(size_t)site * a.Cout + co] = v;
  }
}

// The same gather-GEMM on the f16 matrix pipe (Cin = 32 or 64: the layers that take two thirds of the encoder's time) with the exact
// hi/lo split of conv8h_kernels.h: per kernel offset CIN / 16 steps of v_mfma_f32_32x32x16_f16 x 3 split terms (12 MFMAs of 32 cycles
// at Cin = 64 against 32 of 64 cycles).  Gathered rows are channel-contiguous already, so a thread splits the 8 / 16 channels it
// fetched and writes 16-byte record pieces; the prepared weights [offset][k pair][co][2] are read as 4 float2 per (co, k octet).
// LDS records {CIN hi halves | CIN lo halves | 16 B pad} (an odd number of 16-byte units: conflict-free ds_read_b128 phases).
// Activations carry the running power-of-two scale of conv1x1_f16s_kernel (workgroup max per offset; accumulators rescaled when the
// exponent grows), weights are pre-multiplied by 2^6.
typedef _Float16 sph8_t __attribute__((ext_vector_type(8)));
typedef _Float16 sph2_t __attribute__((ext_vector_type(2)));
typedef float spf2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sp_split8(const float (&v)[8], float mul, uint4& hi, uint4& lo) {  // exact two-term fp16 split of 8 scaled floats
  uint32_t hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a0 = v[2 * i] * mul, a1 = v[2 * i + 1] * mul;
    const sph2_t x = __builtin_convertvector((spf2_t){a0, a1}, sph2_t);
    const sph2_t y = __builtin_convertvector((spf2_t){a0 - (float)x[0], a1 - (float)x[1]}, sph2_t);
    hh[i] = __builtin_bit_cast(uint32_t, x);
    ll[i] = __builtin_bit_cast(uint32_t, y);
  }
  hi = make_uint4(hh[0], hh[1], hh[2], hh[3]);
  lo = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}
template <int CIN>
__global__ __launch_bounds__(256) void sp_conv_f16s_kernel(const SpConvArgs a) {
  static_assert(CIN == 32 || CIN == 64, "split kernel: 32 or 64 input channels");
  constexpr int EPT = CIN / 4, REC = CIN * 4 + 16, LO = CIN * 2, WIT = CIN / 32;  // WIT weight items (co, k octet) per thread
  constexpr float WS = 64.0f;
  __shared__ __align__(16) unsigned char As[64 * REC], Bs[64 * REC];
  __shared__ unsigned int s_mask;
  __shared__ float s_max[2][4];
  fp16_ovfl_clamp();
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, r = l & 31, h = l >> 5;
  const int site0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
  const int sh = wv >> 1, ch = wv & 1;
  const bool co_live = co0 + ch * 32 < a.CoutP;   // wave-uniform
  const int gsite = site0 + l;
  const int c0 = wv * EPT;

  if (tid == 0) s_mask = 0u;
  __syncthreads();
  {
    unsigned int m = 0u;
    if (gsite < a.n_out)
      for (int o = wv; o < a.K; o += 4) m |= (a.nbr[(size_t)o * a.n_out + gsite] >= 0) ? (1u << o) : 0u;
    for (int d = 32; d >= 1; d >>= 1) m |= __shfl_xor(m, d);
    if (l == 0 && m) atomicOr(&s_mask, m);
  }
  __syncthreads();
  unsigned int mask = s_mask;

  f32x16s acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  float va[EPT], vb[WIT][8];
  auto fetch = [&](int o) {
    const int idx = gsite < a.n_out ? a.nbr[(size_t)o * a.n_out + gsite] : -1;
#pragma unroll
    for (int e = 0; e < EPT; ++e) va[e] = 0.f;
    if (idx >= 0) {
      const float* __restrict__ src = a.x + (size_t)idx * CIN + c0;
#pragma unroll
      for (int e = 0; e < EPT; e += 4) {
        const float4 v = *reinterpret_cast<const float4*>(src + e);
        va[e] = v.x; va[e + 1] = v.y; va[e + 2] = v.z; va[e + 3] = v.w;
      }
    }
#pragma unroll
    for (int j = 0; j < WIT; ++j) {
      const int it = tid + 256 * j, co = it & 63, g = it >> 6;   // k octet g: k pairs 4 g .. 4 g + 3
      const bool ok = co0 + co < a.CoutP;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float2 v = ok ? *reinterpret_cast<const float2*>(a.w + (((size_t)o * (CIN / 2) + 4 * g + q) * a.CoutP + co0 + co) * 2) : make_float2(0.f, 0.f);
        vb[j][2 * q] = v.x; vb[j][2 * q + 1] = v.y;
      }
    }
  };
  float xs = 1.0f, run_max = 0.f;
  int par = 0;
  if (mask != 0u) {
    fetch(__builtin_ctz(mask));
    while (true) {
      float mx = 0.f;
#pragma unroll
      for (int e = 0; e < EPT; ++e) mx = fmaxf(mx, fabsf(va[e]));
      mx = wave_max_nonneg(mx);
      if (l == 0) s_max[par][wv] = mx;
      __syncthreads();   // every wave is done with the previous offset's tiles
      run_max = fmaxf(run_max, fmaxf(fmaxf(s_max[par][0], s_max[par][1]), fmaxf(s_max[par][2], s_max[par][3])));
      par ^= 1;
      if (run_max > 0.f) {
        int e2;
        (void)frexpf(run_max, &e2);
        const float ns = ldexpf(1.0f, min(14 - e2, 100));
        if (ns != xs) {
          const float ratio = ns / xs;
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] *= ratio;
          xs = ns;
        }
      }
#pragma unroll
      for (int e = 0; e < EPT; e += 8) {
        const float v8[8] = {va[e], va[e + 1], va[e + 2], va[e + 3], va[e + 4], va[e + 5], va[e + 6], va[e + 7]};
        uint4 hi, lo;
        sp_split8(v8, xs, hi, lo);
        *reinterpret_cast<uint4*>(As + l * REC + (c0 + e) * 2) = hi;
        *reinterpret_cast<uint4*>(As + l * REC + LO + (c0 + e) * 2) = lo;
      }
#pragma unroll
      for (int j = 0; j < WIT; ++j) {
        const int it = tid + 256 * j, co = it & 63, g = it >> 6;
        uint4 hi, lo;
        sp_split8(vb[j], WS, hi, lo);
        *reinterpret_cast<uint4*>(Bs + co * REC + 16 * g) = hi;
        *reinterpret_cast<uint4*>(Bs + co * REC + LO + 16 * g) = lo;
      }
      __syncthreads();
      mask &= mask - 1u;
      if (mask != 0u) fetch(__builtin_ctz(mask));   // next offset's rows and weights travel while the matrix cores run
      if (co_live) {
#pragma unroll
        for (int ks = 0; ks < CIN / 16; ++ks) {
          const unsigned char* ap = As + (sh * 32 + r) * REC + 32 * ks + 16 * h;
          const unsigned char* bp = Bs + (ch * 32 + r) * REC + 32 * ks + 16 * h;
          const sph8_t ah = *reinterpret_cast<const sph8_t*>(ap), al = *reinterpret_cast<const sph8_t*>(ap + LO);
          const sph8_t bh = *reinterpret_cast<const sph8_t*>(bp), bl = *reinterpret_cast<const sph8_t*>(bp + LO);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
        }
      }
      if (mask == 0u) break;
    }
  }
  if (!co_live) return;
  const int co = co0 + ch * 32 + r;
  if (co >= a.Cout) return;
  const float sc = a.scale[co] / (WS * xs), sf = a.shift[co];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int site = site0 + sh * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
    if (site >= a.n_out) continue;
    float v = fmaf(acc[reg], sc, sf);
    if (a.relu) v = fmaxf(v, 0.f);
    a.y[(size_t)site * a.Cout + co] = v;
  }
}

inline int sp_conv_enqueue(const SpConvArgs& a, hipStream_t st) {
  if (a.K > 32) return fail(GC_ERR_ARG, "sparse conv: at most 32 kernel offsets (3 x 3 x 3)");
  const dim3 grid((a.n_out + 63) / 64, (a.CoutP + 63) / 64);
  if (modes_snapshot().split2() && (a.Cin == 32 || a.Cin == 64)) {
    if (a.Cin == 32) sp_conv_f16s_kernel<32><<<grid, 256, 0, st>>>(a);
    else sp_conv_f16s_kernel<64><<<grid, 256, 0, st>>>(a);
    GC_HIP(hipGetLastError());
    return GC_OK;
  }
  if (a.Cin <= 4) sp_conv_kernel<4><<<grid, 256, 0, st>>>(a);
  else if (a.Cin <= 16) sp_conv_kernel<16><<<grid, 256, 0, st>>>(a);
  else if (a.Cin <= 32) sp_conv_kernel<32><<<grid, 256, 0, st>>>(a);
  else if (a.Cin <= 64) sp_conv_kernel<64><<<grid, 256, 0, st>>>(a);
  else return fail(GC_ERR_ARG, "sparse conv: at most 64 input channels (VoxelBackBone8x uses 4 / 16 / 32 / 64)");
  GC_HIP(hipGetLastError());
  return GC_OK;
}
inline int sp_cin_padded(int Cin) { return Cin <= 4 ? 4 : Cin <= 16 ? 16 : Cin <= 32 ? 32 : 64; }

// SparseConvTensor.dense() (height_compression.py:25): out [B][C][D][H][W], zero where inactive; the caller zeroes `out`
__global__ __launch_bounds__(256) void sp_dense_kernel(const float* __restrict__ feat, const long long* __restrict__ keys, int n, int C, const SpGrid g,
                                                       float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)n * C) return;
  const int j = (int)(i / C), c = (int)(i - (long long)j * C);
  const long long key = keys[j];
  if (key == kSpNoKey) return;
  int b, z, y, x;
  sp_decode(g, key, b, z, y, x);
  out[((((size_t)b * C + c) * g.D + z) * g.H + y) * g.W + x] = feat[i];
}

// =====================================================================================================================
// Training of the sparse layers (stage 1 trains the SECOND encoder: sparse_backbone_3d.py:12-31 with BatchNorm1d in train mode).
//   BatchNorm1d over the ACTIVE rows [n][C] (batch statistics, eps 1e-3, momentum 0.01) + ReLU, forward and backward
//   input gradient   = the same gather-GEMM kernel on dy with transposed weights and the inverse rulebook
//                      (SubM layers: the forward rulebook read with mirrored offsets; strided layers: sp_rules_inv_kernel)
//   weight gradient  = dW[o][ci][co] += sum_j x[nbr[o][j]][ci] dy[j][co]: one workgroup per (offset, chunk of output sites),
//                      rows staged through LDS, 16 accumulators per thread, f32 atomics at the end
// =====================================================================================================================
// per-channel sum / sum of squares over rows (f64 atomics): 256 threads = (256 / C) row slots x C channels, C in {16, 32, 64, 128}
__global__ __launch_bounds__(256) void bnrow_stats_kernel(const float* __restrict__ x, double* __restrict__ acc, int n, int C) {
  __shared__ double s_red[256][2];
  const int tid = threadIdx.x, c = tid % C, slot = tid / C, slots = 256 / C;
  double s = 0.0, q = 0.0;
  for (long long r = (long long)blockIdx.x * slots + slot; r < n; r += (long long)gridDim.x * slots) {
    const float v = x[(size_t)r * C + c];
    s += v; q += (double)v * v;
  }
  s_red[tid][0] = s; s_red[tid][1] = q;
  __syncthreads();
  if (slot == 0) {
    for (int k = 1; k < slots; ++k) { s += s_red[k * C + c][0]; q += s_red[k * C + c][1]; }
    atomicAdd(&acc[c * 2], s);
    atomicAdd(&acc[c * 2 + 1], q);
  }
}
__global__ void bn2d_finish_rows_kernel(const double* __restrict__ acc, float* __restrict__ save /*[C][2] mean, rstd*/, float* __restrict__ running_mean,
                                        float* __restrict__ running_var, float momentum, float eps, long long count, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double m = acc[c * 2] / (double)count;
  const double var = fmax(acc[c * 2 + 1] / (double)count - m * m, 0.0);
  save[c * 2] = (float)m;
  save[c * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean != nullptr) {   // nn.BatchNorm1d: running_var takes the unbiased estimate
    const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}
__global__ __launch_bounds__(256) void bnrow_apply_kernel(const float* __restrict__ x, const float* __restrict__ save, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ y, long long total, int C, int relu) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const float v = fmaf((x[i] - save[c * 2]) * save[c * 2 + 1], gamma[c], beta[c]);
  y[i] = relu ? fmaxf(v, 0.f) : v;
}
__global__ __launch_bounds__(256) void bnrow_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                               const float* __restrict__ save, double* __restrict__ acc, int n, int C, int relu) {
  __shared__ double s_red[256][2];
  const int tid = threadIdx.x, c = tid % C, slot = tid / C, slots = 256 / C;
  const float mean = save[c * 2], rstd = save[c * 2 + 1];
  double s = 0.0, q = 0.0;
  for (long long r = (long long)blockIdx.x * slots + slot; r < n; r += (long long)gridDim.x * slots) {
    const size_t e = (size_t)r * C + c;
    const float g = (relu && !(y[e] > 0.f)) ? 0.f : dy[e];
    s += g; q += (double)g * ((x[e] - mean) * rstd);
  }
  s_red[tid][0] = s; s_red[tid][1] = q;
  __syncthreads();
  if (slot == 0) {
    for (int k = 1; k < slots; ++k) { s += s_red[k * C + c][0]; q += s_red[k * C + c][1]; }
    atomicAdd(&acc[c * 2], s);
    atomicAdd(&acc[c * 2 + 1], q);
  }
}
__global__ __launch_bounds__(256) void bnrow_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                              const float* __restrict__ save, const float* __restrict__ gamma, const double* __restrict__ acc,
                                                              float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, long long total,
                                                              long long count, int C, int relu) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < C) {
    if (dgamma != nullptr) dgamma[i] += (float)acc[i * 2 + 1];
    if (dbeta != nullptr) dbeta[i] += (float)acc[i * 2];
  }
  if (i >= total) return;
  const int c = (int)(i % C);
  const float mean = save[c * 2], rstd = save[c * 2 + 1];
  const float mg = (float)(acc[c * 2] / (double)count), mgx = (float)(acc[c * 2 + 1] / (double)count);
  const float g = (relu && !(y[i] > 0.f)) ? 0.f : dy[i];
  const float xh = (x[i] - mean) * rstd;
  dx[i] = gamma[c] * rstd * (g - mg - xh * mgx);
}

// inverse rulebook of a strided SparseConv3d: inv[o][i] = the output site that reads input site i through kernel offset o, or -1
__global__ __launch_bounds__(256) void sp_rules_inv_kernel(const long long* __restrict__ in_keys, int n_in, const long long* __restrict__ out_keys, int n_out,
                                                           const SpConvGeom g, int* __restrict__ inv) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int o = blockIdx.y;
  if (i >= n_in) return;
  const long long key = in_keys[i];
  int res = -1;
  if (key != kSpNoKey) {
    int b, z, y, x;
    sp_decode(g.in, key, b, z, y, x);
    const int kx = o % g.k[2], ky = (o / g.k[2]) % g.k[1], kz = o / (g.k[2] * g.k[1]);
    const int tz = z + g.pad[0] - kz, ty = y + g.pad[1] - ky, tx = x + g.pad[2] - kx;
    if (tz >= 0 && ty >= 0 && tx >= 0 && tz % g.stride[0] == 0 && ty % g.stride[1] == 0 && tx % g.stride[2] == 0) {
      const int oz = tz / g.stride[0], oy = ty / g.stride[1], ox = tx / g.stride[2];
      if (oz < g.out.D && oy < g.out.H && ox < g.out.W) res = sp_find(out_keys, n_out, sp_encode(g.out, b, oz, oy, ox));
    }
  }
  inv[(size_t)o * n_in + i] = res;
}
// weight gradient: dW raw layout 0 ([Cout][K][Cin]); x [n_in][Cin], dy [n_out][Cout], nbr [K][n_out]
struct SpWgradArgs {
  const float* x; const float* dy; const int* nbr; float* dw;
  int n_out, K, Cin, Cout, rows_per_block;
};
__global__ __launch_bounds__(256) void sp_wgrad_kernel(const SpWgradArgs a) {
  __shared__ float sx[64 * 64], sy[64 * 64];   // [row][channel] of up to 64 gathered rows (Cin, Cout <= 64)
  const int o = blockIdx.y, tid = threadIdx.x;
  const int j0 = blockIdx.x * a.rows_per_block, j1 = min(j0 + a.rows_per_block, a.n_out);
  const int Cin = a.Cin, Cout = a.Cout;
  // thread -> accumulators (ci, co0 + 0..15): 256 threads cover 64 x 64 when each owns 16 consecutive co of one ci ... generically:
  const int per = (Cin * Cout + 255) / 256;          // accumulators per thread (<= 16)
  float acc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int jb = j0; jb < j1; jb += 64) {
    const int nr = min(64, j1 - jb);
    __syncthreads();
    for (int i = tid; i < 64 * Cin; i += 256) {
      const int r = i / Cin, c = i - r * Cin;
      float v = 0.f;
      if (r < nr) { const int idx = a.nbr[(size_t)o * a.n_out + jb + r]; if (idx >= 0) v = a.x[(size_t)idx * Cin + c]; }
      sx[r * 64 + c] = v;
    }
    for (int i = tid; i < 64 * Cout; i += 256) {
      const int r = i / Cout, c = i - r * Cout;
      sy[r * 64 + c] = r < nr ? a.dy[(size_t)(jb + r) * Cout + c] : 0.f;
    }
    __syncthreads();
    for (int e = 0; e < per; ++e) {
      const int flat = tid * per + e;
      if (flat >= Cin * Cout) break;
      const int ci = flat / Cout, co = flat - ci * Cout;
      float s = 0.f;
      for (int r = 0; r < 64; ++r) s = fmaf(sx[r * 64 + ci], sy[r * 64 + co], s);
      acc[e] += s;
    }
  }
  for (int e = 0; e < per; ++e) {
    const int flat = tid * per + e;
    if (flat >= Cin * Cout) break;
    const int ci = flat / Cout, co = flat - ci * Cout;
    if (acc[e] != 0.f) atomicAdd(&a.dw[((size_t)co * a.K + o) * Cin + ci], acc[e]);
  }
}

}  // namespace gc
