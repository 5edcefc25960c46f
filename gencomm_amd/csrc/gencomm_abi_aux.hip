// Second translation unit of libgencomm_hip.so: iou3d_nms (reference extension semantics), the point-cloud voxeliser,
// the V2X-ViT attention kernels and the sparse 3-D convolutions of the SECOND encoder.
// Kept apart from gencomm_abi.hip so that the rocPRIM templates do not lengthen the hot path's compile.
#include "../../include/gencomm_hip.h"

#include <algorithm>

#include "common.h"
#include "iou3d_kernels.h"
#include "loss_kernels.h"
#include "sparse_kernels.h"
#include "v2xvit_kernels.h"
#include "voxel_kernels.h"

using namespace gc;

extern "C" {

int gencomm_iou3d_pairwise_fwd(const float* boxes_a, int num_a, const float* boxes_b, int num_b, int mode, float* out, void* stream) {
  GC_CHECK_ARG(num_a >= 0 && num_b >= 0 && (mode == 0 || mode == 1), "bad num_a / num_b / mode");
  if (num_a == 0 || num_b == 0) return GC_OK;
  GC_CHECK_ARG(boxes_a && boxes_b && out, "null pointer");
  const long long total = (long long)num_a * num_b;
  GC_CHECK_ARG((total + 255) / 256 < (1LL << 31), "num_a * num_b too large");
  iou3d_pairwise_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(boxes_a, num_a, boxes_b, num_b, mode, out);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_iou3d_max_boxes(void) { return kIou3dMaxBoxes; }
long long gencomm_iou3d_nms_workspace_bytes(int n) {
  if (n < 0 || n > kIou3dMaxBoxes) { fail(GC_ERR_ARG, "n out of range (gencomm_iou3d_max_boxes)"); return -1; }
  return (long long)align_up((size_t)std::max(n, 1) * ((std::max(n, 1) + 63) / 64) * sizeof(unsigned long long), 256);
}

int gencomm_iou3d_nms_fwd(const float* boxes, int n, float thresh, int normal, long long* keep, int* count,
                          void* workspace, long long workspace_bytes, void* stream) {
  GC_CHECK_ARG(n >= 0 && n <= kIou3dMaxBoxes, "n out of range (gencomm_iou3d_max_boxes)");
  GC_CHECK_ARG(count != nullptr, "null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    GC_HIP(hipMemsetAsync(count, 0, sizeof(int), st));
    return GC_OK;
  }
  GC_CHECK_ARG(boxes && keep && workspace, "null pointer");
  if (gencomm_iou3d_nms_workspace_bytes(n) > workspace_bytes) return fail(GC_ERR_WORKSPACE, "workspace too small (gencomm_iou3d_nms_workspace_bytes)");
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(workspace);
  const int cb = (n + 63) / 64;
  if (normal) iou3d_nms_mask_kernel<true><<<dim3(cb, cb), 64, 0, st>>>(n, thresh, boxes, mask);
  else iou3d_nms_mask_kernel<false><<<dim3(cb, cb), 64, 0, st>>>(n, thresh, boxes, mask);
  iou3d_nms_reduce_kernel<<<1, 64, 0, st>>>(n, mask, keep, count);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

static int voxel_args(VoxelArgs& a, const float* points, int n, int nfeat, const float* voxel_size3, const float* range6, int max_points, int max_voxels) {
  GC_CHECK_ARG(n >= 0 && nfeat >= 3 && max_points >= 1 && max_voxels >= 1 && voxel_size3 && range6, "bad n / nfeat / max_points / max_voxels");
  a.points = points; a.n = n; a.nfeat = nfeat; a.max_points = max_points; a.max_voxels = max_voxels;
  long long cells = 1;
  for (int j = 0; j < 3; ++j) {
    a.vs[j] = voxel_size3[j]; a.r0[j] = range6[j];
    GC_CHECK_ARG(voxel_size3[j] > 0.f, "voxel size must be positive");
    a.grid[j] = (int)lrintf((range6[3 + j] - range6[j]) / voxel_size3[j]);  // np.round((range[3:6] - range[0:3]) / voxel_size), sp_voxel_preprocessor.py:37-39
    GC_CHECK_ARG(a.grid[j] >= 1, "empty grid");
    cells *= a.grid[j];
  }
  GC_CHECK_ARG(cells < 0xFFFFFFFFLL, "grid has too many cells for 32-bit keys");
  return GC_OK;
}

long long gencomm_voxelize_workspace_bytes(int n) {
  if (n < 0) { fail(GC_ERR_ARG, "n must be non-negative"); return -1; }
  return (long long)voxel_ws(n).total;
}

int gencomm_voxelize_fwd(const float* points, int n, int nfeat, const float* voxel_size3, const float* range6, int max_points, int max_voxels,
                         float* voxels, int* coords_zyx, int* num_points, int* count, void* workspace, long long workspace_bytes, void* stream) {
  VoxelArgs a{};
  if (int rc = voxel_args(a, points, n, nfeat, voxel_size3, range6, max_points, max_voxels)) return rc;
  GC_CHECK_ARG(voxels && coords_zyx && num_points && count && workspace && (n == 0 || points), "null pointer");
  if ((long long)voxel_ws(n).total > workspace_bytes) return fail(GC_ERR_WORKSPACE, "workspace too small (gencomm_voxelize_workspace_bytes)");
  return voxelize_enqueue(a, voxels, coords_zyx, num_points, count, (char*)workspace, (hipStream_t)stream);
}

int gencomm_warp_affine_fwd(const float* x, const double* theta, float* out, int n, int C, int H, int W, void* stream) {
  GC_CHECK_ARG(x && theta && out && n >= 1 && n <= 65535 && C >= 1 && H >= 1 && W >= 1, "bad arguments");
  WarpArgs a{x, theta, out, C, H, W};
  warp_affine_kernel<<<dim3((H * W + 255) / 256, n), 256, 0, (hipStream_t)stream>>>(a);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_hgt_attn_fwd(const float* qkv, const int* scene_off, float* out, int B, int heads, int dim_head, int HW, void* stream) {
  GC_CHECK_ARG(qkv && scene_off && out && B >= 1 && B <= 65535 && heads >= 1 && heads <= 65535 && HW >= 1, "bad arguments");
  GC_CHECK_ARG(B <= 65535 / 8, "too many scenes in one launch");
  HgtArgs a{qkv, scene_off, out, heads, HW, 1.0f / sqrtf((float)dim_head)};
  const dim3 grid((HW + 255) / 256, heads, B * 8);  // 8 query-agent slots per scene (v2xvit_kernels.h)
  hipStream_t st = (hipStream_t)stream;
  if ((long long)HW * heads * B >= 131072 && (dim_head == 16 || dim_head == 32)) {   // enough (pixel, head) threads to fill the machine: every value loaded once
    const dim3 gs((HW + 255) / 256, heads, B);
    if (dim_head == 32) hgt_attn_stream_kernel<32><<<gs, 256, 0, st>>>(a);
    else hgt_attn_stream_kernel<16><<<gs, 256, 0, st>>>(a);
    GC_HIP(hipGetLastError());
    return GC_OK;
  }
  if (dim_head == 32) hgt_attn_kernel<32><<<grid, 256, 0, st>>>(a);
  else if (dim_head == 64) hgt_attn_kernel<64><<<grid, 256, 0, st>>>(a);
  else if (dim_head == 16) hgt_attn_kernel<16><<<grid, 256, 0, st>>>(a);
  else if (dim_head == 8) hgt_attn_kernel<8><<<grid, 256, 0, st>>>(a);
  else return fail(GC_ERR_ARG, "hgt attention: dim_head must be 8, 16, 32 or 64");
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_win_attn_fwd(const float* qkv, const float* pos_embedding, float* out, int n, int heads, int dim_head, int window, int H, int W,
                         void* stream) {
  GC_CHECK_ARG(qkv && pos_embedding && out && n >= 1 && n <= 65535 && heads >= 1 && heads <= 65535, "bad arguments");
  GC_CHECK_ARG(H >= window && W >= window && H % window == 0 && W % window == 0, "H and W must be multiples of the window size");
  WinArgs a{qkv, pos_embedding, out, heads, H, W, 1.0f / sqrtf((float)dim_head)};
  hipStream_t st = (hipStream_t)stream;
  if (window == 4 && dim_head == 16) return win_attn_launch<16, 4>(a, n, st);
  if (window == 8 && dim_head == 32) return win_attn_launch<32, 8>(a, n, st);
  if (window == 16 && dim_head == 64) return win_attn_launch<64, 16>(a, n, st);
  if (window == 4 && dim_head == 32) return win_attn_launch<32, 4>(a, n, st);
  if (window == 8 && dim_head == 16) return win_attn_launch<16, 8>(a, n, st);
  if (window == 8 && dim_head == 64) return win_attn_launch<64, 8>(a, n, st);
  if (window == 16 && dim_head == 32) return win_attn_launch<32, 16>(a, n, st);
  return fail(GC_ERR_ARG, "window attention: supported (window, dim_head) pairs are (4,16) (4,32) (8,16) (8,32) (8,64) (16,32) (16,64)");
}

// ---- V2X-ViT backward building blocks (v2xvit_kernels.h) ----------------------------------------------------------------
int gencomm_warp_affine_bwd(const double* theta, const float* dout, float* dx, int n, int C, int H, int W, void* stream) {
  GC_CHECK_ARG(theta && dout && dx && n >= 1 && n <= 65535 && C >= 1 && H >= 1 && W >= 1, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  GC_HIP(hipMemsetAsync(dx, 0, (size_t)n * C * H * W * sizeof(float), st));
  WarpArgs a{nullptr, theta, dx, C, H, W};
  warp_affine_bwd_kernel<<<dim3((H * W + 255) / 256, n), 256, 0, st>>>(a, dout);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
int gencomm_hgt_attn_bwd(const float* qkv, const int* scene_off, const float* dout, float* dqkv, int B, int heads, int dim_head, int HW, void* stream) {
  GC_CHECK_ARG(qkv && scene_off && dout && dqkv && B >= 1 && B <= 65535 && heads >= 1 && heads <= 65535 && HW >= 1, "bad arguments");
  HgtBwdArgs a{qkv, scene_off, dout, dqkv, heads, HW, 1.0f / sqrtf((float)dim_head)};
  const dim3 grid((HW + 255) / 256, heads, B);
  hipStream_t st = (hipStream_t)stream;
  if (dim_head == 32) hgt_attn_bwd_kernel<32><<<grid, 256, 0, st>>>(a);
  else if (dim_head == 64) hgt_attn_bwd_kernel<64><<<grid, 256, 0, st>>>(a);
  else if (dim_head == 16) hgt_attn_bwd_kernel<16><<<grid, 256, 0, st>>>(a);
  else if (dim_head == 8) hgt_attn_bwd_kernel<8><<<grid, 256, 0, st>>>(a);
  else return fail(GC_ERR_ARG, "hgt attention: dim_head must be 16, 32 or 64");
  GC_HIP(hipGetLastError());
  return GC_OK;
}
long long gencomm_win_attn_bwd_scratch_floats(int n, int heads, int window, int H, int W) {
  if (n < 1 || heads < 1 || window < 1 || H < window || W < window || H % window || W % window) { fail(GC_ERR_ARG, "bad arguments"); return -1; }
  const long long T = (long long)window * window, nwin = (long long)(H / window) * (W / window);
  return (long long)n * heads * nwin * 2 * T * T;
}
// dqkv [n][3 inner][H][W] is overwritten; dpos [(2 window - 1)^2] is ACCUMULATED (+=); `out` = the forward's output
int gencomm_win_attn_bwd(const float* qkv, const float* pos_embedding, const float* out, const float* dout, float* dqkv, float* dpos,
                         float* scratch, int n, int heads, int dim_head, int window, int H, int W, void* stream) {
  GC_CHECK_ARG(qkv && pos_embedding && out && dout && dqkv && dpos && scratch && n >= 1 && n <= 65535 && heads >= 1 && heads <= 65535, "bad arguments");
  GC_CHECK_ARG(H >= window && W >= window && H % window == 0 && W % window == 0, "H and W must be multiples of the window size");
  WinBwdArgs a{qkv, pos_embedding, out, dout, dqkv, dpos, scratch, heads, H, W, 1.0f / sqrtf((float)dim_head)};
  hipStream_t st = (hipStream_t)stream;
  if (window == 4 && dim_head == 16) return win_attn_bwd_launch<16, 4>(a, n, st);
  if (window == 8 && dim_head == 32) return win_attn_bwd_launch<32, 8>(a, n, st);
  if (window == 16 && dim_head == 64) return win_attn_bwd_launch<64, 16>(a, n, st);
  if (window == 4 && dim_head == 32) return win_attn_bwd_launch<32, 4>(a, n, st);
  if (window == 8 && dim_head == 16) return win_attn_bwd_launch<16, 8>(a, n, st);
  if (window == 8 && dim_head == 64) return win_attn_bwd_launch<64, 8>(a, n, st);
  if (window == 16 && dim_head == 32) return win_attn_bwd_launch<32, 16>(a, n, st);
  return fail(GC_ERR_ARG, "window attention: supported (window, dim_head) pairs are (4,16) (4,32) (8,16) (8,32) (8,64) (16,32) (16,64)");
}

// radix-3 split attention over three branch maps [n][C][HW] (+ residual): gap -> fc1 -> LayerNorm -> ReLU -> fc2 -> softmax over the
// branches -> weighted sum (sub_modules/split_attn.py:31-62 with radix 3); scratch >= 4 n C floats
int gencomm_split3_attn_fwd(const float* a, const float* b, const float* c, const float* fc1_w, const float* ln_w, const float* ln_b,
                            const float* fc2_w, const float* residual, float* out, float* scratch, int n, int C, int HW, void* stream) {
  GC_CHECK_ARG(a && b && c && fc1_w && ln_w && ln_b && fc2_w && out && scratch, "null pointer");
  GC_CHECK_ARG(n >= 1 && n <= 65535 && C >= 1 && C <= 256 && HW >= 1, "split attention: 1 <= C <= 256");
  hipStream_t st = (hipStream_t)stream;
  float* gap = scratch;
  float* gate = scratch + (size_t)n * C;
  split3_gap_kernel<<<dim3(C, n), 256, 0, st>>>(a, b, c, gap, C, HW);
  Split3GateArgs ga{gap, fc1_w, ln_w, ln_b, fc2_w, gate, C};
  split3_gate_kernel<<<n, 256, 0, st>>>(ga);
  split3_apply_kernel<<<dim3((HW + 255) / 256, C, n), 256, 0, st>>>(a, b, c, gate, residual, out, C, HW);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

// ---- sparse 3-D convolution (SECOND encoder) -------------------------------------------------------------------------
static int sp_grid(SpGrid& g, int B, const int* dims3, const char* what) {
  GC_CHECK_ARG(dims3 != nullptr && B >= 1 && dims3[0] >= 1 && dims3[1] >= 1 && dims3[2] >= 1, what);
  g.B = B; g.D = dims3[0]; g.H = dims3[1]; g.W = dims3[2];
  return GC_OK;
}
static int sp_geom(SpConvGeom& g, int B, const int* in_dims3, const int* kernel3, const int* stride3, const int* pad3) {
  if (int rc = sp_grid(g.in, B, in_dims3, "bad input grid")) return rc;
  GC_CHECK_ARG(kernel3 && stride3 && pad3, "null pointer");
  g.out.B = B;
  int od[3];
  for (int j = 0; j < 3; ++j) {
    GC_CHECK_ARG(kernel3[j] >= 1 && kernel3[j] <= 7 && stride3[j] >= 1 && pad3[j] >= 0, "bad kernel / stride / padding");
    g.k[j] = kernel3[j]; g.stride[j] = stride3[j]; g.pad[j] = pad3[j];
    const int num = in_dims3[j] + 2 * pad3[j] - kernel3[j];
    GC_CHECK_ARG(num >= 0, "kernel larger than the padded grid");
    od[j] = num / stride3[j] + 1;
  }
  g.out.D = od[0]; g.out.H = od[1]; g.out.W = od[2];
  return GC_OK;
}

int gencomm_sp_out_dims(const int* in_dims3, const int* kernel3, const int* stride3, const int* pad3, int* out_dims3) {
  SpConvGeom g{};
  if (int rc = sp_geom(g, 1, in_dims3, kernel3, stride3, pad3)) return rc;
  GC_CHECK_ARG(out_dims3 != nullptr, "null pointer");
  out_dims3[0] = g.out.D; out_dims3[1] = g.out.H; out_dims3[2] = g.out.W;
  return GC_OK;
}

long long gencomm_sp_index_workspace_bytes(int n) {
  if (n < 0) { fail(GC_ERR_ARG, "n must be non-negative"); return -1; }
  return (long long)sp_sort_ws(n).total;
}
int gencomm_sp_index_fwd(const int* coords_bzyx, int n, int B, const int* dims3, long long* keys, int* perm, void* workspace, long long workspace_bytes,
                         void* stream) {
  SpGrid g{};
  if (int rc = sp_grid(g, B, dims3, "bad grid")) return rc;
  GC_CHECK_ARG(n >= 0, "n must be non-negative");
  if (n == 0) return GC_OK;
  GC_CHECK_ARG(coords_bzyx && keys && perm && workspace, "null pointer");
  const SpSortWs w = sp_sort_ws(n);
  if ((long long)w.total > workspace_bytes) return fail(GC_ERR_WORKSPACE, "workspace too small (gencomm_sp_index_workspace_bytes)");
  hipStream_t st = (hipStream_t)stream;
  char* wsp = (char*)workspace;
  long long* key = reinterpret_cast<long long*>(wsp + w.key);
  int* val = reinterpret_cast<int*>(wsp + w.val);
  sp_key_kernel<<<(n + 255) / 256, 256, 0, st>>>(coords_bzyx, n, g, key, val);
  size_t tb = w.temp_bytes;
  GC_HIP(rocprim::radix_sort_pairs(wsp + w.temp, tb, key, keys, val, perm, (size_t)n, 0, 63, st));   // keys are non-negative; out-of-grid rows carry kSpNoKey
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_sp_rules_fwd(const long long* out_keys, int n_out, const long long* in_keys, int n_in, int B, const int* in_dims3, const int* kernel3,
                         const int* stride3, const int* pad3, int* nbr, void* stream) {
  SpConvGeom g{};
  if (int rc = sp_geom(g, B, in_dims3, kernel3, stride3, pad3)) return rc;
  GC_CHECK_ARG(n_out >= 0 && n_in >= 0, "negative count");
  if (n_out == 0) return GC_OK;
  GC_CHECK_ARG(out_keys && nbr && (n_in == 0 || in_keys), "null pointer");
  const int K = g.k[0] * g.k[1] * g.k[2];
  sp_rules_kernel<<<dim3((n_out + 255) / 256, K), 256, 0, (hipStream_t)stream>>>(out_keys, n_out, in_keys, n_in, g, nbr);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

static long long sp_slots(const int* kernel3, const int* stride3) {
  long long s = 1;
  for (int j = 0; j < 3; ++j) s *= (kernel3[j] + stride3[j] - 1) / stride3[j];
  return s;
}
long long gencomm_sp_sites_capacity(int n_in, const int* kernel3, const int* stride3) {
  if (n_in < 0 || !kernel3 || !stride3 || stride3[0] < 1 || stride3[1] < 1 || stride3[2] < 1) { fail(GC_ERR_ARG, "bad arguments"); return -1; }
  return (long long)n_in * sp_slots(kernel3, stride3);
}
long long gencomm_sp_sites_workspace_bytes(int n_in, const int* kernel3, const int* stride3) {
  const long long cap = gencomm_sp_sites_capacity(n_in, kernel3, stride3);
  if (cap < 0) return -1;
  return (long long)sp_sites_ws(cap).total;
}
// out_keys must hold gencomm_sp_sites_capacity entries (the upper bound); *n_out (device) receives the number of output sites
int gencomm_sp_sites_fwd(const long long* in_keys, int n_in, int B, const int* in_dims3, const int* kernel3, const int* stride3, const int* pad3,
                         long long* out_keys, int* n_out, void* workspace, long long workspace_bytes, void* stream) {
  SpConvGeom g{};
  if (int rc = sp_geom(g, B, in_dims3, kernel3, stride3, pad3)) return rc;
  GC_CHECK_ARG(n_in >= 0 && n_out != nullptr, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (n_in == 0) {
    GC_HIP(hipMemsetAsync(n_out, 0, sizeof(int), st));
    return GC_OK;
  }
  GC_CHECK_ARG(in_keys && out_keys && workspace, "null pointer");
  SpSlots sl{};
  for (int j = 0; j < 3; ++j) sl.n[j] = sp_slot_count(g, j);
  const int slots = sl.n[0] * sl.n[1] * sl.n[2];
  const long long nc = (long long)n_in * slots;
  GC_CHECK_ARG(nc < (1LL << 31), "too many candidate sites");
  const SpSitesWs w = sp_sites_ws(nc);
  if ((long long)w.total > workspace_bytes) return fail(GC_ERR_WORKSPACE, "workspace too small (gencomm_sp_sites_workspace_bytes)");
  const long long nokey = (long long)B * g.out.D * g.out.H * g.out.W;   // one past the largest key
  int bits = 1;
  while ((1LL << bits) <= nokey) ++bits;                                // radix passes over the bits in use only
  char* wsp = (char*)workspace;
  long long* cand = reinterpret_cast<long long*>(wsp + w.cand);
  long long* sorted = reinterpret_cast<long long*>(wsp + w.sorted);
  sp_candidates_kernel<<<dim3((n_in + 255) / 256, slots), 256, 0, st>>>(in_keys, n_in, g, sl, nokey, cand);
  size_t tb = w.temp_bytes;
  GC_HIP(rocprim::radix_sort_keys(wsp + w.temp, tb, cand, sorted, (size_t)nc, 0, bits, st));
  tb = w.temp_bytes;
  GC_HIP(rocprim::unique(wsp + w.temp, tb, sorted, out_keys, n_out, (size_t)nc, rocprim::equal_to<long long>(), st));
  sp_fix_count_kernel<<<1, 64, 0, st>>>(out_keys, nokey, n_out);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

long long gencomm_sp_prepared_floats(int K, int Cin, int Cout) {
  if (K < 1 || Cin < 1 || Cin > 64 || Cout < 1) { fail(GC_ERR_ARG, "sparse conv: 1 <= Cin <= 64"); return -1; }
  return (long long)K * sp_cin_padded(Cin) * (long long)align_up((size_t)Cout, 32);
}
int gencomm_sp_prepare(const float* w, float* prepared, int K, int Cin, int Cout, int layout, void* stream) {
  GC_CHECK_ARG(w && prepared && layout >= 0 && layout <= 3, "bad arguments");
  const long long total = gencomm_sp_prepared_floats(K, Cin, Cout);
  if (total < 0) return GC_ERR_ARG;
  sp_prep_w_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(w, prepared, K, Cin, Cout, sp_cin_padded(Cin), (int)align_up((size_t)Cout, 32), layout);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
int gencomm_sp_conv_fwd(const float* x, const int* nbr, const float* prepared, const float* scale, const float* shift, float* y, int n_out, int K, int Cin,
                        int Cout, int relu, void* stream) {
  GC_CHECK_ARG(n_out >= 0 && K >= 1 && Cin >= 1 && Cin <= 64 && Cout >= 1, "bad arguments");
  if (n_out == 0) return GC_OK;
  GC_CHECK_ARG(x && nbr && prepared && scale && shift && y, "null pointer");
  SpConvArgs a{x, nbr, prepared, scale, shift, y, n_out, K, Cin, Cout, (int)align_up((size_t)Cout, 32), relu};
  return sp_conv_enqueue(a, (hipStream_t)stream);
}
// ---- training of the sparse layers -------------------------------------------------------------------------------------
int gencomm_bnrow_train_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* y, float* save,
                            double* scratch, float momentum, float eps, int relu, int n, int C, void* stream) {
  GC_CHECK_ARG(x && gamma && beta && y && save && scratch && n >= 1 && (C == 16 || C == 32 || C == 64 || C == 128), "BatchNorm over rows: C in {16, 32, 64, 128}");
  GC_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "running statistics: both or neither");
  hipStream_t st = (hipStream_t)stream;
  GC_HIP(hipMemsetAsync(scratch, 0, (size_t)C * 2 * sizeof(double), st));
  const int slots = 256 / C;
  bnrow_stats_kernel<<<std::max(1, std::min((n + slots * 16 - 1) / (slots * 16), 512)), 256, 0, st>>>(x, scratch, n, C);
  bn2d_finish_rows_kernel<<<(C + 63) / 64, 64, 0, st>>>(scratch, save, running_mean, running_var, momentum, eps, (long long)n, C);
  const long long total = (long long)n * C;
  bnrow_apply_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(x, save, gamma, beta, y, total, C, relu);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
int gencomm_bnrow_train_bwd(const float* x, const float* y, const float* dy, const float* save, const float* gamma, float* dx, float* dgamma,
                            float* dbeta, double* scratch, int relu, int n, int C, void* stream) {
  GC_CHECK_ARG(x && y && dy && save && gamma && dx && scratch && n >= 1 && (C == 16 || C == 32 || C == 64 || C == 128), "BatchNorm over rows: C in {16, 32, 64, 128}");
  hipStream_t st = (hipStream_t)stream;
  GC_HIP(hipMemsetAsync(scratch, 0, (size_t)C * 2 * sizeof(double), st));
  const int slots = 256 / C;
  bnrow_bwd_reduce_kernel<<<std::max(1, std::min((n + slots * 16 - 1) / (slots * 16), 512)), 256, 0, st>>>(x, y, dy, save, scratch, n, C, relu);
  const long long total = (long long)n * C;
  bnrow_bwd_apply_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(x, y, dy, save, gamma, scratch, dx, dgamma, dbeta, total, (long long)n, C, relu);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
int gencomm_sp_rules_inv_fwd(const long long* in_keys, int n_in, const long long* out_keys, int n_out, int B, const int* in_dims3, const int* kernel3,
                             const int* stride3, const int* pad3, int* inv, void* stream) {
  SpConvGeom g{};
  if (int rc = sp_geom(g, B, in_dims3, kernel3, stride3, pad3)) return rc;
  GC_CHECK_ARG(n_out >= 0 && n_in >= 0, "negative count");
  if (n_in == 0) return GC_OK;
  GC_CHECK_ARG(in_keys && inv && (n_out == 0 || out_keys), "null pointer");
  const int K = g.k[0] * g.k[1] * g.k[2];
  sp_rules_inv_kernel<<<dim3((n_in + 255) / 256, K), 256, 0, (hipStream_t)stream>>>(in_keys, n_in, out_keys, n_out, g, inv);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
// dw: raw layout 0 [Cout][K][Cin], ACCUMULATED (+=)
int gencomm_sp_wgrad(const float* x, const float* dy, const int* nbr, float* dw, int n_out, int K, int Cin, int Cout, void* stream) {
  GC_CHECK_ARG(n_out >= 0 && K >= 1 && K <= 65535 && Cin >= 1 && Cin <= 64 && Cout >= 1 && Cout <= 64, "sparse wgrad: at most 64 channels on either side");
  if (n_out == 0) return GC_OK;
  GC_CHECK_ARG(x && dy && nbr && dw, "null pointer");
  SpWgradArgs a{x, dy, nbr, dw, n_out, K, Cin, Cout, 0};
  a.rows_per_block = 1024;
  while (a.rows_per_block > 64 && (long long)((n_out + a.rows_per_block - 1) / a.rows_per_block) * K < 512) a.rows_per_block >>= 1;
  sp_wgrad_kernel<<<dim3((n_out + a.rows_per_block - 1) / a.rows_per_block, K), 256, 0, (hipStream_t)stream>>>(a);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_sp_dense_fwd(const float* feat, const long long* keys, int n, int C, int B, const int* dims3, float* out, void* stream) {
  SpGrid g{};
  if (int rc = sp_grid(g, B, dims3, "bad grid")) return rc;
  GC_CHECK_ARG(n >= 0 && C >= 1 && out, "bad arguments");
  hipStream_t st = (hipStream_t)stream;
  GC_HIP(hipMemsetAsync(out, 0, (size_t)B * C * g.D * g.H * g.W * sizeof(float), st));
  if (n == 0) return GC_OK;
  GC_CHECK_ARG(feat && keys, "null pointer");
  const long long total = (long long)n * C;
  sp_dense_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(feat, keys, n, C, g, out);
  GC_HIP(hipGetLastError());
  return GC_OK;
}
int gencomm_mean_vfe_fwd(const float* voxels, const int* num_points, const int* perm, float* out, int n, int max_points, int nfeat, void* stream) {
  GC_CHECK_ARG(n >= 0 && max_points >= 1 && nfeat >= 1, "bad arguments");
  if (n == 0) return GC_OK;
  GC_CHECK_ARG(voxels && num_points && out, "null pointer");
  const long long total = (long long)n * nfeat;
  mean_vfe_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(voxels, num_points, perm, out, n, max_points, nfeat);
  GC_HIP(hipGetLastError());
  return GC_OK;
}

int gencomm_head_loss(const float* cls, const float* reg, const float* dir, const float* pos, const float* neg, const float* tgt, float* gcls,
                      float* greg, float* gdir, double* sums, int B, int A, int H, int W, int num_bins, const double* anchor_yaw, double dir_offset,
                      float pos_cls_weight, float gamma, float alpha, float cls_weight, float sigma, float reg_weight, float dir_weight,
                      int batch_size, void* stream) {
  GC_CHECK_ARG(cls && reg && pos && neg && tgt && gcls && greg && sums, "null pointer");
  GC_CHECK_ARG(B >= 1 && A >= 1 && A <= kLossMaxAnchors && H >= 1 && W >= 1 && batch_size >= 1 && sigma > 0.f, "bad dims");
  GC_CHECK_ARG((long long)H * W * A < (1LL << 30) && B <= 65535, "map too large");
  GC_CHECK_ARG(dir == nullptr || (gdir && anchor_yaw && num_bins >= 1 && num_bins <= A), "direction term: gdir, anchor_yaw [A] and 1 <= num_bins <= A");
  HeadLossArgs a{};
  a.cls = cls; a.reg = reg; a.dir = dir; a.pos = pos; a.neg = neg; a.tgt = tgt;
  a.gcls = gcls; a.greg = greg; a.gdir = gdir; a.sums = sums;
  a.B = B; a.A = A; a.HW = H * W; a.has_dir = dir != nullptr; a.num_bins = num_bins;
  a.pos_cls_weight = pos_cls_weight; a.gamma = gamma; a.alpha = alpha; a.cls_weight = cls_weight; a.sigma = sigma; a.reg_weight = reg_weight;
  a.dir_weight = dir_weight; a.inv_bs = 1.0f / (float)batch_size; a.dir_offset = dir_offset;
  for (int i = 0; i < kLossMaxAnchors; ++i) a.anchor_yaw[i] = (dir != nullptr && i < A) ? anchor_yaw[i] : 0.0;
  return head_loss_enqueue(a, (hipStream_t)stream);
}

}  // extern "C"
