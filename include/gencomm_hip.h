/*
 * gencomm_hip.h -- C ABI of libgencomm_hip.so: the MI355X (gfx950) implementation of GenComm's
 * generative-communication hot path.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *   - every tensor pointer is DEVICE memory, float32, contiguous, NCHW unless stated;
 *   - the caller owns every buffer, including the scratch workspace (query *_workspace_bytes);
 *     the library allocates nothing and is re-entrant per stream; its only process-global state is
 *     the mode table (gencomm_set_mode) and the two diagnostics (kernel timer, kernel log), see below;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream); all work is
 *     enqueued on it and nothing synchronises the host, so calls may be captured in a hipGraph;
 *   - every entry point returns 0 on success, non-zero on error (1 bad argument / unsupported
 *     configuration, 2 workspace too small, 3 HIP runtime error) and never exits the process
 *     (contrast the reference's CHECK_* macros, opencood/pcdet_utils/iou3d_nms/src/iou3d_nms.cpp:14-25);
 *     gencomm_last_error() returns a thread-local description of the last failure.
 *
 * What each entry point replaces in the reference (paths relative to the reference checkout):
 *   gencomm_unet_prepare      DiffusionUNet parameter reads + get_timestep_embedding + temb MLP
 *                             (opencood/models/gencomm_modules/unet.py:10-28, :222-228, :309-312)
 *                             and every ResnetBlock.temb_proj (unet.py:124), for all T steps at once
 *   gencomm_unet_fwd          DiffusionUNet.forward (unet.py:307-344) = GenComm.gen_pred
 *                             (opencood/models/gencomm_modules/cond_diff.py:317-319)
 *   gencomm_q_sample_fwd      GenComm.q_sample (cond_diff.py:262-264)
 *   gencomm_denoise_fwd       GenComm.forward eval+train maths: ego repeat (cond_diff.py:332-337),
 *                             q_sample (:262-264, :372), p_sample_loop / p_sample / p_mean_variance /
 *                             q_posterior (:321-329, :302-315, :281-299, :272-279)
 *   gencomm_enhancer_fwd      Enhancer.forward -> Enhancer_block -> FRFN -> SplitAttn
 *                             (opencood/models/gencomm_modules/enhancer.py:367-383, :346-357, :222-250, :315-333)
 *   gencomm_msgext_fwd        MessageExtractorv2.forward (message_extractor_v2.py:103-118; DeformConv2d = torchvision DCNv1)
 *   gencomm_pillar_encode_fwd PointPillar encoder front half: PillarVFE + PointPillarScatter (heter_encoders.py:22-50)
 *   gencomm_warp_attfuse_fwd  AttFusion.forward + warp_affine_simple + ScaledDotProductAttention
 *                             (opencood/models/fuse_modules/fusion_in_one.py:131-151, :41-45;
 *                              opencood/models/sub_modules/torch_transformation_utils.py:323-332)
 *
 * Supported UNet family: ch = 8, ch_mult = all ones (levels <= 4), 1 <= num_res_blocks <= 4,
 * resamp_with_conv = true, dropout = 0, C % 8 == 0 -- a superset of every shipped GenComm yaml
 * (60/60 use ch 8, ch_mult [1,1], 2 res-blocks, attn_resolutions [16] = no AttnBlock).
 * `attn_mask`: bit l set = the AttnBlocks of level l exist (nominal resolution 128 >> l is in
 * attn_resolutions, unet.py:237,:252-253,:286-287); they run as flash-style streaming attention.
 */
#ifndef GENCOMM_HIP_H
#define GENCOMM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define GENCOMM_ABI_VERSION 11

int gencomm_abi_version(void);
const char* gencomm_last_error(void);
/* The code-generation flags the library was compiled with (set by the build: -DGENCOMM_BUILD_FLAGS). The kernels must be
 * built WITHOUT packed fp32 instructions (-Xclang -target-feature -Xclang -packed-fp32-ops): a v_pk_fma_f32 in
 * conv8h_kernel's epilogue returned wrong values under multi-workgroup occupancy; the Python binding
 * refuses a library whose build info lacks that switch and tests/test_abi.py disassembles the code object to check. */
const char* gencomm_build_info(void);

/* Library modes.  Explicit, atomic process-wide settings that every entry point reads ONCE when it is called (they
 * travel with the call from there on); the library reads no environment variable.  gencomm_set_mode returns 0 or 1
 * (unknown key / value out of range); gencomm_get_mode returns the value, -1 for an unknown key.
 *   GENCOMM_MODE_ARITH        0 (default): the hot path's 3x3 / 5x5 / Linear products on the f16 matrix pipe from exact THREE-term
 *                             splits of both fp32 operands (fp16 + fp16 + a bf8 third term: products to 2^-26, fp32 accumulation --
 *                             not narrower than an fp32 FMA); the general convolutions around the path (gencomm_conv2d_fwd, the
 *                             sparse convolutions) on exact-fp32 MFMA kernels; 1: exact-fp32 instruction kernels everywhere;
 *                             3: as 0, plus the general convolutions on the two-term fp16 split of round 2 (22-bit products: faster,
 *                             narrower than fp32 -- opt-in);
 *                             2: bf16 denoise mode (the reference's --half / autocast analogue, train_ddp.py:139-141): the
 *                             UNet's 8-channel intermediates are stored as bf16 and multiplied by single bf16 MFMAs
 *                             (fp32 accumulation, GroupNorm statistics in f64, the sampler's carried state in fp32);
 *                             inference only, no AttnBlock, W divisible by 4 at every level
 *   GENCOMM_MODE_SAMPLER      2: the loop is carried on conv_in's 8-channel output (one fused kernel per step replaces
 *                             conv_out + update + conv_in); 1: literal per-step structure of cond_diff.py:321-329 (tests
 *                             compare the two); 0 (default): automatic -- 2 on maps large enough for the 64x16-tile
 *                             kernels (x_t stays out of HBM), 1 on small, cache-resident maps (shorter dependent chain)
 *   GENCOMM_MODE_TILE_WANT    0 (default): automatic tile choice; > 0: minimum number of 64x16 workgroups before the
 *                             64x16-tile kernels are used (1 forces them onto small maps: tests)
 *   GENCOMM_MODE_ENH_FUSE     Enhancer at C = 64 -- 2 (default): Linear1 + depthwise stage + Linear2 in one kernel;
 *                             1: Linear1 + depthwise stage fused; 0: separate launches
 *   GENCOMM_MODE_CONV8H_MASK  diagnostic bit mask of 8-channel layer variants allowed on the f16 pipe (-1: all)
 *   GENCOMM_MODE_XCD_REMAP    1 (default): workgroup -> tile mapping keeps neighbouring tiles on one XCD; 0: grid order
 *   GENCOMM_MODE_DATAFLOW     1: the body of a UNet call (every layer between conv_in and conv_out) runs as ONE persistent launch
 *                             whose workgroups take (layer, agent, tile) items from per-XCD queues and wait on per-(layer, agent)
 *                             completion counters (no grid barrier); 0 (default): one launch per layer.  Same device functions, same results
 *                             -- EXCEPT when a dependency wait exhausts its bounded spin (a busy or shared GPU): the remaining tiles are
 *                             then skipped so that the grid drains, the entry point still returns GC_OK, and the activations are
 *                             invalid.  A caller that turns this mode on MUST poll gencomm_dataflow_error() after the call.  The mode
 *                             is opt-in, measured 2x slower than the per-layer launches, and kept for that measurement only
 *   GENCOMM_MODE_TILE8        0: 64x16-pixel tiles in every 8-channel layer of the f16 pipe; n > 0: launches with fewer than n
 *                             such workgroups run 64x8 tiles (half the dependent chain per workgroup, 28.8 KB of LDS: the half-resolution
 *                             level of large maps).  Same arithmetic; results identical up to the summation order of the statistics.
 *                             -1 (default): automatic = 256 (a third of a resident round: the half-resolution launches of ONE four-agent
 *                             scene; single-scene latency 10.07 -> 9.78 ms, batched throughput unchanged), off while GENCOMM_MODE_TILE_WANT
 *                             forces a tile size
 *   GENCOMM_MODE_BWD_STREAMS  1 (default): on calls of at least 2^17 pixels (n H W) gencomm_unet_bwd forks its weight-gradient launches onto a
 *                             library-owned side stream of the device (hipEventRecord on the caller's stream / hipStreamWaitEvent) and joins
 *                             them back before its last launches, so they overlap the input-gradient chain (15.2 -> 13.7 ms per training step
 *                             at 4 x 64 x 200 x 704); every buffer is still ordered on the caller's stream when the call returns.  2: on every
 *                             call (costs host time on small maps).  0: every launch on the caller's stream
 *   GENCOMM_MODE_PERSIST      bit mask of the 8-channel f16-pipe layer variants (1 conv1 8 -> 8, 2 conv1 16 -> 8, 4 conv2 + identity residual,
 *                             8 conv2 + 1x1 shortcut, 16 Upsample) that run as a PERSISTENT kernel when a launch has more 64x16 tiles than the device
 *                             has resident slots (3 per CU): every workgroup walks several tiles and requests the next tile's loads one tile
 *                             ahead, so that a slot does not sit through the first-load wait and the store drain of every tile.  Same tile
 *                             function, same arithmetic; results identical up to the summation order of the statistics.  Default 16: measured per
 *                             variant on MI355X only the Upsample convolution gains (38.7 -> 35.0 us per launch); the GroupNorm'd variants
 *                             lose 0 .. 13 % (registers carried across the tile: up to 96 B of scratch)
 *   GENCOMM_MODE_RESFUSE_EMU  0 (default).  1: TIMING EXPERIMENT ONLY -- the 8 -> 8 ResnetBlocks run the launch pattern a fused
 *                             conv1 + conv2 block would have (statistics-only conv1 pass; conv2 pass reading the block input with twice
 *                             the matrix work), an upper bound of that fusion's gain; the outputs are NOT the UNet's */
enum {
  GENCOMM_MODE_ARITH = 0, GENCOMM_MODE_SAMPLER = 1, GENCOMM_MODE_TILE_WANT = 2, GENCOMM_MODE_ENH_FUSE = 3,
  GENCOMM_MODE_CONV8H_MASK = 4, GENCOMM_MODE_XCD_REMAP = 5, GENCOMM_MODE_DATAFLOW = 6, GENCOMM_MODE_RESFUSE_EMU = 7,
  GENCOMM_MODE_TILE8 = 8, GENCOMM_MODE_BWD_STREAMS = 9, GENCOMM_MODE_PERSIST = 10
};
int gencomm_set_mode(int key, long long value);
long long gencomm_get_mode(int key);

/* Diagnostic kernel timer: gencomm_timer_start(family, capacity) arms it for ONE kernel family, gencomm_timer_start_mask
 * for a set (bit f = family f; 0 <= f < gencomm_timer_num_kernels(), name via gencomm_timer_kernel_name); until it is
 * stopped every launch of an armed family (up to `capacity` launches) is bracketed by a pair of HIP events on the stream
 * it is launched on, and its ALGORITHMIC byte count (the tensor bytes the layer has to read and write, computed by the
 * host from the launch shape) is recorded with it. gencomm_timer_stop synchronises those events and returns the summed
 * device time and the number of launches; gencomm_timer_stop_families returns them per family (arrays of n_families
 * entries, any may be NULL) together with the summed algorithmic bytes. Process-global diagnostic state (with the modes
 * above the only one); do not arm it while capturing a graph. */
int gencomm_timer_num_kernels(void);
const char* gencomm_timer_kernel_name(int family);
int gencomm_timer_start(int family, int capacity);
int gencomm_timer_start_mask(unsigned long long family_mask, int capacity);
int gencomm_timer_stop(double* total_ms, int* launches);
int gencomm_timer_stop_families(double* ms, int* launches, double* algorithmic_bytes, int n_families);

/* Diagnostic kernel log: between gencomm_klog_start and gencomm_klog_stop every launch site of the hot path notes the exact
 * kernel instantiation it chose; gencomm_klog_stop writes "name<TAB>count<NEWLINE>" lines into buf (status 1 when cap is
 * too small).  bench.py prints it so that the timed instantiations (in-kernel Philox noise: latent_step_h_kernel<2>,
 * conv_out_h_kernel<2>) are named next to the number.  Process-global diagnostic state like the timer. */
int gencomm_klog_start(void);
int gencomm_klog_stop(char* buf, int cap);

/* Diagnostic: the shader clock the device actually runs at while other work is in flight.  One wave spins for about `spin_us`
 * microseconds of the constant 100 MHz counter (s_memrealtime) and writes {shader-clock ticks (s_memtime), 100 MHz ticks} to
 * out_dev[2] (unsigned long long, device memory): MHz = 100 * ticks[0] / ticks[1].  Launch it on a stream of its own beside the
 * workload (tools/diag/clock_under_load.py). */
int gencomm_clock_probe(unsigned long long* out_dev, int spin_us, void* stream);

/* ----------------------------------------------------------------------------------------------
 * UNet parameters.  The "raw" blob is the concatenation of the module's parameters in EXECUTION
 * order; enumerate it with gencomm_unet_param_info (name = the reference's state_dict key under
 * `denoiser.`, e.g. "down.0.block.1.conv2.weight"; tensors keep the reference layout: conv OIHW,
 * linear [out,in]).  gencomm_unet_prepare turns it into the kernels' layout ([ic][tap][oc] conv
 * weights) and evaluates the timestep path for t = 0..T-1 into a bias table; call it once per
 * weight update.
 * -------------------------------------------------------------------------------------------- */
int gencomm_unet_num_params(int C, int levels, int res_blocks, int attn_mask);
int gencomm_unet_param_info(int C, int levels, int res_blocks, int attn_mask, int index,
                            char* name, int name_cap, long long* numel, long long* offset);
long long gencomm_unet_raw_floats(int C, int levels, int res_blocks, int attn_mask);
long long gencomm_unet_prepared_floats(int C, int levels, int res_blocks, int attn_mask, int T);
int gencomm_unet_prepare(const float* raw, float* prepared, int C, int levels, int res_blocks, int attn_mask, int T,
                         void* stream);

/* One DiffusionUNet call BACKWARDS (the training branch back-propagates through every call: cond_diff.py:342-360; what
 * torch autograd does for unet.py:307-344 in the reference). Re-runs the forward with every intermediate kept, then:
 * grad_xt [n][C][H][W] and grad_cond [n][2][H][W] are OVERWRITTEN with the gradients w.r.t. the two inputs; grad_raw
 * (gencomm_unet_raw_floats floats, the raw blob's layout) must be ZERO on entry and holds the gradient of EVERY parameter
 * on return: conv / norm / nin parameters, and the timestep path (temb.dense.{0,1}, every <block>.temb_proj), which enters
 * the forward only through each ResnetBlock's conv1 bias and is chained from this call's d conv1.bias by one small kernel.
 * `raw` = the parameter blob in the reference's layouts (what gencomm_unet_prepare consumed). attn_mask must be 0. */
long long gencomm_unet_bwd_workspace_bytes(int n, int C, int H, int W, int levels, int res_blocks, int attn_mask);
/* Forward of a call that will be differentiated: as gencomm_unet_fwd, but every intermediate (and its GroupNorm statistics)
 * stays in `workspace` (gencomm_unet_bwd_workspace_bytes; one workspace per call in flight -- MI355X has the HBM for it),
 * to be handed to gencomm_unet_bwd with forward_done = 1. With forward_done = 0 gencomm_unet_bwd re-runs that forward itself. */
int gencomm_unet_fwd_train(const float* prepared, const float* x_t, const float* cond, float* x0_out, int t,
                           int n, int C, int H, int W, int levels, int res_blocks, int attn_mask, int T,
                           void* workspace, long long workspace_bytes, void* stream);
/* ABI v8: gencomm_unet_fwd_train with the sampler's update of step t >= 1 fused into conv_out's epilogue, as the inference loop runs it
 * (cond_diff.py:272-279, :302-315): x_prev = coef1_t x0_hat + coef2_t x_t + sigma_t eps; sched_row = the five schedule constants of
 * timestep t on the device (row t of the table gencomm_denoise_fwd takes); eps = step_noise [n][C][H][W] when given, else the sampler's
 * in-kernel Philox field of (seed, t) -- the field gencomm_denoise_fwd adds. x0_hat is not stored, x_t is only read (x_prev != x_t). */
int gencomm_unet_fwd_train_step(const float* prepared, const float* x_t, const float* cond, float* x_prev, int t, const float* sched_row,
                                const float* step_noise, unsigned long long seed, int n, int C, int H, int W, int levels, int res_blocks,
                                int attn_mask, int T, void* workspace, long long workspace_bytes, void* stream);
int gencomm_unet_bwd(const float* prepared, const float* raw, const float* x_t, const float* cond, int t, const float* grad_x0,
                     float* grad_xt, float* grad_cond, float* grad_raw, int n, int C, int H, int W, int levels, int res_blocks,
                     int attn_mask, int T, int forward_done, void* workspace, long long workspace_bytes, void* stream);

/* ABI v8: gencomm_unet_bwd for a call inside the sampler chain (cond_diff.py:272-315 differentiated): grad_x0 = d_prev = d x_{t-1} as it
 * stands (NOT multiplied by coef1_t), and grad_xt = alpha * (the call's gradient with respect to x_t) + beta * d_prev is formed in the epilogue
 * of conv_in's input-gradient layer (alpha = coef1_t, beta = coef2_t) -- the two elementwise passes around the call disappear. grad_cond and
 * grad_raw are the gradients for grad_x0 as given: multiply them by alpha when accumulating (the call is linear in grad_x0). d_prev may be
 * NULL (then alpha / beta are ignored: gencomm_unet_bwd); grad_xt must not alias d_prev. */
int gencomm_unet_bwd_chain(const float* prepared, const float* raw, const float* x_t, const float* cond, int t, const float* grad_x0,
                           float alpha, float beta, const float* d_prev, float* grad_xt, float* grad_cond, float* grad_raw, int n, int C,
                           int H, int W, int levels, int res_blocks, int attn_mask, int T, int forward_done, void* workspace,
                           long long workspace_bytes, void* stream);

/* One 8 -> 8 channel 3x3 convolution (pad 1, bias) as the UNet's ResnetBlock / Upsample layers run it
 * (unet.py:52, :99-118), without norm or residual: dst[n,8,H,W] = conv(src[n,8,H,W], w[8,8,3,3]) + bias; dstat (nullable)
 * receives per-(sample, channel) {sum, sum of squares} of dst as [n][8][2] doubles. split = 1: the f16-pipe kernel with
 * exact three-term operand splits on 64x16 tiles (conv8h_kernels.h); 0: the exact-fp32 kernel; 0x100 | mask: diagnostic
 * instantiation that issues only the selected terms of the six-instruction product (bit 0 hi w1, 1 lo w1, 2 hi w2, 3 lo w2,
 * 4 hi w3, 5 t wb), so that each operand plane / table is tested alone. scratch >= 4096 floats. Unit-test entry. */
int gencomm_conv8_fwd(const float* src, const float* w_oihw, const float* bias, float* dst, double* dstat,
                      float* scratch, int n, int H, int W, int split, void* stream);

/* ABI v8: 3x3 stride-1 pad-1 convolution 16 -> 16 channels without bias -- FRFN.partial_conv3 at C = 64 (enhancer.py:218, :229-232: the
 * first C / 4 channels of the LayerNorm output) and, transposed = 1, its input gradient (w is the forward weight [16][16][3][3] in both
 * cases) -- on the UNet's exact-fp32 8-channel kernel: x / y are the first 16 channels of [n][x_ct][H][W] / [n][y_ct][H][W] tensors (so
 * the layer reads its slice of the LayerNorm output and writes its slice of Linear1's input without copies); x != y; scratch of
 * gencomm_conv3x3_c16_scratch_floats() floats. The general convolution took 160-250 us for this layer at 4 x 200 x 704 (72 MB of traffic). */
long long gencomm_conv3x3_c16_scratch_floats(void);
int gencomm_conv3x3_c16_fwd(const float* x, int x_ct, const float* w, int transposed, float* y, int y_ct, float* scratch, int n, int H, int W,
                            void* stream);

/* Scratch for one UNet call / the denoise loop on n agents of [C, H, W]. */
long long gencomm_denoise_workspace_bytes(int n, int C, int H, int W, int levels, int res_blocks, int attn_mask);

/* Diagnostic for GENCOMM_MODE_DATAFLOW: the persistent kernel's error word of the last UNet call on this workspace -- 0: every
 * dependency wait was satisfied; k + 1: a workgroup gave up (bounded spin) waiting for the predecessor of body op k, the call's
 * result is then invalid; -1: bad arguments.  Synchronises `stream`. */
int gencomm_dataflow_error(const void* workspace, int n, int C, int H, int W, int levels, int res_blocks, int attn_mask, void* stream);
/* Diagnostic: the kernel's counter block of the last call ([8] tickets taken per XCD queue, [64][n] finished tiles per body op and
 * agent, the error word) copied to host_out[count]. Synchronises `stream`. */
int gencomm_dataflow_words(const void* workspace, int n, int C, int H, int W, int levels, int res_blocks, int attn_mask,
                           unsigned int* host_out, int count, void* stream);

/* x0_hat[n,C,H,W] = UNet(cat[cond[n,2,H,W], x_t[n,C,H,W]], t) for one integer timestep t. */
int gencomm_unet_fwd(const float* prepared, const float* x_t, const float* cond, float* x0_out, int t,
                     int n, int C, int H, int W, int levels, int res_blocks, int attn_mask, int T,
                     void* workspace, long long workspace_bytes, void* stream);

/* The whole sampler.  sched = device float[T][5] rows {sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod,
 * posterior_mean_coef1, posterior_mean_coef2, exp(0.5*posterior_log_variance_clipped)}.
 * x_start of agent i = feat[src_row[i]] (src_row: device int32[n]; the ego row of i's scene).
 * noise0 [n,C,H,W] and step_noise [T,n,C,H,W] (loop order t = T-1..0; entry T-1 unused) are either
 * both given (explicit noise: parity tests) or both NULL (in-kernel Philox4x32-7 + Box-Muller keyed by `seed`; the step noise is sigma_t z rounded to fp16).
 * out [n,C,H,W] receives pred_feature; it is also the in-place x_t buffer of the loop. */
int gencomm_denoise_fwd(const float* prepared, const float* sched,
                        const float* feat, int n_feat_rows, const int* src_row, const float* cond,
                        float* out, const float* noise0, const float* step_noise, unsigned long long seed,
                        int n, int C, int H, int W, int levels, int res_blocks, int attn_mask, int T,
                        void* workspace, long long workspace_bytes, void* stream);

/* gencomm_denoise_fwd with the Philox key optionally read from DEVICE memory at kernel run time (seed_dev != NULL
 * overrides `seed`): a captured hipGraph of the call can be replayed with fresh noise by rewriting one 8-byte word. */
int gencomm_denoise_fwd_dseed(const float* prepared, const float* sched,
                              const float* feat, int n_feat_rows, const int* src_row, const float* cond,
                              float* out, const float* noise0, const float* step_noise, unsigned long long seed,
                              const unsigned long long* seed_dev,
                              int n, int C, int H, int W, int levels, int res_blocks, int attn_mask, int T,
                              void* workspace, long long workspace_bytes, void* stream);

/* q_sample alone (cond_diff.py:262-264 with the ego repeat of :332-337 folded in):
 *   out[i] = sched_row[0] * feat[src_row[i]] + sched_row[1] * eps[i]
 * sched_row: device float[5] (one row of the table above).  noise NULL = Philox stream `stream_id`
 * of `seed` (gencomm_denoise_fwd uses stream_id = T for this draw and t for the step draws). */
int gencomm_q_sample_fwd(const float* sched_row, const float* feat, int n_feat_rows, const int* src_row,
                         const float* noise, unsigned long long seed, unsigned int stream_id,
                         float* out, int n, int C, int H, int W, void* stream);

/* The sampler's in-kernel step noise of (seed, timestep stream_id = t), written out: out[n,C,H,W] = nu_t = fp16(sigma_t z)
 * as float32 -- the exact values gencomm_denoise_fwd adds at step t when noise0 / step_noise are NULL (same device
 * functions, same counter layout, in both sampler structures and every tile size); sched_row = device float[5] row of t
 * ([4] = sigma_t).  unrounded != 0 writes sigma_t z before the fp16 rounding instead (statistics tests).  The reference
 * draws torch.randn per step (cond_diff.py:307, MDD_utils.py:232-235); this entry exists so that a run with in-kernel
 * noise can be replayed through the oracle with explicit noise (step_noise[T-1-t] = nu_t / sigma_t).  q_sample's
 * initial noise is read back with gencomm_q_sample_fwd (zero feat, sched_row {0, 1, ..}, stream_id = T). */
int gencomm_step_noise_fwd(const float* sched_row, unsigned long long seed, unsigned int stream_id, float* out,
                           int n, int C, int H, int W, int unrounded, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Enhancer (live parameters only: block_1.{norm1,norm2,mlp.*}, split_attn.*), raw blob enumerated
 * like the UNet's; names are the reference's state_dict keys under `enhancer.`.
 * -------------------------------------------------------------------------------------------- */
int gencomm_enhancer_num_params(int C);
int gencomm_enhancer_param_info(int C, int index, char* name, int name_cap, long long* numel, long long* offset);
long long gencomm_enhancer_raw_floats(int C);
long long gencomm_enhancer_workspace_bytes(int n, int C, int H, int W);
/* out[n,C,H,W] = split_attn(block_1(x[n,C,H,W])) per agent (agents are independent). */
int gencomm_enhancer_fwd(const float* raw, const float* x, float* out, int n, int C, int H, int W,
                         void* workspace, long long workspace_bytes, void* stream);

/* ----------------------------------------------------------------------------------------------
 * MessageExtractorv2 (opencood/models/gencomm_modules/message_extractor_v2.py:70-120): the producer
 * of `conditions`. out[n,2,H,W] = fuse(dcn(x, offset_conv(x)) * se_gate). x [n,C,H,W], C % 8 == 0.
 * Raw blob enumerated like the UNet's; names = state_dict keys under `message_extractor_m<k>.`
 * (`bev_extractor.{offset1,dcn1,fuse.0,fuse.2,attn.1,attn.3}.{weight,bias}`).
 * -------------------------------------------------------------------------------------------- */
int gencomm_msgext_num_params(int C);
int gencomm_msgext_param_info(int C, int index, char* name, int name_cap, long long* numel, long long* offset);
long long gencomm_msgext_raw_floats(int C);
long long gencomm_msgext_workspace_bytes(int n, int C, int H, int W);
int gencomm_msgext_fwd(const float* raw, const float* x, float* out, int n, int C, int H, int W,
                       void* workspace, long long workspace_bytes, void* stream);

/* ----------------------------------------------------------------------------------------------
 * PointPillars front half, eval mode (opencood/models/heter_encoders.py:22-50 = PillarVFE
 * opencood/models/sub_modules/pillar_vfe.py:105-155 + PointPillarScatter point_pillar_scatter.py:42-76):
 * voxel_features [M,P,4] (P <= 32), voxel_num_points int32 [M], voxel_coords int32 [M,4] = (b,z,y,x)
 * -> out [B,64,ny,nx] (zero where no pillar). linear_w [64,10]; BatchNorm1d running statistics are
 * folded on the device into scratch128 (device float[128]). voxel_size3 / pc_range6 are HOST arrays.
 * The cell index is the reference's `z + y*nx + x`.
 * -------------------------------------------------------------------------------------------- */
int gencomm_pillar_encode_fwd(const float* voxel_features, const int* voxel_num_points, const int* voxel_coords,
                              const float* linear_w, const float* bn_weight, const float* bn_bias,
                              const float* bn_running_mean, const float* bn_running_var,
                              float* out, float* scratch128, int M, int P, int B, int nx, int ny,
                              const float* voxel_size3, const float* pc_range6, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Warp every agent into its scene's ego frame and fuse with per-pixel attention over agents,
 * keeping the ego row.  theta: device double[n][2][3], theta[i] = normalised affine that maps ego
 * grid coordinates to agent i's (= affine_matrix[b, 0, j] of normalize_pairwise_tfm,
 * opencood/utils/transformation_utils.py:68-92).  scene_off: device int32[B+1], agents of scene b
 * are rows scene_off[b] .. scene_off[b+1]-1 of x.  out [B,C,H,W].  At most 8 agents per scene.
 * -------------------------------------------------------------------------------------------- */
int gencomm_warp_attfuse_fwd(const float* x, const double* theta, const int* scene_off, float* out,
                             int B, int n, int C, int H, int W, void* stream);

/* ---- General 2-D convolution for the layers around the hot path -------------------------------------
 * Replaces the nn.Conv2d / nn.ConvTranspose2d (+ BatchNorm2d eval + ReLU) stacks of
 *   BaseBEVBackbone            opencood/models/sub_modules/base_bev_backbone.py:40-92, :94-123
 *   DownsampleConv/DoubleConv  opencood/models/sub_modules/downsample_conv.py:17-24
 *   cls/reg/dir heads          opencood/models/heter_model_baseline_w_gencomm_stage1.py:137-142
 * prepare: OIHW (transposed = 0), ConvTranspose2d IOHW with kernel == stride (transposed = 1), or (transposed = 2) the INPUT-GRADIENT
 *          convolution of a stride-1 layer straight from its forward OIHW weights (Cin = the forward's output channels, Cout = its input
 *          channels; taps flipped, channels transposed: what flip + transpose + contiguous + prepare did in four launches) -> k-major matrix
 *          [Cin*KH*KW][Cout] (resp. [Cin][Cout*KH*KW]) of the same number of floats, followed -- for GEMM shapes the f16-pipe kernel takes
 *          (3x3 / 2x2, Cin >= 16, Cin % 8 == 0, >= 32 GEMM rows; 1x1 and transposed convolutions stay on the exact-fp32 kernel: measured faster) -- by the three-term operand form of the weights and their per-row
 *          power-of-two scales (csrc/conv_h3_kernels.h), written by the same launch.  `prepared` must hold
 *          gencomm_conv2d_prepared_floats(Cin, Cout, KH, KW, transposed) floats (= Cin*Cout*KH*KW for the other shapes; -1: bad dims).
 * fold:    BatchNorm2d (eval) and/or conv bias -> per-channel scale/shift; pass NULL for the four BN tensors
 *          (and/or conv_bias) when absent.
 * fwd:     y[:, out_coff:out_coff+Cout] = act(conv(x) * scale + shift); supported: 3x3 stride 1|2 any pad, 1x1 stride 1;
 *          ups = s > 1 runs ConvTranspose2d(kernel = stride = s) (KH = KW = 1 on the prepared matrix, output H*s x W*s).
 *          KH = KW = 2 with ups = 2 (stride 1, pad 0; output 2H x 2W): the sub-pixel form of a TRANSPOSED 3x3 stride-2 pad-1 convolution,
 *          i.e. the input gradient of base_bev_backbone.py:57-63's stride-2 layers: GEMM row c*4 + a*2 + b = output channel c at
 *          pixel (2u + a, 2v + b), window = input rows u, u + 1 x columns v, v + 1 (zero beyond the map).
 *          y has out_ctotal channels (write into a slice of a concat buffer without a copy).
 *          Arithmetic follows GENCOMM_MODE_ARITH: 0 (default) = shapes whose prepared buffer carries the three-term form run on the f16
 *          matrix pipe with six matrix instructions per product block (operands split exactly into fp16 hi + fp16 lo + a third term:
 *          products accurate to 2^-26, fp32 accumulation; activations under a running power-of-two scale, weights under a per-row one:
 *          any finite fp32 input is safe), every other shape on the exact fp32 MFMA; 1 = exact fp32 MFMA for every shape; 3 (opt-in) =
 *          3x3 with Cin % 8 == 0 and 1x1 / ConvTranspose2d with >= 128 GEMM rows from two-term splits (22-bit products). */
int gencomm_conv2d_prepare(const float* weight, float* prepared, int Cin, int Cout, int KH, int KW, int transposed, void* stream);
long long gencomm_conv2d_prepared_floats(int Cin, int Cout, int KH, int KW, int transposed);
int gencomm_conv2d_fold(const float* bn_weight, const float* bn_bias, const float* bn_running_mean, const float* bn_running_var,
                        const float* conv_bias, float eps, int C, float* scale, float* shift, void* stream);
int gencomm_conv2d_fwd(const float* x, const float* prepared, const float* scale, const float* shift, float* y,
                       int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad, int relu,
                       int ups, int out_ctotal, int out_coff, void* stream);
/* the same (no transposed-conv mode, whole-tensor output) with act in {0 none, 1 ReLU, 2 erf-GELU (nn.GELU), 3 ReLU applied after
 * the residual add (ResNet BasicBlock)} and an optional residual [N][Cout][Ho][Wo]: Linear + GELU, Linear + skip connection,
 * conv + BatchNorm + identity + ReLU in one launch */
int gencomm_conv2d_act_res_fwd(const float* x, const float* prepared, const float* scale, const float* shift, const float* residual,
                               float* y, int N, int Cin, int H, int W, int Cout, int KH, int KW, int stride, int pad, int act,
                               void* stream);

/* ---- Detection tail ---------------------------------------------------------------
 * Replaces, for one agent's head outputs (batch 1), the torch/numpy/shapely chain of
 *   VoxelPostprocessor.post_process   opencood/data_utils/post_processor/voxel_postprocessor.py:1130-1244
 * det_decode: sigmoid(cls) > score_threshold, delta_to_boxes3d (:1351-1396), direction classifier fix (:1159-1175),
 *   boxes_to_corners_3d + project_box3d (box_utils.py:152-204, :278-316), remove_large_pred_bbx / remove_bbx_abnormal_z
 *   (box_utils.py:1062-1112); survivors are APPENDED in anchor order at position *count (device int, in/out -- zero it
 *   before the first agent, call once per agent for late fusion). Layouts: cls [A][H][W], reg [7A][H][W],
 *   dir [A*num_bins][H][W] or NULL, anchors [H][W][A][7] float32, transformation_matrix [16] float32 (device).
 *   corners [capacity][8][3], scores / anchor_index [capacity]; entries beyond capacity are dropped but still counted:
 *   the caller must compare *count with capacity.
 * nms_rotated: box_utils.nms_rotated (:915-960; scores sorted descending, exact ties by descending index; the `top`
 *   best kept, top <= 1024; IoU of the BEV quadrilaterals = first four corners, in float64, suppressed when
 *   (float)iou > iou_threshold), then -- when keep_range6 != NULL -- mask_boxes_outside_range_numpy (:384-421, all 8
 *   corners inside, bounds inclusive). *n_candidates is a DEVICE int (at most gencomm_nms_max_candidates() are
 *   considered). Outputs in pick order; *out_count device int.
 * bbox_overlaps: opencood/utils/box_overlaps.pyx:17-57, (N,4) x (K,4) -> (N,K), bit-identical to the compiled source. */
long long gencomm_det_workspace_bytes(int H, int W, int A);
long long gencomm_nms_workspace_bytes(void);
int gencomm_nms_max_candidates(void);
int gencomm_det_decode_fwd(const float* cls_preds, const float* reg_preds, const float* dir_preds, const float* anchors,
                           const float* transformation_matrix, int H, int W, int A, int num_bins, float score_threshold,
                           float dir_offset, int order_hwl, float* corners, float* scores, int* anchor_index, int* count,
                           int capacity, void* workspace, long long workspace_bytes, void* stream);
int gencomm_nms_rotated_fwd(const float* corners, const float* scores, const int* n_candidates, float iou_threshold, int top,
                            const float* keep_range6, float* out_boxes, float* out_scores, int* out_index, int* out_count,
                            void* workspace, long long workspace_bytes, void* stream);
int gencomm_bbox_overlaps_fwd(const float* boxes, const float* query_boxes, float* overlaps, int N, int K, void* stream);

/* MaxFusion.forward (opencood/models/fuse_modules/fusion_in_one.py:87-124): same warp, element-wise max over the
 * agents of a scene instead of the attention; arguments as gencomm_warp_attfuse_fwd. */
/* Backward of gencomm_warp_attfuse_fwd (what autograd does for fusion_in_one.py:131-151 + F.grid_sample in the reference's
 * training runs): grad_x [n][C][H][W] is OVERWRITTEN with the gradient w.r.t. x given grad_out [B][C][H][W]; the bilinear
 * gather's adjoint uses float atomics (summation order varies in the last bits). */
/* scratch: gencomm_warp_attfuse_bwd_scratch_floats(n, H, W) floats.  The ego's gradient (identity warp) is written once per pixel; the
 * other agents' gradients are GATHERED per source pixel from the few output pixels that sampled it (deterministic, no float atomics)
 * whenever their transform is rigid-like (decided on the device from theta); anything else takes the scatter with float atomics. */
long long gencomm_warp_attfuse_bwd_scratch_floats(int n, int H, int W);
int gencomm_warp_attfuse_bwd(const float* x, const double* theta, const int* scene_off, const float* grad_out, float* grad_x, float* scratch,
                             int B, int n, int C, int H, int W, void* stream);
int gencomm_warp_maxfuse_fwd(const float* x, const double* theta, const int* scene_off, float* out,
                             int B, int n, int C, int H, int W, void* stream);

/* Fast lane for callers that chain Enhancer -> fusion themselves (ScenePipeline): call
 * gencomm_enhancer_fwd with out == NULL (the token-major result and the channel gate stay in the
 * workspace, the NHWC->NCHW transpose launch is skipped), then this entry point with the SAME
 * workspace pointer and (n, C, H, W). Same maths as gencomm_warp_attfuse_fwd(enhancer output). C in {64,128,256}. */
int gencomm_warp_attfuse_tok_fwd(const void* enhancer_workspace, const double* theta, const int* scene_off, float* out,
                                 int B, int n, int C, int H, int W, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Training building blocks (NCHW fp32, correctness-first; the inference path never calls them): what torch autograd's
 * conv / layer_norm / gelu backward functions do in the reference's training runs for the Enhancer
 * (opencood/models/gencomm_modules/enhancer.py:346-357, :222-250) and the dense conv stacks.
 *   gencomm_conv2d_wgrad     dw[Cout][Cin][K][K] += sum dy (x) x, db[Cout] += sum dy (db may be NULL); K = 1 or 3, stride 1 or 2
 *                            (input gradients: gencomm_conv2d_fwd with the transposed, tap-flipped weight)
 *   gencomm_ln_nchw_fwd      LayerNorm over the channel axis of every pixel; residual != 0: out = x + LN(x)
 *   gencomm_ln_nchw_bwd      dx (= or +=), dgamma +=, dbeta +=; scratch: n * HW * 2 floats
 *   gencomm_dwconv3x3_fwd    depthwise 3x3 pad 1 (+ bias); flip != 0: correlation with the flipped taps = input gradient
 *   gencomm_dwconv3x3_wgrad  dw[C][3][3] +=, db[C] += (db may be NULL)
 *   gencomm_gelu_bwd         out = g * GELU'(v), erf form
 *   gencomm_lincomb_fwd      out = a x + b y + c z (y, z may be NULL; out may alias an input; 16-byte aligned pointers): the
 *                            sampler chain's x_{t-1} = c1 x0_hat + c2 x_t + sigma eps (cond_diff.py:272-315) and its adjoint
 * -------------------------------------------------------------------------------------------- */
int gencomm_conv2d_wgrad(const float* dy, const float* x, float* dw, float* db, int N, int Cin, int Hi, int Wi, int Cout,
                         int K, int stride, int pad, void* stream);
/* The same with caller-owned scratch of gencomm_conv2d_wgrad_scratch_floats(...) floats (0 = the shape needs none). Wide 3x3 layers
 * (Cin, Cout >= 32: the BEV backbone, opencood/models/sub_modules/base_bev_backbone.py:40-92) then store per-workgroup partial sums and
 * add them in a fixed order -- deterministic, and 10x faster than the f32 atomics of the scratch-less form on 256-channel layers. */
long long gencomm_conv2d_wgrad_scratch_floats(int N, int Cin, int Hi, int Wi, int Cout, int K, int stride, int pad);
int gencomm_conv2d_wgrad_ws(const float* dy, const float* x, float* dw, float* db, int N, int Cin, int Hi, int Wi, int Cout,
                            int K, int stride, int pad, float* scratch, long long scratch_floats, void* stream);
/* GroupNorm (+ SiLU when silu != 0) over NCHW for any channel count and group size (unet.py:36-37, :31-33): the general-width
 * DiffusionUNet's normalisation; stat = n * groups * 2 floats of caller scratch (mean, rstd per sample and group). */
int gencomm_gn_nchw_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stat, float eps, int silu,
                        int n, int C, int groups, int HW, void* stream);
int gencomm_ln_nchw_fwd(const float* x, const float* gamma, const float* beta, float* out, float eps, int residual, int n, int C, int HW, void* stream);
int gencomm_ln_nchw_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, float* scratch,
                        float eps, int accumulate, int n, int C, int HW, void* stream);
int gencomm_dwconv3x3_fwd(const float* x, const float* w, const float* b, float* y, int n, int C, int H, int W, int flip, void* stream);
int gencomm_dwconv3x3_wgrad(const float* x, const float* dy, float* dw, float* db, int n, int C, int H, int W, void* stream);
/* ABI v8: the same two with the depthwise layer's input taken as the first C channels of an [n][x_ct][H][W] tensor and, act = 1, as
 * GELU(x) evaluated on the values read (enhancer.py:236-243: x1 = GELU(Linear1(.))[:, :hidden] feeds the depthwise convolution) -- the
 * training forward no longer writes the GELU outputs of Linear1, the weight gradient reads Linear1's output instead. act = 0 and
 * x_ct = C are the plain calls above. */
int gencomm_dwconv3x3_act_fwd(const float* x, int x_ct, int act, const float* w, const float* b, float* y, int n, int C, int H, int W, int flip,
                              void* stream);
int gencomm_dwconv3x3_act_wgrad(const float* x, int x_ct, int act, const float* dy, float* dw, float* db, int n, int C, int H, int W, void* stream);
int gencomm_gelu_bwd(const float* v, const float* g, float* out, long long count, void* stream);
int gencomm_lincomb_fwd(float* out, const float* x, const float* y, const float* z, float a, float b, float c, long long count, void* stream);
/* Elementwise pieces of the Enhancer's backward on channel slices (slice = channels [c0, c0 + nch) of an [n][ct][HW] tensor), op:
 *   0 copy: o0 slice = a slice                                    (replaces cat / contiguous)
 *   1 o0 = GELU(a[:, :nch]), o1 = GELU(a[:, nch:])                (a has 2 nch channels; o0, o1 have nch)
 *   2 o0 = GELU(a) * b                                            (gated product, enhancer.py:241-246)
 *   3 o0 = GELU'(a) c b, o1 slice = GELU'(d slice) c GELU(a)      (a = u, b = x2, c = d gated, d = Linear1 output; o1 / d share ct, c0)
 *   4 o0 slice = GELU'(a slice) b                                 (a, o0 share ct, c0; b has nch channels)
 *   5 o0 = GELU(a) * GELU(d slice)                                (ABI v8; d has o1_ct channels, its slice starts at o1_c0: op 2 reading
 *                                                                  x2 = GELU(Linear1 output) from Linear1's output itself)
 *   6 o0 = GELU'(a) c GELU(d slice), o1 slice = GELU'(d slice) c GELU(a)   (ABI v8; op 3 without b: x2 recomputed from d)
 * GELU' = Phi(x) + x phi(x) with the library's erf form (|Phi error| <= 0.85e-7) and one v_exp_f32 for the density.
 * When HW % 4 == 0 and every pointer is 16-byte aligned a lane handles four pixels with 128-bit accesses.
 * nc_scale: out = x * a[n][c] + b[n][c] (b may be NULL);  nc_dot: out[n][c] = sum_p x (* y) with f64 accumulation (out is zeroed). */
int gencomm_ew_slice_fwd(int op, const float* a, const float* b, const float* c, const float* d, float* o0, float* o1, int n, int nch, int HW,
                         int a_ct, int a_c0, int o0_ct, int o0_c0, int o1_ct, int o1_c0, void* stream);
int gencomm_nc_scale_fwd(const float* x, const float* a, const float* b, float* out, int n, int C, int HW, void* stream);
int gencomm_nc_dot_fwd(const float* x, const float* y, float* out, int n, int C, int HW, void* stream);

/* ----------------------------------------------------------------------------------------------
 * iou3d_nms with the reference extension's own semantics (opencood/pcdet_utils/iou3d_nms: src/iou3d_nms_kernel.cu:104-372,
 * src/iou3d_nms.cpp:30-135; Python wrappers iou3d_nms_utils.py:32-46, :147-181, :255-289). Boxes are [n][7] float32
 * (x, y, z, dx, dy, dz, heading). gencomm_iou3d_pairwise_fwd: out[num_a][num_b] = BEV overlap area (mode 0,
 * boxes_overlap_bev_gpu) or BEV IoU (mode 1, boxes_iou_bev_gpu). gencomm_iou3d_nms_fwd: boxes already in descending
 * score order (as nms_gpu / nms_normal_gpu pass them); keep[0..*count) = indices kept, ascending -- the reference returns
 * the count on the host after a blocking copy of the masks, here masks, greedy reduction, keep list and count all stay on
 * the device and the call is asynchronous. normal != 0 ignores the heading (nms_normal_gpu). At most
 * gencomm_iou3d_max_boxes() boxes. (The live GenComm post-processor uses the shapely-semantics NMS below,
 * gencomm_nms_rotated_fwd; this API serves the FPV-RCNN / iou-loss callers of the extension.)
 * -------------------------------------------------------------------------------------------- */
int gencomm_iou3d_pairwise_fwd(const float* boxes_a, int num_a, const float* boxes_b, int num_b, int mode, float* out, void* stream);
int gencomm_iou3d_max_boxes(void);
long long gencomm_iou3d_nms_workspace_bytes(int n);
int gencomm_iou3d_nms_fwd(const float* boxes, int n, float thresh, int normal, long long* keep, int* count,
                          void* workspace, long long workspace_bytes, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Point cloud -> voxels with the semantics of spconv's CPU point-to-voxel as the reference's dataloader uses it
 * (opencood/data_utils/pre_processor/sp_voxel_preprocessor.py:25-29, :54-68): points [n][nfeat] (x, y, z first), voxels
 * in order of first appearance, points in input order, at most max_points per voxel / max_voxels voxels,
 * coords (z, y, x), grid = round((range[3:6] - range[0:3]) / voxel_size). Outputs are sized for max_voxels
 * (voxels [max_voxels][max_points][nfeat] zero-padded, coords_zyx [max_voxels][3], num_points [max_voxels]); *count = the
 * number of voxels produced (device int). Deterministic (stable radix sort + scans, no order-dependent atomics).
 * -------------------------------------------------------------------------------------------- */
long long gencomm_voxelize_workspace_bytes(int n);
int gencomm_voxelize_fwd(const float* points, int n, int nfeat, const float* voxel_size3, const float* range6, int max_points,
                         int max_voxels, float* voxels, int* coords_zyx, int* num_points, int* count,
                         void* workspace, long long workspace_bytes, void* stream);

/* ----------------------------------------------------------------------------------------------
 * V2X-ViT fusion building blocks (opencood/models/fuse_modules/fusion_in_one.py:355-407, sub_modules/hmsa.py:117-150,
 * sub_modules/mswin.py:47-83); NCHW fp32 per agent. The Linear layers around them run through gencomm_conv2d_fwd (1x1),
 * the LayerNorms through gencomm_ln_nchw_fwd; gencomm_amd/v2xvit.py is the module with the reference's state_dict keys.
 *   gencomm_warp_affine_fwd  warp_affine_simple: out[i] = bilinear sample of x[i] on the float64 affine grid theta[i] [2][3]
 *   gencomm_hgt_attn_fwd     qkv [n][3 heads dim_head][HW] (q | k | v blocks) -> out [n][heads dim_head][HW]: per pixel and head,
 *                            softmax attention across the agents of each scene (scene_off [B+1]); relation matrices folded into
 *                            the k / v projections by the caller (all agents have type 0 in GenComm's use)
 *   gencomm_win_attn_fwd     per agent, head and window x window tile: softmax(q k^T / sqrt(dim_head) + pos[dy][dx]) v,
 *                            pos_embedding [2 window - 1][2 window - 1]
 * -------------------------------------------------------------------------------------------- */
int gencomm_warp_affine_fwd(const float* x, const double* theta, float* out, int n, int C, int H, int W, void* stream);
int gencomm_hgt_attn_fwd(const float* qkv, const int* scene_off, float* out, int B, int heads, int dim_head, int HW, void* stream);
int gencomm_win_attn_fwd(const float* qkv, const float* pos_embedding, float* out, int n, int heads, int dim_head, int window, int H, int W,
                         void* stream);
/* Backward of the three blocks above (training with fusion_method v2xvit): gencomm_warp_affine_bwd zeroes dx and scatters dout
 * through the bilinear weights; gencomm_hgt_attn_bwd overwrites dqkv; gencomm_win_attn_bwd overwrites dqkv, ACCUMULATES dpos
 * [(2 window - 1)^2] and needs the forward's output `out` and gencomm_win_attn_bwd_scratch_floats floats of scratch. */
int gencomm_warp_affine_bwd(const double* theta, const float* dout, float* dx, int n, int C, int H, int W, void* stream);
int gencomm_hgt_attn_bwd(const float* qkv, const int* scene_off, const float* dout, float* dqkv, int B, int heads, int dim_head, int HW,
                         void* stream);
long long gencomm_win_attn_bwd_scratch_floats(int n, int heads, int window, int H, int W);
int gencomm_win_attn_bwd(const float* qkv, const float* pos_embedding, const float* out, const float* dout, float* dqkv, float* dpos,
                         float* scratch, int n, int heads, int dim_head, int window, int H, int W, void* stream);
/* radix-3 split attention over the three window branches (sub_modules/split_attn.py:31-62): out = sum_r softmax_r(fc2(ReLU(LN(fc1(
 * mean_HW(a + b + c))))))[r] * branch_r (+ residual); fc1 [C][C], fc2 [3 C][C] without biases; scratch >= 4 n C floats; C <= 256 */
int gencomm_split3_attn_fwd(const float* a, const float* b, const float* c, const float* fc1_w, const float* ln_w, const float* ln_b,
                            const float* fc2_w, const float* residual, float* out, float* scratch, int n, int C, int HW, void* stream);

/* BatchNorm2d with BATCH statistics (training mode; base_bev_backbone.py:47-52: eps 1e-3, momentum 0.01) around the HIP convolutions of
 * the backbone / shrink stacks: y = act(gamma (x - mean_batch) / sqrt(var_batch + eps) + beta) over NCHW, running statistics updated as
 * nn.BatchNorm2d does (unbiased variance; pass null to leave them alone); save [C][2] = (mean, rstd) for the backward; scratch >= 2 C
 * doubles. Backward: dx overwritten, dgamma / dbeta accumulated, y = the forward's output (supplies the ReLU mask).
 * ABI v8: `relu` bit 1 of gencomm_bn2d_train_bwd set = dgamma / dbeta are WRITTEN (the caller need not zero them); clear = accumulated.
 * ABI v9: `relu` bit 2 (value 4) of either call set = `scratch` arrives ZEROED (the caller carved it from a zero-filled pool: no memset
 * launch here); gencomm_bn2d_train_fwd takes `num_batches_tracked` (device int64, may be null) and adds 1 to it inside its own kernel --
 * nn.BatchNorm2d's counter without a launch of its own (batchnorm.py: `self.num_batches_tracked.add_(1)`). */
int gencomm_bn2d_train_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* y, float* save,
                           double* scratch, float momentum, float eps, int relu, int n, int C, int HW, long long* num_batches_tracked,
                           void* stream);
int gencomm_bn2d_train_bwd(const float* x, const float* y, const float* dy, const float* save, const float* gamma, float* dx, float* dgamma,
                           float* dbeta, double* scratch, int relu, int n, int C, int HW, void* stream);
/* ABI v10: one conv -> BatchNorm2d(batch statistics) -> ReLU layer (base_bev_backbone.py:40-83) per call and direction -- compositions of
 * gencomm_conv2d_prepare / _fwd / _wgrad_ws and gencomm_bn2d_train_*, so that a layer costs the host ONE foreign call instead of three or four.
 * fwd: weight OIHW [Cout][Cin][K][K] (K 1 | 3, stride 1 | 2), bias may be null; unit_scale / zero_shift: [Cout] ones / zeros; prepared:
 * scratch of gencomm_conv2d_prepared_floats(Cin, Cout, K, K, 0) floats; pre = the convolution's output (kept for the backward), y, save
 * [Cout][2]; stat_scratch: 2 Cout doubles, ZEROED by the caller.  bwd (stride 1): dpre scratch [N][Cout][Ho][Wo]; dw / dbias (may be null)
 * arrive ZEROED and are accumulated into, dgamma / dbeta written, dx may be null; unit_scale / zero_shift: [Cin]; prepared:
 * gencomm_conv2d_prepared_floats(Cout, Cin, K, K, 2) floats; wgrad_scratch as gencomm_conv2d_wgrad_ws. */
int gencomm_convbn_train_fwd(const float* x, const float* weight, const float* bias, const float* unit_scale, const float* zero_shift,
                             const float* gamma, const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked,
                             float momentum, float eps, int relu, float* prepared, float* pre, float* y, float* save, double* stat_scratch,
                             int N, int Cin, int H, int W, int Cout, int K, int stride, int pad, void* stream);
int gencomm_convbn_train_bwd(const float* x, const float* weight, const float* pre, const float* y, const float* gy, const float* save, const float* gamma,
                             const float* unit_scale, const float* zero_shift, int relu, float* dpre, float* dx, float* dw, float* dbias, float* dgamma,
                             float* dbeta, double* stat_scratch, float* prepared, float* wgrad_scratch, long long wgrad_scratch_floats,
                             int N, int Cin, int H, int W, int Cout, int K, int pad, void* stream);

/* Training path of the PointPillars per-pillar network (pillar_vfe.py:31-54): the Linear layer runs as a 1x1 convolution over the
 * [1, C, 1, M P] point-slot layout (gencomm_conv2d_fwd / _wgrad), BatchNorm1d with batch statistics as gencomm_bn2d_train_*, and the
 * max over the P slots of a pillar here: x [C][M][P] -> out [M][C] with the arg-max slot (first maximum), backward = routing. */
int gencomm_slot_max_fwd(const float* x, float* out, unsigned char* arg, int C, int M, int P, void* stream);
/* The same layer in ONE piece for training mode (BatchNorm1d with batch statistics), without the [M P, C] intermediates (csrc/pfn_kernels.h):
 * the statistics of the Linear output follow exactly from the inputs' first and second moments over all M P slots, ReLU(BN(.)) is
 * monotone in the Linear output, and the dense part of BatchNorm's backward folds into the same moments -- two launches forward, two
 * backward, instead of a 1x1 convolution, BatchNorm and slot-max over five 393-MB tensors at the stage-1 recipe's 48 000 pillars.
 * feats [M][P][F] (slots beyond a pillar's point count zeroed, pillar_vfe.py:96-100), weight [C][F]; F in {9, 10, 11}, C in {32, 64, 128, 256},
 * P <= 255.  fwd: out [M][C], arg [M][C] (the extreme slot), save [C][2] (mean, rstd), moments [gencomm_pfn_moment_doubles(F)] (kept for
 * the backward); running statistics / num_batches_tracked updated as nn.BatchNorm1d does (may be null).  bwd: dweight [C][F], dgamma,
 * dbeta WRITTEN (each may be null); scratch: gencomm_pfn_bwd_scratch_doubles(F, C) doubles (per-block partial sums, added up in a fixed order). */
int gencomm_pfn_train_fwd(const float* feats, const float* weight, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          long long* num_batches_tracked, float momentum, float eps, float* out, unsigned char* arg, float* save, double* moments,
                          int M, int P, int F, int C, void* stream);
int gencomm_pfn_train_bwd(const float* feats, const float* weight, const float* gamma, const float* beta, const float* save, const double* moments,
                          const float* gout, const unsigned char* arg, float* dweight, float* dgamma, float* dbeta, double* scratch,
                          int M, int P, int F, int C, void* stream);
long long gencomm_pfn_moment_doubles(int F);
long long gencomm_pfn_bwd_scratch_doubles(int F, int C);
int gencomm_slot_max_bwd(const float* dout, const unsigned char* arg, float* dx, int C, int M, int P, void* stream);

/* Detection-head terms of the training criterion in one launch, forward and gradients (opencood/loss/point_pillar_loss.py:36-126 with
 * :129-170 and :216-245; called through point_pillar_gencomm_loss.py:16-58): sigmoid focal classification loss, smooth-L1 regression loss
 * on the sin-difference encoding, softmax cross entropy of the heading bin, each weighted and divided by batch_size.
 * cls [B][A][H][W], reg [B][7A][H][W], dir [B][A*A][H][W] (the reference's view(-1, anchor_num): logits of anchor a = channels a*A..a*A+A-1;
 * null: no direction term), pos / neg [B][H][W][A], tgt [B][H][W][7A]; A <= 8.  sums [4] (double, ZEROED BY THE CALLER) += cls, reg, dir loss, their sum;
 * gcls / greg / gdir = d (cls + reg + dir loss) / d map, overwritten.  anchor_yaw: HOST array of A radians (float64 like the reference). */
int gencomm_head_loss(const float* cls, const float* reg, const float* dir, const float* pos, const float* neg, const float* tgt, float* gcls,
                      float* greg, float* gdir, double* sums, int B, int A, int H, int W, int num_bins, const double* anchor_yaw, double dir_offset,
                      float pos_cls_weight, float gamma, float alpha, float cls_weight, float sigma, float reg_weight, float dir_weight,
                      int batch_size, void* stream);

/* Training path of MessageExtractorv2's deformable 3x3 convolution (message_extractor_v2.py:78,:108; DCNv1, padding 1, one offset
 * group), split into its sampling half and its GEMM half so that the backward is GEMMs on the general kernels + one scatter:
 *   gencomm_dcn_sample_fwd   col[n][c * 9 + k][p] = bilinear sample of x[n][c] at tap k's displaced position (zero outside)
 *                            -- the deformable convolution is then gencomm_conv2d_fwd(col, W [64][9 C], 1x1)
 *   gencomm_dcn_scatter_bwd  given dcol: dx += scatter of the bilinear weights (atomics; dx zeroed or pre-filled by the caller),
 *                            doffset [n][18][H][W] = d loss / d offset (overwritten) */
int gencomm_dcn_sample_fwd(const float* x, const float* offset, float* col, int n, int C, int H, int W, void* stream);
int gencomm_dcn_scatter_bwd(const float* x, const float* offset, const float* dcol, float* dx, float* doffset, int n, int C, int H, int W,
                            void* stream);
/* The same with caller-owned scratch of gencomm_dcn_scatter_scratch_floats(n, C, H, W) floats: the input gradient is then formed without
 * global atomics for every bilinear corner within 4 pixels of its sampling pixel's 16 x 16 tile (per-tile regions stored, then summed per
 * output pixel); dx must be zero on entry in both forms. */
long long gencomm_dcn_scatter_scratch_floats(int n, int C, int H, int W);
int gencomm_dcn_scatter_bwd_ws(const float* x, const float* offset, const float* dcol, float* dx, float* doffset, int n, int C, int H, int W,
                               float* scratch, long long scratch_floats, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Sparse 3-D convolutions of the SECOND encoder without spconv (opencood/models/heter_encoders.py:52-81,
 * sub_modules/sparse_backbone_3d.py:33-152, mean_vfe.py:14-33, height_compression.py:10-30). A sparse tensor is
 * (keys int64 [n] ascending, features fp32 [n][C]); key = ((b D + z) H + y) W + x. dims3 = (D, H, W); kernel3 / stride3 /
 * pad3 = (z, y, x). spconv's arithmetic is not part of the reference checkout: parity unpinned, semantics as published
 * (SubMConv3d: output sites = input sites; SparseConv3d: every site whose receptive field holds an active input).
 *   gencomm_sp_index_fwd    coords [n][4] int32 (b, z, y, x) -> keys (sorted) and perm (sorted row -> input row)
 *   gencomm_sp_sites_fwd    output sites of a strided SparseConv3d: out_keys (capacity gencomm_sp_sites_capacity, ascending),
 *                           *n_out (device int)
 *   gencomm_sp_rules_fwd    rulebook nbr [K][n_out] int32: input row of out_coord * stride - pad + offset, -1 if inactive
 *                           (SubMConv3d: out_keys = in_keys, stride 1, pad = k / 2)
 *   gencomm_sp_prepare      weights -> kernel layout; layout 0 = spconv 2.x [Cout][kD][kH][kW][Cin], 1 = spconv 1.x
 *                           [kD][kH][kW][Cin][Cout]; 2 / 3 = the input-gradient convolution of a layout-0 weight (see below)
 *   gencomm_sp_conv_fwd     y[j][co] = act(scale[co] * sum_o sum_ci w[o][ci][co] x[nbr[o][j]][ci] + shift[co]): gather-GEMM on
 *                           v_mfma_f32_32x32x2_f32; scale / shift = folded BatchNorm1d (gencomm_conv2d_fold)
 *   gencomm_sp_dense_fwd    SparseConvTensor.dense(): out [B][C][D][H][W] (zeroed here)
 *   gencomm_mean_vfe_fwd    MeanVFE: out[j] = sum over all point slots of voxel perm[j] / max(num_points, 1)
 * -------------------------------------------------------------------------------------------- */
int gencomm_sp_out_dims(const int* in_dims3, const int* kernel3, const int* stride3, const int* pad3, int* out_dims3);
long long gencomm_sp_index_workspace_bytes(int n);
int gencomm_sp_index_fwd(const int* coords_bzyx, int n, int B, const int* dims3, long long* keys, int* perm,
                         void* workspace, long long workspace_bytes, void* stream);
long long gencomm_sp_sites_capacity(int n_in, const int* kernel3, const int* stride3);   /* entries out_keys must hold */
long long gencomm_sp_sites_workspace_bytes(int n_in, const int* kernel3, const int* stride3);
int gencomm_sp_sites_fwd(const long long* in_keys, int n_in, int B, const int* in_dims3, const int* kernel3, const int* stride3,
                         const int* pad3, long long* out_keys, int* n_out, void* workspace, long long workspace_bytes, void* stream);
int gencomm_sp_rules_fwd(const long long* out_keys, int n_out, const long long* in_keys, int n_in, int B, const int* in_dims3,
                         const int* kernel3, const int* stride3, const int* pad3, int* nbr, void* stream);
long long gencomm_sp_prepared_floats(int K, int Cin, int Cout);
int gencomm_sp_prepare(const float* w, float* prepared, int K, int Cin, int Cout, int layout, void* stream);
int gencomm_sp_conv_fwd(const float* x, const int* nbr, const float* prepared, const float* scale, const float* shift, float* y,
                        int n_out, int K, int Cin, int Cout, int relu, void* stream);
int gencomm_sp_dense_fwd(const float* feat, const long long* keys, int n, int C, int B, const int* dims3, float* out, void* stream);
int gencomm_mean_vfe_fwd(const float* voxels, const int* num_points, const int* perm, float* out, int n, int max_points, int nfeat,
                         void* stream);
/* Training of the sparse layers (stage 1 trains the SECOND encoder): BatchNorm1d over the active rows [n][C] with batch statistics
 * (+ ReLU), forward and backward (as gencomm_bn2d_train_*; C in {16, 32, 64, 128}); gencomm_sp_rules_inv_fwd = inverse rulebook of
 * a strided layer (inv [K][n_in]: the output site that reads input site i through offset o); the input gradient is gencomm_sp_conv_fwd
 * on dy with weights prepared in layout 2 (channels swapped) + the inverse rulebook, or layout 3 (channels swapped, offsets mirrored)
 * + the forward rulebook for SubM layers; gencomm_sp_wgrad accumulates dW in the raw layout 0 [Cout][K][Cin] (<= 64 channels). */
int gencomm_bnrow_train_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* y, float* save,
                            double* scratch, float momentum, float eps, int relu, int n, int C, void* stream);
int gencomm_bnrow_train_bwd(const float* x, const float* y, const float* dy, const float* save, const float* gamma, float* dx, float* dgamma,
                            float* dbeta, double* scratch, int relu, int n, int C, void* stream);
int gencomm_sp_rules_inv_fwd(const long long* in_keys, int n_in, const long long* out_keys, int n_out, int B, const int* in_dims3,
                             const int* kernel3, const int* stride3, const int* pad3, int* inv, void* stream);
int gencomm_sp_wgrad(const float* x, const float* dy, const int* nbr, float* dw, int n_out, int K, int Cin, int Cout, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GENCOMM_HIP_H */
