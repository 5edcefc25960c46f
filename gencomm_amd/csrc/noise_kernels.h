// The sampler's in-kernel noise, written out.  gencomm_step_noise_fwd materialises the CANONICAL step-noise field
// nu_t = fp16(sigma_t z) of a (seed, timestep) with the SAME device functions the sampler kernels call (common.h:
// noise_words / bm_pair_h, the counter layout of latent_step_h_kernel<2>, latent_step_kernel<..,2>,
// conv_out_h_kernel<2> and conv_out_kernel<..,2>), so that a run with in-kernel noise can be replayed through the CPU
// oracle with explicit noise tensors (tests/test_gpu_philox_replay.py): the reference draws a fresh torch.randn per
// step (cond_diff.py:307, MDD_utils.py:232-235), any i.i.d. N(0,1) field is an instance of it, and this is the
// instance the kernels used.  Not on the product path; q_sample's initial noise is read back through
// gencomm_q_sample_fwd itself (zero x_start, sched row {0, 1}).
#pragma once
#include "common.h"

namespace gc {

struct StepNoiseArgs {
  float* out;          // [n][C][H][W]
  const float* sched;  // device [5] row of the timestep: [4] = sigma_t
  unsigned long long seed;
  unsigned int stream_id;  // the timestep t
  int C, H, W;
  int unrounded;  // 1: sigma_t z as fp32 BEFORE the fp16 rounding (statistics tests); 0: the field the kernels add
  long long items;  // n * (C / 2) * H * ceil(W / 4)
};

// one thread per (agent, channel pair, row, aligned pixel quad) = one Philox counter
__global__ __launch_bounds__(256) void step_noise_kernel(const StepNoiseArgs a) {
  const int W4 = (a.W + 3) >> 2;
  const float sg = a.sched[4];
  const float k2 = bm_k2(sg);
  const size_t plane = (size_t)a.H * a.W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.items; i += (long long)gridDim.x * 256) {
    const int q = (int)(i % W4);
    long long r = i / W4;
    const int y = (int)(r % a.H);
    r /= a.H;
    const int cp = (int)(r % (a.C / 2)), n = (int)(r / (a.C / 2));
    const size_t e = ((size_t)n * a.C + 2 * cp) * plane + (size_t)y * a.W + 4 * q;
    float z[8];
    if (a.unrounded) {
      NoiseWords w;
      noise_words((uint64_t)e, a.stream_id, a.seed, w);
#pragma unroll
      for (int j = 0; j < 4; ++j) bm_pair(w.ur[j], w.w[j], k2, z[j], z[4 + j]);
    } else {
      noise_pair_quad((uint64_t)e, a.stream_id, a.seed, k2, z);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (4 * q + j < a.W) {
        a.out[e + j] = z[j];
        a.out[e + plane + j] = z[4 + j];
      }
    }
  }
}

}  // namespace gc
