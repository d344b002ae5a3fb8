// a14 — PPO loss terms and their gradients
// (reference: nnx_ppo/algorithms/ppo.py:456-531).
//
//   target = sg(V + A);  a = normalize ? (A - mean) / (std + 1e-8) : A   (whole [T*mb])
//   r = exp(ll_new - ll_old)
//   actor  = -mean(min(r a, clip(r, 1-eps, 1+eps) a))
//   critic = 0.5 mean((V - target)^2)
//   reg    = mean(reg_elem)
//   total  = actor + w critic + reg
// Streaming elementwise + reductions (HBM-bound, ~28 B/element); reductions use
// fp64 accumulators and a fixed summation order, so the scalars are bitwise
// reproducible.  Gradients are w.r.t. ll_new and V; the regulariser's gradient
// is the constant 1/n handled by the sampler backward.
#include "common.h"

namespace {

constexpr int kThreads = 1024;
constexpr int kMaxBlocks = 256;
// Up to this many elements one block does the whole statistics reduction in ONE
// launch (8 us at the workload's 30 720-element minibatch vs two launches).  The
// loss pass (exp, four fp64 reductions) is spread over ceil(n / 1024) blocks and
// the last block to finish sums the per-block partials in block order
// (mippo::last_block_ticket): one launch, fixed summation order.
constexpr int64_t kSingleBlockMax = 65536;
constexpr int kTicketBytes = 16;  // workspace: [ticket counter | partials[G][4]]

__device__ inline double block_sum(double v, double* scratch) {
  // wave reduce (64 lanes) then across the 16 waves of the block
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) {
    for (int w = 0; w < kThreads / 64; ++w) t += scratch[w];
  }
  return t;  // valid on thread 0
}

// partials: [G][2] = (sum a, sum a^2)
__global__ void __launch_bounds__(kThreads)
adv_stats_partial_kernel(const float* __restrict__ adv, int64_t n, double* partials) {
  __shared__ double scratch[kThreads / 64];
  double s = 0.0, s2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kThreads) {
    const double a = adv[i];
    s += a;
    s2 += a * a;
  }
  const double ts = block_sum(s, scratch);
  const double ts2 = block_sum(s2, scratch);
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = ts;
    partials[2 * blockIdx.x + 1] = ts2;
  }
}

// Small-n fast path: one 1024-thread block computes the final triple directly.
__global__ void __launch_bounds__(kThreads)
adv_stats_single_kernel(const float* __restrict__ adv, int64_t n, double* stats) {
  __shared__ double scratch[kThreads / 64];
  double s = 0.0, s2 = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += kThreads) {
    const double a = adv[i];
    s += a;
    s2 += a * a;
  }
  const double ts = block_sum(s, scratch);
  const double ts2 = block_sum(s2, scratch);
  if (threadIdx.x == 0) {
    stats[0] = ts;
    stats[1] = ts2;
    stats[2] = (double)n;
  }
}

// stats: [3] = (sum, sum of squares, count)
__global__ void adv_stats_finalize_kernel(const double* partials, int G, int64_t n,
                                          double* stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0, s2 = 0.0;
  for (int g = 0; g < G; ++g) {
    s += partials[2 * g];
    s2 += partials[2 * g + 1];
  }
  stats[0] = s;
  stats[1] = s2;
  stats[2] = (double)n;
}

// partials: [G][4] = (sum min-term, sum (V-target)^2, sum reg, count |r-1|>eps)
__global__ void __launch_bounds__(kThreads)
ppo_loss_kernel(const float* __restrict__ ll_new, const float* __restrict__ ll_old,
                const float* __restrict__ adv, const float* __restrict__ values,
                const float* __restrict__ reg, const double* __restrict__ stats,
                float clip, float critic_weight, float* __restrict__ g_ll,
                float* __restrict__ g_v, void* __restrict__ ws,
                float* __restrict__ loss_out, int64_t n) {
  unsigned int* counter = static_cast<unsigned int*>(ws);
  double* partials = reinterpret_cast<double*>(static_cast<char*>(ws) + kTicketBytes);
  float mean = 0.0f, denom = 1.0f;
  if (stats) {
    // ppo.py:477-480: (a - a.mean()) / (a.std() + 1e-8), population std
    const double cnt = stats[2];
    const double m = stats[0] / cnt;
    double var = stats[1] / cnt - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    denom = (float)sqrt(var) + 1e-8f;
  }
  const float inv_n = 1.0f / (float)n;
  const float lo = 1.0f - clip, hi = 1.0f + clip;
  double s_act = 0.0, s_crit = 0.0, s_reg = 0.0, s_clip = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kThreads) {
    const float a_raw = adv[i];
    if (ll_new) {  // actor term (absent for a value head without its own policy term)
      const float a = stats ? (a_raw - mean) / denom : a_raw;
      const float r = expf(ll_new[i] - ll_old[i]);
      const float c1 = r * a;
      const float c2 = fminf(fmaxf(r, lo), hi) * a;
      s_act += (double)fminf(c1, c2);
      // d(-mean min(c1,c2))/d ll_new: the unclipped branch carries a*r, the
      // clipped branch is flat (ties c1 == c2 have r inside the clip range).
      g_ll[i] = c1 <= c2 ? -(a * r) * inv_n : 0.0f;
      s_clip += fabsf(r - 1.0f) > clip ? 1.0 : 0.0;
    }
    if (values) {  // critic term (absent when the advantage is a sum over reward keys)
      const float v = values[i];
      const float target = v + a_raw;  // ppo.py:456-458
      const float diff = v - target;
      s_crit += (double)(diff * diff);
      g_v[i] = critic_weight * diff * inv_n;
    }
    if (reg) s_reg += (double)reg[i];
  }
  // the four sums share one pair of barriers (same wave order as `block_sum`)
  __shared__ double scratch4[4][kThreads / 64];
  double q[4] = {s_act, s_crit, s_reg, s_clip};
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] += __shfl_down(q[k], off, 64);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) scratch4[k][threadIdx.x >> 6] = q[k];
  }
  __syncthreads();
  double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
  if (threadIdx.x == 0) {
    for (int w = 0; w < kThreads / 64; ++w) {
      t0 += scratch4[0][w];
      t1 += scratch4[1][w];
      t2 += scratch4[2][w];
      t3 += scratch4[3][w];
    }
  }
  if (threadIdx.x == 0) {
    double* p = partials + 4 * blockIdx.x;
    p[0] = t0;
    p[1] = t1;
    p[2] = t2;
    p[3] = t3;
  }
  // loss_out: [4] = (actor, critic, regularization, clipping_fraction)
  if (mippo::last_block_ticket(counter) && threadIdx.x < 64) {
    __threadfence();  // acquire
    double s[4] = {0, 0, 0, 0};
    for (int g = threadIdx.x; g < (int)gridDim.x; g += 64)  // lane-strided, then lane order
      for (int k = 0; k < 4; ++k) s[k] += partials[4 * g + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      for (int k = 0; k < 4; ++k) s[k] += __shfl_down(s[k], off, 64);
    if (threadIdx.x == 0) {
      const double dn = (double)n;
      loss_out[0] = (float)(-s[0] / dn);
      loss_out[1] = (float)(0.5 * s[1] / dn);
      loss_out[2] = (float)(s[2] / dn);
      loss_out[3] = (float)(s[3] / dn);
      *counter = 0;
    }
  }
}

int grid_for(int64_t n) {
  int64_t g = mippo::ceil_div(n, kThreads);
  if (g > kMaxBlocks) g = kMaxBlocks;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int64_t mi_ppo_loss_workspace_bytes(int64_t n) {
  if (n < 0) return -EINVAL;
  return kTicketBytes + (int64_t)kMaxBlocks * 4 * (int64_t)sizeof(double);
}

extern "C" int mi_adv_stats_f32(const float* adv, int64_t n, double* stats, void* workspace,
                                mi_stream_t stream) {
  MI_REQUIRE(n >= 1, "mi_adv_stats_f32: n must be >= 1");
  MI_REQUIRE(adv && stats && workspace, "mi_adv_stats_f32: null pointer");
  hipStream_t st = mippo::as_stream(stream);
  if (n <= kSingleBlockMax) {
    hipLaunchKernelGGL(adv_stats_single_kernel, dim3(1), dim3(kThreads), 0, st, adv, n, stats);
    return mippo::check_launch("mi_adv_stats_f32(single)");
  }
  const int G = grid_for(n);
  double* partials = reinterpret_cast<double*>(static_cast<char*>(workspace) + kTicketBytes);
  hipLaunchKernelGGL(adv_stats_partial_kernel, dim3(G), dim3(kThreads), 0, st, adv, n, partials);
  int rc = mippo::check_launch("mi_adv_stats_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(adv_stats_finalize_kernel, dim3(1), dim3(64), 0, st, partials, G, n, stats);
  return mippo::check_launch("mi_adv_stats_f32(finalize)");
}

extern "C" int mi_ppo_loss_f32(const float* ll_new, const float* ll_old, const float* adv,
                               const float* values, const float* reg, const double* adv_stats,
                               float clip_range, float critic_weight, float* g_ll, float* g_v,
                               float* loss_out, void* workspace, int64_t n,
                               mi_stream_t stream) {
  MI_REQUIRE(n >= 1, "mi_ppo_loss_f32: n must be >= 1");
  MI_REQUIRE(adv && loss_out && workspace, "mi_ppo_loss_f32: null pointer");
  MI_REQUIRE((ll_new && ll_old && g_ll) || (!ll_new && !ll_old && !g_ll),
             "mi_ppo_loss_f32: ll_new, ll_old, g_ll go together");
  MI_REQUIRE((values && g_v) || (!values && !g_v), "mi_ppo_loss_f32: values, g_v go together");
  MI_REQUIRE(ll_new || values, "mi_ppo_loss_f32: neither an actor nor a critic term");
  hipStream_t st = mippo::as_stream(stream);
  const int G = grid_for(n);
  hipLaunchKernelGGL(ppo_loss_kernel, dim3(G), dim3(kThreads), 0, st, ll_new, ll_old, adv, values,
                     reg, adv_stats, clip_range, critic_weight, g_ll, g_v, workspace, loss_out, n);
  return mippo::check_launch("mi_ppo_loss_f32");
}
