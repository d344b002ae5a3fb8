// a10 — tanh-Gaussian action sampler: sample / log-prob / entropy-MC regulariser
// forward and backward (reference: nnx_ppo/networks/sampling_layers.py:82-147).
//
// Elementwise over [B, 2A] rows, fp32 only (the ratio exp(ll_new - ll_old) and
// the tanh-Jacobian term are cancellation-prone).  One thread per row: rows are
// 2A contiguous floats, so a wave reads one contiguous span.  Noise comes from
// Philox (philox.h) keyed by a device-resident {seed, offset} pair so that a
// captured HIP graph draws fresh noise on every replay; the backward kernel
// regenerates the entropy noise from the same counter instead of storing it.
#include "sampler_math.h"

namespace {

using namespace mippo_sampler;

constexpr int kThreads = 256;

__global__ void __launch_bounds__(kThreads)
sampler_fwd_kernel(const float* __restrict__ ms, FwdParams p, int64_t B) {
  const int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (b >= B) return;
  fwd_row(ms + b * 2 * p.A, b, p);
}

__global__ void __launch_bounds__(kThreads)
sampler_bwd_kernel(BwdParams p, float* __restrict__ g_ms, int64_t B) {
  const int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (b >= B) return;
  float* grow = g_ms + b * 2 * p.A;
  bwd_row(b, p, [grow](int j, float v) { grow[j] = v; });
}

__global__ void philox_normal_kernel(const uint64_t* rng, uint64_t offset_add,
                                     float* eps, float* eps2, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  float a, b;
  mippo::philox_normal_pair(rng[0], rng[1] + offset_add, (uint64_t)i, a, b);
  if (eps) eps[i] = a;
  if (eps2) eps2[i] = b;
}

__global__ void rng_advance_kernel(uint64_t* rng, uint64_t n) {
  if (threadIdx.x == 0 && blockIdx.x == 0) rng[1] += n;
}

}  // namespace

extern "C" int mi_tanh_gauss_fwd_f32(const float* mean_and_std, const float* extras,
                                     const uint64_t* rng_state, uint64_t offset_add,
                                     const float* eps, const float* eps2, float* raw_out,
                                     float* action, float* loglik, float* reg,
                                     float* mu_out, float* sigma_out, int64_t B,
                                     int64_t A, float min_std, float std_scale,
                                     float entropy_weight, int deterministic,
                                     mi_stream_t stream) {
  MI_REQUIRE(B >= 0 && A >= 1 && A <= (1 << 20), "mi_tanh_gauss_fwd_f32: bad shape");
  if (B == 0) return 0;
  MI_REQUIRE(mean_and_std, "mi_tanh_gauss_fwd_f32: null mean_and_std");
  MI_REQUIRE(rng_state || (eps && eps2),
             "mi_tanh_gauss_fwd_f32: need rng_state or both injected noises");
  FwdParams p = {extras, {rng_state, offset_add, eps, eps2}, raw_out, action, mu_out, sigma_out,
                 loglik, reg, (int)A, min_std, std_scale, entropy_weight, deterministic};
  hipLaunchKernelGGL(sampler_fwd_kernel, dim3((unsigned)mippo::ceil_div(B, kThreads)),
                     dim3(kThreads), 0, mippo::as_stream(stream), mean_and_std, p, B);
  return mippo::check_launch("mi_tanh_gauss_fwd_f32");
}

extern "C" int mi_tanh_gauss_bwd_f32(const float* mean_and_std, const float* extras,
                                     const uint64_t* rng_state, uint64_t offset_add,
                                     const float* eps2, const float* g_loglik, float g_reg,
                                     float* g_mean_and_std, int64_t B, int64_t A,
                                     float min_std, float std_scale, float entropy_weight,
                                     mi_stream_t stream) {
  MI_REQUIRE(B >= 0 && A >= 1 && A <= (1 << 20), "mi_tanh_gauss_bwd_f32: bad shape");
  if (B == 0) return 0;
  MI_REQUIRE(mean_and_std && extras && g_mean_and_std, "mi_tanh_gauss_bwd_f32: null pointer");
  MI_REQUIRE(rng_state || eps2, "mi_tanh_gauss_bwd_f32: need rng_state or injected eps2");
  // the action noise is irrelevant in replay (z is given); alias it to eps2 so
  // Noise::get() never touches Philox when the entropy noise is injected.
  BwdParams p = {mean_and_std, extras, {rng_state, offset_add, eps2, eps2}, g_loglik, g_reg,
                 (int)A, min_std, std_scale, entropy_weight};
  hipLaunchKernelGGL(sampler_bwd_kernel, dim3((unsigned)mippo::ceil_div(B, kThreads)),
                     dim3(kThreads), 0, mippo::as_stream(stream), p, g_mean_and_std, B);
  return mippo::check_launch("mi_tanh_gauss_bwd_f32");
}

extern "C" int mi_philox_normal_f32(const uint64_t* rng_state, uint64_t offset_add,
                                    float* eps, float* eps2, int64_t n,
                                    mi_stream_t stream) {
  MI_REQUIRE(n >= 0 && rng_state, "mi_philox_normal_f32: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(philox_normal_kernel, dim3((unsigned)mippo::ceil_div(n, kThreads)),
                     dim3(kThreads), 0, mippo::as_stream(stream), rng_state, offset_add, eps,
                     eps2, n);
  return mippo::check_launch("mi_philox_normal_f32");
}

extern "C" int mi_rng_advance(uint64_t* rng_state, uint64_t n, mi_stream_t stream) {
  MI_REQUIRE(rng_state, "mi_rng_advance: null rng_state");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, mippo::as_stream(stream),
                     rng_state, n);
  return mippo::check_launch("mi_rng_advance");
}
