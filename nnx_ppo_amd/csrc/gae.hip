// a13 — GAE reverse scan (reference: nnx_ppo/algorithms/ppo.py:351-394).
//
// HBM-bound: 14 B per (t, env) element (+4 B with targets).  One thread per env
// walks time backwards; rows [t, :] are contiguous so every wave access is a
// coalesced 256 B (f32) / 64 B (u8) segment.  The loads of a chunk of CH rows
// do not depend on the recurrence, so they are all issued before the scan of
// that chunk starts (memory-level parallelism instead of T dependent round
// trips).  fp contraction is off so the result is bit-identical to a plain
// fp32 evaluation of the reference expression order.
#include "common.h"

namespace {

template <int CH>
__global__ void __launch_bounds__(64)
gae_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
           const float* __restrict__ last_value, const uint8_t* __restrict__ done,
           const uint8_t* __restrict__ trunc, float* __restrict__ adv,
           float* __restrict__ targets, int64_t T, int64_t N, float gamma,
           float lambda) {
#pragma clang fp contract(off)
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float next_v = last_value[n];
  float next_a = 0.0f;
  for (int64_t t_hi = T; t_hi > 0; t_hi -= CH) {
    float r[CH], v[CH];
    uint8_t d[CH], tr[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int64_t t = t_hi - 1 - i;
      if (t >= 0) {
        const int64_t o = t * N + n;
        r[i] = rewards[o];
        v[i] = values[o];
        d[i] = done[o];
        tr[i] = trunc[o];
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int64_t t = t_hi - 1 - i;
      if (t >= 0) {
        const float nv = d[i] ? 0.0f : next_v;
        float delta = (r[i] + gamma * nv) - v[i];
        delta = tr[i] ? 0.0f : delta;
        const float keep = d[i] ? 0.0f : 1.0f;
        const float a = delta + ((keep * gamma) * lambda) * next_a;
        const int64_t o = t * N + n;
        adv[o] = a;
        if (targets) targets[o] = v[i] + a;
        next_a = a;
        next_v = v[i];
      }
    }
  }
}

}  // namespace

extern "C" int mi_gae_f32(const float* rewards, const float* values,
                          const float* last_value, const uint8_t* done,
                          const uint8_t* truncated, float* advantages,
                          float* targets, int64_t T, int64_t N, float gamma,
                          float lambda, mi_stream_t stream) {
  MI_REQUIRE(T >= 0 && N >= 0, "mi_gae_f32: negative shape T=%lld N=%lld",
             (long long)T, (long long)N);
  if (T == 0 || N == 0) return 0;
  MI_REQUIRE(rewards && values && last_value && done && truncated && advantages,
             "mi_gae_f32: null pointer");
  const int block = 64;
  const int64_t grid = mippo::ceil_div(N, block);
  MI_REQUIRE(grid <= 0x7fffffffLL, "mi_gae_f32: N=%lld too large", (long long)N);
  hipLaunchKernelGGL(gae_kernel<8>, dim3((unsigned)grid), dim3(block), 0,
                     mippo::as_stream(stream), rewards, values, last_value, done,
                     truncated, advantages, targets, T, N, gamma, lambda);
  return mippo::check_launch("mi_gae_f32");
}
