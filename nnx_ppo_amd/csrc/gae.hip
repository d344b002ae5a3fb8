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

// STATS: also reduce (sum A, sum A^2) over the whole [T, N] result — the
// advantage normalisation of the loss (ppo.py:477-480) needs them next, and the
// advantages are in registers here.  fp64 partial per wave, summed in wave order by
// the last wave to finish (fixed order: bitwise reproducible).
// ws: [0] ticket counter (16 B), then partials[G][2].
template <int CH, bool STATS>
__global__ void __launch_bounds__(64)
gae_kernel(const float* __restrict__ rewards, const float* __restrict__ values,
           const float* __restrict__ last_value, const uint8_t* __restrict__ done,
           const uint8_t* __restrict__ trunc, float* __restrict__ adv,
           float* __restrict__ targets, int64_t T, int64_t N, float gamma,
           float lambda, double* __restrict__ stats, void* __restrict__ ws) {
#pragma clang fp contract(off)
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = n < N;
  if (!STATS && !live) return;
  float next_v = live ? last_value[n] : 0.0f;
  float next_a = 0.0f;
  double s = 0.0, s2 = 0.0;
  for (int64_t t_hi = live ? T : 0; t_hi > 0; t_hi -= CH) {
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
        if (STATS) {
          s += (double)a;
          s2 += (double)a * (double)a;
        }
        next_a = a;
        next_v = v[i];
      }
    }
  }
  if constexpr (STATS) {
    unsigned int* counter = static_cast<unsigned int*>(ws);
    double* partials = reinterpret_cast<double*>(static_cast<char*>(ws) + 16);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s += __shfl_down(s, off, 64);
      s2 += __shfl_down(s2, off, 64);
    }
    if (threadIdx.x == 0) {
      partials[2 * blockIdx.x] = s;
      partials[2 * blockIdx.x + 1] = s2;
    }
    if (mippo::last_block_ticket(counter)) {
      __threadfence();  // acquire
      const int G = (int)gridDim.x;
      double t = 0.0, t2 = 0.0;
      for (int g = threadIdx.x; g < G; g += 64) {  // lane-strided, then lane order
        t += partials[2 * g];
        t2 += partials[2 * g + 1];
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        t += __shfl_down(t, off, 64);
        t2 += __shfl_down(t2, off, 64);
      }
      if (threadIdx.x == 0) {
        stats[0] = t;
        stats[1] = t2;
        stats[2] = (double)T * (double)N;
        *counter = 0;
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
  hipLaunchKernelGGL((gae_kernel<8, false>), dim3((unsigned)grid), dim3(block), 0,
                     mippo::as_stream(stream), rewards, values, last_value, done,
                     truncated, advantages, targets, T, N, gamma, lambda,
                     (double*)nullptr, (void*)nullptr);
  return mippo::check_launch("mi_gae_f32");
}

extern "C" int64_t mi_gae_stats_workspace_bytes(int64_t N) {
  if (N < 0) return -EINVAL;
  return 16 + mippo::ceil_div(N, 64) * 2 * (int64_t)sizeof(double);
}

extern "C" int mi_gae_stats_f32(const float* rewards, const float* values,
                                const float* last_value, const uint8_t* done,
                                const uint8_t* truncated, float* advantages,
                                float* targets, int64_t T, int64_t N, float gamma,
                                float lambda, double* adv_stats, void* workspace,
                                mi_stream_t stream) {
  MI_REQUIRE(T >= 1 && N >= 1, "mi_gae_stats_f32: empty shape T=%lld N=%lld",
             (long long)T, (long long)N);
  MI_REQUIRE(rewards && values && last_value && done && truncated && advantages &&
                 adv_stats && workspace,
             "mi_gae_stats_f32: null pointer");
  const int block = 64;
  const int64_t grid = mippo::ceil_div(N, block);
  MI_REQUIRE(grid <= 0x7fffffffLL, "mi_gae_stats_f32: N=%lld too large", (long long)N);
  if (N <= 16384 && T <= 32) {
    // latency-bound size (a minibatch: 16 workgroups): request every row of the scan up
    // front — one memory round trip instead of T/8 — at 32 rows of registers per thread
    hipLaunchKernelGGL((gae_kernel<32, true>), dim3((unsigned)grid), dim3(block), 0,
                       mippo::as_stream(stream), rewards, values, last_value, done,
                       truncated, advantages, targets, T, N, gamma, lambda, adv_stats,
                       workspace);
  } else {
    hipLaunchKernelGGL((gae_kernel<8, true>), dim3((unsigned)grid), dim3(block), 0,
                       mippo::as_stream(stream), rewards, values, last_value, done,
                       truncated, advantages, targets, T, N, gamma, lambda, adv_stats,
                       workspace);
  }
  return mippo::check_launch("mi_gae_stats_f32");
}
