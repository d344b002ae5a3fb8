// a13 + a14 in one launch — the GAE reverse scan, the advantage statistics and the PPO
// loss terms with their gradients (reference: nnx_ppo/algorithms/ppo.py:351-394 `gae`,
// 447-458 advantages / targets, 477-480 normalisation, 482-503 loss terms).
//
// At minibatch size ([T, mb] = [30, 1024]) the two launches this replaces are each bound
// by launch latency and one memory round trip (gae_kernel<32, true> 8.5 us for 0.43 MB,
// ppo_loss_kernel 8.2 us): the advantages a thread has just computed are exactly the ones
// its loss terms need, so they stay in registers, every operand of both phases is
// requested up front (ONE round trip), and the only thing between the phases is the
// (sum, sum of squares) exchange of the mb / 64 workgroups — a counter and a bounded spin
// (the grid is a handful of one-wave workgroups: always co-resident).
//
// Arithmetic is that of gae.hip and loss.hip, expression for expression: the advantages,
// the statistics triple and both gradients are bit-identical to the two-launch path; the
// four loss scalars differ from it only in fp64 summation order.
#include "common.h"

namespace {

constexpr int kMaxT = 32;
constexpr int kMaxBlocks = 256;          // N <= 16384
constexpr int kHeaderBytes = 64;         // [arrive counter | finish ticket | pad]
constexpr unsigned long long kSpinTicks = 200000000ull;  // 2 s of the 100 MHz wall clock

struct Args {
  const float *rewards, *values, *last_value;
  const uint8_t *done, *trunc;
  const float *ll_new, *ll_old, *reg;  // reg nullable
  float *adv_out;                      // nullable
  float *g_ll, *g_v, *loss_out;
  double* stats_out;                   // nullable: the (sum, sum sq, count) triple
  void* ws;
  int64_t T, N;
  float gamma, lambda, clip, critic_weight;
  int normalize;
};

__global__ void __launch_bounds__(64)
gae_loss_kernel(Args a) {
#pragma clang fp contract(off)
  const int64_t N = a.N, T = a.T;
  const int64_t n = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = n < N;
  unsigned int* arrive = static_cast<unsigned int*>(a.ws);
  unsigned int* ticket = arrive + 1;
  double* sp = reinterpret_cast<double*>(static_cast<char*>(a.ws) + kHeaderBytes);  // [G][2]
  double* lp = sp + 2 * kMaxBlocks;                                                 // [G][4]
  const int G = (int)gridDim.x;

  // ---- every operand of both phases, requested before anything is used -------------
  float r[kMaxT], v[kMaxT], lln[kMaxT], llo[kMaxT], rg[kMaxT];
  uint8_t d[kMaxT], tr[kMaxT];
#pragma unroll
  for (int t = 0; t < kMaxT; ++t) {
    r[t] = v[t] = lln[t] = llo[t] = rg[t] = 0.0f;
    d[t] = tr[t] = 0;
    if (live && t < T) {
      const int64_t o = (int64_t)t * N + n;
      r[t] = a.rewards[o];
      v[t] = a.values[o];
      d[t] = a.done[o];
      tr[t] = a.trunc[o];
      lln[t] = a.ll_new[o];
      llo[t] = a.ll_old[o];
      if (a.reg) rg[t] = a.reg[o];
    }
  }
  // ---- phase 1: the reverse scan (gae.hip, same expression order) -------------------
  float adv[kMaxT];
  float next_v = live ? a.last_value[n] : 0.0f;
  float next_a = 0.0f;
  double s = 0.0, s2 = 0.0;
#pragma unroll
  for (int i = 0; i < kMaxT; ++i) {
    const int t = kMaxT - 1 - i;
    adv[t] = 0.0f;
    if (live && t < T) {
      const float nv = d[t] ? 0.0f : next_v;
      float delta = (r[t] + a.gamma * nv) - v[t];
      delta = tr[t] ? 0.0f : delta;
      const float keep = d[t] ? 0.0f : 1.0f;
      const float av = delta + ((keep * a.gamma) * a.lambda) * next_a;
      adv[t] = av;
      if (a.adv_out) a.adv_out[(int64_t)t * N + n] = av;
      s += (double)av;
      s2 += (double)av * (double)av;
      next_a = av;
      next_v = v[t];
    }
  }
  // ---- the statistics of the whole [T, N] block (ppo.py:477-480) ----------------------
  float mean = 0.0f, denom = 1.0f;
  if (a.normalize) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s += __shfl_down(s, off, 64);
      s2 += __shfl_down(s2, off, 64);
    }
    if (threadIdx.x == 0) {
      // write-through (sc1) stores + a drained wave + the arrival: no L2 write-back
      // fence on the critical path (MI355X_MICROARCH: handoff-flag, drained sc1 form)
      __hip_atomic_store(&sp[2 * blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sp[2 * blockIdx.x + 1], s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > kSpinTicks) break;  // never in a healthy launch
      }
    }
    __syncthreads();
    // acquire only (invalidate; nothing of this workgroup needs writing back here): the
    // other workgroups' partials may sit stale in this XCD's L2 from the previous launch
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    // the summation tree of gae_kernel<.., true>'s last block, evaluated by every
    // workgroup: lane-strided, then lane order — the same bits everywhere
    double t1 = 0.0, t2 = 0.0;
    for (int g = threadIdx.x; g < G; g += 64) {
      t1 += __hip_atomic_load(&sp[2 * g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t2 += __hip_atomic_load(&sp[2 * g + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      t1 += __shfl_down(t1, off, 64);
      t2 += __shfl_down(t2, off, 64);
    }
    t1 = __shfl(t1, 0, 64);
    t2 = __shfl(t2, 0, 64);
    const double cnt = (double)T * (double)N;
    if (a.stats_out && blockIdx.x == 0 && threadIdx.x == 0) {
      a.stats_out[0] = t1;
      a.stats_out[1] = t2;
      a.stats_out[2] = cnt;
    }
    // loss.hip: (a - a.mean()) / (a.std() + 1e-8), population std
    const double m = t1 / cnt;
    double var = t2 / cnt - m * m;
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    denom = (float)sqrt(var) + 1e-8f;
  }
  // ---- phase 2: loss terms and gradients (loss.hip, same expressions) ------------------
  const float inv_n = 1.0f / (float)(T * N);
  const float lo = 1.0f - a.clip, hi = 1.0f + a.clip;
  double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < kMaxT; ++t) {
    if (live && t < T) {
      const int64_t o = (int64_t)t * N + n;
      const float a_raw = adv[t];
      const float an = a.normalize ? (a_raw - mean) / denom : a_raw;
      const float rt = expf(lln[t] - llo[t]);
      const float c1 = rt * an;
      const float c2 = fminf(fmaxf(rt, lo), hi) * an;
      q[0] += (double)fminf(c1, c2);
      a.g_ll[o] = c1 <= c2 ? -(an * rt) * inv_n : 0.0f;
      q[3] += fabsf(rt - 1.0f) > a.clip ? 1.0 : 0.0;
      const float target = v[t] + a_raw;  // ppo.py:456-458
      const float diff = v[t] - target;
      q[1] += (double)(diff * diff);
      a.g_v[o] = a.critic_weight * diff * inv_n;
      if (a.reg) q[2] += (double)rg[t];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] += __shfl_down(q[k], off, 64);
  __shared__ bool is_last;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      __hip_atomic_store(&lp[4 * blockIdx.x + k], q[k], __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
              (unsigned)G - 1;
  }
  __syncthreads();
  if (is_last) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double z[4] = {0.0, 0.0, 0.0, 0.0};
    for (int g = threadIdx.x; g < G; g += 64)
      for (int k = 0; k < 4; ++k)
        z[k] += __hip_atomic_load(&lp[4 * g + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      for (int k = 0; k < 4; ++k) z[k] += __shfl_down(z[k], off, 64);
    if (threadIdx.x == 0) {
      const double dn = (double)T * (double)N;
      a.loss_out[0] = (float)(-z[0] / dn);
      a.loss_out[1] = (float)(0.5 * z[1] / dn);
      a.loss_out[2] = (float)(z[2] / dn);
      a.loss_out[3] = (float)(z[3] / dn);
      // every workgroup has passed the arrival spin (it finished): re-arm both counters
      __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace

extern "C" int64_t mi_gae_ppo_loss_workspace_bytes(void) {
  return kHeaderBytes + (int64_t)kMaxBlocks * 6 * (int64_t)sizeof(double);
}

extern "C" int mi_gae_ppo_loss_supported(int64_t T, int64_t N) {
  return T >= 1 && T <= kMaxT && N >= 1 && N <= 64 * (int64_t)kMaxBlocks;
}

extern "C" int mi_gae_ppo_loss_f32(const float* rewards, const float* values,
                                   const float* last_value, const uint8_t* done,
                                   const uint8_t* truncated, const float* ll_new,
                                   const float* ll_old, const float* reg, float gamma,
                                   float lambda, int normalize, float clip_range,
                                   float critic_weight, float* advantages, double* adv_stats,
                                   float* g_ll, float* g_v, float* loss_out, void* workspace,
                                   int64_t T, int64_t N, mi_stream_t stream) {
  MI_REQUIRE(mi_gae_ppo_loss_supported(T, N),
             "mi_gae_ppo_loss_f32: 1 <= T <= %d and 1 <= N <= %d (got T=%lld N=%lld)", kMaxT,
             64 * kMaxBlocks, (long long)T, (long long)N);
  MI_REQUIRE(rewards && values && last_value && done && truncated && ll_new && ll_old && g_ll &&
                 g_v && loss_out && workspace,
             "mi_gae_ppo_loss_f32: null pointer");
  Args a = {rewards, values,  last_value, done,      truncated, ll_new, ll_old,     reg,
            advantages, g_ll, g_v,        loss_out,  adv_stats, workspace, T,       N,
            gamma,   lambda,  clip_range, critic_weight, normalize};
  hipLaunchKernelGGL(gae_loss_kernel, dim3((unsigned)mippo::ceil_div(N, 64)), dim3(64), 0,
                     mippo::as_stream(stream), a);
  return mippo::check_launch("mi_gae_ppo_loss_f32");
}
