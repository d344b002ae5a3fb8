// a13 + a14 in one launch — the GAE reverse scan, the advantage statistics and the PPO
// loss terms with their gradients (reference: nnx_ppo/algorithms/ppo.py:351-394 `gae`,
// 447-458 advantages / targets, 477-480 normalisation, 482-503 loss terms).
//
// At minibatch size ([T, mb] = [30, 1024]) the two launches this replaces are each bound
// by launch latency and one memory round trip (gae_kernel<32, true> 8.5 us for 0.43 MB,
// ppo_loss_kernel 8.2 us): here every operand of both phases is requested up front (ONE
// round trip), the advantages go from the scan to the loss terms through LDS, and the only
// thing between the phases is the (sum, sum of squares) exchange of the mb / 64
// workgroups — a counter and a bounded spin (a handful of workgroups: always co-resident).
// (A first form — one wave per 64 envs doing its 30 loss terms serially — took 22 us
// against 15 us for the two launches: the loss phase wants all T x 64 elements in flight.)
//
// Arithmetic is that of gae.hip and loss.hip, expression for expression: the advantages,
// the statistics triple and both gradients are bit-identical to the two-launch path; the
// four loss scalars differ from it only in fp64 summation order.
#include "common.h"

namespace {

constexpr int kMaxT = 32;
constexpr int kMaxBlocks = 256;          // N <= 16384
constexpr int kHeaderBytes = 64;         // [arrive | finish ticket | sticky timeouts | test hook: us, extra | pad]
constexpr unsigned long long kSpinTicks = 200000000ull;  // 2 s of the 100 MHz wall clock

struct Args {
  const float *rewards, *values, *last_value;
  const uint8_t *done, *trunc;
  const float *ll_new, *ll_old, *reg;  // reg nullable
  float *adv_out;                      // nullable
  float *g_ll, *g_v, *loss_out;        // loss_out null: deferred (partials_out)
  double* partials_out;                // nullable: [G][4] loss partials left for a later sum
  double* stats_out;                   // nullable: the (sum, sum sq, count) triple
  void* ws;
  int64_t T, N;
  float gamma, lambda, clip, critic_weight;
  int normalize;
};

constexpr int kThreads = 256;
constexpr int kEnvs = 64;                        // envs per workgroup (one wave scans them)
constexpr int kPer = kMaxT * kEnvs / kThreads;   // elements per thread: 8

// One workgroup = 64 envs x T steps.  All 256 threads load the block's operands (element
// q = tid + 256 k is step q / 64 of env q % 64: every row coalesced) and stage what the scan
// needs in LDS; wave 0 runs the reverse scan, one env per lane, exactly as gae_kernel does
// (same per-lane accumulation order, same shuffle tree => the same statistics partial);
// after the exchange of the partials all 256 threads evaluate the loss terms of their own
// elements, the advantages coming back out of LDS.
__global__ void __launch_bounds__(kThreads)
gae_loss_kernel(Args a) {
#pragma clang fp contract(off)
  const int64_t N = a.N;
  const int T = (int)a.T;
  const int tid = threadIdx.x;
  const int64_t n0 = (int64_t)blockIdx.x * kEnvs;
  unsigned int* arrive = static_cast<unsigned int*>(a.ws);
  unsigned int* ticket = arrive + 1;
  double* sp = reinterpret_cast<double*>(static_cast<char*>(a.ws) + kHeaderBytes);  // [G][2]
  double* lp = a.partials_out ? a.partials_out : sp + 2 * kMaxBlocks;               // [G][4]
  const int G = (int)gridDim.x;
  __shared__ float s_r[kMaxT][kEnvs], s_v[kMaxT][kEnvs], s_adv[kMaxT][kEnvs];
  __shared__ uint8_t s_d[kMaxT][kEnvs], s_tr[kMaxT][kEnvs];
  __shared__ double s_red[4][kThreads / 64];
  __shared__ float s_norm[2];
  __shared__ bool is_last;

  // ---- every operand of both phases, requested before anything is used -------------
  float r[kPer], v[kPer], lln[kPer], llo[kPer], rg[kPer];
  uint8_t d[kPer], tr[kPer];
  bool ok[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int q = tid + k * kThreads;
    const int t = q / kEnvs, e = q % kEnvs;
    ok[k] = t < T && n0 + e < N;
    r[k] = v[k] = lln[k] = llo[k] = rg[k] = 0.0f;
    d[k] = tr[k] = 0;
    if (ok[k]) {
      const int64_t o = (int64_t)t * N + n0 + e;
      r[k] = a.rewards[o];
      v[k] = a.values[o];
      d[k] = a.done[o];
      tr[k] = a.trunc[o];
      lln[k] = a.ll_new[o];
      llo[k] = a.ll_old[o];
      if (a.reg) rg[k] = a.reg[o];
    }
  }
  float lv = 0.0f;
  if (tid < kEnvs && n0 + tid < N) lv = a.last_value[n0 + tid];
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int q = tid + k * kThreads;
    const int t = q / kEnvs, e = q % kEnvs;
    s_r[t][e] = r[k];
    s_v[t][e] = v[k];
    s_d[t][e] = d[k];
    s_tr[t][e] = tr[k];
  }
  __syncthreads();
  // ---- phase 1: the reverse scan, wave 0, one env per lane (gae.hip, same order) -----
  if (tid < kEnvs) {
    const bool live = n0 + tid < N;
    float next_v = lv, next_a = 0.0f;
    double s = 0.0, s2 = 0.0;
    for (int t = T - 1; t >= 0; --t) {
      float av = 0.0f;
      if (live) {
        const float vt = s_v[t][tid];
        const float nv = s_d[t][tid] ? 0.0f : next_v;
        float delta = (s_r[t][tid] + a.gamma * nv) - vt;
        delta = s_tr[t][tid] ? 0.0f : delta;
        const float keep = s_d[t][tid] ? 0.0f : 1.0f;
        av = delta + ((keep * a.gamma) * a.lambda) * next_a;
        s += (double)av;
        s2 += (double)av * (double)av;
        next_a = av;
        next_v = vt;
      }
      s_adv[t][tid] = av;
    }
    if (a.normalize) {
      bool timed_out = false;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        s2 += __shfl_down(s2, off, 64);
      }
      if (tid == 0) {
        // write-through (sc1) stores + a drained wave + the arrival: no L2 write-back
        // fence on the critical path (MI355X_MICROARCH: handoff-flag, drained sc1 form)
        __hip_atomic_store(&sp[2 * blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sp[2 * blockIdx.x + 1], s2, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // header words 3, 4 (zero in production): the test hook of trunk_ws.hip's hand-over
        const unsigned int hook_us = __builtin_nontemporal_load(arrive + 3);
        const unsigned int hook_extra = __builtin_nontemporal_load(arrive + 4);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <
               (unsigned)G + hook_extra) {
          __builtin_amdgcn_s_sleep(1);
          const unsigned long long limit =
              hook_us ? (unsigned long long)hook_us * 100ull : kSpinTicks;
          if (wall_clock64() - t0 > limit) {  // never in a healthy launch
            // sticky word 2 of the header, read with the iteration's metrics
            // (ops.health_words) and by loop.health_check
            __hip_atomic_fetch_add(arrive + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            timed_out = true;
            break;
          }
        }
      }
      // acquire only (invalidate; nothing of this wave needs writing back here): the
      // other workgroups' partials may sit stale in this XCD's L2 from the previous launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      // the summation tree of gae_kernel<.., true>'s last block: lane-strided, then lane
      // order — the same bits in every workgroup
      double t1 = 0.0, t2 = 0.0;
      for (int g = tid; g < G; g += 64) {
        t1 += __hip_atomic_load(&sp[2 * g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t2 += __hip_atomic_load(&sp[2 * g + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        t1 += __shfl_down(t1, off, 64);
        t2 += __shfl_down(t2, off, 64);
      }
      if (tid == 0) {
        const double cnt = (double)T * (double)N;
        if (a.stats_out && blockIdx.x == 0) {
          a.stats_out[0] = t1;
          a.stats_out[1] = t2;
          a.stats_out[2] = cnt;
        }
        // loss.hip: (a - a.mean()) / (a.std() + 1e-8), population std
        const double m = t1 / cnt;
        double var = t2 / cnt - m * m;
        if (var < 0.0) var = 0.0;
        // incomplete partials after a timed-out hand-over: poison the normalisation
        s_norm[0] = (float)m;
        // (a zero denominator, not a NaN mean: the surrogate's `c1 <= c2` select would turn a
        // NaN advantage into a ZERO gradient; +-inf advantages go through it and reach the
        // parameters as NaN)
        s_norm[1] = timed_out ? 0.0f : (float)sqrt(var) + 1e-8f;
      }
    }
  }
  __syncthreads();
  const float mean = a.normalize ? s_norm[0] : 0.0f;
  const float denom = a.normalize ? s_norm[1] : 1.0f;
  // ---- phase 2: loss terms and gradients of this thread's elements (loss.hip) ----------
  const float inv_n = 1.0f / (float)((int64_t)T * N);
  const float lo = 1.0f - a.clip, hi = 1.0f + a.clip;
  double q4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int q = tid + k * kThreads;
    const int t = q / kEnvs, e = q % kEnvs;
    if (ok[k]) {
      const int64_t o = (int64_t)t * N + n0 + e;
      const float a_raw = s_adv[t][e];
      if (a.adv_out) a.adv_out[o] = a_raw;
      const float an = a.normalize ? (a_raw - mean) / denom : a_raw;
      const float rt = expf(lln[k] - llo[k]);
      const float c1 = rt * an;
      const float c2 = fminf(fmaxf(rt, lo), hi) * an;
      q4[0] += (double)fminf(c1, c2);
      a.g_ll[o] = c1 <= c2 ? -(an * rt) * inv_n : 0.0f;
      q4[3] += fabsf(rt - 1.0f) > a.clip ? 1.0 : 0.0;
      const float target = v[k] + a_raw;  // ppo.py:456-458
      const float diff = v[k] - target;
      q4[1] += (double)(diff * diff);
      a.g_v[o] = a.critic_weight * diff * inv_n;
      if (a.reg) q4[2] += (double)rg[k];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) q4[k] += __shfl_down(q4[k], off, 64);
  if ((tid & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) s_red[k][tid >> 6] = q4[k];
  }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double z = 0.0;
      for (int w = 0; w < kThreads / 64; ++w) z += s_red[k][w];
      __hip_atomic_store(&lp[4 * blockIdx.x + k], z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
              (unsigned)G - 1;
  }
  __syncthreads();
  if (is_last && !a.loss_out) {
    // deferred (mi_policy_loss_finalize_f32 sums the partials): only re-arm the counters
    if (tid == 0) {
      __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (is_last && tid < 64) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double z[4] = {0.0, 0.0, 0.0, 0.0};
    for (int g = tid; g < G; g += 64)
      for (int k = 0; k < 4; ++k)
        z[k] += __hip_atomic_load(&lp[4 * g + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      for (int k = 0; k < 4; ++k) z[k] += __shfl_down(z[k], off, 64);
    if (tid == 0) {
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
                                   float* g_ll, float* g_v, float* loss_out,
                                   double* partials_out, void* workspace, int64_t T, int64_t N,
                                   mi_stream_t stream) {
  MI_REQUIRE(mi_gae_ppo_loss_supported(T, N),
             "mi_gae_ppo_loss_f32: 1 <= T <= %d and 1 <= N <= %d (got T=%lld N=%lld)", kMaxT,
             64 * kMaxBlocks, (long long)T, (long long)N);
  MI_REQUIRE(rewards && values && last_value && done && truncated && ll_new && ll_old && g_ll &&
                 g_v && workspace,
             "mi_gae_ppo_loss_f32: null pointer");
  MI_REQUIRE((loss_out != nullptr) != (partials_out != nullptr),
             "mi_gae_ppo_loss_f32: exactly one of loss_out and partials_out");
  Args a = {rewards, values,  last_value, done,      truncated, ll_new, ll_old,     reg,
            advantages, g_ll, g_v,        loss_out,  partials_out, adv_stats, workspace, T, N,
            gamma,   lambda,  clip_range, critic_weight, normalize};
  hipLaunchKernelGGL(gae_loss_kernel, dim3((unsigned)mippo::ceil_div(N, kEnvs)), dim3(kThreads),
                     0, mippo::as_stream(stream), a);
  return mippo::check_launch("mi_gae_ppo_loss_f32");
}
