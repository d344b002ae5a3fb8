// a8 / a16 — running observation normaliser
// (reference: nnx_ppo/networks/normalizer.py:63-96 forward, 98-136 update).
//
// Forward is a streaming elementwise op (8 B/element, HBM-bound).  The
// statistics update is a column-wise reduction of the rollout's [T*N, F] raw
// observations: one pass over the data with per-thread Welford accumulators,
// Chan-merged in a fixed order (bitwise reproducible run to run), instead of
// the reference's two passes (mean, then sum of squared deviations).
#include "common.h"

namespace {

constexpr int kThreads = 256;

__device__ inline float norm_std(float m2, float counter, float eps) {
  // normalizer.py:76-81,92-96: counter>0 ? sqrt(max(M2/counter, eps)) : 10
  return counter > 0.0f ? sqrtf(fmaxf(m2 / counter, eps)) : 10.0f;
}

// (x_tail: rows M .. M + M_tail - 1 of `out` come from a second source — the bootstrap
// observation behind the T x B rows of a replay, one launch instead of two)
__global__ void __launch_bounds__(kThreads)
normalize_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                     const float* __restrict__ m2, const float* __restrict__ counter,
                     float eps, float* __restrict__ out, int64_t M, int64_t F,
                     const float* __restrict__ x_tail, int64_t M_tail) {
  extern __shared__ float lds[];  // mean[Fc], std[Fc] when F fits
  const float cnt = *counter;
  const bool cached = F <= 2048;
  if (cached) {
    for (int64_t f = threadIdx.x; f < F; f += kThreads) {
      lds[f] = mean[f];
      lds[F + f] = norm_std(m2[f], cnt, eps);
    }
    __syncthreads();
  }
  const int64_t head = M * F, total = (M + M_tail) * F;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int64_t f = i % F;
    const float mu = cached ? lds[f] : mean[f];
    const float sd = cached ? lds[F + f] : norm_std(m2[f], cnt, eps);
    const float v = i < head ? x[i] : x_tail[i - head];
    out[i] = (v - mu) / sd;
  }
}

// d/dx of the forward: g_x = g_out / std (statistics are constants).
__global__ void __launch_bounds__(kThreads)
normalize_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ m2,
                     const float* __restrict__ counter, float eps,
                     float* __restrict__ g_x, int64_t M, int64_t F) {
  const float cnt = *counter;
  const int64_t total = M * F;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    g_x[i] = g_out[i] / norm_std(m2[i % F], cnt, eps);
  }
}

struct Wf {
  float n, mean, m2;
};

__device__ inline void wf_push(Wf& s, float x) {
  s.n += 1.0f;
  const float d = x - s.mean;
  s.mean += d / s.n;
  s.m2 += d * (x - s.mean);
}

// Chan et al. pairwise merge; b into a.
__device__ inline void wf_merge(Wf& a, const Wf& b) {
  if (b.n == 0.0f) return;
  if (a.n == 0.0f) {
    a = b;
    return;
  }
  const float n = a.n + b.n;
  const float d = b.mean - a.mean;
  a.mean += d * (b.n / n);
  a.m2 += b.m2 + d * d * (a.n * b.n / n);
  a.n = n;
}

// grid = (G row chunks, column chunks of width CW = min(F, 256)); R = 256 / CW
// row lanes per block.  partial layout: [G][3][F] (n, mean, M2).
__global__ void __launch_bounds__(kThreads)
welford_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                       int64_t M, int64_t F, int CW, int R) {
  __shared__ Wf red[kThreads];
  const int tid = threadIdx.x;
  const int rl = tid / CW;
  const int64_t c = (int64_t)blockIdx.y * CW + (tid % CW);
  const bool active = rl < R && c < F;
  const int64_t G = gridDim.x;
  const int64_t rows_per = mippo::ceil_div(M, G);
  const int64_t r0 = (int64_t)blockIdx.x * rows_per;
  const int64_t r1 = r0 + rows_per < M ? r0 + rows_per : M;
  Wf s = {0.0f, 0.0f, 0.0f};
  if (active) {
    for (int64_t r = r0 + rl; r < r1; r += R) wf_push(s, x[r * F + c]);
  }
  red[tid] = s;
  __syncthreads();
  if (active && rl == 0) {
    for (int j = 1; j < R; ++j) wf_merge(s, red[j * CW + (tid % CW)]);
    float* p = partial + (int64_t)blockIdx.x * 3 * F;
    p[c] = s.n;
    p[F + c] = s.mean;
    p[2 * F + c] = s.m2;
  }
}

// One WAVE per column: lane l merges partials l, l+64, ... in order, then the 64
// lane results are merged by a fixed butterfly (xor 32, 16, .. 1) — a fixed
// order, so the statistics are bitwise reproducible run to run.
// batch layout: [3][F] (n, mean, M2) — n is identical for every column.
__global__ void __launch_bounds__(kThreads)
welford_finalize_kernel(const float* __restrict__ partial, float* __restrict__ batch,
                        int64_t G, int64_t F) {
  const int lane = threadIdx.x & 63;
  const int64_t c = (int64_t)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (c >= F) return;
  Wf s = {0.0f, 0.0f, 0.0f};
  for (int64_t g = lane; g < G; g += 64) {
    const float* p = partial + g * 3 * F;
    Wf b = {p[c], p[F + c], p[2 * F + c]};
    wf_merge(s, b);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Wf o = {__shfl_xor(s.n, off, 64), __shfl_xor(s.mean, off, 64), __shfl_xor(s.m2, off, 64)};
    // lower lane keeps (self, other) order, upper lane (other, self): both halves of a pair
    // compute the same merged value in the same operand order
    if (lane & off) {
      Wf t = o;
      wf_merge(t, s);
      s = t;
    } else {
      wf_merge(s, o);
    }
  }
  if (lane == 0) {
    batch[c] = s.n;
    batch[F + c] = s.mean;
    batch[2 * F + c] = s.m2;
  }
}

// normalizer.py:110-136 with (n, batch_mean, batch_M2) given:
//   new_count = counter + n; delta = bm - mean; mean += delta * n / new_count
//   M2 += bM2 + delta^2 * counter * n / new_count
// `counter` is advanced by the LAST leaf only (advance_counter != 0): a PyTree
// normaliser shares one counter across leaves (normalizer.py:59).
__global__ void __launch_bounds__(kThreads)
welford_merge_kernel(float* __restrict__ mean, float* __restrict__ m2,
                     const float* __restrict__ counter,
                     const float* __restrict__ batch, int64_t F) {
  const int64_t c = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (c >= F) return;
  const float cnt = *counter;
  const float n = batch[c];
  const float new_count = cnt + n;
  const float frac = n / new_count;
  const float delta = batch[F + c] - mean[c];
  mean[c] = mean[c] + delta * frac;
  m2[c] = m2[c] + batch[2 * F + c] + (delta * delta) * cnt * n / new_count;
}

__global__ void counter_add_kernel(float* counter, const float* batch) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *counter = *counter + batch[0];
}

int stream_grid(int64_t total) {
  int64_t g = mippo::ceil_div(total, kThreads);
  if (g > mippo::kMaxStreamBlocks) g = mippo::kMaxStreamBlocks;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int mi_normalize_fwd_f32(const float* x, const float* mean, const float* m2,
                                    const float* counter, float epsilon, float* out,
                                    int64_t M, int64_t F, mi_stream_t stream) {
  MI_REQUIRE(M >= 0 && F >= 1, "mi_normalize_fwd_f32: bad shape M=%lld F=%lld",
             (long long)M, (long long)F);
  if (M == 0) return 0;
  MI_REQUIRE(x && mean && m2 && counter && out, "mi_normalize_fwd_f32: null pointer");
  const size_t lds = F <= 2048 ? (size_t)(2 * F) * sizeof(float) : 0;
  hipLaunchKernelGGL(normalize_fwd_kernel, dim3(stream_grid(M * F)), dim3(kThreads), lds,
                     mippo::as_stream(stream), x, mean, m2, counter, epsilon, out, M, F, nullptr,
                     (int64_t)0);
  return mippo::check_launch("mi_normalize_fwd_f32");
}

// mi_normalize_fwd_f32 of [x ; x_tail]: out [M + M_tail, F] (the same expression per element).
extern "C" int mi_normalize_fwd_tail_f32(const float* x, const float* x_tail, const float* mean,
                                         const float* m2, const float* counter, float epsilon,
                                         float* out, int64_t M, int64_t M_tail, int64_t F,
                                         mi_stream_t stream) {
  MI_REQUIRE(M >= 0 && M_tail >= 0 && F >= 1, "mi_normalize_fwd_tail_f32: bad shape");
  if (M + M_tail == 0) return 0;
  MI_REQUIRE((x || M == 0) && (x_tail || M_tail == 0) && mean && m2 && counter && out,
             "mi_normalize_fwd_tail_f32: null pointer");
  const size_t lds = F <= 2048 ? (size_t)(2 * F) * sizeof(float) : 0;
  hipLaunchKernelGGL(normalize_fwd_kernel, dim3(stream_grid((M + M_tail) * F)), dim3(kThreads),
                     lds, mippo::as_stream(stream), x, mean, m2, counter, epsilon, out, M, F,
                     x_tail, M_tail);
  return mippo::check_launch("mi_normalize_fwd_tail_f32");
}

extern "C" int mi_normalize_bwd_f32(const float* g_out, const float* m2,
                                    const float* counter, float epsilon, float* g_x,
                                    int64_t M, int64_t F, mi_stream_t stream) {
  MI_REQUIRE(M >= 0 && F >= 1, "mi_normalize_bwd_f32: bad shape");
  if (M == 0) return 0;
  MI_REQUIRE(g_out && m2 && counter && g_x, "mi_normalize_bwd_f32: null pointer");
  hipLaunchKernelGGL(normalize_bwd_kernel, dim3(stream_grid(M * F)), dim3(kThreads), 0,
                     mippo::as_stream(stream), g_out, m2, counter, epsilon, g_x, M, F);
  return mippo::check_launch("mi_normalize_bwd_f32");
}

namespace {
// a17: mean / population std of every column of a small row-major [R][C] matrix (the
// per-gradient-step loss rows of an iteration), fp64 accumulation in row order.  One
// thread per column; R is the number of gradient steps (16), so this is one tiny launch
// in place of one torch `std_mean` launch per metric.
__global__ void __launch_bounds__(64)
col_mean_std_kernel(const float* __restrict__ x, int64_t R, int64_t C, float scale,
                    float* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int64_t r = 0; r < R; ++r) s += (double)(x[r * C + c] * scale);
  const double mean = s / (double)R;
  double q = 0.0;
  for (int64_t r = 0; r < R; ++r) {
    const double d = (double)(x[r * C + c] * scale) - mean;
    q += d * d;
  }
  out[c] = (float)mean;
  out[C + c] = (float)sqrt(q / (double)R);
}
}  // namespace

extern "C" int mi_col_mean_std_f32(const float* x, int64_t R, int64_t C, float scale,
                                   float* out, mi_stream_t stream) {
  MI_REQUIRE(R >= 1 && C >= 1, "mi_col_mean_std_f32: bad shape R=%lld C=%lld", (long long)R,
             (long long)C);
  MI_REQUIRE(x && out, "mi_col_mean_std_f32: null pointer");
  hipLaunchKernelGGL(col_mean_std_kernel, dim3((unsigned)mippo::ceil_div(C, 64)), dim3(64), 0,
                     mippo::as_stream(stream), x, R, C, scale, out);
  return mippo::check_launch("mi_col_mean_std_f32");
}

static int welford_chunks(int64_t M) {
  int64_t g = mippo::ceil_div(M, 256);  // >= 256 rows per partial
  if (g > 512) g = 512;
  return (int)(g < 1 ? 1 : g);
}

extern "C" int64_t mi_welford_workspace_bytes(int64_t M, int64_t F) {
  if (M < 0 || F < 1) return -EINVAL;
  return (int64_t)welford_chunks(M) * 3 * F * (int64_t)sizeof(float);
}

extern "C" int mi_welford_batch_stats_f32(const float* x, float* batch_stats,
                                          void* workspace, int64_t M, int64_t F,
                                          mi_stream_t stream) {
  MI_REQUIRE(M >= 1 && F >= 1, "mi_welford_batch_stats_f32: bad shape M=%lld F=%lld",
             (long long)M, (long long)F);
  MI_REQUIRE(x && batch_stats && workspace, "mi_welford_batch_stats_f32: null pointer");
  const int CW = F < kThreads ? (int)F : kThreads;
  const int R = kThreads / CW;
  const int G = welford_chunks(M);
  const int64_t cchunks = mippo::ceil_div(F, CW);
  MI_REQUIRE(cchunks <= 65535, "mi_welford_batch_stats_f32: F too large");
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(welford_partial_kernel, dim3(G, (unsigned)cchunks), dim3(kThreads), 0,
                     mippo::as_stream(stream), x, partial, M, F, CW, R);
  int rc = mippo::check_launch("mi_welford_batch_stats_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(welford_finalize_kernel,
                     dim3((unsigned)mippo::ceil_div(F, kThreads / 64)), dim3(kThreads), 0,
                     mippo::as_stream(stream), partial, batch_stats, (int64_t)G, F);
  return mippo::check_launch("mi_welford_batch_stats_f32(finalize)");
}

extern "C" int mi_welford_merge_f32(float* mean, float* m2, float* counter,
                                    const float* batch_stats, int64_t F,
                                    int advance_counter, mi_stream_t stream) {
  MI_REQUIRE(F >= 1, "mi_welford_merge_f32: bad F");
  MI_REQUIRE(mean && m2 && counter && batch_stats, "mi_welford_merge_f32: null pointer");
  hipLaunchKernelGGL(welford_merge_kernel, dim3((unsigned)mippo::ceil_div(F, kThreads)),
                     dim3(kThreads), 0, mippo::as_stream(stream), mean, m2, counter,
                     batch_stats, F);
  int rc = mippo::check_launch("mi_welford_merge_f32");
  if (rc || !advance_counter) return rc;
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, mippo::as_stream(stream),
                     counter, batch_stats);
  return mippo::check_launch("mi_welford_merge_f32(counter)");
}
