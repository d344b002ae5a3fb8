// a15 — optimiser step over a flat fp32 parameter arena
// (reference: nnx.Optimizer.update with optax.chain([clip_by_global_norm?],
//  adam | adamw), nnx_ppo/algorithms/ppo.py:316,555-569).  optax / flax are not
//  in the reference tree; the formulas below are the published optax ones and
//  are PARITY UNPINNED (no reference test pins optimiser arithmetic).
//
//   clip_by_global_norm(c): g <- g            if ||g|| < c
//                           g <- g / ||g|| * c otherwise
//   scale_by_adam: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; t = count + 1
//                  u = (m / (1 - b1^t)) / (sqrt(v / (1 - b2^t)) + eps)
//   adamw: u += weight_decay * p ;  p <- p - lr * u
// All parameters, gradients and moments of a network live in four flat fp32
// arenas, so one launch updates the whole network (28 B/param of HBM traffic).
// The step counter and the gradient norm are device-resident so the whole
// update is HIP-graph capturable.
#include "bf16_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPartials = 512;

// Zero the gradient arena and advance the step counter (start of a grad step).
__global__ void __launch_bounds__(kThreads)
begin_step_kernel(float* __restrict__ grads, int64_t n, int64_t* step) {
  const int64_t i0 = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  for (int64_t i = i0; i < n; i += (int64_t)gridDim.x * kThreads) grads[i] = 0.0f;
  if (i0 == 0 && step) *step += 1;
}

__global__ void __launch_bounds__(kThreads)
sumsq_partial_kernel(const float* __restrict__ g, int64_t n, double* partials) {
  __shared__ double scratch[kThreads / 64];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kThreads) {
    const double x = g[i];
    s += x * x;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) t += scratch[w];
    partials[blockIdx.x] = t;
  }
}

__global__ void norm_finalize_kernel(const double* partials, int G, float* norm_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double t = 0.0;
  for (int g = 0; g < G; ++g) t += partials[g];
  *norm_out = (float)sqrt(t);
}

// bf16 shadows of Dense kernels that live in the arena (networks/dense_chain.py): the
// update writes the new value into the four bf16 images too, so no separate
// mi_weights_to_bf16_multi launch sits at the head of the next forward pass.
struct ShadowLeaf {
  int64_t begin;  // flat index of W[0][0] in the arena
  int K, N, ldw, ldwt;
  mippo_bf16::bf16_t* wb;  // [K][ldw]
  mippo_bf16::bf16_t* wt;  // [N][ldwt]
  mippo_bf16::bf16_t* ff;  // forward fragment-major image (gemm_bf16.hip: frag_store)
  mippo_bf16::bf16_t* fb;  // backward fragment-major image
};
constexpr int kMaxShadows = 16;
struct ShadowTable {
  ShadowLeaf leaf[kMaxShadows];
  int n;
};

// index of (column c, reduce element r) in a fragment-major image with R reduce elements
__device__ inline int64_t frag_index(int c, int r, int R) {
  const int KS = (R + 31) / 32;
  const int lane = (c & 15) + 16 * ((r & 31) >> 3);
  return (((int64_t)(c >> 4) * KS + (r >> 5)) * 64 + lane) * 8 + (r & 7);
}

__global__ void __launch_bounds__(kThreads)
adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
            float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
            float weight_decay, int64_t* __restrict__ step,
            const float* __restrict__ grad_norm, float max_norm, unsigned int* ticket,
            ShadowTable shadows) {
  // ticket != null: this launch also opens the NEXT gradient step — it counts itself
  // (t = step + 1, stored by block 0 once every block has signalled that it has read
  // `step`) and leaves the gradient arena zeroed, so no separate
  // mi_begin_grad_step_f32 launch sits between two minibatches.
  const int64_t s0 = *step;
  const float t = (float)(ticket ? s0 + 1 : s0);
  const float bc1 = 1.0f - powf(b1, t);
  const float bc2 = 1.0f - powf(b2, t);
  float gscale = 1.0f;
  bool clip = false;
  float gn = 1.0f;
  if (grad_norm) {
    gn = *grad_norm;
    clip = !(gn < max_norm);
  }
  (void)gscale;
  if (ticket) {
    // `step` has been read by every wave of this block once they pass the barrier
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int64_t pass_begin = (int64_t)blockIdx.x * kThreads; pass_begin < n;
       pass_begin += (int64_t)gridDim.x * kThreads) {
    const int64_t i = pass_begin + threadIdx.x;
    if (i >= n) break;
    float gi = g[i];
    if (clip) gi = gi / gn * max_norm;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * (gi * gi);
    m[i] = mi;
    v[i] = vi;
    float u = (mi / bc1) / (sqrtf(vi / bc2) + eps);
    const float pi = p[i];
    if (weight_decay != 0.0f) u += weight_decay * pi;
    const float pn = pi - lr * u;
    p[i] = pn;
    if (ticket) g[i] = 0.0f;
    // The 256 elements of this pass are one contiguous arena range, so the leaves that
    // overlap it are found with wave-uniform (scalar) tests; inside a leaf the index
    // arithmetic is 32-bit.  (A per-element search over the table with 64-bit divisions
    // made this tiny kernel 14 us long.)
    for (int l = 0; l < shadows.n; ++l) {
      const ShadowLeaf& lf = shadows.leaf[l];
      const unsigned KN = (unsigned)lf.K * (unsigned)lf.N;
      if (lf.begin >= pass_begin + kThreads || lf.begin + (int64_t)KN <= pass_begin) continue;
      const int64_t q64 = i - lf.begin;
      if (q64 >= 0 && q64 < (int64_t)KN) {
        const unsigned q = (unsigned)q64, N = (unsigned)lf.N;
        const unsigned k = q / N, c = q - k * N;
        const mippo_bf16::bf16_t b = (mippo_bf16::bf16_t)pn;
        lf.wb[(size_t)k * lf.ldw + c] = b;
        lf.wt[(size_t)c * lf.ldwt + k] = b;
        if (lf.ff) lf.ff[frag_index((int)c, (int)k, lf.K)] = b;  // columns = outputs, reduce = K
        if (lf.fb) lf.fb[frag_index((int)k, (int)c, lf.N)] = b;  // columns = inputs,  reduce = N
      }
    }
  }
  if (ticket && blockIdx.x == 0 && threadIdx.x == 0) {
    // every block has signalled that it has READ `step` (above); nothing else of this
    // launch is ordered by the counter, so no device-scope fence is needed (the fence of
    // the usual last-block ticket is ~3.5 us — a third of this kernel)
    while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x)
      __builtin_amdgcn_s_sleep(2);
    *step = s0 + 1;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int stream_grid(int64_t n) {
  int64_t g = mippo::ceil_div(n, kThreads);
  if (g > mippo::kMaxStreamBlocks) g = mippo::kMaxStreamBlocks;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int mi_begin_grad_step_f32(float* grads, int64_t n, int64_t* step,
                                      mi_stream_t stream) {
  MI_REQUIRE(n >= 0 && (grads || n == 0), "mi_begin_grad_step_f32: bad arguments");
  hipLaunchKernelGGL(begin_step_kernel, dim3(stream_grid(n)), dim3(kThreads), 0,
                     mippo::as_stream(stream), grads, n, step);
  return mippo::check_launch("mi_begin_grad_step_f32");
}

extern "C" int64_t mi_global_norm_workspace_bytes(int64_t n) {
  if (n < 0) return -EINVAL;
  return (int64_t)kMaxPartials * (int64_t)sizeof(double);
}

extern "C" int mi_global_norm_f32(const float* grads, int64_t n, float* norm_out,
                                  void* workspace, mi_stream_t stream) {
  MI_REQUIRE(n >= 1 && grads && norm_out && workspace, "mi_global_norm_f32: bad arguments");
  int64_t G = mippo::ceil_div(n, (int64_t)kThreads * 8);
  if (G > kMaxPartials) G = kMaxPartials;
  if (G < 1) G = 1;
  double* partials = static_cast<double*>(workspace);
  hipStream_t st = mippo::as_stream(stream);
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)G), dim3(kThreads), 0, st, grads, n,
                     partials);
  int rc = mippo::check_launch("mi_global_norm_f32(partial)");
  if (rc) return rc;
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(1), dim3(64), 0, st, partials, (int)G, norm_out);
  return mippo::check_launch("mi_global_norm_f32(finalize)");
}

extern "C" int mi_adam_step_f32(float* params, float* grads, float* m, float* v, int64_t n,
                                float lr, float b1, float b2, float eps, float weight_decay,
                                int64_t* step, const float* grad_norm, float max_norm,
                                void* begin_next_ticket, int64_t n_shadows,
                                const int64_t* shadow_begin, const int64_t* shadow_K,
                                const int64_t* shadow_N, void* const* w_bf, void* const* wt_bf,
                                void* const* frag_fwd, void* const* frag_bwd,
                                mi_stream_t stream) {
  MI_REQUIRE(n >= 1 && params && grads && m && v && step, "mi_adam_step_f32: bad arguments");
  MI_REQUIRE(n_shadows >= 0 && n_shadows <= kMaxShadows, "mi_adam_step_f32: 0 <= n_shadows <= %d",
             kMaxShadows);
  ShadowTable tab = {};
  tab.n = (int)n_shadows;
  for (int64_t l = 0; l < n_shadows; ++l) {
    MI_REQUIRE(shadow_begin && shadow_K && shadow_N && w_bf && wt_bf && w_bf[l] && wt_bf[l] &&
                   shadow_K[l] >= 1 && shadow_N[l] >= 1 && shadow_begin[l] >= 0 &&
                   shadow_begin[l] + shadow_K[l] * shadow_N[l] <= n,
               "mi_adam_step_f32: bad shadow %lld", (long long)l);
    ShadowLeaf& lf = tab.leaf[l];
    lf.begin = shadow_begin[l];
    lf.K = (int)shadow_K[l];
    lf.N = (int)shadow_N[l];
    lf.ldw = (int)(mippo::ceil_div(shadow_N[l], 8) * 8);
    lf.ldwt = (int)(mippo::ceil_div(shadow_K[l], 8) * 8);
    lf.wb = static_cast<mippo_bf16::bf16_t*>(w_bf[l]);
    lf.wt = static_cast<mippo_bf16::bf16_t*>(wt_bf[l]);
    lf.ff = frag_fwd ? static_cast<mippo_bf16::bf16_t*>(frag_fwd[l]) : nullptr;
    lf.fb = frag_bwd ? static_cast<mippo_bf16::bf16_t*>(frag_bwd[l]) : nullptr;
  }
  // every block takes a ticket when this launch also opens the next step: keep the
  // grid at one block per CU so the tickets do not serialise on the counter's L2 line
  int grid = stream_grid(n);
  if (begin_next_ticket && grid > mippo::kNumCU) grid = mippo::kNumCU;
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(kThreads), 0,
                     mippo::as_stream(stream), params, grads, m, v, n, lr, b1, b2, eps,
                     weight_decay, step, grad_norm, max_norm,
                     static_cast<unsigned int*>(begin_next_ticket), tab);
  return mippo::check_launch("mi_adam_step_f32");
}
