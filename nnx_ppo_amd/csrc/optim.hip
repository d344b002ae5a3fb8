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
#include "optim_common.h"

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

// One launch for the whole network: the element body (optim_common.h: adam_element) is
// shared with the launch that carries the one-shot gradient exchange (comm.hip).
__global__ void __launch_bounds__(kThreads)
adam_kernel(mippo_optim::AdamArgs a) {
  // Every load of a pass is issued before anything waits: gradient, moments, parameter,
  // then the slab sums; the first pass's are in flight while adam_begin reads the step.
  int64_t pass_begin = (int64_t)blockIdx.x * kThreads;
  int64_t i = pass_begin + threadIdx.x;
  float gi = 0.0f, m0 = 0.0f, v0 = 0.0f, p0 = 0.0f;
  auto fetch = [&] {
    if (i < a.n) {
      gi = a.g[i];
      m0 = a.m[i];
      v0 = a.v[i];
      p0 = a.p[i];
      if (a.slabs.n) gi = gi + mippo_optim::slab_sum(a, i, pass_begin);
    }
  };
  const mippo_optim::AdamScalars sc = mippo_optim::adam_read_scalars(a);  // in flight too
  fetch();
  const mippo_optim::AdamStep st = mippo_optim::adam_begin(a, sc);
  while (pass_begin < a.n) {
    if (i < a.n) mippo_optim::adam_element(a, st, i, pass_begin, gi, m0, v0, p0);
    pass_begin += (int64_t)gridDim.x * kThreads;
    i = pass_begin + threadIdx.x;
    if (pass_begin < a.n) fetch();
  }
  mippo_optim::adam_end(a, st);
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

// Adam grid cap: two 256-thread blocks per CU keep the arenas of the reference networks (80 K -
// 700 K floats) to one or two passes, and 512 ticket increments still cost nothing
constexpr int kAdamBlocksPerCU = 2;

// CUs of the device the library runs on (queried once; 256 on MI355X)
static int num_cus() {
  static const int n = [] {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cu < 1) {
      (void)hipGetLastError();
      cu = mippo::kNumCU;
    }
    return cu;
  }();
  return n;
}

extern "C" int mi_adam_step_f32(float* params, float* grads, float* m, float* v, int64_t n,
                                float lr, float b1, float b2, float eps, float weight_decay,
                                int64_t* step, const float* grad_norm, float max_norm,
                                void* begin_next_ticket, int64_t n_shadows,
                                const int64_t* shadow_begin, const int64_t* shadow_K,
                                const int64_t* shadow_N, void* const* w_bf, void* const* wt_bf,
                                void* const* frag_fwd, void* const* frag_bwd,
                                mi_stream_t stream) {
  mippo_optim::AdamArgs a;
  int rc = mippo_optim::fill_adam_args(a, "mi_adam_step_f32", params, grads, m, v, n, lr, b1, b2,
                                       eps, weight_decay, step, grad_norm, max_norm,
                                       begin_next_ticket, n_shadows, shadow_begin, shadow_K,
                                       shadow_N, w_bf, wt_bf, frag_fwd, frag_bwd);
  if (rc) return rc;
  // every block takes a ticket when this launch also opens the next step: keep the
  // grid at one block per CU so the tickets do not serialise on the counter's L2 line
  int grid = stream_grid(n);
  if (begin_next_ticket && grid > kAdamBlocksPerCU * num_cus()) grid = kAdamBlocksPerCU * num_cus();
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(kThreads), 0, mippo::as_stream(stream), a);
  return mippo::check_launch("mi_adam_step_f32");
}

// mi_adam_step_f32 that also reduces the pending split-M slabs of a grouped dW launch
// (mi_dense_bwd_dw_grouped_slabs_bf16) into the gradients it reads: gradient element i of
// problem l's kernel / bias is g[i] + sum_s slab_l[s][...], summed exactly as
// mi_reduce_slabs_grouped_f32 would (bit-identical), without that launch.
extern "C" int mi_adam_step_slabs_f32(
    float* params, float* grads, float* m, float* v, int64_t n, float lr, float b1, float b2,
    float eps, float weight_decay, int64_t* step, const float* grad_norm, float max_norm,
    void* begin_next_ticket, int64_t n_shadows, const int64_t* shadow_begin,
    const int64_t* shadow_K, const int64_t* shadow_N, void* const* w_bf, void* const* wt_bf,
    void* const* frag_fwd, void* const* frag_bwd, int64_t n_slab_leaves,
    const void* const* slab_ptr, const int64_t* n_slabs, const int64_t* slab_K,
    const int64_t* slab_N, const int64_t* gw_offset, const int64_t* gb_offset,
    const int64_t* gb_first, mi_stream_t stream) {
  mippo_optim::AdamArgs a;
  int rc = mippo_optim::fill_adam_args(a, "mi_adam_step_slabs_f32", params, grads, m, v, n, lr, b1,
                                       b2, eps, weight_decay, step, grad_norm, max_norm,
                                       begin_next_ticket, n_shadows, shadow_begin, shadow_K,
                                       shadow_N, w_bf, wt_bf, frag_fwd, frag_bwd);
  if (rc) return rc;
  MI_REQUIRE(n_slab_leaves >= 0 && n_slab_leaves <= mippo_optim::kMaxSlabLeaves,
             "mi_adam_step_slabs_f32: 0 <= n_slab_leaves <= %d", mippo_optim::kMaxSlabLeaves);
  MI_REQUIRE(begin_next_ticket, "mi_adam_step_slabs_f32: the launch must zero the gradients "
                                "(the slabs are consumed here)");
  a.slabs.n = (int)n_slab_leaves;
  for (int64_t l = 0; l < n_slab_leaves; ++l) {
    MI_REQUIRE(slab_ptr && n_slabs && slab_K && slab_N && gw_offset && gb_offset && slab_ptr[l] &&
                   n_slabs[l] >= 1 && slab_K[l] >= 1 && slab_N[l] >= 1 && gw_offset[l] >= 0 &&
                   gw_offset[l] + slab_K[l] * slab_N[l] <= n &&
                   (!gb_first || (gb_first[l] >= 0 && gb_first[l] < slab_N[l])) &&
                   (gb_offset[l] < 0 ||
                    gb_offset[l] + slab_N[l] - (gb_first ? gb_first[l] : 0) <= n),
               "mi_adam_step_slabs_f32: bad slab leaf %lld", (long long)l);
    mippo_optim::SlabLeaf& lf = a.slabs.leaf[l];
    lf.slabs = static_cast<const float*>(slab_ptr[l]);
    lf.S = (int)n_slabs[l];
    lf.KN = (int)(slab_K[l] * slab_N[l]);
    lf.N = (int)slab_N[l];
    lf.gw_off = gw_offset[l];
    lf.gb_off = gb_offset[l];
    lf.b_lo = gb_first ? (int)gb_first[l] : 0;
  }
  int grid = stream_grid(n);
  if (grid > kAdamBlocksPerCU * num_cus()) grid = kAdamBlocksPerCU * num_cus();
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(kThreads), 0, mippo::as_stream(stream), a);
  return mippo::check_launch("mi_adam_step_slabs_f32");
}
