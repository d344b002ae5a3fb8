// The Adam(W) element update shared by the optimiser launch (optim.hip) and the launch
// that carries the one-shot gradient exchange inside it (comm.hip).
// Reference: nnx.Optimizer.update with optax.chain([clip_by_global_norm?], adam | adamw),
// nnx_ppo/algorithms/ppo.py:316,555-569 (optax is third-party: published formulas,
// PARITY UNPINNED).
#pragma once
#include "bf16_common.h"

namespace mippo_optim {

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

// Split-M slabs of a grouped dW launch (gemm_bf16.hip) whose reduction rides on the
// optimiser launch: leaf l adds sum_s slab[s][...] to the gradient of its kernel
// (arena range [gw_off, gw_off + KN)) and bias ([gb_off, gb_off + N), gb_off < 0: none).
// b_lo: first bias column that has a home in the arena — column j >= b_lo of the slab's db
// goes to arena element gb_off + (j - b_lo), columns below it are dropped (a GRU's recurrent
// kernel: of the 3H column sums of dgh only the n gate's H are a parameter's gradient).
struct SlabLeaf {
  const float* slabs;  // [S][KN + N]
  int64_t gw_off, gb_off;
  int S, KN, N, b_lo;
};
constexpr int kMaxSlabLeaves = 8;
struct SlabTable {
  SlabLeaf leaf[kMaxSlabLeaves];
  int n;
};

struct AdamArgs {
  float *p, *g, *m, *v;
  int64_t n;
  float lr, b1, b2, eps, weight_decay;
  int64_t* step;
  const float* grad_norm;  // nullable: clip_by_global_norm input
  float max_norm;
  unsigned int* ticket;    // nullable: this launch also opens the next gradient step
  ShadowTable shadows;
  SlabTable slabs;
};

struct AdamStep {
  int64_t s0;
  float bc1, bc2, gn;
  bool clip;
  unsigned int arrival;  // thread 0: this block's position among the "step read" signals
};

// index of (column c, reduce element r) in a fragment-major image with R reduce elements
__device__ inline int64_t frag_index(int c, int r, int R) {
  const int KS = (R + 31) / 32;
  const int lane = (c & 15) + 16 * ((r & 31) >> 3);
  return (((int64_t)(c >> 4) * KS + (r >> 5)) * 64 + lane) * 8 + (r & 7);
}

// Every thread: read the step count and derive the bias corrections.  With a ticket the
// launch counts itself: each block signals (relaxed, agent scope) that all its waves have
// READ `step`, and the block whose signal arrives last — at which point nobody will read
// the old value again — publishes step + 1 and re-arms the ticket (adam_end).  Nothing else
// is ordered by that counter, so there is no fence and no spin.
// The two scalar reads of adam_begin on their own, so that a kernel can issue them before
// its element loads (they are one more memory round trip otherwise).
struct AdamScalars {
  int64_t s0;
  float gn;
};
__device__ inline AdamScalars adam_read_scalars(const AdamArgs& a) {
  AdamScalars r;
  r.s0 = __hip_atomic_load(a.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  r.gn = a.grad_norm ? *a.grad_norm : 1.0f;
  return r;
}

__device__ inline AdamStep adam_begin(const AdamArgs& a, const AdamScalars& sc) {
  AdamStep st;
  st.s0 = sc.s0;
  const float t = (float)(a.ticket ? st.s0 + 1 : st.s0);
  st.bc1 = 1.0f - powf(a.b1, t);
  st.bc2 = 1.0f - powf(a.b2, t);
  st.clip = false;
  st.gn = 1.0f;
  if (a.grad_norm) {
    st.gn = sc.gn;
    st.clip = !(st.gn < a.max_norm);
  }
  st.arrival = 0;
  if (a.ticket) {
    __syncthreads();  // every wave of this block has read `step`
    if (threadIdx.x == 0)  // the returned position is looked at in adam_end: no wait here
      st.arrival = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return st;
}

__device__ inline AdamStep adam_begin(const AdamArgs& a) {
  return adam_begin(a, adam_read_scalars(a));
}

// Sum over the pending dW slabs that cover arena element i (0 if none): the fixed
// summation order of reduce_slabs_grouped_kernel (s = 0 .. S-1), 16 loads in flight.
__device__ inline float slab_sum(const AdamArgs& a, int64_t i, int64_t pass_begin) {
  float total = 0.0f;
  for (int l = 0; l < a.slabs.n; ++l) {
    const SlabLeaf& lf = a.slabs.leaf[l];
    const int64_t stride = (int64_t)lf.KN + lf.N;
    // wave-uniform range tests first (the pass is one contiguous 256-element range)
    const bool w_hit = lf.gw_off < pass_begin + mippo_bf16::kThreads && lf.gw_off + lf.KN > pass_begin;
    const int nb = lf.N - lf.b_lo;  // bias columns that live in the arena
    const bool b_hit = lf.gb_off >= 0 && lf.gb_off < pass_begin + mippo_bf16::kThreads &&
                       lf.gb_off + nb > pass_begin;
    if (!w_hit && !b_hit) continue;
    int64_t idx = -1;
    if (w_hit && i >= lf.gw_off && i < lf.gw_off + lf.KN) idx = i - lf.gw_off;
    if (b_hit && i >= lf.gb_off && i < lf.gb_off + nb) idx = lf.KN + lf.b_lo + (i - lf.gb_off);
    if (idx < 0) continue;
    float v = 0.0f;
    int s = 0;
    for (; s + 16 <= lf.S; s += 16) {
      float t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = lf.slabs[(int64_t)(s + u) * stride + idx];
#pragma unroll
      for (int u = 0; u < 16; ++u) v += t[u];
    }
    if (s < lf.S) {
      float t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = s + u < lf.S ? lf.slabs[(int64_t)(s + u) * stride + idx] : 0.0f;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (s + u < lf.S) v += t[u];
    }
    total += v;  // one leaf covers an element: this is `g + v` of the reduction kernel
  }
  return total;
}

// Element i with (already reduced) gradient gi.  `pass_begin`: start of the 256-element
// contiguous arena range this wave group is working on (i - threadIdx.x), so the shadow
// leaves that overlap it are found with wave-uniform (scalar) tests; inside a leaf the
// index arithmetic is 32-bit.
// `m0`, `v0`, `p0`: the element's moments and parameter, loaded by the caller (early, so
// the loads overlap whatever produces gi).
__device__ inline void adam_element(const AdamArgs& a, const AdamStep& st, int64_t i,
                                    int64_t pass_begin, float gi, float m0, float v0,
                                    float p0) {
  if (st.clip) gi = gi / st.gn * a.max_norm;
  const float mi = a.b1 * m0 + (1.0f - a.b1) * gi;
  const float vi = a.b2 * v0 + (1.0f - a.b2) * (gi * gi);
  a.m[i] = mi;
  a.v[i] = vi;
  float u = (mi / st.bc1) / (sqrtf(vi / st.bc2) + a.eps);
  const float pi = p0;
  if (a.weight_decay != 0.0f) u += a.weight_decay * pi;
  const float pn = pi - a.lr * u;
  a.p[i] = pn;
  if (a.ticket) a.g[i] = 0.0f;
  for (int l = 0; l < a.shadows.n; ++l) {
    const ShadowLeaf& lf = a.shadows.leaf[l];
    const unsigned KN = (unsigned)lf.K * (unsigned)lf.N;
    if (lf.begin >= pass_begin + mippo_bf16::kThreads || lf.begin + (int64_t)KN <= pass_begin)
      continue;
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

__device__ inline void adam_element(const AdamArgs& a, const AdamStep& st, int64_t i,
                                    int64_t pass_begin, float gi) {
  adam_element(a, st, i, pass_begin, gi, a.m[i], a.v[i], a.p[i]);
}

// The block whose signal arrived last publishes step + 1 and re-arms the ticket (every
// block has read the old value by then; see adam_begin).
__device__ inline void adam_end(const AdamArgs& a, const AdamStep& st) {
  if (a.ticket && threadIdx.x == 0) {
    const unsigned int blocks = gridDim.x * gridDim.y * gridDim.z;
    if (st.arrival == blocks - 1) {
      __hip_atomic_store(a.step, st.s0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Host: validate and pack the C-ABI arguments.
inline int fill_adam_args(AdamArgs& a, const char* who, float* params, float* grads, float* m,
                          float* v, int64_t n, float lr, float b1, float b2, float eps,
                          float weight_decay, int64_t* step, const float* grad_norm,
                          float max_norm, void* begin_next_ticket, int64_t n_shadows,
                          const int64_t* shadow_begin, const int64_t* shadow_K,
                          const int64_t* shadow_N, void* const* w_bf, void* const* wt_bf,
                          void* const* frag_fwd, void* const* frag_bwd) {
  MI_REQUIRE(n >= 1 && params && grads && m && v && step, "%s: bad arguments", who);
  MI_REQUIRE(n_shadows >= 0 && n_shadows <= kMaxShadows, "%s: 0 <= n_shadows <= %d", who,
             kMaxShadows);
  a = {};
  a.p = params;
  a.g = grads;
  a.m = m;
  a.v = v;
  a.n = n;
  a.lr = lr;
  a.b1 = b1;
  a.b2 = b2;
  a.eps = eps;
  a.weight_decay = weight_decay;
  a.step = step;
  a.grad_norm = grad_norm;
  a.max_norm = max_norm;
  a.ticket = static_cast<unsigned int*>(begin_next_ticket);
  a.shadows.n = (int)n_shadows;
  for (int64_t l = 0; l < n_shadows; ++l) {
    MI_REQUIRE(shadow_begin && shadow_K && shadow_N && w_bf && wt_bf && w_bf[l] && wt_bf[l] &&
                   shadow_K[l] >= 1 && shadow_N[l] >= 1 && shadow_begin[l] >= 0 &&
                   shadow_begin[l] + shadow_K[l] * shadow_N[l] <= n,
               "%s: bad shadow %lld", who, (long long)l);
    ShadowLeaf& lf = a.shadows.leaf[l];
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
  return 0;
}

}  // namespace mippo_optim
