// Row function of the tanh-Gaussian sampler forward
// (reference: nnx_ppo/networks/sampling_layers.py:82-147), shared by the
// stand-alone kernel (sampler.hip) and the fused policy step (mlp_bf16.hip) so
// both evaluate the same fp32 expression sequence.
#pragma once
#include "common.h"
#include "philox.h"

namespace mippo_sampler {

constexpr float kLog2 = 0.69314718055994530942f;
constexpr float kHalfLog2Pi = 0.91893853320467274178f;

// The row functions below are one-thread-per-row chains of transcendentals (a few
// thousand cycles whatever the batch size): they sit at the tail of every policy launch.
// exp / log go through the hardware v_exp_f32 / v_log_f32 (about an ulp each, ~3
// instructions) instead of the libm expansions (15-40 instructions each); the results feed
// fp32 sums of O(1) terms, where that is an absolute error of ~1e-7.
__device__ inline float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ inline float flog(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }

__device__ inline float softplus(float x) {
  // jax.nn.softplus = logaddexp(x, 0) = max(x, 0) + log1p(exp(-|x|))
  return fmaxf(x, 0.0f) + flog(1.0f + fexp(-fabsf(x)));
}

__device__ inline float sigmoidf(float x) { return 1.0f / (1.0f + fexp(-x)); }

// tanh z = 1 - 2 / (exp(2 z) + 1); saturates correctly for |z| large
__device__ inline float ftanh(float z) { return 1.0f - 2.0f / (fexp(2.0f * z) + 1.0f); }

__device__ inline float log_det_jac(float z) {
  // sampling_layers.py:133: 2 (log 2 - z - softplus(-2 z))
  return 2.0f * (kLog2 - z - softplus(-2.0f * z));
}

struct Noise {
  const uint64_t* rng;  // {seed, offset}; may be null when both eps are injected
  uint64_t offset_add;
  const float* eps;   // [B, A] injected action noise or null
  const float* eps2;  // [B, A] injected entropy noise or null
  __device__ inline void get(int64_t elem, float& e, float& e2) const {
    if (eps && eps2) {
      e = eps[elem];
      e2 = eps2[elem];
      return;
    }
    float pe, pe2;
    mippo::philox_normal_pair(rng[0], rng[1] + offset_add, (uint64_t)elem, pe, pe2);
    e = eps ? eps[elem] : pe;
    e2 = eps2 ? eps2[elem] : pe2;
  }
  // the entropy noise alone (a replay scores stored actions: the action noise is unused)
  __device__ inline float get_entropy_noise(int64_t elem) const {
    if (eps2) return eps2[elem];
    return mippo::philox_normal_second(rng[0], rng[1] + offset_add, (uint64_t)elem);
  }
};

struct FwdParams {
  const float* extras;  // [B, A] raw actions to score (replay) or null (sample)
  Noise noise;
  float* raw_out;       // each [B, A] or null
  float* action;
  float* mu_out;
  float* sigma_out;
  float* ll;            // [B] or null
  float* reg;           // [B] or null
  int A;
  float min_std, std_scale, entropy_weight;
  int deterministic;
};

// `row`: the 2A floats (mean | pre-softplus std) of batch row b.
__device__ __forceinline__ void fwd_row(const float* row, int64_t b, const FwdParams& p) {
  const int A = p.A;
  float ll_acc = 0.0f, h_acc = 0.0f;
  for (int a = 0; a < A; ++a) {
    const int64_t e = b * A + a;
    const float mu = row[a];
    const float sigma = (softplus(row[A + a]) + p.min_std) * p.std_scale;
    float eps = 0.0f, eps2;
    if (p.extras || p.deterministic)  // the action noise is not needed: one Box-Muller, not two
      eps2 = p.noise.get_entropy_noise(e);
    else
      p.noise.get(e, eps, eps2);
    const float sampled = p.deterministic ? mu : mu + sigma * eps;
    const float z = p.extras ? p.extras[e] : sampled;
    // _loglikelihood, sampling_layers.py:118-135
    const float q = (z - mu) / sigma;
    const float log_sigma = flog(sigma);
    float lp = -0.5f * (q * q) - (kHalfLog2Pi + log_sigma);
    lp -= log_det_jac(z);
    ll_acc += lp;
    // _entropy, sampling_layers.py:137-147
    const float z2 = mu + sigma * eps2;
    h_acc += (0.5f + kHalfLog2Pi + log_sigma) + log_det_jac(z2);
    if (p.raw_out) p.raw_out[e] = z;
    if (p.action) p.action[e] = ftanh(z);
    if (p.mu_out) p.mu_out[e] = mu;
    if (p.sigma_out) p.sigma_out[e] = sigma;
  }
  if (p.ll) p.ll[b] = ll_acc;
  if (p.reg) p.reg[b] = -p.entropy_weight * h_acc;
}

struct BwdParams {
  const float* ms;      // [B, 2A] the forward's input (mean | pre-softplus std)
  const float* extras;  // [B, A] raw actions that were scored
  Noise noise;          // entropy noise (eps2) regenerated from the forward's counter
  const float* g_ll;    // [B] d loss / d log-likelihood, or null
  float g_reg;          // d loss / d regulariser element
  int A;
  float min_std, std_scale, entropy_weight;
};

// Gradient w.r.t. the 2A inputs of row b, written to grow[0..2A) through `store`
// (fp32 global row in the stand-alone kernel, bf16 LDS row in the fused backward).
// `gl` = d loss / d log-likelihood of the row (bwd_row reads it from p.g_ll; the backward
// that evaluates the loss terms itself, trunk_ws.hip GAE, has it in a register).
template <typename Store>
__device__ __forceinline__ void bwd_row_gl(int64_t b, const BwdParams& p, const float gl,
                                           Store store) {
  const int A = p.A;
  const float* row = p.ms + b * 2 * A;
  const float gh = -p.entropy_weight * p.g_reg;  // d loss / d H
  for (int a = 0; a < A; ++a) {
    const int64_t e = b * A + a;
    const float mu = row[a];
    const float s = row[A + a];
    const float sigma = (softplus(s) + p.min_std) * p.std_scale;
    const float eps2 = p.noise.get_entropy_noise(e);
    const float z = p.extras[e];
    const float inv = 1.0f / sigma;
    const float q = (z - mu) * inv;
    // ll: d/dmu = q/sigma ; d/dsigma = (q^2 - 1)/sigma
    float g_mu = gl * q * inv;
    float g_sigma = gl * (q * q - 1.0f) * inv;
    // H: z2 = mu + sigma*eps2 ; d logdetjac / dz2 = -2 tanh(z2)
    const float t2 = ftanh(mu + sigma * eps2);
    g_mu += gh * (-2.0f * t2);
    g_sigma += gh * (inv - 2.0f * t2 * eps2);
    store(a, g_mu);
    store(A + a, g_sigma * sigmoidf(s) * p.std_scale);
  }
}

template <typename Store>
__device__ __forceinline__ void bwd_row(int64_t b, const BwdParams& p, Store store) {
  bwd_row_gl(b, p, p.g_ll ? p.g_ll[b] : 0.0f, store);
}

}  // namespace mippo_sampler
