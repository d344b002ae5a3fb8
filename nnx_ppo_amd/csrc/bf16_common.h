// Shared definitions of the bf16 MFMA kernels (gemm_bf16.hip, mlp_bf16.hip).
#pragma once
#include "common.h"

namespace mippo_bf16 {

using bf16_t = __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using s16x4 = __attribute__((ext_vector_type(4))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

constexpr int kThreads = 256;
constexpr int BK = 64;         // reduce elements per k-tile
constexpr int LROW = BK + 8;   // r-contiguous LDS row length in bf16 (144 B: 16 B pad)

// Transcendentals of the bf16 path use the hardware exp2 / rcp (v_exp_f32,
// v_rcp_f32: ~1 ulp) instead of the libm expansions: the results are rounded to
// bf16 (8 significant bits) anyway, and the inlined libm tanhf / expf made the
// fully unrolled trunk epilogues overflow the instruction cache (22k-line
// kernels ran 5x slower than their MFMA + memory time).
__device__ inline float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ inline float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ inline float fast_sigmoid(float x) { return fast_rcp(1.0f + fast_exp(-x)); }
__device__ inline float fast_tanh(float x) {
  // tanh x = 1 - 2 / (exp(2x) + 1); saturates correctly for |x| large
  return 1.0f - 2.0f * fast_rcp(fast_exp(2.0f * x) + 1.0f);
}

// One uniform branch around a shared exp + rcp (tanh z = 2 sigmoid(2z) - 1,
// swish z = z sigmoid(z)); relu / none are selects.  Keeps unrolled epilogues small.
__device__ inline float act_fwd(float z, int act) {
  float y = act == MI_ACT_RELU ? fmaxf(z, 0.0f) : z;
  if (act >= MI_ACT_TANH) {
    const float s = fast_sigmoid(act == MI_ACT_TANH ? 2.0f * z : z);
    y = act == MI_ACT_TANH ? 2.0f * s - 1.0f : z * s;
  }
  return y;
}

// derivative of the activation; `aux` is the post-activation output for
// relu / tanh and the pre-activation for swish.
__device__ inline float act_grad(float aux, int act) {
  float g = act == MI_ACT_RELU ? (aux > 0.0f ? 1.0f : 0.0f)
                               : (act == MI_ACT_TANH ? 1.0f - aux * aux : 1.0f);
  if (act == MI_ACT_SWISH) {
    const float s = fast_sigmoid(aux);
    g = s * (1.0f + aux * (1.0f - s));
  }
  return g;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int stream_grid(int64_t n) {
  int64_t g = mippo::ceil_div(n, kThreads);
  if (g > mippo::kMaxStreamBlocks) g = mippo::kMaxStreamBlocks;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace mippo_bf16
