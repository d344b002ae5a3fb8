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

__device__ inline float act_fwd(float z, int act) {
  switch (act) {
    case MI_ACT_RELU: return fmaxf(z, 0.0f);
    case MI_ACT_TANH: return tanhf(z);
    case MI_ACT_SWISH: return z / (1.0f + expf(-z));
    default: return z;
  }
}

// derivative of the activation; `aux` is the post-activation output for
// relu / tanh and the pre-activation for swish.
__device__ inline float act_grad(float aux, int act) {
  switch (act) {
    case MI_ACT_RELU: return aux > 0.0f ? 1.0f : 0.0f;
    case MI_ACT_TANH: return 1.0f - aux * aux;
    case MI_ACT_SWISH: {
      const float s = 1.0f / (1.0f + expf(-aux));
      return s * (1.0f + aux * (1.0f - s));
    }
    default: return 1.0f;
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline int stream_grid(int64_t n) {
  int64_t g = mippo::ceil_div(n, kThreads);
  if (g > mippo::kMaxStreamBlocks) g = mippo::kMaxStreamBlocks;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace mippo_bf16
