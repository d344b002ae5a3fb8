// Counter-based RNG for the sampler kernels: Philox4x32-10 (Salmon et al.,
// "Parallel random numbers: as easy as 1, 2, 3", SC'11) + Box-Muller.
//
// The reference draws from jax.random (threefry) through nnx.Rngs
// (nnx_ppo/networks/sampling_layers.py:96,144); those streams cannot be
// reproduced without JAX, so parity is defined on the *scheme*: the oracle
// (oracle/philox.py) regenerates the same integers bit for bit and the same
// normals to fp32 rounding from (seed, offset, element index).
//
//   counter = (elem_lo, elem_hi, offset_lo, offset_hi), key = (seed_lo, seed_hi)
//   eps  = BoxMuller(x0, x1)   -- action noise
//   eps2 = BoxMuller(x2, x3)   -- entropy-estimate noise
// `offset` identifies the sampler call (one per batch forward), `elem` the
// (row, action-dim) element inside it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mippo {

struct U4 {
  uint32_t x, y, z, w;
};

__host__ __device__ inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
    U4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.w = (uint32_t)p0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

// u1 in (0,1), u2 in [0,1), both exact 24-bit fractions.
__device__ inline float box_muller(uint32_t a, uint32_t b) {
  const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
  // cos(2 pi u2) as cospi(2 u2): 2 u2 is exact (u2 is a 24-bit fraction) and cospif reduces
  // its argument exactly, so the result is within an ulp of the real-valued expression the
  // oracle evaluates in fp64 — and, unlike cosf, needs no Payne-Hanek fallback: cosf put
  // 552 bytes of scratch per lane into every kernel that samples, which throttled the
  // waves per CU of the sampling kernels (rocprofv3: 30 us for a 9 us trunk).
  return sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
}

__device__ inline void philox_normal_pair(uint64_t seed, uint64_t offset, uint64_t elem,
                                          float& eps, float& eps2) {
  U4 c = {(uint32_t)elem, (uint32_t)(elem >> 32), (uint32_t)offset,
          (uint32_t)(offset >> 32)};
  const U4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  eps = box_muller(r.x, r.y);
  eps2 = box_muller(r.z, r.w);
}

// eps2 alone (same bits as philox_normal_pair's second output)
__device__ inline float philox_normal_second(uint64_t seed, uint64_t offset, uint64_t elem) {
  U4 c = {(uint32_t)elem, (uint32_t)(elem >> 32), (uint32_t)offset,
          (uint32_t)(offset >> 32)};
  const U4 r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return box_muller(r.z, r.w);
}

}  // namespace mippo
