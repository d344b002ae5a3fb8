// The key scheme's integer mix (nnx_ppo_amd/random.py: splitmix64 finaliser) and the
// synthetic env's observation draw, shared by keys.hip and the fused episode step of
// misc.hip so that both evaluate the very same expressions.
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace mippo_keys {

constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;
constexpr uint64_t kM1 = 0xBF58476D1CE4E5B9ull;
constexpr uint64_t kM2 = 0x94D049BB133111EBull;

__device__ inline uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * kM1;
  z = (z ^ (z >> 27)) * kM2;
  return z ^ (z >> 31);
}

// MockEnv observation column j of an env at step `step`:
// unit_uniform(fold_key(key, step), (O,))[j]  (envs/synthetic.py: MockEnv._obs)
__device__ inline float mock_obs(int64_t key, int64_t step, int j) {
  const uint64_t k = mix((uint64_t)key ^ mix((uint64_t)step + kGolden));
  const uint64_t b = mix(mix(k) ^ ((uint64_t)(j + 1) * kM2));
  const float u = (float)(int64_t)(b >> 40) * (1.0f / 16777216.0f);
  return (u - 0.5f) * 3.4641016151377544f;
}

}  // namespace mippo_keys
