// a4 / a18 — integer key derivation and episode bookkeeping as single launches.
//
// The reference threads jax.random keys through env resets, minibatch
// permutations and the EpisodeWrapper's random initial step counter
// (nnx_ppo/algorithms/ppo.py:271,284-294, rollout.py:57-59,
//  wrappers/episode_wrapper.py:12-31).  Here keys are int64 mixed with
// splitmix64; nnx_ppo_amd/random.py defines the scheme in integer torch ops
// (bit-identical on CPU and GPU) and these kernels evaluate the same integer
// expressions in ONE launch each instead of ~10 elementwise launches — at this
// workload's size the iteration is launch-bound, not arithmetic-bound.
// Integer work: bit-exact by construction (tests/test_keys_gpu.py).
#include <stdlib.h>

#include "common.h"
#include "keys_common.h"

namespace {

using mippo_keys::kGolden;
using mippo_keys::kM2;
using mippo_keys::mix;
constexpr int kThreads = 256;

enum { OUT_SPLIT = 0, OUT_BITS = 1, OUT_RANDINT = 2, OUT_UNIFORM = 3, OUT_UNIT_UNIFORM = 4 };

// keys[n] -> out[n * m]; element (i, j):
//   SPLIT        mix(k_i + (j+1) * GOLDEN)                         (random.split)
//   BITS         mix(mix(k_i) ^ ((j+1) * M2))                      (random.bits)
//   RANDINT      (BITS >> 1) % span + minval                       (random.randint)
//   UNIFORM      (BITS >> 40) * 2^-24                               (random.uniform)
//   UNIT_UNIFORM (UNIFORM - 0.5) * sqrt(12)                         (random.unit_uniform)
__global__ void __launch_bounds__(kThreads)
key_expand_kernel(const int64_t* __restrict__ keys, const int64_t* __restrict__ fold,
                  void* __restrict__ out, int64_t n, int64_t m, int mode, int64_t minval,
                  int64_t span, int child_major) {
  const int64_t total = n * m;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * kThreads) {
    // key-major: out[i*m + j]; child-major: out[j*n + i] (each child set contiguous)
    const int64_t i = child_major ? e % n : e / m;
    const int64_t j = child_major ? e / n : e % m;
    uint64_t k = (uint64_t)keys[i];
    if (fold) k = mix(k ^ mix((uint64_t)fold[i] + kGolden));  // key_fold_kernel
    if (mode == OUT_SPLIT) {
      static_cast<int64_t*>(out)[e] = (int64_t)mix(k + (uint64_t)(j + 1) * kGolden);
      continue;
    }
    const uint64_t b = mix(mix(k) ^ ((uint64_t)(j + 1) * kM2));
    if (mode == OUT_BITS) {
      static_cast<int64_t*>(out)[e] = (int64_t)b;
    } else if (mode == OUT_RANDINT) {
      // torch: (b >>> 1) % span on non-negative int64
      static_cast<int64_t*>(out)[e] = (int64_t)((b >> 1) % (uint64_t)span) + minval;
    } else {
      const float u = (float)(int64_t)(b >> 40) * (1.0f / 16777216.0f);
      static_cast<float*>(out)[e] =
          mode == OUT_UNIFORM ? u : (u - 0.5f) * 3.4641016151377544f;
    }
  }
}

// out[i] = mix(a[i] ^ mix(b[i] + GOLDEN))  — fold a per-env integer into a key
__global__ void __launch_bounds__(kThreads)
key_fold_kernel(const int64_t* __restrict__ a, const int64_t* __restrict__ b,
                int64_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kThreads)
    out[i] = (int64_t)mix((uint64_t)a[i] ^ mix((uint64_t)b[i] + kGolden));
}

// EpisodeWrapper.step, episode_wrapper.py:12-22:
//   counter' = counter + 1 ; truncated = inner_truncated | (counter' >= max_len)
//   done = float(inner_done | truncated)
__global__ void __launch_bounds__(kThreads)
episode_step_kernel(const int64_t* __restrict__ counter, const void* __restrict__ inner_done,
                    int done_is_float, const uint8_t* __restrict__ inner_trunc, int64_t max_len,
                    int64_t* __restrict__ counter_out, uint8_t* __restrict__ trunc_out,
                    float* __restrict__ done_out, uint8_t* __restrict__ flag_out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kThreads) {
    const int64_t c = counter[i] + 1;
    const bool d = done_is_float ? static_cast<const float*>(inner_done)[i] != 0.0f
                                 : static_cast<const uint8_t*>(inner_done)[i] != 0;
    const bool t = (inner_trunc ? inner_trunc[i] != 0 : false) || c >= max_len;
    counter_out[i] = c;
    trunc_out[i] = t ? 1 : 0;
    done_out[i] = (d || t) ? 1.0f : 0.0f;
    if (flag_out) flag_out[i] = (d || t) ? 1 : 0;
  }
}

// The whole step of the synthetic benchmark env (envs/synthetic.py: MockEnv.step, restating
// `nnx_ppo/test_dummies/mock_env.py:25-63`) in ONE launch: step' = step + 1,
// done = step' >= max_steps, obs = unit_uniform(fold_key(key, step'), (O,)) written into
// up to 8 observation leaves (a PyTree observation is the flat draw cut at the leaf
// widths).  Same integer / float expressions as key_expand_kernel (OUT_UNIT_UNIFORM with a
// fold) and episode_step_kernel, which this replaces for that env: bit-identical.
struct MockObs {
  float* leaf[8];
  int width[8];
  int n_leaves, total;
};
__global__ void __launch_bounds__(kThreads)
mock_env_step_kernel(const int64_t* __restrict__ key, const int64_t* __restrict__ count,
                     int64_t max_steps, int64_t* __restrict__ count_out,
                     uint8_t* __restrict__ done_out, MockObs o, int64_t n) {
  const int64_t total = n * o.total;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * kThreads) {
    const int64_t i = e / o.total;
    int j = (int)(e - i * o.total);
    const int64_t step = count[i] + 1;
    if (j == 0) {
      count_out[i] = step;
      done_out[i] = step >= max_steps ? 1 : 0;
    }
    const float v = mippo_keys::mock_obs(key[i], step, j);
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      if (l < o.n_leaves) {
        if (j >= 0 && j < o.width[l]) o.leaf[l][i * o.width[l] + j] = v;
        j -= o.width[l];
      }
    }
  }
}

// Minibatch permutations (ppo.py:284-294): block e sorts the n hashes of key
// fold_in(key, e) and writes the argsort — `random.permutation(fold_in(key, e), n)`
// for every epoch in ONE launch.  (key, index) pairs make the order total, so the
// result equals torch's stable argsort bit for bit.  One workgroup per permutation,
// bitonic network over the padded power of two in LDS (n <= 8192).
constexpr int kSortThreads = 1024;

__global__ void __launch_bounds__(kSortThreads)
key_permutations_kernel(const int64_t* __restrict__ key, int64_t* __restrict__ out, int n, int P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sort_raw[];
  int64_t* hk = reinterpret_cast<int64_t*>(sort_raw);        // [P] hashes
  int* hi = reinterpret_cast<int*>(hk + P);                  // [P] indices
  const uint64_t ke = mix((uint64_t)key[0] ^ mix((uint64_t)blockIdx.x + kGolden));  // fold_in
  const uint64_t mk = mix(ke);
  for (int i = threadIdx.x; i < P; i += kSortThreads) {
    hk[i] = i < n ? (int64_t)mix(mk ^ ((uint64_t)(i + 1) * kM2)) : INT64_MAX;  // random.bits
    hi[i] = i;
  }
  __syncthreads();
  // P/2 compare-exchanges per stage, one per loop trip of a thread (no idle half): pair
  // q -> lower element i = insert a zero bit at position log2(j) of q, partner i + j.
  // For j <= 64 the 64 pairs of a wave's trip stay inside one aligned 128-element chunk
  // through all remaining stages of the merge, so those stages need no workgroup
  // barrier (LDS operations of a wave are ordered): 15 barriers instead of 78 at P = 4096.
  auto stage = [&](int k, int j) {
    for (int q = threadIdx.x; q < (P >> 1); q += kSortThreads) {
      const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1));
      const int l = i + j;
      const int64_t a = hk[i], b = hk[l];
      const int ia = hi[i], ib = hi[l];
      const bool a_gt_b = a > b || (a == b && ia > ib);
      const bool up = (i & k) == 0;
      if (a_gt_b == up) {
        hk[i] = b;
        hk[l] = a;
        hi[i] = ib;
        hi[l] = ia;
      }
    }
  };
  for (int k = 2; k <= P; k <<= 1) {
    int j = k >> 1;
    for (; j > 64; j >>= 1) {
      stage(k, j);
      __syncthreads();
    }
    for (; j > 0; j >>= 1) {
      stage(k, j);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
  }
  int64_t* o = out + (int64_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += kSortThreads) o[i] = hi[i];
}

// The same argsort by RANK: workgroup (x, e) recomputes permutation e's n hashes into LDS
// and counts, for each of its 64 elements (one per lane), the (hash, index) pairs below it
// — that count is the element's position.  The four waves take a quarter of the n
// candidates each.  n^2 compares instead of n log^2 n exchanges, but every CU works
// (n / 64 workgroups per permutation instead of one) and nothing waits on a barrier.
// 64-bit integer compares are slow on the vector unit, so the loop compares the signed
// high words only (one v_cmp + add-with-carry per candidate) and falls to the exact
// (hash, index) order only when some lane sees equal high words.  Same total order, same
// result as the bitonic network.
constexpr int kRankElems = 64;
__global__ void __launch_bounds__(kThreads)
key_permutations_rank_kernel(const int64_t* __restrict__ key, int64_t* __restrict__ out, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sort_raw[];
  int64_t* hk = reinterpret_cast<int64_t*>(sort_raw);  // [n rounded up to 64], pads = max
  __shared__ int part[kThreads / 64][kRankElems];
  const uint64_t ke = mix((uint64_t)key[0] ^ mix((uint64_t)blockIdx.y + kGolden));  // fold_in
  const uint64_t mk = mix(ke);
  const int n2 = (n + 63) & ~63;
  for (int i = threadIdx.x; i < n2; i += kThreads)
    hk[i] = i < n ? (int64_t)mix(mk ^ ((uint64_t)(i + 1) * kM2)) : INT64_MAX;  // random.bits
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * kRankElems + lane;
  const int64_t h = hk[i < n2 ? i : 0];
  const int hhi = (int)(h >> 32);
  // this wave's candidates: chunks of 16, interleaved over the four waves
  int rank = 0;
  for (int j = wave * 16; j < n2; j += 16 * (kThreads / 64)) {
    int64_t t[16];  // requested together (wave-broadcast LDS reads), then compared
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = hk[j + u];
    unsigned nearest = 0xffffffffu;  // min over the candidates of (high word XOR mine)
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int thi = (int)(t[u] >> 32);
      rank += thi < hhi ? 1 : 0;
      nearest = min(nearest, (unsigned)(thi ^ hhi));
    }
    if (__any(nearest == 0u)) {  // rare (and once per lane for the element itself): the exact order
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int thi = (int)(t[u] >> 32);
        if (thi == hhi)
          rank += ((uint32_t)t[u] < (uint32_t)h || ((uint32_t)t[u] == (uint32_t)h && j + u < i))
                      ? 1 : 0;
      }
    }
  }
  part[wave][lane] = rank;
  __syncthreads();
  if (wave == 0 && i < n) {
    int r = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) r += part[w][lane];
    out[(int64_t)blockIdx.y * n + r] = i;
  }
}

int stream_grid(int64_t n) {
  int64_t g = mippo::ceil_div(n, kThreads);
  if (g > mippo::kMaxStreamBlocks) g = mippo::kMaxStreamBlocks;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int mi_key_expand(const int64_t* keys, const int64_t* fold, void* out, int64_t n,
                             int64_t m, int mode, int64_t minval, int64_t maxval,
                             int child_major, mi_stream_t stream) {
  MI_REQUIRE(n >= 0 && m >= 0 && mode >= OUT_SPLIT && mode <= OUT_UNIT_UNIFORM,
             "mi_key_expand: bad arguments");
  if (n == 0 || m == 0) return 0;
  MI_REQUIRE(keys && out, "mi_key_expand: null pointer");
  MI_REQUIRE(mode != OUT_RANDINT || maxval > minval, "mi_key_expand: empty randint range");
  hipLaunchKernelGGL(key_expand_kernel, dim3(stream_grid(n * m)), dim3(kThreads), 0,
                     mippo::as_stream(stream), keys, fold, out, n, m, mode, minval,
                     maxval - minval, child_major);
  return mippo::check_launch("mi_key_expand");
}

extern "C" int mi_key_fold(const int64_t* a, const int64_t* b, int64_t* out, int64_t n,
                           mi_stream_t stream) {
  MI_REQUIRE(n >= 0, "mi_key_fold: bad n");
  if (n == 0) return 0;
  MI_REQUIRE(a && b && out, "mi_key_fold: null pointer");
  hipLaunchKernelGGL(key_fold_kernel, dim3(stream_grid(n)), dim3(kThreads), 0,
                     mippo::as_stream(stream), a, b, out, n);
  return mippo::check_launch("mi_key_fold");
}

extern "C" int mi_episode_step(const int64_t* counter, const void* inner_done, int done_is_float,
                               const uint8_t* inner_truncated, int64_t max_len,
                               int64_t* counter_out, uint8_t* truncated_out, float* done_out,
                               uint8_t* done_flag_out, int64_t n, mi_stream_t stream) {
  MI_REQUIRE(n >= 0, "mi_episode_step: bad n");
  if (n == 0) return 0;
  MI_REQUIRE(counter && inner_done && counter_out && truncated_out && done_out,
             "mi_episode_step: null pointer");
  hipLaunchKernelGGL(episode_step_kernel, dim3(stream_grid(n)), dim3(kThreads), 0,
                     mippo::as_stream(stream), counter, inner_done, done_is_float, inner_truncated,
                     max_len, counter_out, truncated_out, done_out, done_flag_out, n);
  return mippo::check_launch("mi_episode_step");
}

extern "C" int mi_mock_env_step(const int64_t* key, const int64_t* step_count, int64_t max_steps,
                                int64_t* step_count_out, uint8_t* done_out,
                                float* const* obs_leaf, const int64_t* leaf_width,
                                int64_t n_leaves, int64_t n, mi_stream_t stream) {
  MI_REQUIRE(n >= 0 && n_leaves >= 1 && n_leaves <= 8, "mi_mock_env_step: 1 <= n_leaves <= 8");
  if (n == 0) return 0;
  MI_REQUIRE(key && step_count && step_count_out && done_out && obs_leaf && leaf_width,
             "mi_mock_env_step: null pointer");
  MockObs o = {};
  o.n_leaves = (int)n_leaves;
  for (int64_t l = 0; l < n_leaves; ++l) {
    MI_REQUIRE(obs_leaf[l] && leaf_width[l] >= 1 && leaf_width[l] <= (1 << 20),
               "mi_mock_env_step: bad leaf %lld", (long long)l);
    o.leaf[l] = obs_leaf[l];
    o.width[l] = (int)leaf_width[l];
    o.total += (int)leaf_width[l];
  }
  hipLaunchKernelGGL(mock_env_step_kernel, dim3(stream_grid(n * o.total)), dim3(kThreads), 0,
                     mippo::as_stream(stream), key, step_count, max_steps, step_count_out,
                     done_out, o, n);
  return mippo::check_launch("mi_mock_env_step");
}

extern "C" int mi_key_permutations(const int64_t* key, int64_t* out, int64_t n_perm, int64_t n,
                                   mi_stream_t stream) {
  MI_REQUIRE(n_perm >= 0 && n >= 0 && n <= 8192 && n_perm <= 65535,
             "mi_key_permutations: 0 <= n <= 8192 (n=%lld)", (long long)n);
  if (n_perm == 0 || n == 0) return 0;
  MI_REQUIRE(key && out, "mi_key_permutations: null pointer");
  static const bool bitonic = [] {  // MIPPO_PERM_BITONIC=1: the one-workgroup network (A/B)
    const char* e = getenv("MIPPO_PERM_BITONIC");
    return e && e[0] == '1';
  }();
  if (!bitonic) {
    const size_t bytes = (size_t)((n + 63) & ~(int64_t)63) * sizeof(int64_t);  // <= 64 KiB
    hipLaunchKernelGGL(key_permutations_rank_kernel,
                       dim3((unsigned)mippo::ceil_div(n, kRankElems), (unsigned)n_perm),
                       dim3(kThreads), bytes, mippo::as_stream(stream), key, out, (int)n);
    return mippo::check_launch("mi_key_permutations");
  }
  int P = 2;
  while (P < n) P <<= 1;
  const size_t lds = (size_t)P * (sizeof(int64_t) + sizeof(int));
  static const hipError_t attr =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&key_permutations_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 12);
  MI_REQUIRE(attr == hipSuccess, "mi_key_permutations: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL(key_permutations_kernel, dim3((unsigned)n_perm), dim3(kSortThreads), lds,
                     mippo::as_stream(stream), key, out, (int)n, P);
  return mippo::check_launch("mi_key_permutations");
}
