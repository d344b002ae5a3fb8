// a5 / a7 — byte-exact data movement of the hot path:
//   * minibatch gather  x[:, inds]  over time-major [T, N, row] leaves
//     (reference: nnx_ppo/algorithms/ppo.py:297-300)
//   * masked row select  where(done[:, None], on_true, on_false)  used for
//     env / carry reset-on-done (reference: nnx_ppo/algorithms/rollout.py:270-279,
//     ppo.py:411-413)
// Both are dtype-agnostic (rows are moved as 4-byte words when the row size
// allows, bytes otherwise) and therefore bit-exact by construction.
#include "common.h"
#include "keys_common.h"

namespace {

constexpr int kThreads = 256;

// dst[t, j, :] = src[t, idx[j], :]   (W words of type T per row)
template <typename T>
__global__ void __launch_bounds__(kThreads)
gather_cols_kernel(const T* __restrict__ src, const int64_t* __restrict__ idx,
                   T* __restrict__ dst, int64_t Tn, int64_t N, int64_t L, int64_t W) {
  const int64_t total = Tn * L * W;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int64_t w = i % W;
    const int64_t j = (i / W) % L;
    const int64_t t = i / (W * L);
    dst[i] = src[(t * N + idx[j]) * W + w];
  }
}

// out[b, :] = mask[b] ? on_true[b * true_stride : ...] : on_false[b, :]
template <typename T>
__global__ void __launch_bounds__(kThreads)
select_rows_kernel(const uint8_t* __restrict__ mask, const T* __restrict__ on_true,
                   int64_t true_stride, const T* __restrict__ on_false, T* __restrict__ out,
                   int64_t B, int64_t W) {
  const int64_t total = B * W;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int64_t b = i / W, w = i % W;
    out[i] = mask[b] ? on_true[b * true_stride + w] : on_false[i];
  }
}

// Several leaves in one launch (blockIdx.y = leaf): the carry / env state of a
// rollout step has ~10 small leaves, and one launch per leaf is launch-bound.
struct SelectLeaf {
  const void* on_true;
  const void* on_false;
  void* out;
  int64_t true_stride;  // in words; 0 = broadcast one on_true row
  int64_t words;        // words per row
  int word_bytes;       // 4 or 1
  // mi_episode_step_select with a producer: this leaf's on_false values are COMPUTED by the
  // launch (and stored to on_false as the stepped state) instead of read.
  //   1: float32 observation columns col0 .. col0 + words - 1 of the MockEnv draw
  //   2: the int64 inner step counter (two 4-byte words per row)
  int produced;
  int col0;
};
constexpr int kMaxSelectLeaves = 16;
struct SelectTable {
  SelectLeaf leaf[kMaxSelectLeaves];
};

__global__ void __launch_bounds__(kThreads)
select_rows_multi_kernel(const uint8_t* __restrict__ mask, SelectTable tab, int64_t B) {
  const SelectLeaf lf = tab.leaf[blockIdx.y];
  const int64_t total = B * lf.words;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int64_t b = i / lf.words, w = i % lf.words;
    if (lf.word_bytes == 4) {
      const uint32_t* t = static_cast<const uint32_t*>(lf.on_true);
      const uint32_t* f = static_cast<const uint32_t*>(lf.on_false);
      static_cast<uint32_t*>(lf.out)[i] = mask[b] ? t[b * lf.true_stride + w] : f[i];
    } else {
      const uint8_t* t = static_cast<const uint8_t*>(lf.on_true);
      const uint8_t* f = static_cast<const uint8_t*>(lf.on_false);
      static_cast<uint8_t*>(lf.out)[i] = mask[b] ? t[b * lf.true_stride + w] : f[i];
    }
  }
}

// EpisodeWrapper.step + the reset-on-done select of every env-state leaf in ONE launch
// (a rollout step is five launches of 2-10 us; this removes one of them).  The done flag
// of a row is recomputed from the wrapper's three inputs wherever it is needed, so no
// block waits for another.  blockIdx.y == n_leaves: the wrapper's own outputs and the
// three leaves it produces; blockIdx.y < n_leaves: one generic leaf each.
struct EpisodeArgs {
  const int64_t* counter;
  const void* inner_done;
  const uint8_t* inner_trunc;
  int64_t max_len;
  int done_is_float;
  int64_t* counter_out;
  uint8_t* trunc_out;
  float* done_out;
  uint8_t* flag_out;
  const int64_t* reset_counter;
  const uint8_t* reset_trunc;
  const float* reset_done;
  int64_t* counter_sel;
  uint8_t* trunc_sel;
  float* done_sel;
  // producer (MockEnv.step inside this launch; mock_key == nullptr: none): the inner env's
  // step' = mock_count + 1, done = step' >= mock_max_steps, obs = mock_obs(key, step', j)
  const int64_t* mock_key;
  const int64_t* mock_count;
  int64_t mock_max_steps;
};

__device__ inline void episode_row(const EpisodeArgs& e, int64_t b, int64_t& c, bool& t, bool& m) {
  c = e.counter[b] + 1;
  const bool d = e.mock_key ? e.mock_count[b] + 1 >= e.mock_max_steps
                 : e.done_is_float ? static_cast<const float*>(e.inner_done)[b] != 0.0f
                                   : static_cast<const uint8_t*>(e.inner_done)[b] != 0;
  t = (e.inner_trunc ? e.inner_trunc[b] != 0 : false) || c >= e.max_len;
  m = d || t;
}

__global__ void __launch_bounds__(kThreads)
episode_select_kernel(EpisodeArgs e, SelectTable tab, int n_leaves, int64_t B) {
  if ((int)blockIdx.y == n_leaves) {
    for (int64_t b = (int64_t)blockIdx.x * kThreads + threadIdx.x; b < B;
         b += (int64_t)gridDim.x * kThreads) {
      int64_t c;
      bool t, m;
      episode_row(e, b, c, t, m);
      e.counter_out[b] = c;
      e.trunc_out[b] = t ? 1 : 0;
      e.done_out[b] = m ? 1.0f : 0.0f;
      if (e.flag_out) e.flag_out[b] = m ? 1 : 0;
      e.counter_sel[b] = m ? e.reset_counter[b] : c;
      e.trunc_sel[b] = m ? e.reset_trunc[b] : (uint8_t)(t ? 1 : 0);
      e.done_sel[b] = m ? e.reset_done[b] : 0.0f;
    }
    return;
  }
  const SelectLeaf lf = tab.leaf[blockIdx.y];
  const int64_t total = B * lf.words;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int64_t b = i / lf.words, w = i % lf.words;
    int64_t c;
    bool t, m;
    episode_row(e, b, c, t, m);
    if (lf.word_bytes == 4) {
      const uint32_t* tt = static_cast<const uint32_t*>(lf.on_true);
      uint32_t fv;
      if (lf.produced) {  // the inner env's own step: compute, keep as the stepped state
        const int64_t step = e.mock_count[b] + 1;
        if (lf.produced == 1)
          fv = __float_as_uint(mippo_keys::mock_obs(e.mock_key[b], step, lf.col0 + (int)w));
        else
          fv = (uint32_t)((uint64_t)step >> (32 * (int)w));
        static_cast<uint32_t*>(const_cast<void*>(lf.on_false))[i] = fv;
      } else {
        fv = static_cast<const uint32_t*>(lf.on_false)[i];
      }
      static_cast<uint32_t*>(lf.out)[i] = m ? tt[b * lf.true_stride + w] : fv;
    } else {
      const uint8_t* tt = static_cast<const uint8_t*>(lf.on_true);
      const uint8_t* ff = static_cast<const uint8_t*>(lf.on_false);
      static_cast<uint8_t*>(lf.out)[i] = m ? tt[b * lf.true_stride + w] : ff[i];
    }
  }
}

// Several contiguous buffers copied in one launch (blockIdx.y = buffer): the state
// hand-over at the end of a captured iteration is ~10 small tensors, and one
// launch per tensor costs more than the bytes do.
struct CopyLeaf {
  const void* src;
  void* dst;
  int64_t words;
  int word_bytes;
};
struct CopyTable {
  CopyLeaf leaf[kMaxSelectLeaves];
};

__global__ void __launch_bounds__(kThreads)
copy_multi_kernel(CopyTable tab) {
  const CopyLeaf lf = tab.leaf[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < lf.words;
       i += (int64_t)gridDim.x * kThreads) {
    if (lf.word_bytes == 4)
      static_cast<uint32_t*>(lf.dst)[i] = static_cast<const uint32_t*>(lf.src)[i];
    else
      static_cast<uint8_t*>(lf.dst)[i] = static_cast<const uint8_t*>(lf.src)[i];
  }
}

// T per-step tensors of each of n leaves -> n stacked [T, ...] buffers in one launch
// (blockIdx.y = leaf * T + step): a rollout ends with ~12 such stacks, one torch.cat
// launch each.
constexpr int kMaxStackSegments = 448;  // the table must fit the 4 KiB argument segment
struct StackTable {
  const void* src[kMaxStackSegments];
  void* dst[kMaxSelectLeaves];
  int64_t words[kMaxSelectLeaves];
  int word_bytes[kMaxSelectLeaves];
  int T;
};

__global__ void __launch_bounds__(kThreads)
stack_multi_kernel(StackTable tab) {
  const int seg = blockIdx.y, l = seg / tab.T, t = seg % tab.T;
  const int64_t words = tab.words[l];
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < words;
       i += (int64_t)gridDim.x * kThreads) {
    if (tab.word_bytes[l] == 4)
      static_cast<uint32_t*>(tab.dst[l])[t * words + i] =
          static_cast<const uint32_t*>(tab.src[seg])[i];
    else
      static_cast<uint8_t*>(tab.dst[l])[t * words + i] =
          static_cast<const uint8_t*>(tab.src[seg])[i];
  }
}

struct GatherLeaf {
  const void* src;
  void* dst;
  int64_t T;      // time steps of this leaf
  int64_t words;  // words per row
  int word_bytes;
};
struct GatherTable {
  GatherLeaf leaf[kMaxSelectLeaves];
};

// IT = the integer type of the element index arithmetic: uint32_t whenever every leaf has
// fewer than 2^32 elements (always, at the reference's sizes) — four 64-bit divisions per
// element were most of this kernel's time (25 us for 5 M elements at C2)
template <typename IT>
__global__ void __launch_bounds__(kThreads)
gather_cols_multi_kernel(GatherTable tab, const int64_t* __restrict__ idx, int64_t N, int64_t L,
                         int64_t GL) {
  const GatherLeaf lf = tab.leaf[blockIdx.y];
  const IT words = (IT)lf.words, Ls = (IT)L, GLs = (IT)GL;
  const IT total = (IT)(lf.T * L * lf.words);
  const IT stride = (IT)gridDim.x * kThreads;
  for (IT i = (IT)blockIdx.x * kThreads + threadIdx.x; i < total; i += stride) {
    const IT row = i / words, w = i - row * words;
    const IT t = row / Ls, j = row - t * Ls;
    const int64_t s = ((int64_t)t * N + idx[j]) * lf.words + w;
    // group g = j / GL is a contiguous [T][GL][row] block of the output
    const IT g = j / GLs, jj = j - g * GLs;
    const int64_t d = (((int64_t)g * lf.T + t) * GL + jj) * lf.words + w;
    if (lf.word_bytes == 4)
      static_cast<uint32_t*>(lf.dst)[d] = static_cast<const uint32_t*>(lf.src)[s];
    else
      static_cast<uint8_t*>(lf.dst)[d] = static_cast<const uint8_t*>(lf.src)[s];
  }
}

int stream_grid(int64_t n) {
  int64_t g = mippo::ceil_div(n, kThreads);
  if (g > mippo::kMaxStreamBlocks) g = mippo::kMaxStreamBlocks;
  return (int)(g < 1 ? 1 : g);
}

bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3) == 0; }

}  // namespace

extern "C" int mi_gather_cols(const void* src, const int64_t* idx, void* dst, int64_t T,
                              int64_t N, int64_t L, int64_t row_bytes, mi_stream_t stream) {
  MI_REQUIRE(T >= 0 && N >= 0 && L >= 0 && row_bytes >= 1, "mi_gather_cols: bad shape");
  if (T == 0 || L == 0) return 0;
  MI_REQUIRE(src && idx && dst && N >= 1, "mi_gather_cols: null pointer / empty source");
  hipStream_t st = mippo::as_stream(stream);
  if (row_bytes % 4 == 0 && aligned4(src) && aligned4(dst)) {
    const int64_t W = row_bytes / 4;
    hipLaunchKernelGGL(gather_cols_kernel<uint32_t>, dim3(stream_grid(T * L * W)),
                       dim3(kThreads), 0, st, static_cast<const uint32_t*>(src), idx,
                       static_cast<uint32_t*>(dst), T, N, L, W);
  } else {
    hipLaunchKernelGGL(gather_cols_kernel<uint8_t>, dim3(stream_grid(T * L * row_bytes)),
                       dim3(kThreads), 0, st, static_cast<const uint8_t*>(src), idx,
                       static_cast<uint8_t*>(dst), T, N, L, row_bytes);
  }
  return mippo::check_launch("mi_gather_cols");
}

extern "C" int mi_select_rows(const uint8_t* mask, const void* on_true, int64_t true_row_stride_bytes,
                              const void* on_false, void* out, int64_t B, int64_t row_bytes,
                              mi_stream_t stream) {
  MI_REQUIRE(B >= 0 && row_bytes >= 1, "mi_select_rows: bad shape");
  MI_REQUIRE(true_row_stride_bytes == 0 || true_row_stride_bytes == row_bytes,
             "mi_select_rows: on_true stride must be 0 (broadcast row) or row_bytes");
  if (B == 0) return 0;
  MI_REQUIRE(mask && on_true && on_false && out, "mi_select_rows: null pointer");
  hipStream_t st = mippo::as_stream(stream);
  if (row_bytes % 4 == 0 && aligned4(on_true) && aligned4(on_false) && aligned4(out)) {
    const int64_t W = row_bytes / 4;
    hipLaunchKernelGGL(select_rows_kernel<uint32_t>, dim3(stream_grid(B * W)), dim3(kThreads), 0,
                       st, mask, static_cast<const uint32_t*>(on_true), true_row_stride_bytes / 4,
                       static_cast<const uint32_t*>(on_false), static_cast<uint32_t*>(out), B, W);
  } else {
    hipLaunchKernelGGL(select_rows_kernel<uint8_t>, dim3(stream_grid(B * row_bytes)),
                       dim3(kThreads), 0, st, mask, static_cast<const uint8_t*>(on_true),
                       true_row_stride_bytes, static_cast<const uint8_t*>(on_false),
                       static_cast<uint8_t*>(out), B, row_bytes);
  }
  return mippo::check_launch("mi_select_rows");
}

extern "C" int mi_select_rows_multi(const uint8_t* mask, const void* const* on_true,
                                    const int64_t* true_row_stride_bytes,
                                    const void* const* on_false, void* const* out,
                                    const int64_t* row_bytes, int64_t n_leaves, int64_t B,
                                    mi_stream_t stream) {
  MI_REQUIRE(n_leaves >= 0 && n_leaves <= kMaxSelectLeaves && B >= 0,
             "mi_select_rows_multi: 0 <= n_leaves <= %d", kMaxSelectLeaves);
  if (n_leaves == 0 || B == 0) return 0;
  MI_REQUIRE(mask && on_true && on_false && out && row_bytes && true_row_stride_bytes,
             "mi_select_rows_multi: null pointer");
  SelectTable tab = {};
  int64_t max_words = 0;
  for (int64_t l = 0; l < n_leaves; ++l) {
    MI_REQUIRE(on_true[l] && on_false[l] && out[l] && row_bytes[l] >= 1,
               "mi_select_rows_multi: bad leaf %lld", (long long)l);
    MI_REQUIRE(true_row_stride_bytes[l] == 0 || true_row_stride_bytes[l] == row_bytes[l],
               "mi_select_rows_multi: on_true stride must be 0 or row_bytes");
    const bool w4 = row_bytes[l] % 4 == 0 && aligned4(on_true[l]) && aligned4(on_false[l]) &&
                    aligned4(out[l]);
    SelectLeaf& lf = tab.leaf[l];
    lf.on_true = on_true[l];
    lf.on_false = on_false[l];
    lf.out = out[l];
    lf.word_bytes = w4 ? 4 : 1;
    lf.words = row_bytes[l] / lf.word_bytes;
    lf.true_stride = true_row_stride_bytes[l] / lf.word_bytes;
    if (lf.words > max_words) max_words = lf.words;
  }
  dim3 grid((unsigned)stream_grid(B * max_words), (unsigned)n_leaves);
  hipLaunchKernelGGL(select_rows_multi_kernel, grid, dim3(kThreads), 0, mippo::as_stream(stream),
                     mask, tab, B);
  return mippo::check_launch("mi_select_rows_multi");
}

extern "C" int mi_gather_cols_multi(const void* const* src, void* const* dst, const int64_t* T,
                                    const int64_t* row_bytes, int64_t n_leaves,
                                    const int64_t* idx, int64_t N, int64_t L,
                                    int64_t group_len, mi_stream_t stream) {
  MI_REQUIRE(n_leaves >= 0 && n_leaves <= kMaxSelectLeaves && N >= 1 && L >= 0,
             "mi_gather_cols_multi: 0 <= n_leaves <= %d", kMaxSelectLeaves);
  if (n_leaves == 0 || L == 0) return 0;
  MI_REQUIRE(group_len >= 1 && L % group_len == 0,
             "mi_gather_cols_multi: group_len must divide L (L=%lld, group_len=%lld)",
             (long long)L, (long long)group_len);
  MI_REQUIRE(src && dst && T && row_bytes && idx, "mi_gather_cols_multi: null pointer");
  GatherTable tab = {};
  int64_t max_total = 0;
  for (int64_t l = 0; l < n_leaves; ++l) {
    MI_REQUIRE(src[l] && dst[l] && T[l] >= 1 && row_bytes[l] >= 1,
               "mi_gather_cols_multi: bad leaf %lld", (long long)l);
    const bool w4 = row_bytes[l] % 4 == 0 && aligned4(src[l]) && aligned4(dst[l]);
    GatherLeaf& lf = tab.leaf[l];
    lf.src = src[l];
    lf.dst = dst[l];
    lf.T = T[l];
    lf.word_bytes = w4 ? 4 : 1;
    lf.words = row_bytes[l] / lf.word_bytes;
    const int64_t tot = lf.T * L * lf.words;
    if (tot > max_total) max_total = tot;
  }
  dim3 grid((unsigned)stream_grid(max_total), (unsigned)n_leaves);
  // (the grid-stride loop's `i += stride` must not wrap: leave room for one stride)
  if (max_total < (1LL << 32) - (int64_t)grid.x * kThreads)
    hipLaunchKernelGGL(gather_cols_multi_kernel<uint32_t>, grid, dim3(kThreads), 0,
                       mippo::as_stream(stream), tab, idx, N, L, group_len);
  else
    hipLaunchKernelGGL(gather_cols_multi_kernel<uint64_t>, grid, dim3(kThreads), 0,
                       mippo::as_stream(stream), tab, idx, N, L, group_len);
  return mippo::check_launch("mi_gather_cols_multi");
}

extern "C" int mi_copy_multi(const void* const* src, void* const* dst, const int64_t* nbytes,
                             int64_t n_leaves, mi_stream_t stream) {
  MI_REQUIRE(n_leaves >= 0 && n_leaves <= kMaxSelectLeaves, "mi_copy_multi: 0 <= n_leaves <= %d",
             kMaxSelectLeaves);
  if (n_leaves == 0) return 0;
  MI_REQUIRE(src && dst && nbytes, "mi_copy_multi: null pointer");
  CopyTable tab = {};
  int64_t max_words = 0;
  for (int64_t l = 0; l < n_leaves; ++l) {
    MI_REQUIRE(nbytes[l] >= 0 && (nbytes[l] == 0 || (src[l] && dst[l])),
               "mi_copy_multi: bad leaf %lld", (long long)l);
    const bool w4 = nbytes[l] % 4 == 0 && aligned4(src[l]) && aligned4(dst[l]);
    CopyLeaf& lf = tab.leaf[l];
    lf.src = src[l];
    lf.dst = dst[l];
    lf.word_bytes = w4 ? 4 : 1;
    lf.words = nbytes[l] / lf.word_bytes;
    if (lf.words > max_words) max_words = lf.words;
  }
  if (max_words == 0) return 0;
  dim3 grid((unsigned)stream_grid(max_words), (unsigned)n_leaves);
  hipLaunchKernelGGL(copy_multi_kernel, grid, dim3(kThreads), 0, mippo::as_stream(stream), tab);
  return mippo::check_launch("mi_copy_multi");
}

static int episode_step_select_launch(
    const char* who, const int64_t* counter, const void* inner_done, int done_is_float,
    const uint8_t* inner_truncated, int64_t max_len, int64_t* counter_out,
    uint8_t* truncated_out, float* done_out, uint8_t* done_flag_out,
    const int64_t* reset_counter, const uint8_t* reset_truncated, const float* reset_done,
    int64_t* counter_sel, uint8_t* truncated_sel, float* done_sel, const void* const* on_true,
    const int64_t* true_row_stride_bytes, const void* const* on_false, void* const* out,
    const int64_t* row_bytes, int64_t n_leaves, int64_t B, const int64_t* mock_key,
    const int64_t* mock_count, int64_t mock_max_steps, const int64_t* produced,
    const int64_t* produced_col0, mi_stream_t stream) {
  MI_REQUIRE(n_leaves >= 0 && n_leaves <= kMaxSelectLeaves && B >= 0,
             "%s: 0 <= n_leaves <= %d", who, kMaxSelectLeaves);
  if (B == 0) return 0;
  MI_REQUIRE(counter && (inner_done || mock_key) && counter_out && truncated_out && done_out &&
                 reset_counter && reset_truncated && reset_done && counter_sel &&
                 truncated_sel && done_sel,
             "%s: null pointer", who);
  MI_REQUIRE(!mock_key || (mock_count && produced && produced_col0),
             "%s: the producer needs its step counter and the leaf kinds", who);
  MI_REQUIRE(n_leaves == 0 || (on_true && on_false && out && row_bytes && true_row_stride_bytes),
             "%s: null leaf table", who);
  EpisodeArgs e = {counter,       inner_done,    inner_truncated, max_len,       done_is_float,
                   counter_out,   truncated_out, done_out,        done_flag_out, reset_counter,
                   reset_truncated, reset_done,  counter_sel,     truncated_sel, done_sel,
                   mock_key,      mock_count,    mock_max_steps};
  SelectTable tab = {};
  int64_t max_words = 1;
  for (int64_t l = 0; l < n_leaves; ++l) {
    MI_REQUIRE(on_true[l] && on_false[l] && out[l] && row_bytes[l] >= 1, "%s: bad leaf %lld", who,
               (long long)l);
    MI_REQUIRE(true_row_stride_bytes[l] == 0 || true_row_stride_bytes[l] == row_bytes[l],
               "%s: on_true stride must be 0 or row_bytes", who);
    const bool w4 = row_bytes[l] % 4 == 0 && aligned4(on_true[l]) && aligned4(on_false[l]) &&
                    aligned4(out[l]);
    SelectLeaf& lf = tab.leaf[l];
    lf.on_true = on_true[l];
    lf.on_false = on_false[l];
    lf.out = out[l];
    lf.word_bytes = w4 ? 4 : 1;
    lf.words = row_bytes[l] / lf.word_bytes;
    lf.true_stride = true_row_stride_bytes[l] / lf.word_bytes;
    if (mock_key && produced[l]) {
      MI_REQUIRE(w4 && (produced[l] == 1 || (produced[l] == 2 && row_bytes[l] == 8)) &&
                     produced_col0[l] >= 0,
                 "%s: produced leaf %lld must be float32 columns (1) or the int64 counter (2)",
                 who, (long long)l);
      lf.produced = (int)produced[l];
      lf.col0 = (int)produced_col0[l];
    }
    if (lf.words > max_words) max_words = lf.words;
  }
  dim3 grid((unsigned)stream_grid(B * max_words), (unsigned)n_leaves + 1);
  hipLaunchKernelGGL(episode_select_kernel, grid, dim3(kThreads), 0, mippo::as_stream(stream), e,
                     tab, (int)n_leaves, B);
  return mippo::check_launch(who);
}

extern "C" int mi_episode_step_select(
    const int64_t* counter, const void* inner_done, int done_is_float,
    const uint8_t* inner_truncated, int64_t max_len, int64_t* counter_out,
    uint8_t* truncated_out, float* done_out, uint8_t* done_flag_out,
    const int64_t* reset_counter, const uint8_t* reset_truncated, const float* reset_done,
    int64_t* counter_sel, uint8_t* truncated_sel, float* done_sel, const void* const* on_true,
    const int64_t* true_row_stride_bytes, const void* const* on_false, void* const* out,
    const int64_t* row_bytes, int64_t n_leaves, int64_t B, mi_stream_t stream) {
  return episode_step_select_launch(
      "mi_episode_step_select", counter, inner_done, done_is_float, inner_truncated, max_len,
      counter_out, truncated_out, done_out, done_flag_out, reset_counter, reset_truncated,
      reset_done, counter_sel, truncated_sel, done_sel, on_true, true_row_stride_bytes, on_false,
      out, row_bytes, n_leaves, B, nullptr, nullptr, 0, nullptr, nullptr, stream);
}

// mi_episode_step_select with the synthetic env's own step (mi_mock_env_step) inside the
// launch: the inner done flag is step' >= mock_max_steps with step' = mock_count + 1, and the
// leaves marked in `produced` (1: float32 observation columns starting at produced_col0[l]
// of the env's flat draw; 2: the int64 step counter) are computed — stored to on_false[l]
// as the stepped state AND selected against on_true[l] — instead of read.  Bit-identical to
// mi_mock_env_step followed by mi_episode_step_select.
extern "C" int mi_mock_episode_step_select(
    const int64_t* mock_key, const int64_t* mock_count, int64_t mock_max_steps,
    const int64_t* produced, const int64_t* produced_col0, const int64_t* counter,
    const uint8_t* inner_truncated, int64_t max_len, int64_t* counter_out,
    uint8_t* truncated_out, float* done_out, uint8_t* done_flag_out,
    const int64_t* reset_counter, const uint8_t* reset_truncated, const float* reset_done,
    int64_t* counter_sel, uint8_t* truncated_sel, float* done_sel, const void* const* on_true,
    const int64_t* true_row_stride_bytes, const void* const* on_false, void* const* out,
    const int64_t* row_bytes, int64_t n_leaves, int64_t B, mi_stream_t stream) {
  MI_REQUIRE(mock_key && mock_count, "mi_mock_episode_step_select: null producer");
  return episode_step_select_launch(
      "mi_mock_episode_step_select", counter, nullptr, 0, inner_truncated, max_len, counter_out,
      truncated_out, done_out, done_flag_out, reset_counter, reset_truncated, reset_done,
      counter_sel, truncated_sel, done_sel, on_true, true_row_stride_bytes, on_false, out,
      row_bytes, n_leaves, B, mock_key, mock_count, mock_max_steps, produced, produced_col0,
      stream);
}

extern "C" int mi_stack_multi(const void* const* src, void* const* dst, const int64_t* nbytes,
                              int64_t n_leaves, int64_t T, mi_stream_t stream) {
  MI_REQUIRE(n_leaves >= 0 && n_leaves <= kMaxSelectLeaves && T >= 0 &&
                 n_leaves * T <= kMaxStackSegments,
             "mi_stack_multi: n_leaves <= %d and n_leaves * T <= %d", kMaxSelectLeaves,
             kMaxStackSegments);
  if (n_leaves == 0 || T == 0) return 0;
  MI_REQUIRE(src && dst && nbytes, "mi_stack_multi: null pointer");
  StackTable tab = {};
  tab.T = (int)T;
  int64_t max_words = 0;
  for (int64_t l = 0; l < n_leaves; ++l) {
    MI_REQUIRE(dst[l] && nbytes[l] >= 0, "mi_stack_multi: bad leaf %lld", (long long)l);
    bool w4 = nbytes[l] % 4 == 0 && aligned4(dst[l]);
    for (int64_t t = 0; t < T; ++t) {
      MI_REQUIRE(src[l * T + t] || nbytes[l] == 0, "mi_stack_multi: null source");
      w4 = w4 && aligned4(src[l * T + t]);
      tab.src[l * T + t] = src[l * T + t];
    }
    tab.dst[l] = dst[l];
    tab.word_bytes[l] = w4 ? 4 : 1;
    tab.words[l] = nbytes[l] / tab.word_bytes[l];
    if (tab.words[l] > max_words) max_words = tab.words[l];
  }
  if (max_words == 0) return 0;
  dim3 grid((unsigned)stream_grid(max_words), (unsigned)(n_leaves * T));
  hipLaunchKernelGGL(stack_multi_kernel, grid, dim3(kThreads), 0, mippo::as_stream(stream), tab);
  return mippo::check_launch("mi_stack_multi");
}
