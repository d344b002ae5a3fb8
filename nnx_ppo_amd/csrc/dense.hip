// a9 — Dense layer forward / backward as fp32 MFMA GEMMs
// (reference: nnx_ppo/networks/feedforward.py:42-51, y = act(x @ W + b),
//  W: [in, out] as flax.nnx.Linear stores it; backward = what nnx.grad derives).
//
// fp32 path: v_mfma_f32_32x32x2_f32 — exact fp32 products with fp32
// accumulation (bit-for-bit a k-ordered fmaf chain), so this path is the
// tight-tolerance twin of the reference's fp32 XLA dot.  One workgroup = 4 waves
// (64 lanes each); every wave owns a 32 x (32*NACC) accumulator tile; operands
// are staged through LDS k-major so that each MFMA operand read is one
// conflict-free ds_read_b32 per lane (MFMA lane map: A[i = lane&31][k = lane>>5],
// B[k = lane>>5][j = lane&31]).  Layer shapes here are tiny in K and N
// (5..512) and huge in M (T*minibatch = 30 720), so tiles are tall: 128 x 64.
//
//   fwd :  Y[M,N]  = act(X[M,K] @ W[K,N] + b)          (aux = pre-activation for swish)
//   dX  :  gX[M,K] = (gY ⊙ act'(aux))[M,N] @ W^T       (act' fused into the A-operand load)
//   dW  :  gW[K,N] = X^T @ (gY ⊙ act'(aux)), gb = colsum(...)
//          split over M into per-slab partials, then a fixed-order reduction
//          (bitwise reproducible; no float atomics).
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kThreads = 256;
constexpr int BK = 16;
constexpr int PAD = 4;

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

// Generic block GEMM: C_tile += sum_r A(row, r) * B(r, col), r in [r_begin, r_end).
// A_RC / B_RC: operand memory is contiguous along the reduce index (true) or
// along the row / col index (false) — decides the thread -> element map of the
// staging loads so that global reads stay coalesced either way.
template <int WM, int WN, int NACC, bool A_RC, bool B_RC, class FA, class FB, class FS>
__device__ inline void block_gemm(int64_t r_begin, int64_t r_end, FA load_a, FB load_b,
                                  FS on_b_tile, f32x16 (&acc)[NACC]) {
  constexpr int BM = 32 * WM;
  constexpr int BN = 32 * NACC * WN;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  __shared__ float As[BK][BM + PAD];
  __shared__ float Bs[BK][BN + PAD];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: scalar unit
  const int wm = wave / WN;
  const int wn = wave % WN;
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

  for (int64_t r0 = r_begin; r0 < r_end; r0 += BK) {
#pragma unroll
    for (int i = tid; i < BM * BK; i += kThreads) {
      const int row = A_RC ? i / BK : i % BM;
      const int rr = A_RC ? i % BK : i / BM;
      const int64_t r = r0 + rr;
      As[rr][row] = r < r_end ? load_a(row, r) : 0.0f;
    }
#pragma unroll
    for (int i = tid; i < BN * BK; i += kThreads) {
      const int col = B_RC ? i / BK : i % BN;
      const int rr = B_RC ? i % BK : i / BN;
      const int64_t r = r0 + rr;
      Bs[rr][col] = r < r_end ? load_b(r, col) : 0.0f;
    }
    __syncthreads();
    on_b_tile(Bs);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const int k = 2 * kk + (lane >> 5);
      const float a = As[k][wm * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        const float b = Bs[k][(wn * NACC + j) * 32 + (lane & 31)];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
}

// C/D lane map of the 32x32 MFMA: col = lane & 31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
template <int WM, int WN, int NACC, class FE>
__device__ inline void for_each_out(const f32x16 (&acc)[NACC], FE emit) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int wm = wave / WN;
  const int wn = wave % WN;
#pragma unroll
  for (int j = 0; j < NACC; ++j) {
    const int col = (wn * NACC + j) * 32 + (lane & 31);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      emit(row, col, acc[j][e]);
    }
  }
}

struct NoTileHook {
  template <class T>
  __device__ void operator()(const T&) const {}
};

template <int NACC>
__global__ void __launch_bounds__(kThreads)
dense_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W,
                 const float* __restrict__ bias, float* __restrict__ Y,
                 float* __restrict__ preact, int64_t M, int64_t K, int64_t N, int act) {
  constexpr int WM = 4, WN = 1;
  constexpr int BM = 32 * WM, BN = 32 * NACC * WN;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int64_t n0 = (int64_t)blockIdx.y * BN;
  f32x16 acc[NACC];
  block_gemm<WM, WN, NACC, true, false>(
      0, K,
      [&](int row, int64_t k) { return m0 + row < M ? X[(m0 + row) * K + k] : 0.0f; },
      [&](int64_t k, int col) { return n0 + col < N ? W[k * N + n0 + col] : 0.0f; },
      NoTileHook{}, acc);
  for_each_out<WM, WN, NACC>(acc, [&](int row, int col, float v) {
    const int64_t m = m0 + row, n = n0 + col;
    if (m < M && n < N) {
      const float z = v + (bias ? bias[n] : 0.0f);
      if (preact) preact[m * N + n] = z;
      Y[m * N + n] = act_fwd(z, act);
    }
  });
}

template <int NACC>
__global__ void __launch_bounds__(kThreads)
dense_dx_kernel(const float* __restrict__ gY, const float* __restrict__ aux,
                const float* __restrict__ W, float* __restrict__ gX, int64_t M, int64_t K,
                int64_t N, int act) {
  constexpr int WM = 4, WN = 1;
  constexpr int BM = 32 * WM, BN = 32 * NACC * WN;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int64_t k0 = (int64_t)blockIdx.y * BN;
  f32x16 acc[NACC];
  block_gemm<WM, WN, NACC, true, true>(
      0, N,
      [&](int row, int64_t n) {
        const int64_t m = m0 + row;
        if (m >= M) return 0.0f;
        const float g = gY[m * N + n];
        return act == MI_ACT_NONE ? g : g * act_grad(aux[m * N + n], act);
      },
      [&](int64_t n, int col) { return k0 + col < K ? W[(k0 + col) * N + n] : 0.0f; },
      NoTileHook{}, acc);
  for_each_out<WM, WN, NACC>(acc, [&](int row, int col, float v) {
    const int64_t m = m0 + row, k = k0 + col;
    if (m < M && k < K) gX[m * K + k] = v;
  });
}

// grid = (K tiles, N tiles, S splits over M).  Slab layout: [S][K*N + N]
// (weights then bias column sums).
template <int NACC>
__global__ void __launch_bounds__(kThreads)
dense_dw_kernel(const float* __restrict__ X, const float* __restrict__ gY,
                const float* __restrict__ aux, float* __restrict__ slabs, int64_t M,
                int64_t K, int64_t N, int act, int64_t rows_per_split) {
  constexpr int WM = 2, WN = 2;
  constexpr int BM = 32 * WM, BN = 32 * NACC * WN;
  const int64_t k0 = (int64_t)blockIdx.x * BM;
  const int64_t n0 = (int64_t)blockIdx.y * BN;
  const int64_t s = blockIdx.z;
  const int64_t m_begin = s * rows_per_split;
  const int64_t m_end = m_begin + rows_per_split < M ? m_begin + rows_per_split : M;
  float* slab = slabs + s * (K * N + N);
  float colsum = 0.0f;
  const bool do_bias = blockIdx.x == 0;
  f32x16 acc[NACC];
  block_gemm<WM, WN, NACC, false, false>(
      m_begin, m_end,
      [&](int row, int64_t m) { return k0 + row < K ? X[m * K + k0 + row] : 0.0f; },
      [&](int64_t m, int col) {
        const int64_t n = n0 + col;
        if (n >= N) return 0.0f;
        const float g = gY[m * N + n];
        return act == MI_ACT_NONE ? g : g * act_grad(aux[m * N + n], act);
      },
      [&](const float (&Bs)[BK][BN + PAD]) {
        if (do_bias && threadIdx.x < BN) {
#pragma unroll
          for (int rr = 0; rr < BK; ++rr) colsum += Bs[rr][threadIdx.x];
        }
      },
      acc);
  for_each_out<WM, WN, NACC>(acc, [&](int row, int col, float v) {
    const int64_t k = k0 + row, n = n0 + col;
    if (k < K && n < N) slab[k * N + n] = v;
  });
  if (do_bias && threadIdx.x < BN && n0 + threadIdx.x < N)
    slab[K * N + n0 + threadIdx.x] = colsum;
}

// out[i] (+)= sum_s slabs[s][i], fixed order.  Loads are issued 8 slabs at a time
// (independent addresses) so the sum is bandwidth- rather than latency-paced.
__global__ void __launch_bounds__(kThreads)
reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ gW,
                    float* __restrict__ gb, int64_t S, int64_t KN, int64_t N,
                    int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  const int64_t stride = KN + N;
  if (i >= stride) return;
  float v = 0.0f;
  int64_t s = 0;
  for (; s + 8 <= S; s += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = slabs[(s + u) * stride + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; s < S; ++s) v += slabs[s * stride + i];
  if (i < KN) {
    gW[i] = accumulate ? gW[i] + v : v;
  } else if (gb) {
    gb[i - KN] = accumulate ? gb[i - KN] + v : v;
  }
}

int64_t dw_splits(int64_t M, int64_t K, int64_t N) {
  // enough (tile x split) workgroups to cover the chip a few times over, with
  // at least 256 rows per split so the slab traffic stays small.
  const int64_t tiles = mippo::ceil_div(K, 64) * mippo::ceil_div(N, 64);
  int64_t s = mippo::ceil_div((int64_t)4 * mippo::kNumCU, tiles);
  const int64_t max_s = mippo::ceil_div(M, 256);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 65535) s = 65535;
  return s;
}

}  // namespace

namespace {
struct RedProblem {
  const float* slabs;
  float* gw;
  float* gb;
  int64_t S, KN, N;
};
struct RedTable {
  RedProblem p[8];
};

// blockIdx.y = problem; same fixed-order sum as reduce_slabs_kernel.
__global__ void __launch_bounds__(kThreads)
reduce_slabs_grouped_kernel(RedTable tab, int accumulate) {
  const RedProblem pr = tab.p[blockIdx.y];
  const int64_t stride = pr.KN + pr.N;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < stride;
       i += (int64_t)gridDim.x * kThreads) {
    float v = 0.0f;
    int64_t s = 0;
    for (; s + 16 <= pr.S; s += 16) {  // 16 loads in flight: S = 32..64 is 2..4 round trips
      float t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = pr.slabs[(s + u) * stride + i];
#pragma unroll
      for (int u = 0; u < 16; ++u) v += t[u];
    }
    if (s < pr.S) {  // the remainder as ONE predicated batch (it was one round trip per slab)
      float t[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) t[u] = s + u < pr.S ? pr.slabs[(s + u) * stride + i] : 0.0f;
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (s + u < pr.S) v += t[u];
    }
    if (i < pr.KN) {
      pr.gw[i] = accumulate ? pr.gw[i] + v : v;
    } else if (pr.gb) {
      pr.gb[i - pr.KN] = accumulate ? pr.gb[i - pr.KN] + v : v;
    }
  }
}
}  // namespace

namespace mippo {
int reduce_slabs_grouped(int n, const float* const* slabs, const int64_t* S, const int64_t* KN,
                         const int64_t* N, float* const* g_w, float* const* g_b, int accumulate,
                         hipStream_t st) {
  RedTable tab = {};
  int64_t max_stride = 0;
  for (int l = 0; l < n; ++l) {
    tab.p[l] = RedProblem{slabs[l], g_w[l], g_b ? g_b[l] : nullptr, S[l], KN[l], N[l]};
    if (KN[l] + N[l] > max_stride) max_stride = KN[l] + N[l];
  }
  int64_t gx = ceil_div(max_stride, kThreads);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(reduce_slabs_grouped_kernel, dim3((unsigned)gx, (unsigned)n), dim3(kThreads),
                     0, st, tab, accumulate);
  return check_launch("reduce_slabs_grouped");
}

int reduce_slabs(const float* slabs, float* g_w, float* g_b, int64_t S, int64_t KN, int64_t N,
                 int accumulate, hipStream_t st) {
  const int64_t total = KN + N;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)ceil_div(total, kThreads)),
                     dim3(kThreads), 0, st, slabs, g_w, g_b, S, KN, N, accumulate);
  return check_launch("reduce_slabs");
}
}  // namespace mippo

extern "C" int mi_dense_fwd_f32(const float* x, const float* w, const float* bias, float* y,
                                float* preact, int64_t M, int64_t K, int64_t N, int act,
                                mi_stream_t stream) {
  MI_REQUIRE(M >= 0 && K >= 1 && N >= 1, "mi_dense_fwd_f32: bad shape M=%lld K=%lld N=%lld",
             (long long)M, (long long)K, (long long)N);
  MI_REQUIRE(act >= MI_ACT_NONE && act <= MI_ACT_SWISH, "mi_dense_fwd_f32: bad act %d", act);
  if (M == 0) return 0;
  MI_REQUIRE(x && w && y, "mi_dense_fwd_f32: null pointer");
  hipStream_t st = mippo::as_stream(stream);
  const int64_t gm = mippo::ceil_div(M, 128);
  MI_REQUIRE(gm <= 0x7fffffffLL, "mi_dense_fwd_f32: M too large");
  if (N > 32) {
    dim3 grid((unsigned)gm, (unsigned)mippo::ceil_div(N, 64));
    hipLaunchKernelGGL(dense_fwd_kernel<2>, grid, dim3(kThreads), 0, st, x, w, bias, y, preact,
                       M, K, N, act);
  } else {
    dim3 grid((unsigned)gm, 1);
    hipLaunchKernelGGL(dense_fwd_kernel<1>, grid, dim3(kThreads), 0, st, x, w, bias, y, preact,
                       M, K, N, act);
  }
  return mippo::check_launch("mi_dense_fwd_f32");
}

extern "C" int mi_dense_bwd_dx_f32(const float* g_y, const float* aux, const float* w,
                                   float* g_x, int64_t M, int64_t K, int64_t N, int act,
                                   mi_stream_t stream) {
  MI_REQUIRE(M >= 0 && K >= 1 && N >= 1, "mi_dense_bwd_dx_f32: bad shape");
  MI_REQUIRE(act >= MI_ACT_NONE && act <= MI_ACT_SWISH, "mi_dense_bwd_dx_f32: bad act");
  if (M == 0) return 0;
  MI_REQUIRE(g_y && w && g_x && (aux || act == MI_ACT_NONE), "mi_dense_bwd_dx_f32: null pointer");
  hipStream_t st = mippo::as_stream(stream);
  const int64_t gm = mippo::ceil_div(M, 128);
  MI_REQUIRE(gm <= 0x7fffffffLL, "mi_dense_bwd_dx_f32: M too large");
  if (K > 32) {
    dim3 grid((unsigned)gm, (unsigned)mippo::ceil_div(K, 64));
    hipLaunchKernelGGL(dense_dx_kernel<2>, grid, dim3(kThreads), 0, st, g_y, aux, w, g_x, M, K,
                       N, act);
  } else {
    dim3 grid((unsigned)gm, 1);
    hipLaunchKernelGGL(dense_dx_kernel<1>, grid, dim3(kThreads), 0, st, g_y, aux, w, g_x, M, K,
                       N, act);
  }
  return mippo::check_launch("mi_dense_bwd_dx_f32");
}

extern "C" int64_t mi_dense_bwd_dw_workspace_bytes(int64_t M, int64_t K, int64_t N) {
  if (M < 0 || K < 1 || N < 1) return -EINVAL;
  return dw_splits(M, K, N) * (K * N + N) * (int64_t)sizeof(float);
}

extern "C" int mi_dense_bwd_dw_f32(const float* x, const float* g_y, const float* aux,
                                   float* g_w, float* g_b, void* workspace, int64_t M,
                                   int64_t K, int64_t N, int act, int accumulate,
                                   mi_stream_t stream) {
  MI_REQUIRE(M >= 1 && K >= 1 && N >= 1, "mi_dense_bwd_dw_f32: bad shape");
  MI_REQUIRE(act >= MI_ACT_NONE && act <= MI_ACT_SWISH, "mi_dense_bwd_dw_f32: bad act");
  MI_REQUIRE(x && g_y && g_w && workspace && (aux || act == MI_ACT_NONE),
             "mi_dense_bwd_dw_f32: null pointer");
  hipStream_t st = mippo::as_stream(stream);
  const int64_t S = dw_splits(M, K, N);
  const int64_t rows = mippo::ceil_div(M, S);
  float* slabs = static_cast<float*>(workspace);
  dim3 grid((unsigned)mippo::ceil_div(K, 64), (unsigned)mippo::ceil_div(N, 64), (unsigned)S);
  MI_REQUIRE(grid.y <= 65535, "mi_dense_bwd_dw_f32: N too large");
  hipLaunchKernelGGL(dense_dw_kernel<1>, grid, dim3(kThreads), 0, st, x, g_y, aux, slabs, M, K, N,
                     act, rows);
  int rc = mippo::check_launch("mi_dense_bwd_dw_f32(partial)");
  if (rc) return rc;
  return mippo::reduce_slabs(slabs, g_w, g_b, S, K * N, N, accumulate, st);
}
