// a9 (bf16 path) — the three GEMMs of a Dense layer as ONE "NT" kernel on bf16
// MFMA with fp32 accumulation (v_mfma_f32_16x16x32_bf16):
//
//     C[i][j] = sum_r A[i][r] * B[j][r]          A: [I][lda], B: [J][ldb], r contiguous
//
//   forward  Y[M,N]  = X[M,K]   . Wt[N,K]^T      (Wt = bf16 transposed shadow of the fp32 master)
//   dX       gX[M,K] = dZ[M,N]  . W[K,N]^T       (W  = bf16 shadow, flax [in,out] layout)
//   dW       gW[K,N] = Xt[K,M]  . dZt[N,M]^T     (both operands are TRANSPOSED activation
//                                                 copies written by the producing epilogues)
// so every MFMA operand fragment (8 consecutive reduce elements per lane) is one
// 16-byte ds_read_b128 from an r-contiguous LDS row.  Workgroup = 4 waves of 64;
// each wave owns TM x TN accumulator tiles of 16x16; A/B k-tiles of 64 are
// register-staged (global_load_dwordx4 -> ds_write_b128) one tile ahead of the
// MFMAs, LDS double-buffered, one barrier per k-tile; LDS rows are padded by
// 16 B to spread the ds_read_b128 lane groups over banks.
//
// Epilogues fuse what would otherwise be separate HBM passes:
//   FWD: + bias, activation, and up to three stores of the same tile — fp32 [M][N]
//        (chain output), bf16 [M][ld] (next layer's A operand / act' input) and
//        bf16 TRANSPOSED [N][ld] (the dW operand of the next layer);
//   DX : multiply by act'(previous layer output) and store dZ_prev as bf16 and
//        bf16 transposed (the next dX / dW operands) — the reference's chain rule
//        through `act(x @ W + b)` (nnx_ppo/networks/feedforward.py:42-51);
//   DW : split over the reduce dimension into fp32 slabs (+ bias row sums),
//        reduced in fixed order by reduce_slabs (dense.hip) — no float atomics.
#include "common.h"

namespace {

using bf16_t = __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

constexpr int kThreads = 256;
constexpr int BK = 64;         // reduce elements per k-tile
constexpr int LROW = BK + 8;   // LDS row length in bf16 (144 B: 16 B pad)

enum { EPI_FWD = 0, EPI_DX = 1, EPI_DW = 2 };

__device__ inline float act_fwd(float z, int act) {
  switch (act) {
    case MI_ACT_RELU: return fmaxf(z, 0.0f);
    case MI_ACT_TANH: return tanhf(z);
    case MI_ACT_SWISH: return z / (1.0f + expf(-z));
    default: return z;
  }
}

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

struct Epi {
  // FWD
  const float* bias;     // [J] or null
  int act;
  float* out_f32;        // [I][ld_f32] or null
  int64_t ld_f32;
  bf16_t* out_bf;        // [I][ld_bf] or null
  int64_t ld_bf;
  bf16_t* out_bft;       // [J][ld_bft] (transposed) or null
  int64_t ld_bft;
  bf16_t* aux_bf;        // FWD: pre-activation (swish) [I][ld_bf] or null
  // DX
  const bf16_t* prev;    // [I][ld_prev]: previous layer's output (or pre-act for swish), or null
  int64_t ld_prev;
  int prev_act;
  // DW
  float* slabs;          // [S][I*J + J]
  int64_t rows_per_split;
};

template <int WM, int WN, int TM, int TN, int EPI>
__global__ void __launch_bounds__(kThreads)
nt_gemm_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B,
               int64_t ldb, int64_t I, int64_t J, int64_t R, Epi ep) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int BM = WM * TM * 16;
  constexpr int BN = WN * TN * 16;
  constexpr int A_CH = BM * (BK / 8);                       // 16-byte chunks per A tile
  constexpr int B_CH = BN * (BK / 8);
  constexpr int A_PT = (A_CH + kThreads - 1) / kThreads;    // chunks per thread
  constexpr int B_PT = (B_CH + kThreads - 1) / kThreads;
  __shared__ __attribute__((aligned(16))) bf16_t As[2][BM][LROW];
  __shared__ __attribute__((aligned(16))) bf16_t Bs[2][BN][LROW];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int64_t i0 = (int64_t)blockIdx.x * BM;
  const int64_t j0 = (int64_t)blockIdx.y * BN;

  int64_t r_begin = 0, r_end = R;
  if (EPI == EPI_DW) {
    r_begin = (int64_t)blockIdx.z * ep.rows_per_split;
    r_end = r_begin + ep.rows_per_split < R ? r_begin + ep.rows_per_split : R;
  }

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[A_PT], rb[B_PT];
  auto load_tile = [&](int64_t r0) {
#pragma unroll
    for (int p = 0; p < A_PT; ++p) {
      const int c = tid + p * kThreads;
      const int row = c / (BK / 8), kc = c % (BK / 8);
      const int64_t gi = i0 + row, gr = r0 + kc * 8;
      ra[p] = u32x4{0u, 0u, 0u, 0u};
      if (c < A_CH && gi < I && gr < r_end)
        ra[p] = *reinterpret_cast<const u32x4*>(A + gi * lda + gr);
    }
#pragma unroll
    for (int p = 0; p < B_PT; ++p) {
      const int c = tid + p * kThreads;
      const int row = c / (BK / 8), kc = c % (BK / 8);
      const int64_t gj = j0 + row, gr = r0 + kc * 8;
      rb[p] = u32x4{0u, 0u, 0u, 0u};
      if (c < B_CH && gj < J && gr < r_end)
        rb[p] = *reinterpret_cast<const u32x4*>(B + gj * ldb + gr);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int p = 0; p < A_PT; ++p) {
      const int c = tid + p * kThreads;
      if (c < A_CH)
        *reinterpret_cast<u32x4*>(&As[buf][c / (BK / 8)][(c % (BK / 8)) * 8]) = ra[p];
    }
#pragma unroll
    for (int p = 0; p < B_PT; ++p) {
      const int c = tid + p * kThreads;
      if (c < B_CH)
        *reinterpret_cast<u32x4*>(&Bs[buf][c / (BK / 8)][(c % (BK / 8)) * 8]) = rb[p];
    }
  };

  float bias_sum = 0.0f;  // EPI_DW: row sums of B (= column sums of dZ)
  const bool do_bias = (EPI == EPI_DW) && blockIdx.x == 0;

  // NOTE: split boundaries are multiples of 8 (host guarantees), so a 16-byte
  // chunk never straddles r_end.
  load_tile(r_begin);
  store_tile(0);
  __syncthreads();
  int buf = 0;
  for (int64_t r0 = r_begin; r0 < r_end; r0 += BK) {
    const bool more = r0 + BK < r_end;
    if (more) load_tile(r0 + BK);
    if (do_bias && tid < BN) {
#pragma unroll 8
      for (int k = 0; k < BK; ++k) bias_sum += (float)Bs[buf][tid][k];
    }
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8 af[TM], bfr[TN];
      const int kof = ks * 32 + 8 * (lane >> 4);
#pragma unroll
      for (int a = 0; a < TM; ++a)
        af[a] = *reinterpret_cast<const bf16x8*>(
            &As[buf][(wm * TM + a) * 16 + (lane & 15)][kof]);
#pragma unroll
      for (int b = 0; b < TN; ++b)
        bfr[b] = *reinterpret_cast<const bf16x8*>(
            &Bs[buf][(wn * TN + b) * 16 + (lane & 15)][kof]);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- epilogue.  C/D map of the 16x16 MFMA: col = lane & 15, row = 4*(lane>>4) + e.
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int64_t j = j0 + (wn * TN + b) * 16 + (lane & 15);
      const int64_t ib = i0 + (wm * TM + a) * 16 + 4 * (lane >> 4);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[a][b][e];
      if (EPI == EPI_FWD) {
        if (j < J) {
          const float bj = ep.bias ? ep.bias[j] : 0.0f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int64_t i = ib + e;
            const float z = v[e] + bj;
            v[e] = act_fwd(z, ep.act);
            if (i < I) {
              if (ep.aux_bf) ep.aux_bf[i * ep.ld_bf + j] = (bf16_t)z;
              if (ep.out_f32) ep.out_f32[i * ep.ld_f32 + j] = v[e];
              if (ep.out_bf) ep.out_bf[i * ep.ld_bf + j] = (bf16_t)v[e];
            }
          }
        }
      } else if (EPI == EPI_DX) {
        if (j < J) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int64_t i = ib + e;
            if (i < I) {
              if (ep.prev && ep.prev_act != MI_ACT_NONE)
                v[e] *= act_grad((float)ep.prev[i * ep.ld_prev + j], ep.prev_act);
              if (ep.out_f32) ep.out_f32[i * ep.ld_f32 + j] = v[e];
              if (ep.out_bf) ep.out_bf[i * ep.ld_bf + j] = (bf16_t)v[e];
            }
          }
        }
      } else {  // EPI_DW
        if (j < J) {
          float* slab = ep.slabs + (int64_t)blockIdx.z * (I * J + J);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ib + e < I) slab[(ib + e) * J + j] = v[e];
        }
      }
      if (EPI != EPI_DW && ep.out_bft && j < J) {
        // transposed store: 4 consecutive i for one j -> 8 contiguous bytes
        bf16_t* dst = ep.out_bft + j * ep.ld_bft + ib;
        if (ib + 3 < I) {
          typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
          bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          *reinterpret_cast<bf16x4*>(dst) = pk;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ib + e < I) dst[e] = (bf16_t)v[e];
        }
      }
    }
  }
  if (do_bias && tid < BN && j0 + tid < J) {
    float* slab = ep.slabs + (int64_t)blockIdx.z * (I * J + J);
    slab[I * J + j0 + tid] = bias_sum;
  }
}

// fp32 [M][F] (optionally times act'(aux)) -> bf16 [M][ld] (zero-padded to ld)
// and bf16 transposed [F][ldt].
__global__ void __launch_bounds__(kThreads)
cast_pad_kernel(const float* __restrict__ x, const bf16_t* __restrict__ aux, int64_t ldaux,
                int act, bf16_t* __restrict__ out, int64_t ld, bf16_t* __restrict__ out_t,
                int64_t ldt, int64_t M, int64_t F) {
  const int64_t total = M * ld;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int64_t m = i / ld, f = i % ld;
    float v = f < F ? x[m * F + f] : 0.0f;
    if (aux && f < F) v *= act_grad((float)aux[m * ldaux + f], act);
    if (out) out[i] = (bf16_t)v;
    if (out_t && f < F) out_t[f * ldt + m] = (bf16_t)v;
  }
}

// fp32 master W [K][N] -> bf16 W [K][ldw] and bf16 Wt [N][ldwt], zero padded.
__global__ void __launch_bounds__(kThreads)
weights_to_bf16_kernel(const float* __restrict__ w, bf16_t* __restrict__ wb, int64_t ldw,
                       bf16_t* __restrict__ wt, int64_t ldwt, int64_t K, int64_t N) {
  const int64_t n1 = K * ldw, n2 = N * ldwt;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n1 + n2;
       i += (int64_t)gridDim.x * kThreads) {
    if (i < n1) {
      const int64_t k = i / ldw, n = i % ldw;
      wb[i] = (bf16_t)(n < N ? w[k * N + n] : 0.0f);
    } else {
      const int64_t q = i - n1;
      const int64_t n = q / ldwt, k = q % ldwt;
      wt[q] = (bf16_t)(k < K ? w[k * N + n] : 0.0f);
    }
  }
}

// ---------------------------------------------------------------------------
// Fused MLP chain forward: up to 8 Dense layers in ONE launch.  A workgroup owns
// 64 rows (envs / samples) and walks them through every layer; the activations
// never leave LDS (two ping-pong [64][W] bf16 buffers), only the layer weights
// stream through (bf16 W^T k-tiles from L2, register-staged + double-buffered as
// in nt_gemm_kernel).  Per-layer launch latency and the HBM round trip of every
// intermediate activation are what bound the per-layer path at this workload's
// shapes (K, N <= 512, M = 1k..30k), not MFMA rate.
// Inference: only the fp32 output of the last layer is stored.  Training: each
// layer also stores y (bf16), y^T (bf16, the dW operand) and, for swish, the
// pre-activation — exactly the buffers the per-layer backward kernels consume.
constexpr int CH_BM = 64;
constexpr int CH_BN = 64;
constexpr int CH_MAXL = 8;

struct ChainLayer {
  const bf16_t* wt;     // [N][ldwt] bf16 W^T
  const float* bias;    // [N] or null
  bf16_t* y_bf;         // [M][ldy] or null
  bf16_t* yt_bf;        // [N][ldyt] or null
  bf16_t* pre_bf;       // [M][ldy] or null
  int64_t ldwt, ldy, ldyt;
  int K, N, act;
};
struct Chain {
  ChainLayer layer[CH_MAXL];
  const float* x;       // [M][K0] fp32
  float* out;           // [M][N_last] fp32
  bf16_t* xt_bf;        // [K0][ldxt] transposed bf16 copy of the input, or null
  int64_t ldxt;
  int64_t M;
  int L;
};

struct Step {
  int l, n0, k0;
};

__device__ inline bool step_valid(const Chain& c, const Step& s) { return s.l < c.L; }

// (layer, n-tile, k-tile) in execution order: k fastest, then n, then layer.
__device__ inline Step step_next(const Chain& c, Step s) {
  const int Kp = (c.layer[s.l].K + 31) / 32 * 32;
  s.k0 += BK;
  if (s.k0 >= Kp) {
    s.k0 = 0;
    s.n0 += CH_BN;
    if (s.n0 >= c.layer[s.l].N) {
      s.n0 = 0;
      s.l += 1;
    }
  }
  return s;
}

// The whole trunk is ONE software pipeline over its flattened (layer, n-tile,
// k-tile) steps: the weight tile of step i+2 is in flight (global -> registers)
// while step i computes, and is parked in a 3-slot LDS ring one step later — so
// the L2 latency of a weight fetch is covered by two steps of MFMA work even
// across n-tile and LAYER boundaries (weights do not depend on activations).
template <int MAXW>
__global__ void __launch_bounds__(kThreads)
mlp_fwd_kernel(Chain c) {
  constexpr int AROW = MAXW + 8;
  constexpr int NSLOT = 3;
  __shared__ __attribute__((aligned(16))) bf16_t act[2][CH_BM][AROW];
  __shared__ __attribute__((aligned(16))) bf16_t Bs[NSLOT][CH_BN][LROW];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * CH_BM;
  const int K0 = c.layer[0].K;
  const int K0p = (K0 + 31) / 32 * 32;

  auto load_b = [&](const Step& s, u32x4 (&r)[2]) {
    const ChainLayer& ly = c.layer[s.l];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int cidx = tid + p * kThreads;
      const int row = cidx / 8, kc = cidx % 8;
      const int64_t gj = s.n0 + row, gr = s.k0 + kc * 8;
      r[p] = u32x4{0u, 0u, 0u, 0u};
      if (gj < ly.N && gr < ly.ldwt)
        r[p] = *reinterpret_cast<const u32x4*>(ly.wt + gj * ly.ldwt + gr);
    }
  };
  auto store_b = [&](int slot, const u32x4 (&r)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int cidx = tid + p * kThreads;
      *reinterpret_cast<u32x4*>(&Bs[slot][cidx / 8][(cidx % 8) * 8]) = r[p];
    }
  };

  Step s0 = {0, 0, 0};
  Step s1 = step_next(c, s0);
  u32x4 ra[2], rb[2];
  load_b(s0, ra);
  if (step_valid(c, s1)) load_b(s1, rb);

  // stage 0: input tile fp32 -> bf16 (zero padded to a multiple of 32 columns)
  for (int i = tid; i < CH_BM * K0p; i += kThreads) {
    const int row = i / K0p, k = i % K0p;
    const int64_t gi = i0 + row;
    const float v = (gi < c.M && k < K0) ? c.x[gi * K0 + k] : 0.0f;
    act[0][row][k] = (bf16_t)v;
    if (c.xt_bf && gi < c.M && k < K0) c.xt_bf[(int64_t)k * c.ldxt + gi] = (bf16_t)v;
  }
  store_b(0, ra);
  __syncthreads();

  f32x4 acc[4];
  // one pipeline step: compute `s` from ring slot `slot`; `ld` receives the tile of
  // step s+2, `st` (loaded one step ago, tile of step s+1) is parked in slot+1.
  auto run_step = [&](const Step& s, int slot, u32x4 (&ld)[2], const u32x4 (&st)[2]) {
    const ChainLayer& ly = c.layer[s.l];
    const int Kp = (ly.K + 31) / 32 * 32;
    const int cur = s.l & 1, nxt = cur ^ 1;
    const bool last = s.l == c.L - 1;
    const Step s1n = step_next(c, s);
    const Step s2n = step_valid(c, s1n) ? step_next(c, s1n) : s1n;
    if (step_valid(c, s1n) && step_valid(c, s2n)) load_b(s2n, ld);
    if (s.k0 == 0) {
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (s.n0 == 0 && !last) {
        const int Np = (ly.N + 31) / 32 * 32;
        if (Np != ly.N) {
          for (int i = tid; i < CH_BM * (Np - ly.N); i += kThreads)
            act[nxt][i / (Np - ly.N)][ly.N + i % (Np - ly.N)] = (bf16_t)0.0f;
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      if (s.k0 + ks * 32 < Kp) {
        const int kof = ks * 32 + 8 * (lane >> 4);
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(
            &act[cur][wave * 16 + (lane & 15)][s.k0 + kof]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const bf16x8 bfr =
              *reinterpret_cast<const bf16x8*>(&Bs[slot][b * 16 + (lane & 15)][kof]);
          acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr, acc[b], 0, 0, 0);
        }
      }
    }
    if (s.k0 + BK >= Kp) {  // last k-tile of this n-tile: epilogue
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int j = s.n0 + b * 16 + (lane & 15);
        if (j < ly.N) {
          const float bj = ly.bias ? ly.bias[j] : 0.0f;
          const int rb0 = wave * 16 + 4 * (lane >> 4);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = acc[b][e] + bj;
            v[e] = act_fwd(z, ly.act);
            const int64_t gi = i0 + rb0 + e;
            if (!last) act[nxt][rb0 + e][j] = (bf16_t)v[e];
            if (gi < c.M) {
              if (ly.pre_bf) ly.pre_bf[gi * ly.ldy + j] = (bf16_t)z;
              if (ly.y_bf) ly.y_bf[gi * ly.ldy + j] = (bf16_t)v[e];
              if (last) c.out[gi * ly.N + j] = v[e];
            }
          }
          if (ly.yt_bf) {
            bf16_t* dst = ly.yt_bf + (int64_t)j * ly.ldyt + i0 + rb0;
            if (i0 + rb0 + 3 < c.M) {
              typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
              bf16x4 pk = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
              *reinterpret_cast<bf16x4*>(dst) = pk;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (i0 + rb0 + e < c.M) dst[e] = (bf16_t)v[e];
            }
          }
        }
      }
    }
    if (step_valid(c, s1n)) store_b((slot + 1) % NSLOT, st);
    __syncthreads();
  };

  // steps alternate between the two register sets: (load -> ra, park rb), then
  // (load -> rb, park ra); at entry rb holds the tile of step 1.
  Step s = s0;
  int slot = 0;
  while (step_valid(c, s)) {
    run_step(s, slot, ra, rb);
    s = step_next(c, s);
    slot = (slot + 1) % NSLOT;
    if (!step_valid(c, s)) break;
    run_step(s, slot, rb, ra);
    s = step_next(c, s);
    slot = (slot + 1) % NSLOT;
  }
}

// ---------------------------------------------------------------------------
// Latency-optimised trunk for SMALL M (rollout step / bootstrap: 1k-8k rows).
// With so few rows the chip is filled by giving each workgroup only 16 rows (one
// MFMA row tile) and splitting every layer's OUTPUT COLUMNS over the 4 waves.
// A wave's weight fragments are then used by exactly one MFMA row tile, so they
// go global -> VGPR directly (no LDS staging, no per-k-tile barrier); they do not
// depend on activations, so the next chunk's fragments — across layer boundaries
// too — are in flight while the current chunk computes.  Activations (16 x W
// bf16, 8 KB) ping-pong in LDS; ONE barrier per layer.
constexpr int IF_BM = 16;
constexpr int IF_KC = 128;   // reduce elements per pipeline step (4 MFMA k-steps)

struct IStep {
  int l, p, kc;  // layer, column pass (256 columns per pass), k-chunk
};

__device__ inline IStep istep_next(const Chain& c, IStep s) {
  const int Kp = (c.layer[s.l].K + 31) / 32 * 32;
  s.kc += IF_KC;
  if (s.kc >= Kp) {
    s.kc = 0;
    s.p += 1;
    if (s.p * 256 >= c.layer[s.l].N) {
      s.p = 0;
      s.l += 1;
    }
  }
  return s;
}

struct BFrags {
  bf16x8 f[4][4];  // [k-step][column tile]
};

template <int MAXW>
__global__ void __launch_bounds__(kThreads)
mlp_infer_kernel(Chain c) {
  constexpr int AROW = MAXW + 8;
  __shared__ __attribute__((aligned(16))) bf16_t act[2][IF_BM][AROW];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * IF_BM;
  const int K0 = c.layer[0].K;
  const int K0p = (K0 + 31) / 32 * 32;

  // column tile b of this wave in pass p starts at column ((p*4 + b)*4 + wave) * 16
  auto load_frags = [&](const IStep& s, BFrags& B) {
    const ChainLayer& ly = c.layer[s.l];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int col = ((s.p * 4 + b) * 4 + wave) * 16 + (lane & 15);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int k = s.kc + ks * 32 + 8 * (lane >> 4);
        u32x4 r = u32x4{0u, 0u, 0u, 0u};
        if (col < ly.N && k < ly.ldwt)
          r = *reinterpret_cast<const u32x4*>(ly.wt + (int64_t)col * ly.ldwt + k);
        B.f[ks][b] = __builtin_bit_cast(bf16x8, r);
      }
    }
  };

  IStep s = {0, 0, 0};
  BFrags Ba, Bb;
  load_frags(s, Ba);

  for (int i = tid; i < IF_BM * K0p; i += kThreads) {
    const int row = i / K0p, k = i % K0p;
    const int64_t gi = i0 + row;
    act[0][row][k] = (bf16_t)((gi < c.M && k < K0) ? c.x[gi * K0 + k] : 0.0f);
  }
  __syncthreads();

  f32x4 acc[4];
  auto run_step = [&](const IStep& st, const BFrags& B, BFrags& Bnext) {
    const ChainLayer& ly = c.layer[st.l];
    const int Kp = (ly.K + 31) / 32 * 32;
    const int cur = st.l & 1, nxt = cur ^ 1;
    const bool last = st.l == c.L - 1;
    const IStep sn = istep_next(c, st);
    if (sn.l < c.L) load_frags(sn, Bnext);
    if (st.kc == 0) {
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (st.kc + ks * 32 < Kp) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(
            &act[cur][lane & 15][st.kc + ks * 32 + 8 * (lane >> 4)]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (((st.p * 4 + b) * 4 + wave) * 16 < ly.N)
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, B.f[ks][b], acc[b], 0, 0, 0);
        }
      }
    }
    if (st.kc + IF_KC >= Kp) {  // this wave's columns of this pass are complete
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int j = ((st.p * 4 + b) * 4 + wave) * 16 + (lane & 15);
        if (j < ly.N) {
          const float bj = ly.bias ? ly.bias[j] : 0.0f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int row = 4 * (lane >> 4) + e;
            const float v = act_fwd(acc[b][e] + bj, ly.act);
            if (!last) {
              act[nxt][row][j] = (bf16_t)v;
            } else if (i0 + row < c.M) {
              c.out[(i0 + row) * ly.N + j] = v;
            }
          }
        }
      }
      if (sn.l != st.l) {  // layer finished: zero the pad columns, then publish the buffer
        if (!last) {
          const int Np = (ly.N + 31) / 32 * 32;
          if (Np != ly.N) {
            for (int i = tid; i < IF_BM * (Np - ly.N); i += kThreads)
              act[nxt][i / (Np - ly.N)][ly.N + i % (Np - ly.N)] = (bf16_t)0.0f;
          }
        }
        __syncthreads();
      }
    }
  };

  while (s.l < c.L) {
    run_step(s, Ba, Bb);
    s = istep_next(c, s);
    if (s.l >= c.L) break;
    run_step(s, Bb, Ba);
    s = istep_next(c, s);
  }
}

int stream_grid(int64_t n) {
  int64_t g = mippo::ceil_div(n, kThreads);
  if (g > mippo::kMaxStreamBlocks) g = mippo::kMaxStreamBlocks;
  return (int)(g < 1 ? 1 : g);
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Tile configuration by output width J: wide outputs get 128x128 (2x2 waves of
// 4x4 tiles), medium 128x64, narrow heads (J <= 16) 128x16.
template <int EPI>
int launch_nt(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int64_t I, int64_t J,
              int64_t R, const Epi& ep, int64_t splits, hipStream_t st) {
  if (J > 64) {
    dim3 grid((unsigned)mippo::ceil_div(I, 128), (unsigned)mippo::ceil_div(J, 128),
              (unsigned)splits);
    hipLaunchKernelGGL((nt_gemm_kernel<2, 2, 4, 4, EPI>), grid, dim3(kThreads), 0, st, A, lda, B,
                       ldb, I, J, R, ep);
  } else if (J > 16) {
    dim3 grid((unsigned)mippo::ceil_div(I, 128), (unsigned)mippo::ceil_div(J, 64),
              (unsigned)splits);
    hipLaunchKernelGGL((nt_gemm_kernel<4, 1, 2, 4, EPI>), grid, dim3(kThreads), 0, st, A, lda, B,
                       ldb, I, J, R, ep);
  } else {
    dim3 grid((unsigned)mippo::ceil_div(I, 128), 1, (unsigned)splits);
    hipLaunchKernelGGL((nt_gemm_kernel<4, 1, 2, 1, EPI>), grid, dim3(kThreads), 0, st, A, lda, B,
                       ldb, I, J, R, ep);
  }
  return mippo::check_launch("nt_gemm_bf16");
}

int64_t dw_splits_bf16(int64_t M, int64_t K, int64_t N) {
  const int64_t bn = N > 64 ? 128 : (N > 16 ? 64 : 16);
  const int64_t tiles = mippo::ceil_div(K, 128) * mippo::ceil_div(N, bn);
  int64_t s = mippo::ceil_div((int64_t)2 * mippo::kNumCU, tiles);
  const int64_t max_s = mippo::ceil_div(M, 512);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  return s;
}

}  // namespace

extern "C" int mi_cast_pad_bf16(const float* x, const void* aux_bf, int64_t ldaux, int act,
                                void* out, int64_t ld, void* out_t, int64_t ldt, int64_t M,
                                int64_t F, mi_stream_t stream) {
  MI_REQUIRE(act >= MI_ACT_NONE && act <= MI_ACT_SWISH, "mi_cast_pad_bf16: bad act");
  MI_REQUIRE(M >= 0 && F >= 1 && ld >= F && ld % 8 == 0, "mi_cast_pad_bf16: bad shape");
  MI_REQUIRE(!out_t || (ldt >= M && ldt % 8 == 0), "mi_cast_pad_bf16: bad transposed ld");
  if (M == 0) return 0;
  MI_REQUIRE(x && (out || out_t), "mi_cast_pad_bf16: null pointer");
  hipLaunchKernelGGL(cast_pad_kernel, dim3(stream_grid(M * ld)), dim3(kThreads), 0,
                     mippo::as_stream(stream), x,
                     act == MI_ACT_NONE ? nullptr : static_cast<const bf16_t*>(aux_bf), ldaux, act,
                     static_cast<bf16_t*>(out), ld, static_cast<bf16_t*>(out_t), ldt, M, F);
  return mippo::check_launch("mi_cast_pad_bf16");
}

extern "C" int mi_weights_to_bf16(const float* w, void* w_bf, int64_t ldw, void* wt_bf,
                                  int64_t ldwt, int64_t K, int64_t N, mi_stream_t stream) {
  MI_REQUIRE(K >= 1 && N >= 1 && ldw >= N && ldw % 8 == 0 && ldwt >= K && ldwt % 8 == 0,
             "mi_weights_to_bf16: bad shape");
  MI_REQUIRE(w && w_bf && wt_bf, "mi_weights_to_bf16: null pointer");
  hipLaunchKernelGGL(weights_to_bf16_kernel, dim3(stream_grid(K * ldw + N * ldwt)),
                     dim3(kThreads), 0, mippo::as_stream(stream), w, static_cast<bf16_t*>(w_bf),
                     ldw, static_cast<bf16_t*>(wt_bf), ldwt, K, N);
  return mippo::check_launch("mi_weights_to_bf16");
}

extern "C" int mi_dense_fwd_bf16(const void* x_bf, int64_t ldx, const void* wt_bf, int64_t ldwt,
                                 const float* bias, float* y_f32, void* y_bf, int64_t ldy,
                                 void* yt_bf, int64_t ldyt, void* preact_bf, int64_t M,
                                 int64_t K, int64_t N, int act, mi_stream_t stream) {
  MI_REQUIRE(M >= 0 && K >= 1 && N >= 1, "mi_dense_fwd_bf16: bad shape");
  MI_REQUIRE(act >= MI_ACT_NONE && act <= MI_ACT_SWISH, "mi_dense_fwd_bf16: bad act");
  MI_REQUIRE(ldx % 8 == 0 && ldwt % 8 == 0 && ldx >= K && ldwt >= K,
             "mi_dense_fwd_bf16: operand ld must be a multiple of 8 and >= K");
  MI_REQUIRE(!y_bf || ldy >= N, "mi_dense_fwd_bf16: ldy < N");
  MI_REQUIRE(!yt_bf || (ldyt >= M && ldyt % 4 == 0), "mi_dense_fwd_bf16: bad ldyt");
  if (M == 0) return 0;
  MI_REQUIRE(x_bf && wt_bf && (y_f32 || y_bf), "mi_dense_fwd_bf16: null pointer");
  MI_REQUIRE(al16(x_bf) && al16(wt_bf), "mi_dense_fwd_bf16: operands must be 16-byte aligned");
  Epi ep = {};
  ep.bias = bias;
  ep.act = act;
  ep.out_f32 = y_f32;
  ep.ld_f32 = N;
  ep.out_bf = static_cast<bf16_t*>(y_bf);
  ep.ld_bf = ldy;
  ep.out_bft = static_cast<bf16_t*>(yt_bf);
  ep.ld_bft = ldyt;
  ep.aux_bf = static_cast<bf16_t*>(preact_bf);
  // reduce length = K rounded up to the operand padding (zeros beyond K)
  const int64_t R = mippo::ceil_div(K, 8) * 8;
  return launch_nt<EPI_FWD>(static_cast<const bf16_t*>(x_bf), ldx,
                            static_cast<const bf16_t*>(wt_bf), ldwt, M, N, R, ep, 1,
                            mippo::as_stream(stream));
}

extern "C" int mi_dense_bwd_dx_bf16(const void* dz_bf, int64_t lddz, const void* w_bf,
                                    int64_t ldw, const void* prev_bf, int64_t ldprev,
                                    int prev_act, float* gx_f32, void* gx_bf, int64_t ldgx,
                                    void* gxt_bf, int64_t ldgxt, int64_t M, int64_t K, int64_t N,
                                    mi_stream_t stream) {
  MI_REQUIRE(M >= 0 && K >= 1 && N >= 1, "mi_dense_bwd_dx_bf16: bad shape");
  MI_REQUIRE(prev_act >= MI_ACT_NONE && prev_act <= MI_ACT_SWISH, "mi_dense_bwd_dx_bf16: bad act");
  MI_REQUIRE(lddz % 8 == 0 && ldw % 8 == 0 && lddz >= N && ldw >= N,
             "mi_dense_bwd_dx_bf16: operand ld must be a multiple of 8 and >= N");
  MI_REQUIRE(!gx_bf || ldgx >= K, "mi_dense_bwd_dx_bf16: ldgx < K");
  MI_REQUIRE(!gxt_bf || (ldgxt >= M && ldgxt % 4 == 0), "mi_dense_bwd_dx_bf16: bad ldgxt");
  if (M == 0) return 0;
  MI_REQUIRE(dz_bf && w_bf && (gx_f32 || gx_bf), "mi_dense_bwd_dx_bf16: null pointer");
  MI_REQUIRE(prev_act == MI_ACT_NONE || prev_bf, "mi_dense_bwd_dx_bf16: prev output needed");
  MI_REQUIRE(al16(dz_bf) && al16(w_bf), "mi_dense_bwd_dx_bf16: operands must be 16-byte aligned");
  Epi ep = {};
  ep.out_f32 = gx_f32;
  ep.ld_f32 = K;
  ep.out_bf = static_cast<bf16_t*>(gx_bf);
  ep.ld_bf = ldgx;
  ep.out_bft = static_cast<bf16_t*>(gxt_bf);
  ep.ld_bft = ldgxt;
  ep.prev = static_cast<const bf16_t*>(prev_bf);
  ep.ld_prev = ldprev;
  ep.prev_act = prev_act;
  const int64_t R = mippo::ceil_div(N, 8) * 8;
  return launch_nt<EPI_DX>(static_cast<const bf16_t*>(dz_bf), lddz,
                           static_cast<const bf16_t*>(w_bf), ldw, M, K, R, ep, 1,
                           mippo::as_stream(stream));
}

extern "C" int64_t mi_dense_bwd_dw_bf16_workspace_bytes(int64_t M, int64_t K, int64_t N) {
  if (M < 0 || K < 1 || N < 1) return -EINVAL;
  return dw_splits_bf16(M, K, N) * (K * N + N) * (int64_t)sizeof(float);
}

extern "C" int mi_dense_bwd_dw_bf16(const void* xt_bf, int64_t ldxt, const void* dzt_bf,
                                    int64_t lddzt, float* g_w, float* g_b, void* workspace,
                                    int64_t M, int64_t K, int64_t N, int accumulate,
                                    mi_stream_t stream);

// reduce_slabs lives in dense.hip
namespace mippo {
int reduce_slabs(const float* slabs, float* g_w, float* g_b, int64_t S, int64_t KN, int64_t N,
                 int accumulate, hipStream_t st);
}

extern "C" int mi_dense_bwd_dw_bf16(const void* xt_bf, int64_t ldxt, const void* dzt_bf,
                                    int64_t lddzt, float* g_w, float* g_b, void* workspace,
                                    int64_t M, int64_t K, int64_t N, int accumulate,
                                    mi_stream_t stream) {
  MI_REQUIRE(M >= 1 && K >= 1 && N >= 1, "mi_dense_bwd_dw_bf16: bad shape");
  MI_REQUIRE(ldxt % 8 == 0 && lddzt % 8 == 0 && ldxt >= M && lddzt >= M,
             "mi_dense_bwd_dw_bf16: transposed operand ld must be a multiple of 8 and >= M");
  MI_REQUIRE(xt_bf && dzt_bf && g_w && workspace, "mi_dense_bwd_dw_bf16: null pointer");
  MI_REQUIRE(al16(xt_bf) && al16(dzt_bf), "mi_dense_bwd_dw_bf16: operands must be 16-byte aligned");
  hipStream_t st = mippo::as_stream(stream);
  const int64_t S = dw_splits_bf16(M, K, N);
  // split boundaries on multiples of 64 so 16-byte chunks never straddle a split
  const int64_t rows = mippo::ceil_div(mippo::ceil_div(M, S), 64) * 64;
  const int64_t S_eff = mippo::ceil_div(M, rows);
  Epi ep = {};
  ep.slabs = static_cast<float*>(workspace);
  ep.rows_per_split = rows;
  // reduce length: M rounded up to 8 (the transposed copies are zero padded to ld)
  const int64_t R = mippo::ceil_div(M, 8) * 8;
  int rc = launch_nt<EPI_DW>(static_cast<const bf16_t*>(xt_bf), ldxt,
                             static_cast<const bf16_t*>(dzt_bf), lddzt, K, N, R, ep, S_eff, st);
  if (rc) return rc;
  return mippo::reduce_slabs(ep.slabs, g_w, g_b, S_eff, K * N, N, accumulate, st);
}

extern "C" int mi_mlp_fwd_bf16(const float* x, int64_t M, int64_t L, const void* const* wt_bf,
                               const float* const* bias, const int64_t* dims,
                               const int64_t* acts, float* out, void* const* y_bf,
                               void* const* yt_bf, void* const* pre_bf, void* xt_bf,
                               mi_stream_t stream) {
  MI_REQUIRE(M >= 0 && L >= 1 && L <= CH_MAXL, "mi_mlp_fwd_bf16: 1 <= L <= %d", CH_MAXL);
  if (M == 0) return 0;
  MI_REQUIRE(x && wt_bf && dims && acts && out, "mi_mlp_fwd_bf16: null pointer");
  Chain c = {};
  c.x = x;
  c.out = out;
  c.M = M;
  c.L = (int)L;
  c.xt_bf = static_cast<bf16_t*>(xt_bf);
  const int64_t Mp = mippo::ceil_div(M, 8) * 8;
  c.ldxt = Mp;
  int maxw = 0;
  for (int l = 0; l < L; ++l) {
    const int64_t K = dims[l], N = dims[l + 1];
    MI_REQUIRE(K >= 1 && N >= 1 && K <= 512 && N <= 512,
               "mi_mlp_fwd_bf16: layer widths must be in [1, 512]");
    MI_REQUIRE(acts[l] >= MI_ACT_NONE && acts[l] <= MI_ACT_SWISH, "mi_mlp_fwd_bf16: bad act");
    MI_REQUIRE(wt_bf[l] && al16(wt_bf[l]), "mi_mlp_fwd_bf16: weights must be 16-byte aligned");
    ChainLayer& ly = c.layer[l];
    ly.wt = static_cast<const bf16_t*>(wt_bf[l]);
    ly.ldwt = mippo::ceil_div(K, 8) * 8;
    ly.bias = bias ? bias[l] : nullptr;
    ly.K = (int)K;
    ly.N = (int)N;
    ly.act = (int)acts[l];
    ly.ldy = mippo::ceil_div(N, 8) * 8;
    ly.ldyt = Mp;
    ly.y_bf = y_bf ? static_cast<bf16_t*>(y_bf[l]) : nullptr;
    ly.yt_bf = yt_bf ? static_cast<bf16_t*>(yt_bf[l]) : nullptr;
    ly.pre_bf = pre_bf ? static_cast<bf16_t*>(pre_bf[l]) : nullptr;
    const int w = (int)(mippo::ceil_div(K > N ? K : N, 32) * 32);
    if (w > maxw) maxw = w;
  }
  hipStream_t st = mippo::as_stream(stream);
  const bool training = y_bf || yt_bf || pre_bf || xt_bf;
  if (!training && M <= 16384) {
    // inference at small M: the latency-optimised 16-row kernel
    dim3 igrid((unsigned)mippo::ceil_div(M, IF_BM));
    if (maxw <= 256) {
      hipLaunchKernelGGL(mlp_infer_kernel<256>, igrid, dim3(kThreads), 0, st, c);
    } else {
      hipLaunchKernelGGL(mlp_infer_kernel<512>, igrid, dim3(kThreads), 0, st, c);
    }
    return mippo::check_launch("mi_mlp_fwd_bf16(infer)");
  }
  dim3 grid((unsigned)mippo::ceil_div(M, CH_BM));
  if (maxw <= 256) {
    hipLaunchKernelGGL(mlp_fwd_kernel<256>, grid, dim3(kThreads), 0, st, c);
  } else {
    hipLaunchKernelGGL(mlp_fwd_kernel<512>, grid, dim3(kThreads), 0, st, c);
  }
  return mippo::check_launch("mi_mlp_fwd_bf16");
}
