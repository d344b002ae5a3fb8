// a9 (bf16 path) — whole MLP trunks in one launch (see gemm_bf16.hip for the
// per-layer kernels and the operand conventions).
#include "bf16_common.h"

namespace {

using namespace mippo_bf16;

// ---------------------------------------------------------------------------
// Fused MLP chain forward: up to 8 Dense layers in ONE launch.  A workgroup owns
// 64 rows (envs / samples) and walks them through every layer; the activations
// never leave LDS (two ping-pong [64][W] bf16 buffers), only the layer weights
// stream through (bf16 W^T k-tiles from L2, register-staged + double-buffered as
// in nt_gemm_kernel).  Per-layer launch latency and the HBM round trip of every
// intermediate activation are what bound the per-layer path at this workload's
// shapes (K, N <= 512, M = 1k..30k), not MFMA rate.
// Inference: only the fp32 output of the last layer is stored.  Training: each
// layer also stores y (bf16 row-major, copied out of the LDS activation buffer
// in coalesced 16-byte rows) and, for swish, the pre-activation — exactly the
// buffers the per-layer backward kernels consume.
constexpr int CH_BM = 64;
constexpr int CH_BN = 64;
constexpr int CH_MAXL = 8;

struct ChainLayer {
  const bf16_t* wt;     // [N][ldwt] bf16 W^T
  const float* bias;    // [N] or null
  bf16_t* y_bf;         // [M][ldy] or null
  bf16_t* pre_bf;       // [M][ldy] or null
  int64_t ldwt, ldy;
  int K, N, act;
};
struct Chain {
  ChainLayer layer[CH_MAXL];
  const float* x;       // [M][K0] fp32
  float* out;           // [M][N_last] fp32
  bf16_t* x_bf;         // [M][ldx] bf16 copy of the input (dW operand of layer 0), or null
  int64_t ldx;
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
  }
  store_b(0, ra);
  __syncthreads();

  // coalesced copy of a published activation buffer to its bf16 global image
  auto flush_act = [&](int bufi, bf16_t* dst, int64_t ld) {
    const int nch = (int)(ld / 8);
    for (int cidx = tid; cidx < CH_BM * nch; cidx += kThreads) {
      const int row = cidx / nch, cc = cidx % nch;
      const int64_t gi = i0 + row;
      if (gi < c.M)
        *reinterpret_cast<u32x4*>(dst + gi * ld + cc * 8) =
            *reinterpret_cast<const u32x4*>(&act[bufi][row][cc * 8]);
    }
  };

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
    if (s.k0 == 0 && s.n0 == 0) {
      // act[cur] is complete and published: copy it out if the backward needs it
      if (s.l == 0) {
        if (c.x_bf) flush_act(cur, c.x_bf, c.ldx);
      } else if (c.layer[s.l - 1].y_bf) {
        flush_act(cur, c.layer[s.l - 1].y_bf, c.layer[s.l - 1].ldy);
      }
    }
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
              if (last) {
                c.out[gi * ly.N + j] = v[e];
                if (ly.y_bf) ly.y_bf[gi * ly.ldy + j] = (bf16_t)v[e];
              }
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

}  // namespace

extern "C" int mi_mlp_fwd_bf16(const float* x, int64_t M, int64_t L, const void* const* wt_bf,
                               const float* const* bias, const int64_t* dims,
                               const int64_t* acts, float* out, void* const* y_bf,
                               void* const* pre_bf, void* x_bf, mi_stream_t stream) {
  MI_REQUIRE(M >= 0 && L >= 1 && L <= CH_MAXL, "mi_mlp_fwd_bf16: 1 <= L <= %d", CH_MAXL);
  if (M == 0) return 0;
  MI_REQUIRE(x && wt_bf && dims && acts && out, "mi_mlp_fwd_bf16: null pointer");
  Chain c = {};
  c.x = x;
  c.out = out;
  c.M = M;
  c.L = (int)L;
  c.x_bf = static_cast<bf16_t*>(x_bf);
  c.ldx = mippo::ceil_div(dims[0], 8) * 8;
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
    ly.y_bf = y_bf ? static_cast<bf16_t*>(y_bf[l]) : nullptr;
    ly.pre_bf = pre_bf ? static_cast<bf16_t*>(pre_bf[l]) : nullptr;
    const int w = (int)(mippo::ceil_div(K > N ? K : N, 32) * 32);
    if (w > maxw) maxw = w;
  }
  hipStream_t st = mippo::as_stream(stream);
  const bool training = y_bf || pre_bf || x_bf;
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
