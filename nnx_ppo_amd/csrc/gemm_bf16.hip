// a9 (bf16 path) — the three GEMMs of a Dense layer on bf16 MFMA with fp32
// accumulation (v_mfma_f32_16x16x32_bf16).  Every activation is kept ONCE, as a
// row-major bf16 matrix [M][pad8 N]; master weights stay fp32, with two bf16
// shadows (W [K][pad8 N] and W^T [N][pad8 K]).
//
//   forward  Y[M,N]  = X[M,K]  . Wt[N,K]^T   "NT": both operands reduce-contiguous,
//   dX       gX[M,K] = dZ[M,N] . W[K,N]^T          fragments are one ds_read_b128
//   dW       gW[K,N] = X[M,K]^T . dZ[M,N]    "TN": both operands are REDUCE-MAJOR
//            (the reduce index m is the row index of both).  The tiles are staged
//            into LDS as they lie in memory (coalesced 16-byte chunks) and the MFMA
//            fragments are gathered by gfx950's transposing LDS read
//            (ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block
//            delivered column-major), so no transposed copy of any activation is
//            ever written to HBM.
//
// Workgroup = 4 waves of 64; each wave owns TM x TN accumulator tiles of 16x16;
// k-tiles of 64 are register-staged (global_load_dwordx4 -> ds_write_b128) one
// tile ahead of the MFMAs, LDS double-buffered, one barrier per k-tile.
//
// Epilogues fuse what would otherwise be separate HBM passes, and stage the
// output tile through LDS so global stores are coalesced 16-byte rows (the MFMA
// accumulator layout holds 4 rows x 1 column per lane — storing it directly costs
// 2-byte scattered stores, which is what bounded the first version of this kernel):
//   FWD: + bias, activation -> bf16 Y (next layer's operand and act' input), optional
//        fp32 Y (chain output), optional bf16 pre-activation (swish backward);
//   DX : x act'(previous layer output) -> bf16 dZ_prev — the reference's chain rule
//        through `act(x @ W + b)` (nnx_ppo/networks/feedforward.py:42-51);
//   DW : split over the reduce dimension into fp32 slabs (+ bias column sums),
//        reduced in fixed order by reduce_slabs (dense.hip) — no float atomics.
#include <stdlib.h>

#include "bf16_common.h"
#include "gemm_epi.h"

namespace {

using namespace mippo_bf16;

using mippo_gemm::Epi;
using mippo_gemm::EPI_FWD;
using mippo_gemm::EPI_DX;

template <int WM, int WN, int TM, int TN, int EPI>
__global__ void __launch_bounds__(kThreads)
nt_gemm_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B,
               int64_t ldb, int64_t I, int64_t J, int64_t R, Epi ep) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int BM = WM * TM * 16;
  constexpr int BN = WN * TN * 16;
  constexpr int A_CH = BM * (BK / 8);  // 16-byte chunks per A tile
  constexpr int B_CH = BN * (BK / 8);
  constexpr int A_PT = (A_CH + kThreads - 1) / kThreads;
  constexpr int B_PT = (B_CH + kThreads - 1) / kThreads;
  constexpr int CROW = BN + 8;  // staged output tile row (bf16)
  constexpr int LOOP_BYTES = 2 * (BM + BN) * LROW * 2;
  constexpr int TILE_BYTES = 2 * BM * CROW * 2;
  constexpr int SMEM = LOOP_BYTES > TILE_BYTES ? LOOP_BYTES : TILE_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  auto As = reinterpret_cast<bf16_t(*)[BM][LROW]>(smem);
  auto Bs = reinterpret_cast<bf16_t(*)[BN][LROW]>(smem + 2 * BM * LROW * 2);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: scalar unit
  const int wm = wave / WN, wn = wave % WN;
  const int64_t i0 = (int64_t)blockIdx.x * BM;
  const int64_t j0 = (int64_t)blockIdx.y * BN;

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
      if (c < A_CH && gi < I && gr < R)
        ra[p] = *reinterpret_cast<const u32x4*>(A + gi * lda + gr);
    }
#pragma unroll
    for (int p = 0; p < B_PT; ++p) {
      const int c = tid + p * kThreads;
      const int row = c / (BK / 8), kc = c % (BK / 8);
      const int64_t gj = j0 + row, gr = r0 + kc * 8;
      rb[p] = u32x4{0u, 0u, 0u, 0u};
      if (c < B_CH && gj < J && gr < R)
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

  load_tile(0);
  store_tile(0);
  __syncthreads();
  int buf = 0;
  for (int64_t r0 = 0; r0 < R; r0 += BK) {
    const bool more = r0 + BK < R;
    if (more) load_tile(r0 + BK);
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

  // ---- epilogue: the loop buffers are dead; stage the output tile in LDS ----
  auto tC = reinterpret_cast<bf16_t(*)[CROW]>(smem);                  // [BM][CROW]
  auto tZ = reinterpret_cast<bf16_t(*)[CROW]>(smem + BM * CROW * 2);  // [BM][CROW]
  constexpr int C_CH = BM * (BN / 8);  // 16-byte chunks of the output tile
  const bool use_prev = EPI == EPI_DX && ep.prev && ep.prev_act != MI_ACT_NONE;
  if (use_prev) {
    for (int c = tid; c < C_CH; c += kThreads) {
      const int row = c / (BN / 8), cc = c % (BN / 8);
      const int64_t gi = i0 + row, gj = j0 + cc * 8;
      u32x4 v = u32x4{0u, 0u, 0u, 0u};
      if (gi < I && gj < ep.ld_prev)
        v = *reinterpret_cast<const u32x4*>(ep.prev + gi * ep.ld_prev + gj);
      *reinterpret_cast<u32x4*>(&tC[row][cc * 8]) = v;
    }
    __syncthreads();
  }
  // C/D map of the 16x16 MFMA: col = lane & 15, row = 4*(lane>>4) + e.
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = (wn * TN + b) * 16 + (lane & 15);
      const int64_t gj = j0 + col;
      const bool jin = gj < J;
      const float bj = (EPI == EPI_FWD && ep.bias && jin) ? ep.bias[gj] : 0.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = (wm * TM + a) * 16 + 4 * (lane >> 4) + e;
        const int64_t gi = i0 + row;
        float v = acc[a][b][e];
        if (EPI == EPI_FWD) {
          const float z = jin ? v + bj : 0.0f;
          v = jin ? act_fwd(z, ep.act) : 0.0f;
          if (ep.aux_bf) tZ[row][col] = (bf16_t)z;
        } else {
          if (use_prev) v *= act_grad((float)tC[row][col], ep.prev_act);
          if (!jin) v = 0.0f;
        }
        if (ep.out_f32 && jin && gi < I) ep.out_f32[gi * ep.ld_f32 + gj] = v;
        tC[row][col] = (bf16_t)v;
      }
    }
  }
  if (ep.out_bf || ep.aux_bf) {
    __syncthreads();
    for (int c = tid; c < C_CH; c += kThreads) {
      const int row = c / (BN / 8), cc = c % (BN / 8);
      const int64_t gi = i0 + row, gj = j0 + cc * 8;
      if (gi < I && gj < ep.ld_bf) {
        if (ep.out_bf)
          *reinterpret_cast<u32x4*>(ep.out_bf + gi * ep.ld_bf + gj) =
              *reinterpret_cast<const u32x4*>(&tC[row][cc * 8]);
        if (ep.aux_bf)
          *reinterpret_cast<u32x4*>(ep.aux_bf + gi * ep.ld_bf + gj) =
              *reinterpret_cast<const u32x4*>(&tZ[row][cc * 8]);
      }
    }
  }
}

// dW: C[i][j] = sum_m A[m][i] * B[m][j] over the split's rows; A = X [M][lda],
// B = dZ [M][ldb], both reduce-major.  Slab layout: [S][I*J + J].
// One dW problem of a grouped launch.  All layers of an MLP trunk (of one tile
// class) go out in ONE launch: their dW GEMMs are independent once the dX chain
// has produced every dZ, and together they fill the chip (a single 64x64 layer
// only makes 60 workgroups).
struct DwProblem {
  const bf16_t* A;   // X  [M][lda]
  const bf16_t* B;   // dZ [M][ldb]
  float* slabs;      // [S][I*J + J]
  int64_t lda, ldb, I, J;
  int z_begin;       // first blockIdx.z of this problem (its splits follow)
};
constexpr int kMaxDwProblems = 8;
struct DwTable {
  DwProblem p[kMaxDwProblems];
  int n;
  int gx, gy;        // tile grid of the largest problem; the launch is 1-D: gx * gy * splits
  int64_t M, rows_per_split;
};

// Workgroups are dispatched round-robin over the 8 XCDs, each with its own L2.
// Map the hardware id so that CONSECUTIVE logical ids share an XCD: the tiles of one
// row split (same X / dZ rows, read by gx * gy workgroups) then hit in one L2
// instead of being fetched into several.
__device__ inline unsigned xcd_swizzle(unsigned hw, unsigned total) {
  constexpr unsigned kXcd = 8;
  const unsigned chunk = total / kXcd, rem = total % kXcd;
  const unsigned xcd = hw % kXcd, idx = hw / kXcd;
  return xcd * chunk + (xcd < rem ? xcd : rem) + idx;
}

template <int WM, int WN, int TM, int TN>
constexpr int dw_lds_bytes() {
  return 2 * BK * ((WM * TM * 16 + 16) + (WN * TN * 16 + 16)) * (int)sizeof(bf16_t);
}

// -DMIPPO_TRACE (tools/trace_policy.py): per-workgroup phase stamps of the dW kernel
// (start, first tile in LDS, loop done, end; wall clock start / end; class; iterations).
#ifdef MIPPO_TRACE
constexpr int TRD_EV = 12, TRD_WG = 2048;
__device__ unsigned long long g_trace_dw[TRD_WG * TRD_EV];
#define MI_TRD(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define MI_TRD(var) do {} while (0)
#endif

// One tile class of a dW launch: workgroup `hw` of `total` of this class.
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void dw_body(const DwTable& tab, unsigned hw, unsigned total,
                                        unsigned char* lds) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  const unsigned logical = xcd_swizzle(hw, total);
  const int bx = (int)(logical % (unsigned)tab.gx);
  const int by = (int)((logical / (unsigned)tab.gx) % (unsigned)tab.gy);
  const int bz = (int)(logical / (unsigned)(tab.gx * tab.gy));
  int pi = 0;
#pragma unroll
  for (int q = 1; q < kMaxDwProblems; ++q)
    if (q < tab.n && bz >= tab.p[q].z_begin) pi = q;
  const DwProblem pr = tab.p[pi];
  const bf16_t* __restrict__ A = pr.A;
  const bf16_t* __restrict__ B = pr.B;
  const int64_t lda = pr.lda, ldb = pr.ldb, I = pr.I, J = pr.J, M = tab.M;
  const int64_t rows_per_split = tab.rows_per_split;
  float* __restrict__ slabs = pr.slabs;
  const int zsplit = bz - pr.z_begin;
  if ((int64_t)bx * (WM * TM * 16) >= I || (int64_t)by * (WN * TN * 16) >= J)
    return;  // this problem has fewer tiles than the group's grid
  constexpr int BM = WM * TM * 16;
  constexpr int BN = WN * TN * 16;
  constexpr int AROW = BM + 16;  // LDS rows are the reduce index: [BK][BM + pad]
  constexpr int BROW = BN + 16;
  constexpr int A_CH = BK * (BM / 8);
  constexpr int B_CH = BK * (BN / 8);
  constexpr int A_PT = (A_CH + kThreads - 1) / kThreads;
  constexpr int B_PT = (B_CH + kThreads - 1) / kThreads;
  auto As = reinterpret_cast<bf16_t(*)[BK][AROW]>(lds);
  auto Bs = reinterpret_cast<bf16_t(*)[BK][BROW]>(lds + 2 * BK * AROW * sizeof(bf16_t));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: scalar unit
  const int wm = wave / WN, wn = wave % WN;
  const int64_t i0 = (int64_t)bx * BM;
  const int64_t j0 = (int64_t)by * BN;
  const int64_t r_begin = (int64_t)zsplit * rows_per_split;
  const int64_t r_end = r_begin + rows_per_split < M ? r_begin + rows_per_split : M;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Two register stages (tiles i+1 and i+2 in flight while tile i is multiplied): with
  // one stage the loads of a tile were in flight for a single 64-row iteration, less than
  // the memory latency under load, and every iteration ended waiting for them.  Loads are
  // unconditional (clamped addresses, invalid chunks zeroed by a select) so that the
  // compiler can count them and wait for the OLDER stage only.
  struct Stage {
    u32x4 a[A_PT], b[B_PT];
  };
  Stage st0, st1;
  const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};
  const int64_t r_last = r_end - 1;  // r_end > r_begin for every launched split
  // Full tiles (every row below r_end — all but a split's last one or two) load through
  // per-slot RUNNING POINTERS advanced by BK rows per call (the per-tile 64-bit
  // row * ld products were ~90 VALU instructions per tile beside 40 MFMAs) and are stored
  // without the zeroing select when the tile lies inside both operands' columns.
  // Threads beyond a tile's chunk count read rows below the tile (valid while the whole
  // A_ROWS / B_ROWS window is below r_end; never stored).
  // (slot p of a thread is kThreads / (row chunks) rows below slot 0: ONE running pointer
  // per operand, the slots at uniform offsets)
  constexpr int A_RPS = kThreads / (BM / 8), B_RPS = kThreads / (BN / 8);  // rows per slot
  static_assert(kThreads % (BM / 8) == 0 && kThreads % (BN / 8) == 0, "slot stride");
  constexpr int A_ROWS = A_PT * A_RPS, B_ROWS = B_PT * B_RPS;  // rows a call touches
  const int64_t gi_t = i0 + (tid % (BM / 8)) * 8, gj_t = j0 + (tid % (BN / 8)) * 8;
  const bf16_t* pa0 = A + (r_begin + tid / (BM / 8)) * lda + (gi_t < lda ? gi_t : 0);
  const bf16_t* pb0 = B + (r_begin + tid / (BN / 8)) * ldb + (gj_t < ldb ? gj_t : 0);
  const bool all_cols = i0 + BM <= lda && j0 + BN <= ldb;
  auto load_tile = [&](int64_t r0, Stage& sg) {
    if (r0 + A_ROWS <= r_end && r0 + B_ROWS <= r_end) {
#pragma unroll
      for (int p = 0; p < A_PT; ++p)
        sg.a[p] = *reinterpret_cast<const u32x4*>(pa0 + (int64_t)(p * A_RPS) * lda);
#pragma unroll
      for (int p = 0; p < B_PT; ++p)
        sg.b[p] = *reinterpret_cast<const u32x4*>(pb0 + (int64_t)(p * B_RPS) * ldb);
    } else {
#pragma unroll
      for (int p = 0; p < A_PT; ++p) {
        const int c = tid + p * kThreads;
        const int r = c / (BM / 8), ic = c % (BM / 8);
        const int64_t gr = r0 + r, gi = i0 + ic * 8;
        sg.a[p] = *reinterpret_cast<const u32x4*>(A + (gr < r_end ? gr : r_last) * lda +
                                                  (gi < lda ? gi : 0));
      }
#pragma unroll
      for (int p = 0; p < B_PT; ++p) {
        const int c = tid + p * kThreads;
        const int r = c / (BN / 8), jc = c % (BN / 8);
        const int64_t gr = r0 + r, gj = j0 + jc * 8;
        sg.b[p] = *reinterpret_cast<const u32x4*>(B + (gr < r_end ? gr : r_last) * ldb +
                                                  (gj < ldb ? gj : 0));
      }
    }
    // the calls walk the tiles in order: r_begin, + BK, + 2 BK, ...
    pa0 += BK * lda;
    pb0 += BK * ldb;
  };
  // (the zeroing select happens here, when the values are consumed, so that only the raw
  // loaded registers stay live across an iteration)
  auto store_tile = [&](int buf, const Stage& sg, int64_t r0) {
    const bool plain = all_cols && r0 + BK <= r_end;
#pragma unroll
    for (int p = 0; p < A_PT; ++p) {
      const int c = tid + p * kThreads;
      const int r = c / (BM / 8), ic = c % (BM / 8);
      const bool ok = plain || (r0 + r < r_end && i0 + ic * 8 < lda);
      if (c < A_CH) *reinterpret_cast<u32x4*>(&As[buf][r][ic * 8]) = ok ? sg.a[p] : zero4;
    }
#pragma unroll
    for (int p = 0; p < B_PT; ++p) {
      const int c = tid + p * kThreads;
      const int r = c / (BN / 8), jc = c % (BN / 8);
      const bool ok = plain || (r0 + r < r_end && j0 + jc * 8 < ldb);
      if (c < B_CH) *reinterpret_cast<u32x4*>(&Bs[buf][r][jc * 8]) = ok ? sg.b[p] : zero4;
    }
  };

  // transposing fragment gather: lane 4q+p of each 16-lane group addresses row
  // (rowbase + q), columns colbase + 4p..4p+3; lane i receives column colbase + i
  // of rows rowbase..rowbase+3.  Two reads give the 8 consecutive reduce elements
  // of the MFMA operand map (A[i = lane&15][k = 8*(lane>>4) + 0..7]).
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  auto frag = [&](const bf16_t* tile, int rowstride, int rowbase, int colbase) -> bf16x8 {
    using lds_s16x4 = __attribute__((address_space(3))) s16x4;
    const bf16_t* p0 = tile + (rowbase + tq) * rowstride + colbase + 4 * tp;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0 + 4 * rowstride));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  // Bias gradient db[j] = sum_r dZ[r][j] rides on the matrix cores: the dZ fragments
  // against an all-ones operand (wave row 0 of the first row-tile column only).  Summing
  // the column out of LDS element by element cost ~200 instructions per iteration on the
  // waves that did it — more than their MFMA work.
  const bool do_bias = bx == 0 && wm == 0;
  f32x4 accb[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) accb[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short ones_s16x8;
  const bf16x8 ones = __builtin_bit_cast(
      bf16x8, ones_s16x8{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80});

  auto multiply = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8 af[TM], bfr[TN];
      const int rowbase = ks * 32 + 8 * (lane >> 4);
#pragma unroll
      for (int a = 0; a < TM; ++a)
        af[a] = frag(&As[buf][0][0], AROW, rowbase, (wm * TM + a) * 16);
#pragma unroll
      for (int b = 0; b < TN; ++b)
        bfr[b] = frag(&Bs[buf][0][0], BROW, rowbase, (wn * TN + b) * 16);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          // transposed tile (dZ fragment as the A operand): a lane then holds 4
          // CONSECUTIVE output columns of one row — 16-byte slab stores below
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int b = 0; b < TN; ++b)
          accb[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b], ones, accb[b], 0, 0, 0);
      }
    }
  };

#ifdef MIPPO_TRACE
  const unsigned long long trd_w0 = wall_clock64();
#endif
  MI_TRD(trd_t0);
  load_tile(r_begin, st0);
  store_tile(0, st0, r_begin);
  __syncthreads();
  MI_TRD(trd_t1);
  load_tile(r_begin + BK, st0);      // tile 1
  // tile 2 strictly after tile 1: the in-loop waits count on "stage X older than stage Y"
  __builtin_amdgcn_sched_barrier(0);
  load_tile(r_begin + 2 * BK, st1);  // tile 2
  __builtin_amdgcn_sched_barrier(0);
  // Both halves always run (no early exit: a straight-line body keeps the accumulators in
  // place); a split with an odd number of tiles multiplies one tile of zeros at the end.
#ifdef MIPPO_TRACE
  unsigned long long trd_acc[4] = {0, 0, 0, 0};  // multiply / store (+ its wait) / load issue / barrier
#define MI_TRD_ACC(k, a, b) trd_acc[k] += (b) - (a)
#else
#define MI_TRD_ACC(k, a, b) do {} while (0)
#endif
  for (int64_t r0 = r_begin; r0 < r_end; r0 += 2 * BK) {
    // LDS buffer 0 holds tile r0; st0 = tile r0 + BK, st1 = tile r0 + 2 BK
    MI_TRD(q0);
    multiply(0);
    MI_TRD(q1);
    store_tile(1, st0, r0 + BK);
    MI_TRD(q2);
    load_tile(r0 + 3 * BK, st0);
    MI_TRD(q3);
    __syncthreads();
    MI_TRD(q4);
    MI_TRD_ACC(0, q0, q1);
    MI_TRD_ACC(1, q1, q2);
    MI_TRD_ACC(2, q2, q3);
    MI_TRD_ACC(3, q3, q4);
    // LDS buffer 1 holds tile r0 + BK; st1 = tile r0 + 2 BK, st0 = tile r0 + 3 BK
    multiply(1);
    store_tile(0, st1, r0 + 2 * BK);
    load_tile(r0 + 4 * BK, st1);
    __syncthreads();
  }

  MI_TRD(trd_t2);
  float* slab = slabs + (int64_t)zsplit * (I * J + J);
  const int In = (int)I, Jn = (int)J;  // <= 512: 32-bit index arithmetic
  // rows of the slab are 16-byte aligned
  const bool vec = (Jn & 3) == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int i = (int)i0 + (wm * TM + a) * 16 + (lane & 15);
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int j = (int)j0 + (wn * TN + b) * 16 + 4 * (lane >> 4);
      if (i < In && j < Jn) {
        float* dst = slab + i * Jn + j;
        if (vec) {  // j % 4 == 0 and J % 4 == 0: the four columns exist
          *reinterpret_cast<f32x4*>(dst) = acc[a][b];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j + e < Jn) dst[e] = acc[a][b][e];
        }
      }
    }
  }
  if (do_bias && (lane & 15) == 0) {  // every row of the ones-tile holds the same sums
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int j = (int)j0 + (wn * TN + b) * 16 + 4 * (lane >> 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (j + e < Jn) slab[(int64_t)In * Jn + j + e] = accb[b][e];
    }
  }
#ifdef MIPPO_TRACE
  if (tid == 0 && blockIdx.x < TRD_WG) {
    unsigned long long* o = g_trace_dw + (size_t)blockIdx.x * TRD_EV;
    o[0] = trd_t0;
    o[1] = trd_t1;
    o[2] = trd_t2;
    o[3] = __builtin_amdgcn_s_memtime();
    o[4] = trd_w0;
    o[5] = wall_clock64();
    o[6] = (unsigned long long)(WM * 100 + TN);  // tile class tag
    o[7] = (unsigned long long)((r_end - r_begin + BK - 1) / BK);
    for (int k = 0; k < 4; ++k) o[8 + k] = trd_acc[k];  // sums over the EVEN tiles
  }
#endif
}

// ALL tile classes of a grouped dW in one launch: the workgroups of the 128x128 class
// come first, then 128x64, then 128x16 (longest first); narrow classes run beside the
// wide one instead of after it, and three launches (graph nodes) become one.
struct DwAll {
  DwTable cls[3];
  unsigned begin[4];  // workgroup ranges of the classes
};

__global__ void __launch_bounds__(kThreads, 2)  // two workgroups per CU: <= 256 VGPRs
tn_gemm_dw_all_kernel(DwAll all) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dw_lds[];
  const unsigned hw = blockIdx.x;
  if (hw < all.begin[1]) {
    dw_body<2, 2, 4, 4>(all.cls[0], hw - all.begin[0], all.begin[1] - all.begin[0], dw_lds);
  } else if (hw < all.begin[2]) {
    dw_body<4, 1, 2, 4>(all.cls[1], hw - all.begin[1], all.begin[2] - all.begin[1], dw_lds);
  } else {
    dw_body<4, 1, 2, 1>(all.cls[2], hw - all.begin[2], all.begin[3] - all.begin[2], dw_lds);
  }
}

// fp32 [M][F] (optionally times act'(aux)) -> bf16 [M][ld], zero padded to ld.
__global__ void __launch_bounds__(kThreads)
cast_pad_kernel(const float* __restrict__ x, const bf16_t* __restrict__ aux, int64_t ldaux,
                int act, bf16_t* __restrict__ out, int64_t ld, int64_t M, int64_t F) {
  const int64_t total = M * ld;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int64_t m = i / ld, f = i % ld;
    float v = f < F ? x[m * F + f] : 0.0f;
    if (aux && f < F) v *= act_grad((float)aux[m * ldaux + f], act);
    out[i] = (bf16_t)v;
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

struct WLeaf {
  const float* w;
  bf16_t* wb;   // [K][ldw]  row-major W (dX operand of the per-layer kernel)
  bf16_t* wt;   // [N][ldwt] row-major W^T (forward operand of the per-layer kernel)
  bf16_t* ff;   // forward fragment-major image (whole-trunk kernels), or null
  bf16_t* fb;   // backward fragment-major image, or null
  int64_t K, N, ldw, ldwt;
};
struct WTable {
  WLeaf leaf[16];
};

// Fragment-major image of an [R x C] "NT" operand (C output columns, R reduce
// elements): block (ct, ks) holds, for lane l of a wave, the 8 bf16 of column
// ct*16 + (l & 15), reduce elements ks*32 + 8*(l >> 4) .. +7 — i.e. exactly the
// MFMA 16x16x32 B-operand fragment — so a wave-wide fragment load is ONE
// contiguous 1 KiB read instead of 16 strided 64-byte segments.
__device__ inline void frag_store(bf16_t* dst, int64_t idx, int64_t C, int64_t R,
                                  const float* w, int64_t N, bool transposed_src) {
  // idx -> (ct, ks, lane, e)
  const int e = (int)(idx & 7);
  const int lane = (int)((idx >> 3) & 63);
  const int64_t blk = idx >> 9;
  const int64_t KS = (R + 31) / 32;
  const int64_t ks = blk % KS, ct = blk / KS;
  const int64_t c = ct * 16 + (lane & 15);
  const int64_t r = ks * 32 + 8 * (lane >> 4) + e;
  float v = 0.0f;
  if (c < C && r < R) v = transposed_src ? w[r * N + c] : w[c * N + r];
  dst[idx] = (bf16_t)v;
}

// all layers of a network in one launch (blockIdx.y = layer)
__global__ void __launch_bounds__(kThreads)
weights_to_bf16_multi_kernel(WTable tab) {
  const WLeaf lf = tab.leaf[blockIdx.y];
  const int64_t n1 = lf.K * lf.ldw, n2 = lf.N * lf.ldwt;
  // forward image: columns = N outputs, reduce = K; source W[k][n] -> (c = n, r = k)
  const int64_t n3 = lf.ff ? ((lf.N + 15) / 16) * ((lf.K + 31) / 32) * 512 : 0;
  // backward image: columns = K outputs, reduce = N; source W[k][n] -> (c = k, r = n)
  const int64_t n4 = lf.fb ? ((lf.K + 15) / 16) * ((lf.N + 31) / 32) * 512 : 0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n1 + n2 + n3 + n4;
       i += (int64_t)gridDim.x * kThreads) {
    if (i < n1) {
      const int64_t k = i / lf.ldw, n = i % lf.ldw;
      lf.wb[i] = (bf16_t)(n < lf.N ? lf.w[k * lf.N + n] : 0.0f);
    } else if (i < n1 + n2) {
      const int64_t q = i - n1;
      const int64_t n = q / lf.ldwt, k = q % lf.ldwt;
      lf.wt[q] = (bf16_t)(k < lf.K ? lf.w[k * lf.N + n] : 0.0f);
    } else if (i < n1 + n2 + n3) {
      frag_store(lf.ff, i - n1 - n2, lf.N, lf.K, lf.w, lf.N, true);
    } else {
      frag_store(lf.fb, i - n1 - n2 - n3, lf.K, lf.N, lf.w, lf.N, false);
    }
  }
}

// Tile configuration by output width J: wide outputs get 128x128 (2x2 waves of
// 4x4 tiles), medium 128x64, narrow heads (J <= 16) 128x16.
template <int EPI>
int launch_nt(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int64_t I, int64_t J,
              int64_t R, const Epi& ep, hipStream_t st) {
  // matrix-core-bound shapes: 256 x 256 tiles with direct-to-LDS loads (gemm256_bf16.hip);
  // bit-identical to the kernels below
  const int took = mippo_gemm::nt256_launch(EPI, A, lda, B, ldb, I, J, R, ep, st);
  if (took) return took < 0 ? took : 0;
  if (J > 64) {
    dim3 grid((unsigned)mippo::ceil_div(I, 128), (unsigned)mippo::ceil_div(J, 128));
    hipLaunchKernelGGL((nt_gemm_kernel<2, 2, 4, 4, EPI>), grid, dim3(kThreads), 0, st, A, lda, B,
                       ldb, I, J, R, ep);
  } else if (J > 16) {
    dim3 grid((unsigned)mippo::ceil_div(I, 128), (unsigned)mippo::ceil_div(J, 64));
    hipLaunchKernelGGL((nt_gemm_kernel<4, 1, 2, 4, EPI>), grid, dim3(kThreads), 0, st, A, lda, B,
                       ldb, I, J, R, ep);
  } else {
    dim3 grid((unsigned)mippo::ceil_div(I, 128), 1);
    hipLaunchKernelGGL((nt_gemm_kernel<4, 1, 2, 1, EPI>), grid, dim3(kThreads), 0, st, A, lda, B,
                       ldb, I, J, R, ep);
  }
  return mippo::check_launch("nt_gemm_bf16");
}

int64_t dw_tile_n(int64_t N) { return N > 64 ? 128 : (N > 16 ? 64 : 16); }

}  // namespace

namespace mippo {
int reduce_slabs(const float* slabs, float* g_w, float* g_b, int64_t S, int64_t KN, int64_t N,
                 int accumulate, hipStream_t st);
}

extern "C" int mi_cast_pad_bf16(const float* x, const void* aux_bf, int64_t ldaux, int act,
                                void* out, int64_t ld, int64_t M, int64_t F,
                                mi_stream_t stream) {
  MI_REQUIRE(act >= MI_ACT_NONE && act <= MI_ACT_SWISH, "mi_cast_pad_bf16: bad act");
  MI_REQUIRE(M >= 0 && F >= 1 && ld >= F && ld % 8 == 0, "mi_cast_pad_bf16: bad shape");
  if (M == 0) return 0;
  MI_REQUIRE(x && out, "mi_cast_pad_bf16: null pointer");
  MI_REQUIRE(act == MI_ACT_NONE || aux_bf, "mi_cast_pad_bf16: aux needed for act'");
  hipLaunchKernelGGL(cast_pad_kernel, dim3(stream_grid(M * ld)), dim3(kThreads), 0,
                     mippo::as_stream(stream), x,
                     act == MI_ACT_NONE ? nullptr : static_cast<const bf16_t*>(aux_bf), ldaux, act,
                     static_cast<bf16_t*>(out), ld, M, F);
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
                                 void* preact_bf, int64_t M, int64_t K, int64_t N, int act,
                                 mi_stream_t stream) {
  MI_REQUIRE(M >= 0 && K >= 1 && N >= 1, "mi_dense_fwd_bf16: bad shape");
  MI_REQUIRE(act >= MI_ACT_NONE && act <= MI_ACT_SWISH, "mi_dense_fwd_bf16: bad act");
  MI_REQUIRE(ldx % 8 == 0 && ldwt % 8 == 0 && ldx >= K && ldwt >= K,
             "mi_dense_fwd_bf16: operand ld must be a multiple of 8 and >= K");
  MI_REQUIRE(!(y_bf || preact_bf) || (ldy >= N && ldy % 8 == 0), "mi_dense_fwd_bf16: bad ldy");
  if (M == 0) return 0;
  MI_REQUIRE(x_bf && wt_bf && (y_f32 || y_bf), "mi_dense_fwd_bf16: null pointer");
  MI_REQUIRE(al16(x_bf) && al16(wt_bf) && al16(y_bf) && al16(preact_bf),
             "mi_dense_fwd_bf16: bf16 buffers must be 16-byte aligned");
  Epi ep = {};
  ep.bias = bias;
  ep.act = act;
  ep.out_f32 = y_f32;
  ep.ld_f32 = N;
  ep.out_bf = static_cast<bf16_t*>(y_bf);
  ep.ld_bf = ldy;
  ep.aux_bf = static_cast<bf16_t*>(preact_bf);
  // reduce length = K rounded up to the operand padding (zeros beyond K)
  const int64_t R = mippo::ceil_div(K, 8) * 8;
  return launch_nt<EPI_FWD>(static_cast<const bf16_t*>(x_bf), ldx,
                            static_cast<const bf16_t*>(wt_bf), ldwt, M, N, R, ep,
                            mippo::as_stream(stream));
}

extern "C" int mi_dense_bwd_dx_bf16(const void* dz_bf, int64_t lddz, const void* w_bf,
                                    int64_t ldw, const void* prev_bf, int64_t ldprev,
                                    int prev_act, float* gx_f32, void* gx_bf, int64_t ldgx,
                                    int64_t M, int64_t K, int64_t N, mi_stream_t stream) {
  MI_REQUIRE(M >= 0 && K >= 1 && N >= 1, "mi_dense_bwd_dx_bf16: bad shape");
  MI_REQUIRE(prev_act >= MI_ACT_NONE && prev_act <= MI_ACT_SWISH, "mi_dense_bwd_dx_bf16: bad act");
  MI_REQUIRE(lddz % 8 == 0 && ldw % 8 == 0 && lddz >= N && ldw >= N,
             "mi_dense_bwd_dx_bf16: operand ld must be a multiple of 8 and >= N");
  MI_REQUIRE(!gx_bf || (ldgx >= K && ldgx % 8 == 0), "mi_dense_bwd_dx_bf16: bad ldgx");
  if (M == 0) return 0;
  MI_REQUIRE(dz_bf && w_bf && (gx_f32 || gx_bf), "mi_dense_bwd_dx_bf16: null pointer");
  MI_REQUIRE(prev_act == MI_ACT_NONE || (prev_bf && ldprev >= K && ldprev % 8 == 0),
             "mi_dense_bwd_dx_bf16: prev output needed");
  MI_REQUIRE(al16(dz_bf) && al16(w_bf) && al16(gx_bf) && al16(prev_bf),
             "mi_dense_bwd_dx_bf16: bf16 buffers must be 16-byte aligned");
  Epi ep = {};
  ep.out_f32 = gx_f32;
  ep.ld_f32 = K;
  ep.out_bf = static_cast<bf16_t*>(gx_bf);
  ep.ld_bf = ldgx;
  ep.prev = static_cast<const bf16_t*>(prev_bf);
  ep.ld_prev = ldprev;
  ep.prev_act = prev_act;
  const int64_t R = mippo::ceil_div(N, 8) * 8;
  return launch_nt<EPI_DX>(static_cast<const bf16_t*>(dz_bf), lddz,
                           static_cast<const bf16_t*>(w_bf), ldw, M, K, R, ep,
                           mippo::as_stream(stream));
}

// rows per split / number of splits shared by every problem of a launch (they
// share M): enough workgroups to cover the chip about twice, >= 128 rows each.
// `total_tiles`: output tiles of EVERY problem of the launch, whatever their class: one
// split count for all of them.  A 64-row tile costs a workgroup 2 400-3 000 cycles in
// every class (tools/trace_policy.py: the iteration is bound by its latency chain, not by
// the tile's MFMA count), so equal tiles per workgroup make the classes finish together —
// per-class targets left the 128x128 class with 15 tiles per workgroup beside 7-10 in the
// narrow ones and the kernel as long as its slowest class.  The launch stays inside ONE
// resident wave of workgroups (2 per CU: 73 KB of LDS each) and every split also adds a
// K x N fp32 slab that is written and read again; measured on bench.py (C2: 13 tiles)
// with MIPPO_DW_BLOCKS = 416 .. 640: 416 (32 splits, 15 row tiles per workgroup) is best,
// 455-500 (35-39 splits) cost +19-25 % in this kernel, 544+ need a second wave.
// MIPPO_DW_BLOCKS overrides the target (tuning aid).
static void dw_split_plan(int64_t M, int64_t total_tiles, int64_t* rows, int64_t* S) {
  static const int64_t override_target = [] {
    const char* e = getenv("MIPPO_DW_BLOCKS");
    return e ? (int64_t)atoi(e) : (int64_t)0;
  }();
  const int64_t target =
      override_target > 0 ? override_target : (int64_t)13 * mippo::kNumCU / 8;
  const int64_t tiles = total_tiles < 1 ? 1 : total_tiles;
  int64_t s = mippo::ceil_div(target, tiles);
  // Whole splits per XCD: workgroup ids are split-major and dealt to the 8 XCDs in equal runs,
  // so with S a multiple of 8 every XCD's L2 serves the tiles of ITS splits only (they read the
  // same rows of X and dZ).  Off that grid the launch is a third slower (C2, 13 tiles: S = 32
  // 28 us; 26 / 40 splits 37-40 us; C4, 14 tiles: S = 30 -> 32 is 4.6 % of the iteration).  Up
  // if the launch still fits the chip's 2 x 256 workgroup slots, else down.
  if (override_target <= 0 && s >= 8) {
    const int64_t up = mippo::ceil_div(s, 8) * 8, down = s / 8 * 8;
    s = tiles * up <= 2 * (int64_t)mippo::kNumCU ? up : down;
  }
  const int64_t max_s = mippo::ceil_div(M, 128);
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 512) s = 512;
  *rows = mippo::ceil_div(mippo::ceil_div(M, s), 64) * 64;
  *S = mippo::ceil_div(M, *rows);
}

static int64_t dw_tiles(int64_t K, int64_t N) {
  return mippo::ceil_div(K, 128) * mippo::ceil_div(N, dw_tile_n(N));
}

namespace mippo {
int reduce_slabs_grouped(int n, const float* const* slabs, const int64_t* S, const int64_t* KN,
                         const int64_t* N, float* const* g_w, float* const* g_b, int accumulate,
                         hipStream_t st);
}

extern "C" int64_t mi_dense_bwd_dw_grouped_bf16_workspace_bytes(int64_t n, const int64_t* K,
                                                               const int64_t* N, int64_t M) {
  if (n < 1 || n > kMaxDwProblems || !K || !N || M < 1) return -EINVAL;
  // worst case: every problem alone in its launch
  int64_t total = 0;
  for (int64_t l = 0; l < n; ++l) {
    int64_t rows, S;
    dw_split_plan(M, dw_tiles(K[l], N[l]), &rows, &S);
    if (mippo_gemm::dw256_takes(K[l], N[l], M) || mippo_gemm::dw256_candidate(K[l], N[l], M)) {
      // (the largest split it can get: a group of exactly four tiles of the 256-row kernel)
      int64_t r2, S2;
      const int64_t t = mippo_gemm::dw256_tiles(K[l], N[l]);
      mippo_gemm::dw256_plan(M, t > 4 ? t : 4, &r2, &S2);
      if (S2 > S) S = S2;
    }
    total += S * (K[l] * N[l] + N[l]);
  }
  return total * (int64_t)sizeof(float);
}

// The dW launch of a group (all tile classes in one launch); reports where each problem's
// slabs sit in `workspace` and how many there are.
static int dw_grouped_launch(const char* who, int64_t n, const void* const* x_bf,
                             const void* const* dz_bf, const int64_t* K, const int64_t* N,
                             int64_t M, void* workspace, const float** slab_ptr, int64_t* Sv,
                             hipStream_t st) {
  MI_REQUIRE(n >= 1 && n <= kMaxDwProblems && M >= 1, "%s: 1 <= n <= %d", who, kMaxDwProblems);
  MI_REQUIRE(x_bf && dz_bf && K && N && workspace, "%s: null pointer", who);
  float* ws = static_cast<float*>(workspace);
  int64_t KNv[kMaxDwProblems];
  for (int64_t l = 0; l < n; ++l) {
    MI_REQUIRE(x_bf[l] && dz_bf[l] && K[l] >= 1 && N[l] >= 1 && al16(x_bf[l]) && al16(dz_bf[l]),
               "%s: bad problem %lld", who, (long long)l);
    KNv[l] = K[l] * N[l];
  }
  // matrix-core-bound problems (both dimensions >= 128 at training sizes) go out first, in
  // one launch of the 256 x 256-tile kernel (gemm256_bf16.hip); the rest below
  bool big[kMaxDwProblems];
  int n_big = 0;
  {
    const bf16_t* xa[kMaxDwProblems];
    const bf16_t* za[kMaxDwProblems];
    float* sl[kMaxDwProblems];
    int64_t Kb[kMaxDwProblems], Nb[kMaxDwProblems];
    int64_t tiles_big = 0;
    // (a square 256-wide layer is one tile: on its own the chip-filling split would cost more
    // in slabs than the tile kernel does, beside three or more other tiles it does not)
    for (int64_t l = 0; l < n; ++l) {
      big[l] = mippo_gemm::dw256_takes(K[l], N[l], M) || mippo_gemm::dw256_candidate(K[l], N[l], M);
      if (big[l]) tiles_big += mippo_gemm::dw256_tiles(K[l], N[l]);
    }
    if (tiles_big < 4) {  // (a problem `dw256_takes` on its own has four tiles itself)
      tiles_big = 0;
      for (int64_t l = 0; l < n; ++l) big[l] = false;
    }
    if (tiles_big) {
      int64_t rows_b, S_b;
      mippo_gemm::dw256_plan(M, tiles_big, &rows_b, &S_b);
      for (int64_t l = 0; l < n; ++l) {
        if (!big[l]) continue;
        xa[n_big] = static_cast<const bf16_t*>(x_bf[l]);
        za[n_big] = static_cast<const bf16_t*>(dz_bf[l]);
        Kb[n_big] = K[l];
        Nb[n_big] = N[l];
        sl[n_big] = ws;
        slab_ptr[l] = ws;
        Sv[l] = S_b;
        ws += S_b * (KNv[l] + N[l]);
        ++n_big;
      }
      const int rc = mippo_gemm::dw256_launch(n_big, xa, za, Kb, Nb, M, sl, rows_b, S_b, st);
      if (rc) return rc;
      if (n_big == n) return 0;
    }
  }
  // one workgroup range per tile class (by output width), problems of a class side by
  // side; all classes go out in ONE launch
  DwAll all = {};
  unsigned next = 0;
  int64_t tiles_all = 0;
  for (int64_t l = 0; l < n; ++l)
    if (!big[l]) tiles_all += dw_tiles(K[l], N[l]);
  int64_t rows, S;
  dw_split_plan(M, tiles_all, &rows, &S);
  {
    // the same tiles and splits through the DMA-staged 128 x 128 kernel (gemm256_bf16.hip:
    // tn128_kernel) when the rows come in whole 32-row slots: bit-identical slabs
    int n_rest = 0;
    int64_t wide = 0;  // tiles of outputs wider than 64 columns
    for (int64_t l = 0; l < n; ++l) {
      if (big[l]) continue;
      ++n_rest;
      if (N[l] > 64) wide += dw_tiles(K[l], N[l]);
    }
    if (mippo_gemm::dw128_takes(M, rows, n_rest, wide, tiles_all)) {
      const bf16_t* xa[kMaxDwProblems];
      const bf16_t* za[kMaxDwProblems];
      float* sl[kMaxDwProblems];
      int64_t Kr[kMaxDwProblems], Nr[kMaxDwProblems];
      int q = 0;
      for (int64_t l = 0; l < n; ++l) {
        if (big[l]) continue;
        xa[q] = static_cast<const bf16_t*>(x_bf[l]);
        za[q] = static_cast<const bf16_t*>(dz_bf[l]);
        Kr[q] = K[l];
        Nr[q] = N[l];
        sl[q] = ws;
        slab_ptr[l] = ws;
        Sv[l] = S;
        ws += S * (KNv[l] + N[l]);
        ++q;
      }
      return mippo_gemm::dw128_launch(n_rest, xa, za, Kr, Nr, M, sl, rows, S, st);
    }
  }
  size_t lds = 0;
  for (int cls = 0; cls < 3; ++cls) {
    DwTable& tab = all.cls[cls];
    int idx[kMaxDwProblems];
    int64_t gx = 0, gy = 0;
    for (int64_t l = 0; l < n; ++l) {
      const int c = N[l] > 64 ? 0 : (N[l] > 16 ? 1 : 2);
      if (c != cls || big[l]) continue;
      idx[tab.n++] = (int)l;
      const int64_t tx = mippo::ceil_div(K[l], 128), ty = mippo::ceil_div(N[l], dw_tile_n(N[l]));
      if (tx > gx) gx = tx;
      if (ty > gy) gy = ty;
    }
    all.begin[cls] = next;
    if (tab.n == 0) continue;
    tab.M = M;
    tab.rows_per_split = rows;
    for (int q = 0; q < tab.n; ++q) {
      const int l = idx[q];
      DwProblem& pr = tab.p[q];
      pr.A = static_cast<const bf16_t*>(x_bf[l]);
      pr.B = static_cast<const bf16_t*>(dz_bf[l]);
      pr.lda = mippo::ceil_div(K[l], 8) * 8;
      pr.ldb = mippo::ceil_div(N[l], 8) * 8;
      pr.I = K[l];
      pr.J = N[l];
      pr.slabs = ws;
      pr.z_begin = (int)(q * S);
      slab_ptr[l] = ws;
      Sv[l] = S;
      ws += S * (KNv[l] + N[l]);
    }
    tab.gx = (int)gx;
    tab.gy = (int)gy;
    const int64_t blocks = gx * gy * tab.n * S;
    MI_REQUIRE(next + blocks <= 0x7fffffffLL, "%s: grid too large", who);
    next += (unsigned)blocks;
    const size_t need = cls == 0 ? dw_lds_bytes<2, 2, 4, 4>()
                                 : (cls == 1 ? dw_lds_bytes<4, 1, 2, 4>() : dw_lds_bytes<4, 1, 2, 1>());
    if (need > lds) lds = need;
  }
  all.begin[3] = next;
  static const hipError_t attr = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&tn_gemm_dw_all_kernel),
      hipFuncAttributeMaxDynamicSharedMemorySize, dw_lds_bytes<2, 2, 4, 4>());
  MI_REQUIRE(attr == hipSuccess, "%s: cannot raise the LDS limit", who);
  hipLaunchKernelGGL(tn_gemm_dw_all_kernel, dim3(next), dim3(kThreads), lds, st, all);
  return mippo::check_launch(who);
}

extern "C" int mi_dense_bwd_dw_grouped_bf16(int64_t n, const void* const* x_bf,
                                            const void* const* dz_bf, float* const* g_w,
                                            float* const* g_b, const int64_t* K,
                                            const int64_t* N, int64_t M, void* workspace,
                                            int accumulate, mi_stream_t stream) {
  MI_REQUIRE(g_w, "mi_dense_bwd_dw_grouped_bf16: null pointer");
  hipStream_t st = mippo::as_stream(stream);
  const float* slab_ptr[kMaxDwProblems];
  int64_t Sv[kMaxDwProblems], KNv[kMaxDwProblems];
  int rc = dw_grouped_launch("mi_dense_bwd_dw_grouped_bf16", n, x_bf, dz_bf, K, N, M, workspace,
                             slab_ptr, Sv, st);
  if (rc) return rc;
  for (int64_t l = 0; l < n; ++l) {
    MI_REQUIRE(g_w[l], "mi_dense_bwd_dw_grouped_bf16: null gradient %lld", (long long)l);
    KNv[l] = K[l] * N[l];
  }
  return mippo::reduce_slabs_grouped((int)n, slab_ptr, Sv, KNv, N, g_w, g_b, accumulate, st);
}

// The two halves of mi_dense_bwd_dw_grouped_bf16 on their own: the dW launch that leaves
// the split-M slabs in `workspace` (slab_ptr_out[l], n_slabs_out[l] say where), and their
// fixed-order reduction into the gradients.  A caller that reduces the slabs elsewhere —
// mi_adam_step_slabs_f32 sums them while it reads the gradient arena — skips the second.
extern "C" int mi_dense_bwd_dw_grouped_slabs_bf16(int64_t n, const void* const* x_bf,
                                                  const void* const* dz_bf, const int64_t* K,
                                                  const int64_t* N, int64_t M, void* workspace,
                                                  const void** slab_ptr_out,
                                                  int64_t* n_slabs_out, mi_stream_t stream) {
  MI_REQUIRE(slab_ptr_out && n_slabs_out, "mi_dense_bwd_dw_grouped_slabs_bf16: null pointer");
  const float* slab_ptr[kMaxDwProblems];
  int64_t Sv[kMaxDwProblems];
  int rc = dw_grouped_launch("mi_dense_bwd_dw_grouped_slabs_bf16", n, x_bf, dz_bf, K, N, M,
                             workspace, slab_ptr, Sv, mippo::as_stream(stream));
  if (rc) return rc;
  for (int64_t l = 0; l < n; ++l) {
    slab_ptr_out[l] = slab_ptr[l];
    n_slabs_out[l] = Sv[l];
  }
  return 0;
}

extern "C" int mi_reduce_slabs_grouped_f32(int64_t n, const void* const* slab_ptr,
                                           const int64_t* n_slabs, const int64_t* K,
                                           const int64_t* N, float* const* g_w,
                                           float* const* g_b, int accumulate,
                                           mi_stream_t stream) {
  MI_REQUIRE(n >= 1 && n <= kMaxDwProblems && slab_ptr && n_slabs && K && N && g_w,
             "mi_reduce_slabs_grouped_f32: bad arguments");
  const float* sp[kMaxDwProblems];
  int64_t KNv[kMaxDwProblems];
  for (int64_t l = 0; l < n; ++l) {
    MI_REQUIRE(slab_ptr[l] && g_w[l] && n_slabs[l] >= 1 && K[l] >= 1 && N[l] >= 1,
               "mi_reduce_slabs_grouped_f32: bad problem %lld", (long long)l);
    sp[l] = static_cast<const float*>(slab_ptr[l]);
    KNv[l] = K[l] * N[l];
  }
  return mippo::reduce_slabs_grouped((int)n, sp, n_slabs, KNv, N, g_w, g_b, accumulate,
                                     mippo::as_stream(stream));
}

extern "C" int64_t mi_dense_bwd_dw_bf16_workspace_bytes(int64_t M, int64_t K, int64_t N) {
  return mi_dense_bwd_dw_grouped_bf16_workspace_bytes(1, &K, &N, M);
}

extern "C" int mi_dense_bwd_dw_bf16(const void* x_bf, int64_t ldx, const void* dz_bf,
                                    int64_t lddz, float* g_w, float* g_b, void* workspace,
                                    int64_t M, int64_t K, int64_t N, int accumulate,
                                    mi_stream_t stream) {
  MI_REQUIRE(M >= 1 && K >= 1 && N >= 1, "mi_dense_bwd_dw_bf16: bad shape");
  MI_REQUIRE(ldx == mippo::ceil_div(K, 8) * 8 && lddz == mippo::ceil_div(N, 8) * 8,
             "mi_dense_bwd_dw_bf16: operand ld must be pad8(K) / pad8(N)");
  return mi_dense_bwd_dw_grouped_bf16(1, &x_bf, &dz_bf, &g_w, &g_b, &K, &N, M, workspace,
                                      accumulate, stream);
}

extern "C" int mi_weights_to_bf16_multi(int64_t n_layers, const float* const* w, void* const* w_bf,
                                        void* const* wt_bf, void* const* frag_fwd,
                                        void* const* frag_bwd, const int64_t* K,
                                        const int64_t* N, mi_stream_t stream) {
  MI_REQUIRE(n_layers >= 0 && n_layers <= 16, "mi_weights_to_bf16_multi: 0 <= n_layers <= 16");
  if (n_layers == 0) return 0;
  MI_REQUIRE(w && w_bf && wt_bf && K && N, "mi_weights_to_bf16_multi: null pointer");
  WTable tab = {};
  int64_t max_total = 0;
  for (int64_t l = 0; l < n_layers; ++l) {
    MI_REQUIRE(w[l] && w_bf[l] && wt_bf[l] && K[l] >= 1 && N[l] >= 1,
               "mi_weights_to_bf16_multi: bad layer %lld", (long long)l);
    WLeaf& lf = tab.leaf[l];
    lf.w = w[l];
    lf.wb = static_cast<bf16_t*>(w_bf[l]);
    lf.wt = static_cast<bf16_t*>(wt_bf[l]);
    lf.ff = frag_fwd ? static_cast<bf16_t*>(frag_fwd[l]) : nullptr;
    lf.fb = frag_bwd ? static_cast<bf16_t*>(frag_bwd[l]) : nullptr;
    lf.K = K[l];
    lf.N = N[l];
    lf.ldw = mippo::ceil_div(N[l], 8) * 8;
    lf.ldwt = mippo::ceil_div(K[l], 8) * 8;
    int64_t tot = lf.K * lf.ldw + lf.N * lf.ldwt;
    if (lf.ff) tot += mippo::ceil_div(N[l], 16) * mippo::ceil_div(K[l], 32) * 512;
    if (lf.fb) tot += mippo::ceil_div(K[l], 16) * mippo::ceil_div(N[l], 32) * 512;
    if (tot > max_total) max_total = tot;
  }
  dim3 grid((unsigned)stream_grid(max_total), (unsigned)n_layers);
  hipLaunchKernelGGL(weights_to_bf16_multi_kernel, grid, dim3(kThreads), 0,
                     mippo::as_stream(stream), tab);
  return mippo::check_launch("mi_weights_to_bf16_multi");
}

#ifdef MIPPO_TRACE
extern "C" int mi_debug_trace_dw(unsigned long long* host_out, int64_t n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_trace_dw), (size_t)n * 8);
}
extern "C" int mi_debug_trace_dw_clear() {
  void* p = nullptr;
  hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_trace_dw));
  return e != hipSuccess ? (int)e : (int)hipMemset(p, 0, sizeof(g_trace_dw));
}
#endif
