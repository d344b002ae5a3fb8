// a9 (bf16 path, matrix-core-bound shapes) — the per-layer NT GEMM on 256 x 256 workgroup
// tiles with direct-to-LDS loads.
//
//   forward  Y[M,N]  = X[M,K]  . Wt[N,K]^T  (+ bias, activation -> bf16 image)
//   dX       gX[M,K] = dZ[M,N] . W[K,N]^T   (x act'(previous layer's output) -> bf16 image)
// (`feedforward.py:42-51` and its derivative; operands as gemm_bf16.hip keeps them: row-major
// bf16, reduce-contiguous on both sides.)
//
// BASELINE config 3 (CheetahRun-shaped: actor 4 x 256, critic 2 x 512, M = 61 440 rows per
// gradient step) is the one config whose layers are matrix-core bound (arithmetic intensity
// ~300 flop/B).  The whole-trunk walk of mlp_bf16.hip re-fetches a 512 x 512 layer's 512 KB of
// weights per 64-row tile and ran at ~200 TF/s; gemm_bf16.hip's 128 x 128 register-staged
// kernel at ~370.  This kernel is the CDNA4 GEMM shape for such layers:
//   * one 8-wave workgroup per CU owns a 256 x 256 output tile; wave (wm, wn) of 2 x 4 owns
//     128 x 64 of it = 8 x 4 MFMA tiles (v_mfma_f32_16x16x32_bf16), 128 accumulator registers;
//   * operands go global -> LDS directly (global_load_lds_dwordx4: no VGPR staging, no
//     ds_write pass), in k-slots of 32 reduce elements, FOUR slots of 32 KB in a ring: the
//     loads of slot t + 3 are issued before slot t is multiplied and the wait is the COUNTED
//     `s_waitcnt vmcnt(8)` — two slots stay in flight across the barrier, which is a raw
//     s_barrier (a __syncthreads() would drain them);
//   * the LDS image is lane-linear (the DMA writes base + lane * 16), so the bank swizzle sits
//     on the SOURCE address: the 16-byte chunk stored at position p of row r is chunk
//     p ^ (-(r >> 2) & 3) of that row, and the fragment reads apply the same XOR — every
//     ds_read_b128 of a fragment then touches 16 distinct 16-byte slots of the 256-byte bank
//     row (conflict-free, derived in DESIGN §3);
//   * the MFMA computes the transposed tile (weights as the first operand), so a lane ends
//     with 4 CONSECUTIVE output columns of one row: bias / activation on packed registers,
//     8-byte stores into the staged output tile, which leaves LDS in whole 16-byte row chunks
//     (the dX epilogue multiplies by relu' of the previous layer's image chunk on the way).
// Same products, same k order, same epilogue expressions as gemm_bf16.hip's nt_gemm_kernel:
// results are bit-identical to it (tests/test_gemm256_gpu.py), so the dispatch inside
// mi_dense_fwd_bf16 / mi_dense_bwd_dx_bf16 is invisible.
//
// Shape class (everything else keeps the 128-row kernel): R % 32 == 0, J % 8 == 0, J >= 128,
// I >= 2048, bf16 output only (no fp32 chain output, no pre-activation image: relu / tanh /
// none), operands 16-byte aligned with ld % 8 == 0.
#include <stdlib.h>

#include "gemm_epi.h"

namespace {

using namespace mippo_bf16;
using mippo_gemm::Epi;
using mippo_gemm::EPI_DX;
using mippo_gemm::EPI_FWD;

constexpr int kT = 512;          // threads: 8 waves, 2 (rows) x 4 (columns)
constexpr int TB = 256;          // tile rows = tile columns
constexpr int KB = 32;           // reduce elements per slot = one MFMA k-step
constexpr int NSLOT = 4;
constexpr int kOpBytes = TB * KB * 2;        // one operand of one slot: 16 KB, 64-byte rows
constexpr int kSlotBytes = 2 * kOpBytes;     // A then B
constexpr int kLdsBytes = NSLOT * kSlotBytes;  // 128 KB
constexpr int TM = 8, TN = 4;    // MFMA tiles per wave
constexpr int CROW = TB + 8;     // staged output row (bf16): 16 bytes of padding

// chunk position <-> source chunk of a 64-byte row (4 chunks): an involution
__device__ __forceinline__ int swz(int chunk, int row) { return chunk ^ ((-(row >> 2)) & 3); }

using lds_ptr_t = __attribute__((address_space(3))) void*;
using glb_ptr_t = const __attribute__((address_space(1))) void*;

template <int EPI>
__global__ void __launch_bounds__(kT, 2)
nt256_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
             int64_t I, int64_t J, int64_t R, Epi ep, int tiles_j) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int li = lane & 15, lq = lane >> 4;

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs; consecutive
  // LOGICAL ids (the column tiles of one row block, which read the same activation rows)
  // share an XCD's L2
  const unsigned total = gridDim.x;
  const unsigned hw = blockIdx.x;
  const unsigned chunk = total / 8, rem = total % 8;
  const unsigned xcd = hw % 8, idx = hw / 8;
  const unsigned logical = xcd * chunk + (xcd < rem ? xcd : rem) + idx;
  const int64_t i0 = (int64_t)(logical / (unsigned)tiles_j) * TB;
  const int64_t j0 = (int64_t)(logical % (unsigned)tiles_j) * TB;

  // ---- staging: wave w issues DMA instructions 2w, 2w + 1 of each operand; instruction q
  // covers rows 16q .. 16q + 15 (64 lanes x 16 bytes = 16 rows of 64 bytes) ----------------
  const char* ga[2];
  const char* gb[2];
  unsigned la[2];  // LDS byte offset of the instruction's base inside an operand image
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = (wave * 2 + q) * 16 + (lane >> 2);
    const int src_chunk = swz(lane & 3, row);
    int64_t gi = i0 + row;
    gi = gi < I ? gi : I - 1;      // clamped rows are computed but never stored
    int64_t gj = j0 + row;
    gj = gj < J ? gj : J - 1;
    ga[q] = reinterpret_cast<const char*>(A + gi * lda) + src_chunk * 16;
    gb[q] = reinterpret_cast<const char*>(B + gj * ldb) + src_chunk * 16;
    la[q] = (unsigned)((wave * 2 + q) * 16 * 64);
  }
  const int nk = (int)(R / KB);
  auto stage = [&](int t) {  // k-slot t (clamped: the tail re-loads the last slot, unread)
    const int tc = t < nk ? t : nk - 1;
    unsigned char* slot = lds + (t & (NSLOT - 1)) * kSlotBytes;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(ga[q] + (int64_t)tc * (KB * 2)),
                                       (lds_ptr_t)(slot + la[q]), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(gb[q] + (int64_t)tc * (KB * 2)),
                                       (lds_ptr_t)(slot + kOpBytes + la[q]), 16, 0, 0);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: row (.. + li), chunk lq at its swizzled position — the XOR depends on
  // li only (tile bases are multiples of 16 rows), so one base per operand + immediates
  const int pos = swz(lq, li);
  const unsigned fa = (unsigned)((wm * 128 + li) * 64 + pos * 16);
  const unsigned fb = (unsigned)(kOpBytes + (wn * 64 + li) * 64 + pos * 16);

  stage(0);
  stage(1);
  stage(2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // slot 0 landed (this wave's part)
  __builtin_amdgcn_s_barrier();                      // ... and everybody's
  asm volatile("" ::: "memory");

  for (int t = 0; t < nk; ++t) {
    stage(t + 3);  // into the slot read in iteration t - 1: every wave is past that barrier
    const unsigned char* slot = lds + (t & (NSLOT - 1)) * kSlotBytes;
    bf16x8 af[TM], bfr[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b)
      bfr[b] = *reinterpret_cast<const bf16x8*>(slot + fb + b * (16 * 64));
#pragma unroll
    for (int a = 0; a < TM; ++a)
      af[a] = *reinterpret_cast<const bf16x8*>(slot + fa + a * (16 * 64));
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
        // transposed tile: weights as the first operand -> lane holds 4 consecutive columns
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
    // the emitted order: the 4 weight fragments and the first row fragment, then every group
    // of 4 MFMAs (one row tile) with the NEXT row tile's fragment read beside it — the LDS
    // reads hide behind the matrix pipe instead of in front of it (all 12 reads first, then
    // `lgkmcnt(0)`, left the pipe idle for the whole read phase of both waves of a SIMD)
    // (one row fragment AHEAD: the read of row tile a + 2 is issued before the MFMAs of row
    // tile a, so it has two groups of MFMAs to land)
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int a = 0; a < TM - 2; ++a) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
    // slot t + 1 (issued two iterations ago) has landed once at most the 8 younger DMAs of
    // slots t + 2 and t + 3 are still in flight
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");  // no LDS read of the next slot moves above the barrier
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail's unread re-loads
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // ---- epilogue: two passes of 128 rows (the staged tile does not fit beside itself) ------
  bf16_t* tC = reinterpret_cast<bf16_t*>(lds);  // [128][CROW]
  const bool use_prev = EPI == EPI_DX && ep.prev && ep.prev_act != MI_ACT_NONE;
  f32x4 bias4[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    bias4[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t gj = j0 + wn * 64 + b * 16 + 4 * lq;
    if (EPI == EPI_FWD && ep.bias && gj + 3 < J)
      bias4[b] = *reinterpret_cast<const f32x4*>(ep.bias + gj);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (wm == h) {
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const int row = a * 16 + li;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int col = wn * 64 + b * 16 + 4 * lq;
          bf16x4 vo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[a][b][e];
            if (EPI == EPI_FWD) v = act_fwd(v + bias4[b][e], ep.act);
            vo[e] = (bf16_t)v;
          }
          *reinterpret_cast<bf16x4*>(tC + row * CROW + col) = vo;
        }
      }
    }
    __syncthreads();
    // 128 rows x 32 chunks of 16 bytes, 8 per thread; a wave-instruction covers two whole rows
    constexpr int CPR = TB / 8;
#pragma unroll
    for (int p = 0; p < 128 * CPR / kT; ++p) {
      const int c = tid + p * kT;
      const int row = c / CPR, cc = c % CPR;
      const int64_t gi = i0 + h * 128 + row, gj = j0 + cc * 8;
      if (gi < I && gj < ep.ld_bf) {
        u32x4 v = *reinterpret_cast<const u32x4*>(tC + row * CROW + cc * 8);
        if (gj >= J) v = u32x4{0u, 0u, 0u, 0u};  // padding columns of the image
        if (use_prev) {
          // x act'(previous layer's output): relu' / tanh' of the matching image chunk
          const bf16x8 pv = *reinterpret_cast<const bf16x8*>(ep.prev + gi * ep.ld_prev + gj);
          bf16x8 dv = __builtin_bit_cast(bf16x8, v);
#pragma unroll
          for (int e = 0; e < 8; ++e)
            dv[e] = (bf16_t)((float)dv[e] * act_grad((float)pv[e], ep.prev_act));
          v = __builtin_bit_cast(u32x4, dv);
        }
        *reinterpret_cast<u32x4*>(ep.out_bf + gi * ep.ld_bf + gj) = v;
      }
    }
    __syncthreads();
  }
}

// ---- dW on 256 x 256 tiles: gW[K,N] = X[M,K]^T . dZ[M,N], split over M into fp32 slabs ----
// Both operands are REDUCE-major (the reduce index m is their row index): a slot is 32 rows of
// 512 bytes of each, DMA'd as they lie in memory (one instruction = two rows); the MFMA
// fragments are gathered by the transposing LDS read (ds_read_b64_tr_b16, as gemm_bf16.hip's
// dW).  The bank swizzle is again on the source address: 16-byte chunk c of row r sits at
// position c ^ 2 f(r), f(r) = (r & 3) | ((r >> 3 & 1) << 2) — the eight rows a 32-lane half
// of a transposing read touches (4 rows of two 16-lane groups, 8 rows apart) then fall into
// the eight different 32-byte slots of the 256-byte bank row.
struct Dw256Problem {
  const bf16_t* A;  // X  [M][lda]
  const bf16_t* B;  // dZ [M][ldb]
  float* slabs;     // [S][I * J + J]
  int64_t lda, ldb, I, J;
  int tile_begin;   // first tile index of this problem; tiles_j column tiles per row tile
  int tiles_j;
};
constexpr int kMaxDw256 = 8;
struct Dw256Table {
  Dw256Problem p[kMaxDw256];
  int n, tiles;     // problems, tiles of all problems
  int64_t M, rows_per_split;
};

__device__ __forceinline__ int swz_tn(int row) { return 2 * ((row & 3) | (((row >> 3) & 1) << 2)); }

__global__ void __launch_bounds__(kT, 2)
tn256_kernel(Dw256Table tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // split-major ids (the tiles of one row range are neighbours), then the XCD remap
  const unsigned total = gridDim.x;
  const unsigned hw = blockIdx.x;
  const unsigned chunk = total / 8, rem = total % 8;
  const unsigned xcd = hw % 8, idx = hw / 8;
  const unsigned logical = xcd * chunk + (xcd < rem ? xcd : rem) + idx;
  const int zsplit = (int)(logical / (unsigned)tab.tiles);
  const int tile = (int)(logical % (unsigned)tab.tiles);
  int pi = 0;
#pragma unroll
  for (int q = 1; q < kMaxDw256; ++q)
    if (q < tab.n && tile >= tab.p[q].tile_begin) pi = q;
  const Dw256Problem pr = tab.p[pi];
  const int lt = tile - pr.tile_begin;
  const int64_t i0 = (int64_t)(lt / pr.tiles_j) * TB;
  const int64_t j0 = (int64_t)(lt % pr.tiles_j) * TB;
  const int64_t I = pr.I, J = pr.J;
  const int64_t r_begin = (int64_t)zsplit * tab.rows_per_split;
  int64_t r_end = r_begin + tab.rows_per_split;
  r_end = r_end < tab.M ? r_end : tab.M;
  const int nk = r_end > r_begin ? (int)((r_end - r_begin) / KB) : 0;  // whole slots (M % 32 == 0)

  // ---- staging: instruction q of an operand = rows 2q, 2q + 1 of the slot (2 x 512 bytes);
  // wave w issues q = 2w, 2w + 1 ------------------------------------------------------------
  const char* ga[2];
  const char* gb[2];
  unsigned la[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = (wave * 2 + q) * 2 + (lane >> 5);
    const int src_chunk = (lane & 31) ^ swz_tn(row);
    // column chunks beyond the operand's row stay inside it (their products land in output
    // rows / columns that are never stored)
    const int64_t ca = i0 + src_chunk * 8 < pr.lda ? i0 + src_chunk * 8 : 0;
    const int64_t cb = j0 + src_chunk * 8 < pr.ldb ? j0 + src_chunk * 8 : 0;
    ga[q] = reinterpret_cast<const char*>(pr.A + (r_begin + row) * pr.lda + ca);
    gb[q] = reinterpret_cast<const char*>(pr.B + (r_begin + row) * pr.ldb + cb);
    la[q] = (unsigned)((wave * 2 + q) * 1024);
  }
  const int64_t stride_a = (int64_t)KB * pr.lda * 2, stride_b = (int64_t)KB * pr.ldb * 2;
  auto stage = [&](int t) {
    const int tc = t < nk ? t : (nk > 0 ? nk - 1 : 0);
    unsigned char* slot = lds + (t & (NSLOT - 1)) * kSlotBytes;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(ga[q] + tc * stride_a),
                                       (lds_ptr_t)(slot + la[q]), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(gb[q] + tc * stride_b),
                                       (lds_ptr_t)(slot + kOpBytes + la[q]), 16, 0, 0);
  };

  f32x4 acc[TM][TN], accb[TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < TN; ++b) accb[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = i0 == 0 && wm == 0;  // column sums of dZ: once per column tile
  typedef __attribute__((ext_vector_type(8))) short ones_s16x8;
  const bf16x8 ones = __builtin_bit_cast(
      bf16x8, ones_s16x8{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80});

  // transposing fragment gather (gemm_bf16.hip: frag): lane 4q + p of each 16-lane group g
  // addresses row 8g + q (+ 4 for the second read), columns c0 + 4p .. 4p + 3
  const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  const int xo = swz_tn(8 * g + tq);  // the same for row + 4
  unsigned oa[TM], ob[TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
    oa[a] = (unsigned)((8 * g + tq) * 512 + (((wm * 16 + 2 * a + (tp >> 1)) ^ xo) * 16) +
                       8 * (tp & 1));
#pragma unroll
  for (int b = 0; b < TN; ++b)
    ob[b] = (unsigned)(kOpBytes + (8 * g + tq) * 512 +
                       (((wn * 8 + 2 * b + (tp >> 1)) ^ xo) * 16) + 8 * (tp & 1));
  using lds_s16x4 = __attribute__((address_space(3))) s16x4;
  auto frag = [&](const unsigned char* slot, unsigned off) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(slot + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(slot + off + 4 * 512));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  stage(0);
  stage(1);
  stage(2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (int t = 0; t < nk; ++t) {
    stage(t + 3);
    const unsigned char* slot = lds + (t & (NSLOT - 1)) * kSlotBytes;
    bf16x8 af[TM], bfr[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) bfr[b] = frag(slot, ob[b]);
#pragma unroll
    for (int a = 0; a < TM; ++a) af[a] = frag(slot, oa[a]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
        // dZ fragment first: a lane ends with 4 CONSECUTIVE output columns of one row
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
    if (do_bias) {
#pragma unroll
      for (int b = 0; b < TN; ++b)
        accb[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b], ones, accb[b], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  float* slab = pr.slabs + (int64_t)zsplit * (I * J + J);
  const int li = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int64_t i = i0 + wm * 128 + a * 16 + li;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int64_t j = j0 + wn * 64 + b * 16 + 4 * lq;
      if (i < I && j < J) *reinterpret_cast<f32x4*>(slab + i * J + j) = acc[a][b];  // J % 8 == 0
    }
  }
  if (do_bias && li == 0) {  // every row of the ones-tile holds the same sums
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int64_t j = j0 + wn * 64 + b * 16 + 4 * lq;
      if (j < J) *reinterpret_cast<f32x4*>(slab + I * J + j) = accb[b];
    }
  }
}


// ---- dW on 128 x 128 tiles with the SAME staging: the small problems of a training-size group ----
// (the MLP trunks' 64- and 256-wide layers, heads, first layers).  gemm_bf16.hip's tile kernel
// stages a 64-row tile global -> registers -> ds_write and spends 880 + 370 of its ~3 000 cycles
// per tile on that (store-to-LDS with the wait for its loads, load issue) beside 1 170 of
// multiplying; here a slot is 32 rows of 256 bytes of each operand, DMA'd straight into LDS
// (one instruction = four rows), four 16-KB slots in a ring (64 KB: two workgroups per CU), the
// wait a counted vmcnt and the barrier raw — tn256_kernel's pipeline with 4 waves (2 x 2, each
// 64 x 64 = 4 x 4 MFMA tiles).  A row is exactly one 256-byte bank row, so the chunk swizzle of
// the 512-byte rows carries over unchanged: chunk c of row r at position c ^ 2 f(r).  Same
// split plan, same k order (32 rows per MFMA k-step, ascending), same operand order as the
// tile kernel: the slabs are bit-identical.
constexpr int kT128 = 256;
constexpr int TB128 = 128;
constexpr int kRow128 = TB128 * 2;             // bytes of a staged row
constexpr int kOp128 = KB * kRow128;           // one operand of one slot: 8 KB
constexpr int kSlot128 = 2 * kOp128;
constexpr int kLds128 = NSLOT * kSlot128;      // 64 KB
constexpr int TM128 = 4, TN128 = 4;

__global__ void __launch_bounds__(kT128, 2)
tn128_kernel(Dw256Table tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // split-major ids (the tiles of one row range are neighbours), then the XCD remap
  const unsigned total = gridDim.x;
  const unsigned hw = blockIdx.x;
  const unsigned chunk = total / 8, rem = total % 8;
  const unsigned xcd = hw % 8, idx = hw / 8;
  const unsigned logical = xcd * chunk + (xcd < rem ? xcd : rem) + idx;
  const int zsplit = (int)(logical / (unsigned)tab.tiles);
  const int tile = (int)(logical % (unsigned)tab.tiles);
  int pi = 0;
#pragma unroll
  for (int q = 1; q < kMaxDw256; ++q)
    if (q < tab.n && tile >= tab.p[q].tile_begin) pi = q;
  const Dw256Problem pr = tab.p[pi];
  const int lt = tile - pr.tile_begin;
  const int64_t i0 = (int64_t)(lt / pr.tiles_j) * TB128;
  const int64_t j0 = (int64_t)(lt % pr.tiles_j) * TB128;
  const int64_t I = pr.I, J = pr.J;
  const int64_t r_begin = (int64_t)zsplit * tab.rows_per_split;
  int64_t r_end = r_begin + tab.rows_per_split;
  r_end = r_end < tab.M ? r_end : tab.M;
  const int nk = r_end > r_begin ? (int)((r_end - r_begin) / KB) : 0;  // whole slots (M % 32 == 0)

  // staging: instruction q of an operand = rows 4q .. 4q + 3 of the slot; wave w issues 2w, 2w + 1
  const char* ga[2];
  const char* gb[2];
  unsigned la[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = (wave * 2 + q) * 4 + (lane >> 4);
    const int src_chunk = (lane & 15) ^ swz_tn(row);
    // column chunks beyond the operand's row stay inside it (their products land in output
    // rows / columns that are never stored)
    const int64_t ca = i0 + src_chunk * 8 < pr.lda ? i0 + src_chunk * 8 : 0;
    const int64_t cb = j0 + src_chunk * 8 < pr.ldb ? j0 + src_chunk * 8 : 0;
    ga[q] = reinterpret_cast<const char*>(pr.A + (r_begin + row) * pr.lda + ca);
    gb[q] = reinterpret_cast<const char*>(pr.B + (r_begin + row) * pr.ldb + cb);
    la[q] = (unsigned)((wave * 2 + q) * 1024);
  }
  const int64_t stride_a = (int64_t)KB * pr.lda * 2, stride_b = (int64_t)KB * pr.ldb * 2;
  auto stage = [&](int t) {
    const int tc = t < nk ? t : (nk > 0 ? nk - 1 : 0);
    unsigned char* slot = lds + (t & (NSLOT - 1)) * kSlot128;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(ga[q] + tc * stride_a),
                                       (lds_ptr_t)(slot + la[q]), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(gb[q] + tc * stride_b),
                                       (lds_ptr_t)(slot + kOp128 + la[q]), 16, 0, 0);
  };

  f32x4 acc[TM128][TN128], accb[TN128];
#pragma unroll
  for (int a = 0; a < TM128; ++a)
#pragma unroll
    for (int b = 0; b < TN128; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < TN128; ++b) accb[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = i0 == 0 && wm == 0;  // column sums of dZ: once per column tile
  typedef __attribute__((ext_vector_type(8))) short ones_s16x8;
  const bf16x8 ones = __builtin_bit_cast(
      bf16x8, ones_s16x8{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80});

  // transposing fragment gather: lane 4q + p of each 16-lane group g addresses row 8g + q
  // (+ 4 for the second read), columns c0 + 4p .. 4p + 3
  const int g = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  const int xo = swz_tn(8 * g + tq);  // the same for row + 4
  unsigned oa[TM128], ob[TN128];
#pragma unroll
  for (int a = 0; a < TM128; ++a)
    oa[a] = (unsigned)((8 * g + tq) * kRow128 + (((wm * 8 + 2 * a + (tp >> 1)) ^ xo) * 16) +
                       8 * (tp & 1));
#pragma unroll
  for (int b = 0; b < TN128; ++b)
    ob[b] = (unsigned)(kOp128 + (8 * g + tq) * kRow128 +
                       (((wn * 8 + 2 * b + (tp >> 1)) ^ xo) * 16) + 8 * (tp & 1));
  using lds_s16x4 = __attribute__((address_space(3))) s16x4;
  auto frag = [&](const unsigned char* slot, unsigned off) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(slot + off));
    const s16x4 hi =
        __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(slot + off + 4 * kRow128));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  stage(0);
  stage(1);
  stage(2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (int t = 0; t < nk; ++t) {
    stage(t + 3);
    const unsigned char* slot = lds + (t & (NSLOT - 1)) * kSlot128;
    bf16x8 af[TM128], bfr[TN128];
#pragma unroll
    for (int b = 0; b < TN128; ++b) bfr[b] = frag(slot, ob[b]);
#pragma unroll
    for (int a = 0; a < TM128; ++a) af[a] = frag(slot, oa[a]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < TM128; ++a)
#pragma unroll
      for (int b = 0; b < TN128; ++b)
        // dZ fragment first: a lane ends with 4 CONSECUTIVE output columns of one row
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
    if (do_bias) {
#pragma unroll
      for (int b = 0; b < TN128; ++b)
        accb[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b], ones, accb[b], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  float* slab = pr.slabs + (int64_t)zsplit * (I * J + J);
  const int li = lane & 15, lq = lane >> 4;
  const int In = (int)I, Jn = (int)J;
  const bool vec = (Jn & 3) == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0;
#pragma unroll
  for (int a = 0; a < TM128; ++a) {
    const int i = (int)i0 + wm * 64 + a * 16 + li;
#pragma unroll
    for (int b = 0; b < TN128; ++b) {
      const int j = (int)j0 + wn * 64 + b * 16 + 4 * lq;
      if (i < In && j < Jn) {
        float* dst = slab + i * Jn + j;
        if (vec) {
          *reinterpret_cast<f32x4*>(dst) = acc[a][b];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j + e < Jn) dst[e] = acc[a][b][e];
        }
      }
    }
  }
  if (do_bias && li == 0) {  // every row of the ones-tile holds the same sums
#pragma unroll
    for (int b = 0; b < TN128; ++b) {
      const int j = (int)j0 + wn * 64 + b * 16 + 4 * lq;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (j + e < Jn) slab[(int64_t)In * Jn + j + e] = accb[b][e];
    }
  }
}

}  // namespace

namespace mippo_gemm {

int nt256_launch(int epi, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int64_t I,
                 int64_t J, int64_t R, const Epi& ep, hipStream_t st) {
  static const int enabled = [] {  // MIPPO_GEMM256=0: the 128-row kernel everywhere (A/B)
    const char* e = getenv("MIPPO_GEMM256");
    return !(e && e[0] == '0');
  }();
  if (!enabled) return 0;
  if (R < KB || R % KB || J < 128 || J % 8 || I < 2048) return 0;
  if (ep.out_f32 || ep.aux_bf || !ep.out_bf) return 0;
  if (epi == EPI_FWD && ep.act == MI_ACT_SWISH) return 0;
  // (the dX epilogue applies act' to the ROUNDED gradient chunk: exact for relu' in {0, 1},
  // a second rounding for anything else — those keep the 128-row kernel)
  if (epi == EPI_DX && ep.prev && ep.prev_act != MI_ACT_NONE && ep.prev_act != MI_ACT_RELU)
    return 0;
  if (lda % 8 || ldb % 8 || ep.ld_bf % 8 || !al16(A) || !al16(B) || !al16(ep.out_bf)) return 0;
  if (epi == EPI_DX && ep.prev && (ep.ld_prev % 8 || !al16(ep.prev) || ep.ld_prev < ep.ld_bf))
    return 0;
  if (epi == EPI_FWD && ep.bias && (reinterpret_cast<uintptr_t>(ep.bias) & 15)) return 0;
  // the padding columns J .. ld_bf - 1 of the image are written (as zero) by the last column
  // tile only if they fall inside it
  const int64_t tiles_i = mippo::ceil_div(I, TB), tiles_j = mippo::ceil_div(J, TB);
  if (ep.ld_bf > tiles_j * TB) return 0;
  static bool attr_set[2] = {false, false};
  const void* fn = epi == EPI_FWD ? reinterpret_cast<const void*>(&nt256_kernel<EPI_FWD>)
                                  : reinterpret_cast<const void*>(&nt256_kernel<EPI_DX>);
  if (!attr_set[epi]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) !=
        hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    attr_set[epi] = true;
  }
  const dim3 grid((unsigned)(tiles_i * tiles_j));
  if (epi == EPI_FWD)
    hipLaunchKernelGGL((nt256_kernel<EPI_FWD>), grid, dim3(kT), kLdsBytes, st, A, lda, B, ldb, I,
                       J, R, ep, (int)tiles_j);
  else
    hipLaunchKernelGGL((nt256_kernel<EPI_DX>), grid, dim3(kT), kLdsBytes, st, A, lda, B, ldb, I,
                       J, R, ep, (int)tiles_j);
  const int rc = mippo::check_launch("nt256_gemm_bf16");
  return rc ? rc : 1;
}

// The dW problems of a grouped launch that the 256 x 256 kernel takes: K and N multiples of 8
// covering at least four tiles, M a multiple of 32 and >= 8192.  dw256_plan decides the split for
// the taken problems together; dw256_launch runs them in ONE launch and reports slab
// positions / counts like the 128-row path.
// a problem the kernel can take as part of a group of >= 4 tiles: whole 256-row / 256-column
// tiles (a partial tile multiplies padding), training-size M
bool dw256_candidate(int64_t K, int64_t N, int64_t M) {
  static const int enabled = [] {
    const char* e = getenv("MIPPO_GEMM256");
    return !(e && e[0] == '0');
  }();
  return enabled && K % TB == 0 && N % TB == 0 && M >= 8192 && M % KB == 0;
}

bool dw256_takes(int64_t K, int64_t N, int64_t M) {
  static const int enabled = [] {
    const char* e = getenv("MIPPO_GEMM256");
    return !(e && e[0] == '0');
  }();
  // at least four 256 x 256 tiles: with fewer, filling the chip takes so many M-splits that
  // the fp32 slabs (S x K x N x 4 bytes, written and read again) outweigh the operands —
  // measured at M = 61 440 (tools/microbench_dw256.py): 512 x 512 alone 72 -> 64 us, inside the
  // 17-512-512-1 critic's group 137 -> 107 us; 256 x 256 alone 34 -> 42 us (the 128-row kernel
  // keeps it)
  return enabled && K % 8 == 0 && N % 8 == 0 && M >= 8192 && M % KB == 0 &&
         mippo::ceil_div(K, TB) * mippo::ceil_div(N, TB) >= 4;
}

void dw256_plan(int64_t M, int64_t tiles, int64_t* rows, int64_t* S) {
  static const int64_t target = [] {  // MIPPO_DW256_BLOCKS: workgroups aimed at (tuning aid)
    const char* e = getenv("MIPPO_DW256_BLOCKS");
    return e ? (int64_t)atoi(e) : (int64_t)mippo::kNumCU;
  }();
  int64_t s = target / (tiles < 1 ? 1 : tiles);
  if (s < 1) s = 1;
  const int64_t max_s = M / 256 < 1 ? 1 : M / 256;  // at least 8 slots per workgroup
  if (s > max_s) s = max_s;
  *rows = mippo::ceil_div(mippo::ceil_div(M, s), KB) * KB;
  *S = mippo::ceil_div(M, *rows);
}

int64_t dw256_tiles(int64_t K, int64_t N) { return mippo::ceil_div(K, TB) * mippo::ceil_div(N, TB); }

int dw256_launch(int n, const bf16_t* const* x_bf, const bf16_t* const* dz_bf, const int64_t* K,
                 const int64_t* N, int64_t M, float* const* slabs, int64_t rows, int64_t S,
                 hipStream_t st) {
  if (n < 1 || n > kMaxDw256) return -EINVAL;
  Dw256Table tab = {};
  tab.n = n;
  tab.M = M;
  tab.rows_per_split = rows;
  int tiles = 0;
  for (int l = 0; l < n; ++l) {
    Dw256Problem& pr = tab.p[l];
    pr.A = x_bf[l];
    pr.B = dz_bf[l];
    pr.slabs = slabs[l];
    pr.lda = mippo::ceil_div(K[l], 8) * 8;
    pr.ldb = mippo::ceil_div(N[l], 8) * 8;
    pr.I = K[l];
    pr.J = N[l];
    pr.tile_begin = tiles;
    pr.tiles_j = (int)mippo::ceil_div(N[l], TB);
    tiles += (int)dw256_tiles(K[l], N[l]);
  }
  tab.tiles = tiles;
  static const hipError_t attr = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&tn256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
      kLdsBytes);
  if (attr != hipSuccess) {
    mippo::set_error("dw256_launch: cannot raise the LDS limit");
    return -EIO;
  }
  hipLaunchKernelGGL(tn256_kernel, dim3((unsigned)(tiles * S)), dim3(kT), kLdsBytes, st, tab);
  return mippo::check_launch("tn256_gemm_dw_bf16");
}

// The 128-tile form (tn128_kernel) for the rest of a group: M a multiple of 32, splits of whole
// slots; MIPPO_DW128_DMA=0 keeps the register-staged tile kernel (A/B, bit-identity tests).
bool dw128_takes(int64_t M, int64_t rows_per_split, int n, int64_t wide_tiles, int64_t tiles) {
  const char* e = getenv("MIPPO_DW128_DMA");  // read per launch: the tests switch it
  const bool enabled = !(e && e[0] == '0');
  // short splits only: the pipeline's prologue and epilogue are cheaper than the register-staged
  // kernel's, its steady state (a barrier per 32 rows) is not — at 960 rows per workgroup the
  // launches tie (C2) or this one wins (C4: +2.6 % of the iteration), at 3 840 rows (N = 16 384)
  // the tile kernel is 11 % of the iteration ahead.  MIPPO_DW128_DMA=2 forces it for any length.
  // ... and groups made mostly of full-width tiles: every tile costs this kernel a 128 x 128
  // tile's work, while the tile kernel walks outputs of at most 64 / 16 columns on cheaper
  // classes (C2: 7 of 13 tiles are such: the tile kernel is 0.8 us ahead; C4: 4 of 14)
  const bool any_len = e && e[0] == '2';
  return enabled && n >= 1 && n <= kMaxDw256 && M >= KB && M % KB == 0 &&
         rows_per_split % KB == 0 &&
         (any_len || (rows_per_split <= 1024 && 2 * wide_tiles > tiles));
}

int dw128_launch(int n, const bf16_t* const* x_bf, const bf16_t* const* dz_bf, const int64_t* K,
                 const int64_t* N, int64_t M, float* const* slabs, int64_t rows, int64_t S,
                 hipStream_t st) {
  if (n < 1 || n > kMaxDw256) return -EINVAL;
  Dw256Table tab = {};
  tab.n = n;
  tab.M = M;
  tab.rows_per_split = rows;
  int tiles = 0;
  for (int l = 0; l < n; ++l) {
    Dw256Problem& pr = tab.p[l];
    pr.A = x_bf[l];
    pr.B = dz_bf[l];
    pr.slabs = slabs[l];
    pr.lda = mippo::ceil_div(K[l], 8) * 8;
    pr.ldb = mippo::ceil_div(N[l], 8) * 8;
    pr.I = K[l];
    pr.J = N[l];
    pr.tile_begin = tiles;
    pr.tiles_j = (int)mippo::ceil_div(N[l], TB128);
    tiles += (int)(mippo::ceil_div(K[l], TB128) * mippo::ceil_div(N[l], TB128));
  }
  tab.tiles = tiles;
  static const hipError_t attr = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&tn128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
      kLds128);
  if (attr != hipSuccess) {
    mippo::set_error("dw128_launch: cannot raise the LDS limit");
    return -EIO;
  }
  hipLaunchKernelGGL(tn128_kernel, dim3((unsigned)(tiles * S)), dim3(kT128), kLds128, st, tab);
  return mippo::check_launch("tn128_gemm_dw_bf16");
}

}  // namespace mippo_gemm
