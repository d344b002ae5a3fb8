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
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
        // transposed tile: weights as the first operand -> lane holds 4 consecutive columns
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
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

}  // namespace mippo_gemm
