// The forward half of the weights-stationary trunk kernels (trunk_ws.hip's header comment has
// the design): chain descriptors, wave geometry, LDS layout and `ws_fwd_body`, shared by
// trunk_ws.hip (one policy step / one replay per launch) and rollout_ws.hip (a whole T-step
// rollout per launch).  Everything sits in an anonymous namespace: each translation unit gets
// its own copy.
#pragma once
#include <stdlib.h>

#include "bf16_common.h"
#include "sampler_math.h"

namespace {

using namespace mippo_bf16;

constexpr int WS_MAXL = 8;

// -DMIPPO_TRACE (tools/trace_ws.py): thread 0 of every workgroup stamps the shader clock at
// the phase boundaries of its first two row tiles; compiled out of the product build.
#ifdef MIPPO_TRACE
constexpr int WST_EV = 32, WST_WG = 512;
__device__ unsigned long long g_ws_trace[WST_WG * WST_EV];
#define WS_TR_BLOCK blockIdx.x
#define WS_TR()                                                                        \
  do {                                                                                 \
    if (tid == 0 && ev_ < WST_EV && WS_TR_BLOCK < WST_WG)                              \
      g_ws_trace[WS_TR_BLOCK * WST_EV + ev_] = __builtin_amdgcn_s_memtime();           \
    ++ev_;                                                                             \
  } while (0)
#else
#define WS_TR() do {} while (0)
#endif

struct WsLayer {
  const bf16_t* w;    // forward fragment-major image (gemm_bf16.hip: frag_store)
  const float* bias;  // [N] or null
  bf16_t* out_bf;     // [M][ldo] bf16 image of this layer's output, or null
  int64_t ldo;
  // relu' of this layer's output, 4 bits per lane of the transposed MFMA tile, or null:
  // 16 x 16 tile (row tile rt, column tile ct) has byte [(((rt >> 2) * N/16 + ct) * 64 + lane)
  // * 4 + (rt & 3)], bit e = y[16 rt + (lane & 15)][16 ct + 4 (lane >> 4) + e] > 0 — the four
  // row tiles of a 64-row block side by side, so a wave that owns RTW of them moves RTW bytes
  // per lane in one access.  The backward reads ONE byte per lane and tile instead of 8 bytes
  // of the bf16 image (trunk_ws backward, MASK).
  unsigned char* mask_out;
};

struct WsChain {
  WsLayer layer[WS_MAXL];  // layer 0: K0 -> H; 1..NH: H -> H; NH+1: H -> N_out
  const float* x;          // [M_head][K0] fp32
  const float* x_tail;     // rows M_head .. M-1 come from here ([M - M_head][K0]), or null
  int64_t M_head;          // = M without a tail
  bf16_t* x_bf;            // [M][ldx] bf16 image of the (normalised) input, or null
  int64_t ldx;
  float* out;              // [M][N_out] fp32
  int64_t M;
  int K0, N_out;
  // policy step (mi_policy_ws_fwd_bf16): the running-statistics normaliser in the input
  // stage (normalizer.py:76-96) and, for the action trunk, the sampler on the head's rows
  const float* norm_mean;  // [K0] or null
  const float* norm_m2;
  const float* norm_count;
  float norm_eps;
  mippo_sampler::FwdParams samp;  // samp.A == 0: no sampler
};

// block (ct, ks) of a fragment-major image with KS k-steps per column tile: the 16 bytes of
// this lane
__device__ inline bf16x8 ws_frag(const bf16_t* w, unsigned ct, unsigned ks, unsigned KS, int lane) {
  const char* p = reinterpret_cast<const char*>(w) + (((size_t)ct * KS + ks) << 10) + lane * 16;
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
}

// relu' masks (WsLayer::mask_out): the RTW bytes of row tiles rt0 .. rt0 + RTW - 1 (rt0 a
// multiple of RTW, RTW in {1, 2, 4}) of column tile ct, for this lane, as one access.
template <int RTW>
__device__ inline size_t ws_mask_index(int64_t rt0, int n_ct, int ct, int lane) {
  static_assert(RTW == 1 || RTW == 2 || RTW == 4, "a wave owns 1, 2 or 4 row tiles");
  return (size_t)((((rt0 >> 2) * n_ct + ct) * 64 + lane) * 4 + (rt0 & 3));
}
template <int RTW>
__device__ inline void ws_mask_store(unsigned char* m, int64_t rt0, int n_ct, int ct, int lane,
                                     unsigned pack) {
  unsigned char* p = m + ws_mask_index<RTW>(rt0, n_ct, ct, lane);
  if constexpr (RTW == 4) *reinterpret_cast<unsigned int*>(p) = pack;
  else if constexpr (RTW == 2) *reinterpret_cast<unsigned short*>(p) = (unsigned short)pack;
  else *p = (unsigned char)pack;
}
template <int RTW>
__device__ inline unsigned ws_mask_load(const unsigned char* m, int64_t rt0, int n_ct, int ct,
                                        int lane) {
  const unsigned char* p = m + ws_mask_index<RTW>(rt0, n_ct, ct, lane);
  if constexpr (RTW == 4) return *reinterpret_cast<const unsigned int*>(p);
  else if constexpr (RTW == 2) return *reinterpret_cast<const unsigned short*>(p);
  else return *p;
}

// H: hidden width; NH: number of H x H layers; RT: 16-row tiles per row tile.
//
// 512 threads = 8 waves, two per SIMD (<= 256 registers each): while one wave of a SIMD
// is in an epilogue, a copy-out or the input stage, its partner multiplies.  The waves
// split a layer's output tile CW column groups x RW row groups (CW * RW = 8): 256 columns
// -> 8 x 1 with two 16-column tiles per wave, 128 -> 8 x 1, 64 -> 4 x 2.
template <int H>
struct WsGeom {
  static constexpr int CW = H / 16 < 8 ? H / 16 : 8;  // column groups
  static constexpr int RW = 8 / CW;                   // row groups
  static constexpr int TPW = (H / 16) / CW;           // column tiles per wave
};
constexpr int kWsThreads = 512;

// SAMP: the trunk ends in the sampler (its own instantiations: the transcendental row
// function does not belong in the register budget of the plain trunks).
// The body is a device function of (chain, workgroup index, workgroup count) so that one
// launch can run two trunks side by side (policy_ws_dual_kernel below); its LDS is
// function-scope, i.e. allocated per kernel that reaches it.
// LDS of one forward body (carved out of the kernel's one array, so a two-trunk launch
// needs the larger of the two, not their sum)
template <int H, int RT, bool SAMP>
struct WsFwdLds {
  static constexpr size_t kRows = 16 * RT;
  static constexpr size_t bufX = 0;
  static constexpr size_t bufA = bufX + kRows * (32 + 8) * 2;
  static constexpr size_t bufB = bufA + kRows * (H + 8) * 2;
  static constexpr size_t mean = bufB + kRows * (H + 8) * 2;
  static constexpr size_t sd = mean + 32 * 4;
  static constexpr size_t stash = sd + 32 * 4;
  static constexpr size_t wo = stash + (SAMP ? 4096 : 4) * 4;  // head fragments: H/32 KiB
  static constexpr size_t bytes = wo + (H / 32) * 1024;
};

// FULL: a training launch whose every row tile is whole (M, and the head's M_head, multiples
// of 16 RT) and that writes every image and mask — the guards around the tile loop's
// stores are then compile-time true.  That matters beyond the guards' own cost: vmcnt counts
// loads and stores in order, and the wait for the NEXT tile's prefetched input may leave as
// many younger instructions in flight as EVERY control-flow path issued; behind a data- or
// pointer-dependent guard one path has no stores at all, so the wait became vmcnt(0) and each
// tile's input stage sat until the previous tile's ~20 copy-out stores had reached memory.
template <int H, int NH, int RT, bool SAMP, bool FULL = false>
__device__ __forceinline__ void ws_fwd_body(const WsChain& c, const int bid, const int nblk,
                                            unsigned char* smem) {
  static_assert(H == 64 || H == 128 || H == 256, "hidden width: 64, 128 or 256");
  using G = WsGeom<H>;
  constexpr int CW = G::CW, RW = G::RW, TPW = G::TPW;
  static_assert(RT % RW == 0 && RT <= 8, "row tiles must split over the row groups");
  constexpr int RTW = RT / RW;  // row tiles per wave
  constexpr int ROWS = 16 * RT;
  constexpr int KSH = H / 32;   // k-steps of an H-deep reduce
  constexpr int AROW = H + 8;   // LDS row (bf16): 16 bytes of padding
  constexpr int XROW = 32 + 8;  // input tile row: K0 <= 32 columns
  using Lds = WsFwdLds<H, RT, SAMP>;
  bf16_t* const bufX = reinterpret_cast<bf16_t*>(smem + Lds::bufX);  // [ROWS][XROW]
  bf16_t* const bufA = reinterpret_cast<bf16_t*>(smem + Lds::bufA);  // [ROWS][AROW]
  bf16_t* const bufB = reinterpret_cast<bf16_t*>(smem + Lds::bufB);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % CW, wr = wave / CW;
  const int li = lane & 15, lq = lane >> 4;
  const int64_t M = c.M;
  const int K0 = c.K0, N_out = c.N_out;
  const int64_t ntiles = (M + ROWS - 1) / ROWS;
#ifdef MIPPO_TRACE
  int ev_ = 0;
#endif
  WS_TR();  // 0: start

  // ---- input tile: fp32 [ROWS][K0] is one contiguous run; element e of it = (row e / K0,
  // column e % K0) -------------------------------------------------------------------------
  constexpr int IN_PT = (ROWS * 32 + kWsThreads - 1) / kWsThreads;  // K0 <= 32
  const int nel = ROWS * K0;
  const float rcpK0 = 1.0f / (float)K0;
  float xin[IN_PT];
  const int64_t M_head = c.M_head;
  auto request_input = [&](int64_t tile) {
    const int64_t i0 = tile * ROWS;
#pragma unroll
    for (int u = 0; u < IN_PT; ++u) {
      const int e = tid + u * kWsThreads;
      const int row = (int)(((float)e + 0.5f) * rcpK0);
      const int64_t gi = i0 + row;
      xin[u] = 0.0f;
      if (e < nel && gi < M) {
        const int k = e - row * K0;
        xin[u] = gi < M_head ? c.x[gi * K0 + k] : c.x_tail[(gi - M_head) * K0 + k];
      }
    }
  };
  // normaliser statistics of the K0 input columns, once (same fp32 expressions as
  // normalize_fwd_kernel / the input stage of mlp_bf16.hip)
  float* const s_mean = reinterpret_cast<float*>(smem + Lds::mean);  // [32]
  float* const s_sd = reinterpret_cast<float*>(smem + Lds::sd);
  const bool norm = c.norm_mean != nullptr;
  if (norm && tid < K0) {
    const float cnt = *c.norm_count;
    s_mean[tid] = c.norm_mean[tid];
    s_sd[tid] = cnt > 0.0f ? sqrtf(fmaxf(c.norm_m2[tid] / cnt, c.norm_eps)) : 10.0f;
  }
  int64_t tile = bid;
  // the first input tile and layer 0's fragments go out FIRST: the (much larger) rest of the
  // trunk arrives while the first row tile's input stage and layer 0 run
  if (tile < ntiles) request_input(tile);

  // ---- the trunk, once: weight fragments and biases of this wave's column tiles --------
  bf16x8 W0[TPW];
  bf16x8 WH[NH > 0 ? NH : 1][TPW][KSH];
  f32x4 B0[TPW], BH[NH > 0 ? NH : 1][TPW], BO;
  // the head's fragments (one column tile, KSH k-steps) wait in LDS: only RT of the 8 waves
  // multiply the head, and 4 KSH registers per lane in every wave are what pushes the
  // 256-wide trunk over the register file
  bf16_t* const wo_s = reinterpret_cast<bf16_t*>(smem + Lds::wo);  // [KSH][64 lanes][8]
  if (tid < KSH * 64)
    *reinterpret_cast<u32x4*>(wo_s + tid * 8) =
        *reinterpret_cast<const u32x4*>(c.layer[NH + 1].w + tid * 8);
#pragma unroll
  for (int b = 0; b < TPW; ++b) {
    const unsigned ct = (unsigned)(wc + CW * b);
    W0[b] = ws_frag(c.layer[0].w, ct, 0, 1, lane);
    B0[b] = c.layer[0].bias
                ? *reinterpret_cast<const f32x4*>(c.layer[0].bias + ct * 16 + 4 * lq)
                : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int b = 0; b < TPW; ++b) {
    const unsigned ct = (unsigned)(wc + CW * b);
#pragma unroll
    for (int l = 0; l < NH; ++l) {
#pragma unroll
      for (int ks = 0; ks < KSH; ++ks) WH[l][b][ks] = ws_frag(c.layer[1 + l].w, ct, ks, KSH, lane);
      BH[l][b] = c.layer[1 + l].bias
                     ? *reinterpret_cast<const f32x4*>(c.layer[1 + l].bias + ct * 16 + 4 * lq)
                     : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  BO = f32x4{0.f, 0.f, 0.f, 0.f};
  if (c.layer[NH + 1].bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * lq + e < N_out) BO[e] = c.layer[NH + 1].bias[4 * lq + e];
  }
  // the input buffer's pad columns K0..31 are zero for the whole kernel (nothing else is
  // ever written there)
  for (int row = tid >> 5; row < ROWS; row += kWsThreads >> 5)
    for (int k = K0 + (tid & 31); k < 32; k += 32) bufX[row * XROW + k] = (bf16_t)0.0f;
  __syncthreads();  // s_mean / s_sd
  WS_TR();  // 1: trunk and first input requested

  // sampling_layers.py:82-147 on the action trunk's rows.  One thread per row is a
  // ~6 000-cycle chain of transcendentals whatever the number of rows, so the head's fp32
  // rows of several row tiles are stashed in LDS and the sampler runs on all of them at
  // once (up to 512 rows: every thread busy), not once per 64-row tile.
  constexpr int kStashFloats = SAMP ? 4096 : 4;         // 16 KB
  constexpr int kStashTilesMax = kWsThreads / ROWS;     // one row per thread at most
  float* const ms_s = reinterpret_cast<float*>(smem + Lds::stash);  // [kStashFloats]
  constexpr bool has_samp = SAMP;
  int64_t stash_first = 0;  // row tile of stash slot 0; slot q holds tile + q * nblk
  int stash_n = 0;
  int stash_cap = has_samp ? kStashFloats / (ROWS * N_out) : 1;
  if (stash_cap > kStashTilesMax) stash_cap = kStashTilesMax;
  auto run_sampler = [&]() {
    const int slot = tid / ROWS, row = tid % ROWS;
    if (SAMP && slot < stash_n) {
      const int64_t gi = (stash_first + (int64_t)slot * nblk) * ROWS + row;
      if (gi < M) mippo_sampler::fwd_row(ms_s + (slot * ROWS + row) * N_out, gi, c.samp);
    }
    __syncthreads();  // the stash is free again
  };

  for (; tile < ntiles; tile += nblk) {
    const int64_t i0 = tile * ROWS;
    WS_TR();  // tile + 0
    // stage 0: the requested input tile -> bf16 in bufX
#pragma unroll
    for (int u = 0; u < IN_PT; ++u) {
      const int e = tid + u * kWsThreads;
      if (e < nel) {
        const int row = (int)(((float)e + 0.5f) * rcpK0);
        const int k = e - row * K0;
        float v = xin[u];
        if (norm && i0 + row < M) v = (v - s_mean[k]) / s_sd[k];
        bufX[row * XROW + k] = (bf16_t)v;
      }
    }
    // the next row tile's input is in flight while this one is computed
    if (tile + nblk < ntiles) request_input(tile + nblk);
    WS_TR();  // tile + 1: input staged
    __syncthreads();
    WS_TR();  // tile + 2
    if ((FULL || (c.x_bf && i0 + tid < M)) && tid < ROWS) {  // bf16 image of the input (dW operand)
      const int64_t ldx = c.ldx;                 // pad8(K0) <= 32 columns: 16-byte chunks
      for (int k = 0; k < (int)ldx; k += 8)
        *reinterpret_cast<u32x4*>(c.x_bf + (i0 + tid) * ldx + k) =
            *reinterpret_cast<const u32x4*>(bufX + tid * XROW + k);
    }

    f32x4 acc[RTW][TPW];
    // ---- layer 0: K0 (<= 32, one k-step) -> H ------------------------------------------
    {
      bf16x8 af[RTW];
#pragma unroll
      for (int r = 0; r < RTW; ++r)
        af[r] = *reinterpret_cast<const bf16x8*>(bufX + ((wr * RTW + r) * 16 + li) * XROW + 8 * lq);
#pragma unroll
      for (int b = 0; b < TPW; ++b)
#pragma unroll
        for (int r = 0; r < RTW; ++r)
          acc[r][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              W0[b], af[r], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    WS_TR();  // tile + 3: layer 0 multiplied
    // bias + relu + round to bf16 -> the next layer's LDS operand
    auto epilogue_hidden = [&](const f32x4(&bias)[TPW], bf16_t* nbuf, unsigned char* mask) {
#pragma unroll
      for (int b = 0; b < TPW; ++b) {
        const int col = (wc + CW * b) * 16 + 4 * lq;
        unsigned pack = 0;  // byte r: the mask of this wave's row tile r
#pragma unroll
        for (int r = 0; r < RTW; ++r) {
          bf16x4 vo;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            vo[e] = (bf16_t)fmaxf(acc[r][b][e] + bias[b][e], 0.0f);
            pack |= ((float)vo[e] > 0.0f ? 1u : 0u) << (8 * r + e);
          }
          *reinterpret_cast<bf16x4*>(nbuf + ((wr * RTW + r) * 16 + li) * AROW + col) = vo;
        }
        if (FULL || mask)
          ws_mask_store<RTW>(mask, (i0 >> 4) + wr * RTW, H / 16, wc + CW * b, lane, pack);
      }
    };
    // The layer's bf16 image, out of the published LDS buffer in whole rows: 16 bytes per
    // lane, a wave-instruction covers 1 KiB of consecutive addresses.  All the LDS reads of
    // a thread first, then its stores; the stores drain while the next layer multiplies.
    auto copy_out = [&](const bf16_t* buf, const WsLayer& ly) {
      if (!FULL && !ly.out_bf) return;
      constexpr int CPR = H / 8;               // 16-byte chunks per row
      constexpr int RPP = kWsThreads / CPR;    // rows per pass
      constexpr int NP = (ROWS + RPP - 1) / RPP;
      const int cc = tid % CPR, r0 = tid / CPR;
      u32x4 v[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int row = r0 + p * RPP < ROWS ? r0 + p * RPP : ROWS - 1;
        v[p] = *reinterpret_cast<const u32x4*>(buf + row * AROW + cc * 8);
      }
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int row = r0 + p * RPP;
        if ((ROWS % RPP == 0 || row < ROWS) && (FULL || i0 + row < M))
          *reinterpret_cast<u32x4*>(ly.out_bf + (i0 + row) * ly.ldo + cc * 8) = v[p];
      }
    };
    epilogue_hidden(B0, bufB, c.layer[0].mask_out);
    WS_TR();  // tile + 4: layer 0 epilogue
    __syncthreads();
    WS_TR();  // tile + 5
    copy_out(bufB, c.layer[0]);
    WS_TR();  // tile + 6: copy-out issued
    // ---- hidden layers: H -> H, activations ping-pong bufB -> bufA -> ... ---------------
    bf16_t* cur = bufB;
    bf16_t* nxt = bufA;
#pragma unroll
    for (int l = 0; l < NH; ++l) {
#pragma unroll
      for (int r = 0; r < RTW; ++r)
#pragma unroll
        for (int b = 0; b < TPW; ++b) acc[r][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSH; ++ks) {
        bf16x8 af[RTW];
#pragma unroll
        for (int r = 0; r < RTW; ++r)
          af[r] = *reinterpret_cast<const bf16x8*>(cur + ((wr * RTW + r) * 16 + li) * AROW +
                                                   ks * 32 + 8 * lq);
#pragma unroll
        for (int b = 0; b < TPW; ++b)
#pragma unroll
          for (int r = 0; r < RTW; ++r)
            acc[r][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WH[l][b][ks], af[r], acc[r][b],
                                                                0, 0, 0);
      }
      WS_TR();  // hidden: multiplied
      epilogue_hidden(BH[l], nxt, c.layer[1 + l].mask_out);
      WS_TR();  // hidden: epilogue
      __syncthreads();
      WS_TR();
      copy_out(nxt, c.layer[1 + l]);
      WS_TR();  // hidden: copy-out issued
      bf16_t* t = cur;
      cur = nxt;
      nxt = t;
    }
    // ---- head: H -> N_out (<= 16: one column tile); wave w takes row tile w -------------
    if (wave < RT) {
      f32x4 ah = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSH; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(cur + (wave * 16 + li) * AROW +
                                                          ks * 32 + 8 * lq);
        const bf16x8 wo = *reinterpret_cast<const bf16x8*>(wo_s + (ks * 64 + lane) * 8);
        ah = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo, a, ah, 0, 0, 0);
      }
      const int row = wave * 16 + li;
      const int64_t gi = i0 + row;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (4 * lq + e < N_out) {
          const float v = ah[e] + BO[e];
          if (FULL || gi < M) c.out[gi * N_out + 4 * lq + e] = v;
          // the sampler's input rows wait in LDS until the stash is full (below)
          if (has_samp) ms_s[(stash_n * ROWS + row) * N_out + 4 * lq + e] = v;
        }
      }
    }
    if (has_samp) {
      if (stash_n == 0) stash_first = tile;
      ++stash_n;
      if (stash_n == stash_cap || tile + nblk >= ntiles) {
        __syncthreads();  // the head's rows are in the stash
        run_sampler();
        stash_n = 0;
      }
    }
    WS_TR();  // head done
    __syncthreads();  // bufA / bufB / bufX are free for the next row tile
    WS_TR();  // tile end
  }
}

// ---- host side shared by the launchers ------------------------------------------------------
int ws_grid(int64_t ntiles) {
  static const int cus = [] {
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cu < 1) {
      (void)hipGetLastError();
      cu = mippo::kNumCU;
    }
    return cu;
  }();
  return (int)(ntiles < cus ? ntiles : cus);
}

// (value trunk, action trunk) pairs the one-launch form is instantiated for: the equal-width
// trunks `make_mlp_actor_critic` is usually called with, and BASELINE C2's 2x256 / 4x64.
#define WS_DUAL_MENU(X) \
  X(256, 1, 64, 3) X(256, 1, 64, 2) X(256, 1, 64, 1) X(256, 1, 256, 1) X(256, 0, 256, 0) \
  X(128, 1, 128, 1) X(128, 2, 128, 2) X(128, 1, 64, 1) X(64, 1, 64, 1) X(64, 2, 64, 2)    \
  X(64, 3, 64, 3)

bool ws_dual_has(int64_t hv, int64_t nhv, int64_t ha, int64_t nha) {
#define X(a, b, c, d) if (hv == a && nhv == b && ha == c && nha == d) return true;
  WS_DUAL_MENU(X)
#undef X
  return false;
}

}  // namespace
