// a9 (bf16 path) — a whole MLP trunk in ONE launch, forward or backward (dX
// chain).  See gemm_bf16.hip for the per-layer kernels and operand conventions.
//
// At this workload's shapes (K, N <= 512) a layer is one or two k-tiles of work:
// per-layer kernels are bound by launch latency and by the HBM round trip of
// every intermediate activation, not by MFMA rate.  Here a workgroup owns
// 16*RT rows (envs / samples) and walks them through every layer:
//   * activations ping-pong between two LDS buffers and never leave the CU; the
//     buffers are sized to the trunk's real width (dynamic LDS), so narrow trunks
//     keep many workgroups resident per CU;
//   * each layer's OUTPUT COLUMNS are split over the 4 waves, so a wave's weight
//     fragments are used by its own MFMAs only and go global -> VGPR directly
//     (no LDS staging, no per-k-tile barrier); weights do not depend on
//     activations, so the next chunk's fragments — across layer boundaries too —
//     are in flight while the current chunk computes; ONE barrier per layer;
//   * the MFMA computes the TRANSPOSED tile (weights as the A operand, activations
//     as B): a lane then holds 4 CONSECUTIVE output columns of one row, so the
//     epilogue moves 8-byte vectors (LDS store, act' operand load, swish
//     pre-activation store) instead of 2-byte scalars;
//   * RT row tiles per workgroup reuse every weight fragment RT times: RT = 1
//     fills the chip at rollout sizes (M = 1k..8k rows), RT = 4 at training sizes
//     (M = T * minibatch = 30 720) — measured in tools/microbench_trunk.py; a trunk
//     no wider than 64 runs RT = 16 with ONE column tile per wave (NB = 1) there.
//   * the kernel is bound by instruction issue and dependent latency, not by MFMA or
//     HBM (tools/trace_policy.py: per-workgroup phase timeline): loads of a step are
//     unconditional and counted, biases and the input tile share one memory round trip,
//     kernel-argument scalars are read once, copies out ride one layer behind.
// Forward:  v = act(acc + bias); training also copies each layer's output (and the
//           bf16 input) out of LDS in coalesced 16-byte rows for the backward, while
//           the NEXT layer's MFMAs run.
// Backward: the same walk over the transposed problem: dz_{l-1} = (dz_l . W_l^T)
//           (.) act'_{l-1}(y_{l-1}); the y_{l-1} values a lane needs are loaded
//           global -> VGPR before the layer's last MFMAs; every dz_l is copied out
//           (bf16) for the grouped dW launch.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "bf16_common.h"
#include "sampler_math.h"

namespace {

using namespace mippo_bf16;

constexpr int CH_MAXL = 8;
constexpr int IF_KC = 64;  // reduce elements per pipeline step (2 MFMA k-steps)
constexpr int IF_KS = IF_KC / 32;

struct ChainLayer {
  const bf16_t* w;      // streamed operand, FRAGMENT-MAJOR (gemm_bf16.hip: frag_store):
                        // block (ct, ks) = the 64-lane MFMA fragment of column tile ct, k-step ks
  const float* bias;    // forward: [N] or null
  bf16_t* out_bf;       // [M][ldo] bf16 image of this layer's output, or null
  bf16_t* pre_bf;       // forward/swish: pre-activation [M][ldo], or null
  const bf16_t* aux;    // backward: tensor act' is evaluated on, [M][ldo], or null
  int64_t ldo;
  int K, N, act;        // reduce width, output width, activation (fwd) / act' kind (bwd)
};
struct Chain {
  ChainLayer layer[CH_MAXL];
  const float* x;       // [M][K0] fp32 input (backward: gradient of the chain output)
  const float* x_tail;  // rows M_head .. M-1 come from here ([M - M_head][K0]), or null
  int64_t M_head;       // = M without a tail
  const bf16_t* aux0;   // backward: act' operand of the input, or null
  int64_t ldaux0;
  int act0;
  float* out;           // [M][N_last] fp32 or null
  bf16_t* x_bf;         // [M][ldx] bf16 image of the (processed) input, or null
  int64_t ldx;
  int64_t M;
  int L;
  int arow;             // LDS row length in bf16: widest (32-rounded) layer + 8
  int bias_off;         // forward: LDS byte offset of the fp32 bias rows (32-padded, zero filled)
  // copies of layer[l].bias / layer[l].N side by side: the bias staging reads all of them
  // at kernel start, and 8 descriptors 64 bytes apart are 8 scalar-cache lines
  const float* bias_all[CH_MAXL];
  int n_all[CH_MAXL];
};

// Fused policy step (mi_policy_fwd_bf16): blockIdx.y picks one of two trunks that
// read the same input (0 = action port, 1 = value port); the input stage applies the
// running-statistics normaliser and the action trunk ends in the sampler.
struct PolicyExtra {
  const float* norm_mean;   // [K0] or null: no normalisation
  const float* norm_m2;
  const float* norm_count;  // [1]
  float norm_eps;
  mippo_sampler::FwdParams samp;  // forward: samp.A == 0: no sampler
  int ms_off;               // LDS byte offset of the fp32 [ROWS][2A] sampler input
  mippo_sampler::BwdParams sbwd;  // backward: the action trunk's output gradient comes
                                  // from the sampler backward (sbwd.A == 0: from c.x)
};
struct PolicyArgs {
  Chain c[2];
  PolicyExtra px;
};

struct IStep {
  int l, p, kc;  // layer, column pass (256 columns per pass), k-chunk
};

// Kp: the layer's reduce width rounded up to 32, N: its output width, kstep: reduce
// elements per step, pass_cols: output columns of a column pass (64 per fragment slot)
__device__ inline IStep istep_next(int Kp, int N, int kstep, int pass_cols, IStep s) {
  s.kc += kstep;
  if (s.kc >= Kp) {
    s.kc = 0;
    s.p += 1;
    if (s.p * pass_cols >= N) {
      s.p = 0;
      s.l += 1;
    }
  }
  return s;
}

// -DMIPPO_TRACE (tools/trace_policy.py): thread 0 of every workgroup stamps the shader
// clock at each phase boundary; compiled out of the product build.
#ifdef MIPPO_TRACE
constexpr int TR_EV = 64, TR_WG = 2048;
__device__ unsigned long long g_trace[TR_WG * TR_EV];
// stamps go to LDS and are copied out at the end: a global store per stamp would sit in
// the wave's vmcnt queue and add a store round trip to every later wait
#define MI_TR()                                                                      \
  do {                                                                               \
    if (tid == 0 && ev_ < TR_EV - 2) tr_s[ev_] = __builtin_amdgcn_s_memtime();       \
    ++ev_;                                                                           \
  } while (0)
#define MI_TR_BEGIN()                                                                \
  __shared__ unsigned long long tr_s[TR_EV];                                         \
  int ev_ = 0;                                                                       \
  const unsigned long long tr_t0_ = wall_clock64();
#define MI_TR_END()                                                                  \
  do {                                                                               \
    const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                        \
    if (tid == 0 && wg_ < TR_WG) {                                                   \
      for (int i_ = 0; i_ < TR_EV - 2; ++i_)                                         \
        g_trace[wg_ * TR_EV + i_] = i_ < ev_ ? tr_s[i_] : 0ull;                      \
      g_trace[wg_ * TR_EV + TR_EV - 2] = tr_t0_;                                     \
      g_trace[wg_ * TR_EV + TR_EV - 1] = wall_clock64();                             \
    }                                                                                \
  } while (0)
#else
#define MI_TR() do {} while (0)
#define MI_TR_BEGIN() do {} while (0)
#define MI_TR_END() do {} while (0)
#endif

template <int NB>
struct BFrags {
  bf16x8 f[IF_KS][NB];  // [k-step][column tile]
};

// NB = column tiles (fragment slots) per wave and pass.  NB = 4: a pass covers 256 columns
// (the general form).  NB = 1 with RT = 16: a pass covers 64 columns and the workgroup owns
// 256 rows — for trunks no wider than 64, where a wave has ONE column tile anyway and
// three of four slots idled: the same accumulator and operand registers hold 4x the rows,
// so a quarter of the workgroups pay the per-workgroup latency chain.
// INPLACE: ONE activation buffer instead of the ping-pong pair (half the LDS, so more
// workgroups per CU): a layer's outputs overwrite its inputs, which needs every wave to
// have finished READING the inputs first — one more barrier per layer, in front of the
// epilogue — and every layer to be a single column pass (N <= 64 * NB; checked on the host).
template <int RT, bool BWD, bool POLICY, int NB = 4, bool INPLACE = false>
__device__ __forceinline__ void chain_body(const Chain& c, const PolicyExtra* px) {
  constexpr int ROWS = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int arow = c.arow;
  bf16_t* const act0 = reinterpret_cast<bf16_t*>(lds_raw);
  bf16_t* const act1 = INPLACE ? act0 : act0 + ROWS * arow;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // wave-uniform by construction; telling the compiler so keeps the column-tile
  // arithmetic (and with it the weight addresses and the epilogue's tile tests) on
  // the scalar unit — the kernel is bound by VALU instruction issue, not by MFMA
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int64_t i0 = (int64_t)blockIdx.x * ROWS;
  // kernel-argument scalars the loop needs, read ONCE: left as `c.field` the compiler
  // re-reads them from the argument segment at every use (an s_load + lgkmcnt(0) wait,
  // ~90 of them in the loop body)
  const int64_t cM = c.M;
  const int cL = c.L;
  float* const c_out = c.out;
  const float* const bias_s = reinterpret_cast<const float*>(lds_raw + c.bias_off);
  if (i0 >= cM) return;  // a launch covers the longer of two trunks (policy step)
  const int K0 = c.layer[0].K;
  const int K0p = (K0 + 31) / 32 * 32;
  MI_TR_BEGIN();
  MI_TR();

  // column tile b of this wave in pass p starts at column ((p*4 + b)*4 + wave) * 16
  // The 8 loads of a step are UNCONDITIONAL straight-line code (tile / k-step indices
  // past the layer's range are clamped; those fragments are never multiplied): with a
  // branch around each load the compiler cannot count the loads in flight and waits
  // for vmcnt(0) in front of every MFMA group — i.e. for the fragments it has just
  // requested, which turns the prefetch into one exposed memory round trip per step
  // (tools/trace_policy.py: 2 000-2 800 cycles per step whatever the step computes).
  // KS / NT (k-steps and column tiles of the layer) are kept per layer, and the block
  // index is 32-bit unsigned: the step's address arithmetic is a few scalar instructions
  // per load (it was ~130 dependent scalar instructions per step, ~1 000 cycles).
  // A layer with ONE column tile (N <= 16: value heads, the backward's first layer) runs
  // "deep": the 8 fragment slots of a step hold 8 consecutive k-steps of that tile, so
  // K = 256 is one step instead of four (same accumulation order, same sums).
  auto load_frags = [&](const IStep& s, const bf16_t* w, unsigned KS, unsigned NT,
                        BFrags<NB>& B) {
    const char* const wb = reinterpret_cast<const char*>(w) + lane * 16;
    // deep: slot (ks, b) = k-step b*IF_KS + ks
    const unsigned bstride = (!BWD && NB == 4 && NT == 1) ? IF_KS : 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      unsigned ct = ((unsigned)s.p * NB + b) * 4 + wave;
      ct = ct < NT ? ct : NT - 1;
      const unsigned base = ct * KS;
#pragma unroll
      for (int ks = 0; ks < IF_KS; ++ks) {  // one contiguous 1 KiB per wave-instruction
        unsigned kg = ((unsigned)s.kc >> 5) + b * bstride + ks;
        kg = kg < KS ? kg : KS - 1;
        B.f[ks][b] = __builtin_bit_cast(
            bf16x8, *reinterpret_cast<const u32x4*>(wb + ((size_t)(base + kg) << 10)));
      }
    }
  };
  // coalesced copy of a published LDS buffer (rows x ld columns) to global
  // (a power-of-two number of threads per row: shifts instead of divisions)
  auto flush = [&](const bf16_t* buf, bf16_t* dst, int64_t ld) {
    const int nch = (int)(ld / 8);            // 16-byte chunks per row, <= 64
    int sh = 0;
    while ((1 << sh) < nch) ++sh;             // scalar
    const int cc = tid & ((1 << sh) - 1);
    const int rstep = kThreads >> sh;
    if (cc < nch) {
      for (int row = tid >> sh; row < ROWS; row += 4 * rstep) {
        u32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // clamped: in-bounds LDS reads, stores predicated below
          const int ru = row + u * rstep < ROWS ? row + u * rstep : ROWS - 1;
          v[u] = *reinterpret_cast<const u32x4*>(buf + ru * arow + cc * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int ru = row + u * rstep;
          const int64_t gi = i0 + ru;
          if (ru < ROWS && gi < cM) *reinterpret_cast<u32x4*>(dst + gi * ld + cc * 8) = v[u];
        }
      }
    }
  };

  // Layer descriptors live in the kernel-argument segment; indexing them per step
  // costs a scalar-memory round trip in front of every step (PMC: ~100 s_load per
  // wave, WAIT_ANY 53 % of wave cycles).  The current and the next layer's
  // descriptors are kept in registers instead and rolled forward at layer changes,
  // so the load for layer l+2 is issued a whole layer before it is needed.
  ChainLayer Lc = c.layer[0];
  ChainLayer Ln = c.layer[cL > 1 ? 1 : 0];
  IStep s = {0, 0, 0};
  BFrags<NB> B, Bn;
  unsigned KSc = (unsigned)(Lc.K + 31) >> 5, NTc = (unsigned)(Lc.N + 15) >> 4;
  unsigned KSn = (unsigned)(Ln.K + 31) >> 5, NTn = (unsigned)(Ln.N + 15) >> 4;
  load_frags(s, Lc.w, KSc, NTc, B);
  MI_TR();

  // Every XCD's L2 starts a kernel cold and the workgroups of a launch walk the layers
  // in lock-step, so each layer's first fetch would be an L2 miss for all of them at
  // once.  Workgroups 8l .. 8l+7 (one per XCD: consecutive workgroups go to consecutive
  // XCDs) touch one dword per 128-byte line of layer l's image at kernel start; the
  // values are folded into `warm` after the input stage and never used.
  unsigned warm = 0;
  if (blockIdx.x < 8 * (unsigned)cL) {
    const ChainLayer& wl = c.layer[blockIdx.x >> 3];
    const int64_t bytes = (int64_t)((wl.N + 15) / 16) * ((wl.K + 31) / 32) * 1024;
    unsigned t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t off = (int64_t)(tid + u * kThreads) * 128;
      t[u] = off < bytes ? *reinterpret_cast<const unsigned*>(
                               reinterpret_cast<const char*>(wl.w) + off)
                         : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) warm ^= t[u];
  }
  MI_TR();
  // bias rows -> LDS (each padded to a multiple of 32 columns with zeros): the epilogue
  // then has no global load of its own.  Every layer's values are requested here and
  // stored after the input tile below, so both share one memory round trip.
  float bv[CH_MAXL][2];
  int bN[CH_MAXL];
  if constexpr (!BWD) {
    // all CH_MAXL descriptors are read unconditionally (unused ones are zero filled), so
    // the scalar loads of the argument segment go out as one batch with one wait — a
    // branch on `l < L` in front of each serialised them (~500 cycles apiece)
    const float* bl[CH_MAXL];
#pragma unroll
    for (int l = 0; l < CH_MAXL; ++l) {
      bl[l] = c.bias_all[l];
      bN[l] = c.n_all[l];
    }
#pragma unroll
    for (int l = 0; l < CH_MAXL; ++l) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        bv[l][u] = 0.0f;
        if (bl[l] && tid + u * kThreads < bN[l]) bv[l][u] = bl[l][tid + u * kThreads];
      }
    }
  }

  // the sampler's own inputs (counter, replayed actions) are first needed after the last
  // layer: request their lines now so that the tail does not start with a cold miss
  unsigned tw0 = 0, tw1 = 0, tw2 = 0;
  if constexpr (POLICY && !BWD) {
    const mippo_sampler::FwdParams& sp = px->samp;
    if (sp.A > 0 && blockIdx.y == 0 && tid < ROWS && i0 + tid < cM) {
      const int64_t e = (i0 + tid) * sp.A;
      if (sp.noise.rng) tw0 = (unsigned)sp.noise.rng[1];
      if (sp.extras) tw1 = __float_as_uint(sp.extras[e]);
      if (sp.noise.eps2) tw2 = __float_as_uint(sp.noise.eps2[e]);
    }
  }
  MI_TR();
  // stage 0: fp32 input tile (x act'(aux0) in the backward) -> bf16, zero padded
  bool from_sampler = false;
  if constexpr (POLICY && BWD) from_sampler = px->sbwd.A > 0 && blockIdx.y == 0;
  if constexpr (POLICY && BWD) {
    if (from_sampler) {
      // sampling_layers.py:82-147 differentiated: one thread per row writes the 2A
      // gradient columns (K0 == 2A); the others clear the pad columns
      for (int row = tid; row < ROWS; row += kThreads) {
        bf16_t* dst = act0 + row * arow;
        if (i0 + row < cM) {
          mippo_sampler::bwd_row(i0 + row, px->sbwd, [dst](int j, float v) { dst[j] = (bf16_t)v; });
        } else {
          for (int k = 0; k < K0; ++k) dst[k] = (bf16_t)0.0f;
        }
      }
      for (int row = tid >> 5; row < ROWS; row += kThreads >> 5)
        for (int k = K0 + (tid & 31); k < K0p; k += 32) act0[row * arow + k] = (bf16_t)0.0f;
    }
  }
  if (!from_sampler) {
    // The tile's ROWS x K0 real elements are one contiguous run of the input: thread t
    // takes elements t, t + 256, ... and ALL loads of a batch are issued before the first
    // is used (one memory round trip for the tile instead of one per row group).  The
    // row of an element comes from a float reciprocal (exact: e < 2^15, and (e + 1/2) / K0
    // stays 1/(2 K0) away from an integer) — an integer division here is ~40 instructions
    // per element on a kernel that is bound by instruction issue.
    // elements per thread and batch: one batch covers the tile for K0 <= 8
    constexpr int SB = ROWS > 64 ? 8 : 2;
    const int nel = ROWS * K0;
    const float rcpK0 = 1.0f / (float)K0;
    float cnt = 0.0f;
    bool norm = false;
    if constexpr (POLICY && !BWD) {
      norm = px->norm_mean != nullptr;
      if (norm) cnt = *px->norm_count;
    }
    for (int e0 = 0; e0 < nel; e0 += SB * kThreads) {
      float xv[SB], av[SB], mean[SB], m2[SB];
      int rw[SB], kk[SB];
      bool ok[SB];
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        const int e = e0 + u * kThreads + tid;
        rw[u] = (int)(((float)e + 0.5f) * rcpK0);
        kk[u] = e - rw[u] * K0;
        const int64_t gi = i0 + rw[u];
        ok[u] = e < nel && gi < cM;
        xv[u] = 0.0f;
        av[u] = 0.0f;
        mean[u] = 0.0f;
        m2[u] = 1.0f;
        if (ok[u]) {
          xv[u] = gi < c.M_head ? c.x[gi * K0 + kk[u]] : c.x_tail[(gi - c.M_head) * K0 + kk[u]];
          if constexpr (POLICY && !BWD) {
            if (norm) {
              mean[u] = px->norm_mean[kk[u]];
              m2[u] = px->norm_m2[kk[u]];
            }
          }
          if constexpr (BWD) {
            if (c.aux0) av[u] = (float)c.aux0[gi * c.ldaux0 + kk[u]];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        float v = xv[u];
        if constexpr (POLICY && !BWD) {
          // normalizer.py:76-81,92-96 — the same fp32 expression as normalize_fwd_kernel
          if (norm && ok[u]) {
            const float sd = cnt > 0.0f ? sqrtf(fmaxf(m2[u] / cnt, px->norm_eps)) : 10.0f;
            v = (v - mean[u]) / sd;
          }
        }
        if constexpr (BWD) {
          if (c.aux0 && ok[u]) v *= act_grad(av[u], c.act0);
        }
        if (e0 + u * kThreads + tid < nel) act0[rw[u] * arow + kk[u]] = (bf16_t)v;
      }
    }
    // zero the pad columns the first layer reduces over (no division: 8 rows x 32 columns
    // of threads sweep the tile)
    for (int row = tid >> 5; row < ROWS; row += kThreads >> 5)
      for (int k = K0 + (tid & 31); k < K0p; k += 32) act0[row * arow + k] = (bf16_t)0.0f;
  }
  warm ^= tw0 ^ tw1 ^ tw2;
  if constexpr (!BWD) {
    float* const bias_w = reinterpret_cast<float*>(lds_raw + c.bias_off);
    int boff = 0;
#pragma unroll
    for (int l = 0; l < CH_MAXL; ++l) {
      const int Np32 = (bN[l] + 31) / 32 * 32;  // 0 for the unused descriptors
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (tid + u * kThreads < Np32) bias_w[boff + tid + u * kThreads] = bv[l][u];
      boff += Np32;
    }
  }
  MI_TR();
  __syncthreads();
  MI_TR();
  // Copies to global are deferred by one layer: the bf16 image of layer l's output is
  // written out while layer l+1's MFMAs run (its LDS buffer stays intact until layer
  // l+2's epilogue), so the stores have a whole layer to be acknowledged before any
  // later wait of this wave has to pass them (vmcnt counts loads and stores in order).
  const bf16_t* pf_buf = act0;
  bf16_t* pf_dst = c.x_bf;
  int64_t pf_ld = c.ldx;
  MI_TR();

  bool samp_here = false;     // this workgroup's trunk ends in the sampler
  float* ms_base = nullptr;
  if constexpr (POLICY && !BWD) {
    samp_here = px->samp.A > 0 && blockIdx.y == 0;
    ms_base = reinterpret_cast<float*>(lds_raw + px->ms_off);
  }
  int boff_c = 0;     // offset of the current layer's bias row in the LDS bias area
  f32x4 acc[RT][NB];  // acc[r][b][e]: row r*16 + li, column tile b, column 4*lq + e
  s16x4 auxr[RT][NB]; // backward: the act' operands of the same elements
  // Unrolled epilogue of one column pass.  TRANS selects, at compile time, the
  // variant with transcendentals (tanh / swish) so that the common relu / none
  // variant stays a handful of VALU ops per element; the kernel keeps exactly ONE
  // instance of the step body (the two weight-fragment register sets are swapped by
  // register moves) — fully unrolled epilogues in several instances overflowed the
  // instruction cache and ran 5x slower than their MFMA + memory time.
  auto epilogue = [&](const IStep& st, const ChainLayer& ly, auto trans_tag) {
    constexpr bool TRANS = decltype(trans_tag)::value;
    bf16_t* const nbuf = (st.l & 1) ? act0 : act1;
    const bool last = st.l == cL - 1;
    const bool keep = !last || ly.out_bf;
    const bool to_out = last && c_out;
    // action trunk of a policy step: the sampler reads the fp32 output row from LDS
    bool to_ms = false;
    float* ms_s = nullptr;
    if constexpr (POLICY && !BWD) {
      to_ms = last && samp_here;
      ms_s = ms_base;
    }
    const bool store_pre = !BWD && TRANS && ly.pre_bf;
    const bool use_aux = BWD && ly.aux && ly.act != MI_ACT_NONE;
    const int Np = (ly.N + 31) / 32 * 32;  // the next layer reduces over Np columns
    const int relu = ly.act == MI_ACT_RELU;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int ct = (st.p * NB + b) * 4 + wave;  // scalar: the tests on it are uniform
      if (ct * 16 >= Np) continue;
      const int j0 = ct * 16 + 4 * lq;
      const bool full = ct * 16 + 16 <= ly.N;    // no pad column in this tile
      f32x4 bj = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (!BWD)  // staged at kernel start (zeros where the layer has no bias)
        bj = *reinterpret_cast<const f32x4*>(bias_s + boff_c + j0);
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const int row = r * 16 + li;
        f32x4 v4 = acc[r][b];
        bf16x4 vo, zo;
        if constexpr (BWD) {
          if (use_aux) {
            const bf16x4 a4 = __builtin_bit_cast(bf16x4, auxr[r][b]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float a = (float)a4[e];
              if constexpr (TRANS) {  // swish: aux is the pre-activation
                const float sg = fast_sigmoid(a);
                v4[e] *= sg * (1.0f + a * (1.0f - sg));
              } else {
                v4[e] *= relu ? (a > 0.0f ? 1.0f : 0.0f) : 1.0f - a * a;
              }
            }
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float z = v4[e] + bj[e];
            if constexpr (TRANS) {
              const float sg = fast_sigmoid(ly.act == MI_ACT_TANH ? 2.0f * z : z);
              v4[e] = ly.act == MI_ACT_TANH ? 2.0f * sg - 1.0f : z * sg;
              zo[e] = (bf16_t)z;
            } else {
              v4[e] = relu ? fmaxf(z, 0.0f) : z;
            }
          }
        }
        if (!full) {  // pad columns (weights are zero there)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (j0 + e >= ly.N) v4[e] = 0.0f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) vo[e] = (bf16_t)v4[e];
        if (keep) *reinterpret_cast<bf16x4*>(nbuf + row * arow + j0) = vo;
        if constexpr (!BWD && TRANS) {
          const int64_t gi = i0 + row;
          if (store_pre && gi < cM && j0 < ly.ldo)
            *reinterpret_cast<bf16x4*>(ly.pre_bf + gi * ly.ldo + j0) = zo;
        }
        // fp32 results of the chain's last layer (a handful of columns): kept out
        // of the loop above so that the other layers do not pay for its predicates
        if (to_out) {
          const int64_t gi = i0 + row;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gi < cM && j0 + e < ly.N) c_out[gi * ly.N + j0 + e] = v4[e];
        }
        if constexpr (POLICY && !BWD) {
          if (to_ms) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (j0 + e < ly.N) ms_s[row * ly.N + j0 + e] = v4[e];
          }
        }
      }
    }
  };

  // Hidden relu layer whose width is a multiple of 32 — the common case: no pad columns,
  // no fp32 / sampler / pre-activation outputs, so none of the per-(tile, row) uniform
  // tests of the general epilogue (each a compare + taken branch) is needed.  The trunk
  // kernels are bound by instruction issue; this is ~12 instead of ~30 instructions per
  // (tile, row).  Same fp32 expressions as the general path.
  auto epilogue_hidden_relu = [&](const IStep& st, const ChainLayer& ly) {
    bf16_t* const wbase = ((st.l & 1) ? act0 : act1) + li * arow + 4 * lq;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int ct = (st.p * NB + b) * 4 + wave;
      if (ct * 16 >= ly.N) continue;
      f32x4 bj = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (!BWD) bj = *reinterpret_cast<const f32x4*>(bias_s + boff_c + ct * 16 + 4 * lq);
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        f32x4 v4 = acc[r][b];
        bf16x4 vo;
        if constexpr (BWD) {
          const bf16x4 a4 = __builtin_bit_cast(bf16x4, auxr[r][b]);
#pragma unroll
          for (int e = 0; e < 4; ++e) v4[e] *= ((float)a4[e] > 0.0f ? 1.0f : 0.0f);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v4[e] = fmaxf(v4[e] + bj[e], 0.0f);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) vo[e] = (bf16_t)v4[e];
        *reinterpret_cast<bf16x4*>(wbase + r * 16 * arow + ct * 16) = vo;
      }
    }
  };

  // Linear head of a forward chain (no activation, no bf16 image): its fp32 columns go to
  // the chain output and / or the sampler's LDS rows and nothing else — the general
  // epilogue's per-element predicates and pad handling cost ~800 cycles per row tile here.
  auto epilogue_head = [&](const IStep& st, const ChainLayer& ly, bool to_ms) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int ct = (st.p * NB + b) * 4 + wave;
      if (ct * 16 >= ly.N) continue;
      const int j0 = ct * 16 + 4 * lq;
      const f32x4 bj = *reinterpret_cast<const f32x4*>(bias_s + boff_c + j0);
      if (j0 >= ly.N) continue;  // per lane: none of its 4 columns exists
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const int row = r * 16 + li;
        const int64_t gi = i0 + row;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[r][b][e] + bj[e];
          if (j0 + e < ly.N) {
            if (to_ms) ms_base[row * ly.N + j0 + e] = v;
            if (c_out && gi < cM) c_out[gi * ly.N + j0 + e] = v;
          }
        }
      }
    }
  };

  while (s.l < cL) {
    const IStep st = s;
    const ChainLayer& ly = Lc;
    const int Kp = (int)KSc * 32;
    const bf16_t* const cbuf = (st.l & 1) ? act1 : act0;
    const bool deep = !BWD && NB == 4 && NTc == 1;  // (the backward's single-tile layers have K <= 64)
    const int kstep = deep ? 4 * IF_KC : IF_KC;
    const bool pass_done = st.kc + kstep >= Kp;  // this step completes the wave's columns
    const IStep sn = istep_next(Kp, ly.N, kstep, NB * 64, st);
    {
      // next step's fragments; the last step re-reads its own (never used) so that the
      // loads of a step stay unconditional
      const bool same = sn.l == st.l || sn.l >= cL;
      const IStep sl = sn.l < cL ? sn : st;
      load_frags(sl, same ? Lc.w : Ln.w, same ? KSc : KSn, same ? NTc : NTn, Bn);
    }
    MI_TR();
    if constexpr (BWD) {
      if (st.kc == 0 && ly.aux && ly.act != MI_ACT_NONE) {
        // act' operands of this pass: 8 bytes per (row, column tile), requested at the
        // pass's FIRST step — in flight during all its MFMAs (they were requested at the
        // last step and the epilogue of a 4-step layer then waited ~3 000 cycles)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int j0 = ((st.p * NB + b) * 4 + wave) * 16 + 4 * lq;
#pragma unroll
          for (int r = 0; r < RT; ++r) {
            const int64_t gi = i0 + r * 16 + li;
            s16x4 a = s16x4{0, 0, 0, 0};
            if (gi < cM && j0 < ly.ldo)
              a = *reinterpret_cast<const s16x4*>(ly.aux + gi * ly.ldo + j0);
            auxr[r][b] = a;
          }
        }
      }
    }
    if (st.kc == 0) {
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[r][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (deep) {
      if (wave == 0) {  // the layer's only column tile
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int ks = 0; ks < IF_KS; ++ks) {
            const int ko = st.kc + (b * IF_KS + ks) * 32;
            if (ko < Kp) {
#pragma unroll
              for (int r = 0; r < RT; ++r) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(cbuf + (r * 16 + li) * arow +
                                                                  ko + 8 * lq);
                acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B.f[ks][b], a, acc[r][0],
                                                                    0, 0, 0);
              }
            }
            // keep the scheduler from hoisting all 8 x RT operand reads (128 VGPRs at RT = 4)
            __builtin_amdgcn_sched_barrier(0);
          }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < IF_KS; ++ks) {
        if (st.kc + ks * 32 < Kp) {
          bf16x8 af[RT];
#pragma unroll
          for (int r = 0; r < RT; ++r)
            af[r] = *reinterpret_cast<const bf16x8*>(cbuf + (r * 16 + li) * arow + st.kc +
                                                     ks * 32 + 8 * lq);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            if (((st.p * NB + b) * 4 + wave) * 16 < ly.N) {
#pragma unroll
              for (int r = 0; r < RT; ++r)  // transposed tile: weights are the A operand
                acc[r][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B.f[ks][b], af[r],
                                                                    acc[r][b], 0, 0, 0);
            }
          }
        }
      }
    }
    MI_TR();
    if (pf_dst) {  // the previous layer's image, while this step's MFMAs run
      flush(pf_buf, pf_dst, pf_ld);
      pf_dst = nullptr;
    }
    MI_TR();
    if (pass_done) {
      const bool trans = BWD ? ly.act == MI_ACT_SWISH : ly.act >= MI_ACT_TANH;
      // (also the backward's last layer: it only keeps its bf16 image for the dW launch)
      const bool plain_out = st.l != cL - 1 || (BWD && !c_out && ly.out_bf);
      const bool hidden_relu = plain_out && (ly.N & 31) == 0 && ly.act == MI_ACT_RELU &&
                               (!BWD || ly.aux);
      bool head = false;
      if constexpr (!BWD)
        head = st.l == cL - 1 && ly.act == MI_ACT_NONE && !ly.out_bf && !ly.pre_bf;
      if constexpr (INPLACE) __syncthreads();  // every wave has read this layer's inputs
      if (hidden_relu) {
        epilogue_hidden_relu(st, ly);
      } else if (head) {
        epilogue_head(st, ly, samp_here);
      } else if (trans) {
        epilogue(st, ly, std::true_type{});
      } else {
        epilogue(st, ly, std::false_type{});
      }
      MI_TR();
      if (sn.l != st.l) {  // layer finished: publish; the copy out rides on the next step
        __syncthreads();
        MI_TR();
        pf_buf = (st.l & 1) ? act0 : act1;
        pf_dst = ly.out_bf;
        pf_ld = ly.ldo;
      }
    }
    // hand the prefetched fragments to the next step (register moves)
#pragma unroll
    for (int ks = 0; ks < IF_KS; ++ks)
#pragma unroll
      for (int b = 0; b < NB; ++b) B.f[ks][b] = Bn.f[ks][b];
    if (sn.l != st.l) {  // roll the descriptors: the load for layer l+2 starts now
      boff_c += (Lc.N + 31) / 32 * 32;
      Lc = Ln;
      KSc = KSn;
      NTc = NTn;
      if (sn.l + 1 < cL) {
        Ln = c.layer[sn.l + 1];
        KSn = (unsigned)(Ln.K + 31) >> 5;
        NTn = (unsigned)(Ln.N + 15) >> 4;
      }
    }
    s = sn;
  }
  if (pf_dst) flush(pf_buf, pf_dst, pf_ld);
  if (warm == 0x9e3779b9u && cM < 0) act0[tid] = (bf16_t)0.0f;  // keeps the warm-up loads
  if constexpr (POLICY && !BWD) {
    // sampling_layers.py:82-147 on this workgroup's rows (the last layer's barrier
    // has published ms_s); one thread per row, as in sampler_fwd_kernel
    if (samp_here) {
      const float* ms_s = ms_base;
      for (int row = tid; row < ROWS; row += kThreads)
        if (i0 + row < cM) mippo_sampler::fwd_row(ms_s + row * 2 * px->samp.A, i0 + row, px->samp);
    }
  }
  MI_TR();
  MI_TR_END();
}

template <int RT, bool BWD>
// (kThreads, 2): at least two waves per SIMD, i.e. at most 256 VGPRs
__global__ void __launch_bounds__(kThreads, 2)
mlp_chain_kernel(Chain c) {
  chain_body<RT, BWD, false>(c, nullptr);
}

// blockIdx.y = 0: action trunk with (RT_A, NB_A), 1: value trunk with (RT_V, NB_V).  Equal
// shapes share ONE instance of the body (instruction cache).  WPS = waves per SIMD the
// register allocation must leave room for (2: <= 256 VGPRs, 3: <= 168, 4: <= 128).
template <bool BWD, int RT_A, int NB_A, int RT_V, int NB_V, bool INPLACE>
__device__ __forceinline__ void policy_trunk(const PolicyArgs& a) {
  if constexpr (RT_A == RT_V && NB_A == NB_V) {
    chain_body<RT_A, BWD, true, NB_A, INPLACE>(a.c[blockIdx.y], &a.px);
  } else if (blockIdx.y == 0) {
    chain_body<RT_A, BWD, true, NB_A, INPLACE>(a.c[0], &a.px);
  } else {
    chain_body<RT_V, BWD, true, NB_V, INPLACE>(a.c[1], &a.px);
  }
}

template <int RT_A, int NB_A, int RT_V, int NB_V, bool INPLACE = false, int WPS = 2>
__global__ void __launch_bounds__(kThreads, WPS)
policy_kernel(PolicyArgs a) {
  policy_trunk<false, RT_A, NB_A, RT_V, NB_V, INPLACE>(a);
}

template <int RT_A, int NB_A, int RT_V, int NB_V, bool INPLACE = false, int WPS = 2>
__global__ void __launch_bounds__(kThreads, WPS)
policy_bwd_kernel(PolicyArgs a) {
  policy_trunk<true, RT_A, NB_A, RT_V, NB_V, INPLACE>(a);
}

// Tuning aid: MIPPO_POLICY_SHAPE="RT_A,NB_A,RT_V,NB_V,INPLACE,WPS" picks the instantiation
// of the training-size policy kernels (builds with -DMIPPO_POLICY_VARIANTS carry the
// whole menu; tools/microbench_policy.py sweeps it).
struct PolicyShape {
  int rt_a, nb_a, rt_v, nb_v, inplace, wps;
  bool set;
};
const PolicyShape& policy_shape_override() {
  static const PolicyShape ps = [] {
    PolicyShape p = {0, 0, 0, 0, 0, 0, false};
    const char* e = getenv("MIPPO_POLICY_SHAPE");
    if (e && sscanf(e, "%d,%d,%d,%d,%d,%d", &p.rt_a, &p.nb_a, &p.rt_v, &p.nb_v, &p.inplace,
                    &p.wps) == 6)
      p.set = true;
    return p;
  }();
  return ps;
}

// A trunk no wider than 64 has one column tile per wave: at training sizes it runs with
// 256 rows per workgroup (RT = 16, NB = 1).  MIPPO_NARROW_TRUNK=0 disables (tuning aid).
bool narrow_trunk_enabled() {
  static const bool on = [] {
    const char* e = getenv("MIPPO_NARROW_TRUNK");
    return !(e && e[0] == '0');
  }();
  return on;
}

// fp32 words of the LDS bias area of a forward chain (one 32-padded row per layer)
int bias_words(const Chain& c) {
  int n = 0;
  for (int l = 0; l < c.L; ++l) n += (c.layer[l].N + 31) / 32 * 32;
  return n;
}

#ifdef MIPPO_TRACE
constexpr int kLdsMax = 160 * 1024 - 2048;  // room for the static stamp buffers
#else
constexpr int kLdsMax = 160 * 1024;
#endif

template <int RT, bool BWD>
int launch_rt(Chain& c, hipStream_t st) {
  const int act_bytes = 2 * 16 * RT * c.arow * (int)sizeof(bf16_t);
  c.bias_off = act_bytes;
  const size_t lds = (size_t)act_bytes + (BWD ? 0 : (size_t)bias_words(c) * sizeof(float));
  constexpr int kWant = 2 * 16 * RT * (512 + 8) * (int)sizeof(bf16_t) + CH_MAXL * 512 * 4;
  constexpr int kCap = kWant < kLdsMax ? kWant : kLdsMax;
  static const hipError_t attr = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&mlp_chain_kernel<RT, BWD>),
      hipFuncAttributeMaxDynamicSharedMemorySize, kCap);
  MI_REQUIRE(attr == hipSuccess, "mlp_chain: cannot raise the dynamic LDS limit: %s",
             hipGetErrorString(attr));
  MI_REQUIRE(lds <= (size_t)kCap, "mlp_chain: %zu bytes of LDS needed, %d available", lds, kCap);
  hipLaunchKernelGGL((mlp_chain_kernel<RT, BWD>), dim3((unsigned)mippo::ceil_div(c.M, 16 * RT)),
                     dim3(kThreads), lds, st, c);
  return mippo::check_launch(BWD ? "mi_mlp_bwd_dx_bf16" : "mi_mlp_fwd_bf16");
}

template <bool BWD>
int launch_chain(Chain& c, int maxw, hipStream_t st) {
  c.arow = maxw + 8;
  // RT = 1 fills the chip at rollout sizes; larger M reuses each weight fragment
  // across 4 row tiles (2 for trunks wider than 256: LDS).  Measured with
  // tools/microbench_trunk.py; MIPPO_TRUNK_RT=1|2|4 overrides (tuning aid).
  static const int rt_override = [] {
    const char* e = getenv("MIPPO_TRUNK_RT");
    return e ? atoi(e) : 0;
  }();
  const int rt = rt_override ? rt_override : (c.M <= 8192 ? 1 : 4);
  if (rt == 1) return launch_rt<1, BWD>(c, st);
  if (rt == 4 && maxw <= 256) return launch_rt<4, BWD>(c, st);
  return launch_rt<2, BWD>(c, st);
}

}  // namespace

namespace {

// Fills a forward Chain from the C-ABI arrays (shared by mi_mlp_fwd_bf16 and
// mi_policy_fwd_bf16); *maxw receives the widest 32-rounded layer.
int fill_fwd_chain(Chain& c, const char* who, const float* x, int64_t M, int64_t L,
                   const void* const* wt_bf, const float* const* bias, const int64_t* dims,
                   const int64_t* acts, float* out, void* const* y_bf, void* const* pre_bf,
                   void* x_bf, int* maxw_out) {
  MI_REQUIRE(L >= 1 && L <= CH_MAXL, "%s: 1 <= L <= %d", who, CH_MAXL);
  MI_REQUIRE(x && wt_bf && dims && acts, "%s: null pointer", who);
  c = {};
  c.x = x;
  c.x_tail = nullptr;
  c.M_head = M;
  c.out = out;
  c.M = M;
  c.L = (int)L;
  c.x_bf = static_cast<bf16_t*>(x_bf);
  c.ldx = mippo::ceil_div(dims[0], 8) * 8;
  int maxw = 0;
  for (int l = 0; l < L; ++l) {
    const int64_t K = dims[l], N = dims[l + 1];
    MI_REQUIRE(K >= 1 && N >= 1 && K <= 512 && N <= 512, "%s: layer widths must be in [1, 512]",
               who);
    MI_REQUIRE(acts[l] >= MI_ACT_NONE && acts[l] <= MI_ACT_SWISH, "%s: bad act", who);
    MI_REQUIRE(wt_bf[l] && al16(wt_bf[l]), "%s: weights must be 16-byte aligned", who);
    ChainLayer& ly = c.layer[l];
    ly.w = static_cast<const bf16_t*>(wt_bf[l]);
    ly.bias = bias ? bias[l] : nullptr;
    c.bias_all[l] = ly.bias;
    c.n_all[l] = (int)N;
    ly.K = (int)K;
    ly.N = (int)N;
    ly.act = (int)acts[l];
    ly.ldo = mippo::ceil_div(N, 8) * 8;
    ly.out_bf = y_bf ? static_cast<bf16_t*>(y_bf[l]) : nullptr;
    ly.pre_bf = pre_bf ? static_cast<bf16_t*>(pre_bf[l]) : nullptr;
    MI_REQUIRE(al16(ly.out_bf) && al16(ly.pre_bf), "%s: outputs must be 16-byte aligned", who);
    const int w = (int)(mippo::ceil_div(K > N ? K : N, 32) * 32);
    if (w > maxw) maxw = w;
  }
  *maxw_out = maxw;
  return 0;
}

template <int RT_A, int NB_A, int RT_V, int NB_V, bool INPLACE = false, int WPS = 2>
int launch_policy(PolicyArgs& a, int wa, int wc, hipStream_t st) {
  constexpr int ROWS_A = 16 * RT_A, ROWS_V = 16 * RT_V;
  constexpr int NBUF = INPLACE ? 1 : 2;
  if (INPLACE)
    MI_REQUIRE(wa <= 64 * NB_A && wc <= 64 * NB_V,
               "policy_kernel: the in-place form needs single-pass layers (%d / %d columns)",
               64 * NB_A, 64 * NB_V);
  const int act_a = NBUF * ROWS_A * (wa + 8) * (int)sizeof(bf16_t);
  const int act_v = NBUF * ROWS_V * (wc + 8) * (int)sizeof(bf16_t);
  a.px.ms_off = act_a;  // only the action trunk has the sampler's fp32 rows
  const int ms_bytes = (ROWS_A * 2 * a.px.samp.A * (int)sizeof(float) + 15) / 16 * 16;
  a.c[0].bias_off = act_a + ms_bytes;
  a.c[1].bias_off = act_v;
  const size_t lds_a = (size_t)act_a + ms_bytes + (size_t)bias_words(a.c[0]) * sizeof(float);
  const size_t lds_v = (size_t)act_v + (size_t)bias_words(a.c[1]) * sizeof(float);
  const size_t lds = lds_a > lds_v ? lds_a : lds_v;
  constexpr int kRowsMax = ROWS_A > ROWS_V ? ROWS_A : ROWS_V;
  constexpr int kWant = 2 * kRowsMax * (512 + 8) * (int)sizeof(bf16_t) +
                        kRowsMax * 128 * (int)sizeof(float) + CH_MAXL * 512 * 4;
  constexpr int kCap = kWant < kLdsMax ? kWant : kLdsMax;
  static const hipError_t attr = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&policy_kernel<RT_A, NB_A, RT_V, NB_V, INPLACE, WPS>),
      hipFuncAttributeMaxDynamicSharedMemorySize, kCap);
  MI_REQUIRE(attr == hipSuccess, "policy_kernel: cannot raise the dynamic LDS limit: %s",
             hipGetErrorString(attr));
  MI_REQUIRE(lds <= (size_t)kCap, "policy_kernel: %zu bytes of LDS needed, %d available", lds, kCap);
  const int64_t ga = mippo::ceil_div(a.c[0].M, ROWS_A), gv = mippo::ceil_div(a.c[1].M, ROWS_V);
  hipLaunchKernelGGL((policy_kernel<RT_A, NB_A, RT_V, NB_V, INPLACE, WPS>),
                     dim3((unsigned)(ga > gv ? ga : gv), 2), dim3(kThreads), lds, st, a);
  return mippo::check_launch("mi_policy_fwd_bf16");
}

// the menu of training-size instantiations (see policy_shape_override)
#define MI_POLICY_MENU(X)     \
  X(16, 1, 4, 4, false, 2)    \
  X(4, 4, 4, 4, false, 2)
#ifdef MIPPO_POLICY_VARIANTS
#define MI_POLICY_MENU_EXTRA(X) \
  X(16, 1, 4, 4, true, 2)       \
  X(16, 1, 4, 4, true, 3)       \
  X(16, 1, 4, 4, false, 3)      \
  X(8, 1, 2, 4, true, 4)        \
  X(8, 1, 3, 4, true, 3)        \
  X(16, 1, 6, 4, true, 2)       \
  X(12, 1, 6, 4, true, 2)       \
  X(8, 1, 4, 4, true, 3)
#else
#define MI_POLICY_MENU_EXTRA(X)
#endif

}  // namespace

extern "C" int mi_mlp_fwd_bf16(const float* x, int64_t M, int64_t L, const void* const* wt_bf,
                               const float* const* bias, const int64_t* dims,
                               const int64_t* acts, float* out, void* const* y_bf,
                               void* const* pre_bf, void* x_bf, mi_stream_t stream) {
  MI_REQUIRE(M >= 0, "mi_mlp_fwd_bf16: bad M");
  if (M == 0) return 0;
  // (out may be null when the last layer leaves its bf16 image: a caller that only reads that)
  MI_REQUIRE(out || (y_bf && L >= 1 && y_bf[L - 1]), "mi_mlp_fwd_bf16: null pointer");
  Chain c;
  int maxw = 0;
  int rc = fill_fwd_chain(c, "mi_mlp_fwd_bf16", x, M, L, wt_bf, bias, dims, acts, out, y_bf,
                          pre_bf, x_bf, &maxw);
  if (rc) return rc;
  return launch_chain<false>(c, maxw, mippo::as_stream(stream));
}

extern "C" int mi_policy_fwd_bf16(
    const float* obs, int64_t M, const float* norm_mean, const float* norm_m2,
    const float* norm_count, float norm_eps, int64_t La, const void* const* a_w,
    const float* const* a_bias, const int64_t* a_dims, const int64_t* a_acts, int64_t Lc,
    const void* const* c_w, const float* const* c_bias, const int64_t* c_dims,
    const int64_t* c_acts, const float* extras, const uint64_t* rng_state, uint64_t offset_add,
    const float* eps, const float* eps2, float min_std, float std_scale, float entropy_weight,
    int deterministic, float* mean_and_std, float* raw_out, float* action, float* loglik,
    float* reg, float* mu_out, float* sigma_out, float* value, void* const* a_y_bf,
    void* const* a_pre_bf, void* a_x_bf, void* const* c_y_bf, void* const* c_pre_bf,
    void* c_x_bf, const float* value_tail_obs, int64_t M_tail, mi_stream_t stream) {
  MI_REQUIRE(M >= 0 && M_tail >= 0 && (M_tail == 0 || value_tail_obs),
             "mi_policy_fwd_bf16: bad M / tail");
  if (M == 0) return 0;
  MI_REQUIRE(obs && value && a_dims && c_dims, "mi_policy_fwd_bf16: null pointer");
  MI_REQUIRE(a_dims[0] == c_dims[0], "mi_policy_fwd_bf16: both trunks read the same input");
  MI_REQUIRE(!norm_mean || (norm_m2 && norm_count), "mi_policy_fwd_bf16: incomplete normaliser");
  MI_REQUIRE(rng_state || (eps && eps2),
             "mi_policy_fwd_bf16: need rng_state or both injected noises");
  PolicyArgs a;
  int wa = 0, wc = 0;
  int rc = fill_fwd_chain(a.c[0], "mi_policy_fwd_bf16(action)", obs, M, La, a_w, a_bias, a_dims,
                          a_acts, mean_and_std, a_y_bf, a_pre_bf, a_x_bf, &wa);
  if (rc) return rc;
  rc = fill_fwd_chain(a.c[1], "mi_policy_fwd_bf16(value)", obs, M + M_tail, Lc, c_w, c_bias,
                      c_dims, c_acts, value, c_y_bf, c_pre_bf, c_x_bf, &wc);
  if (rc) return rc;
  a.c[1].x_tail = value_tail_obs;
  a.c[1].M_head = M;
  const int64_t A2 = a_dims[La];
  MI_REQUIRE(A2 >= 2 && A2 % 2 == 0 && A2 <= 128,
             "mi_policy_fwd_bf16: the action trunk must end in 2A <= 128 columns");
  a.c[0].arow = wa + 8;
  a.c[1].arow = wc + 8;
  a.px = {};
  a.px.norm_mean = norm_mean;
  a.px.norm_m2 = norm_m2;
  a.px.norm_count = norm_count;
  a.px.norm_eps = norm_eps;
  a.px.samp = {extras, {rng_state, offset_add, eps, eps2}, raw_out, action, mu_out, sigma_out,
               loglik, reg, (int)(A2 / 2), min_std, std_scale, entropy_weight, deterministic};
  const int maxw = wa > wc ? wa : wc;
  hipStream_t st = mippo::as_stream(stream);
  if (M + M_tail <= 8192) return launch_policy<1, 4, 1, 4>(a, wa, wc, st);
  // a trunk wider than 256 walks 32 rows per workgroup (its two activation buffers of 64 rows
  // would not leave room for a second workgroup on the CU), the other keeps its 64
  if (maxw > 256) {
    if (wa > 256 && wc > 256) return launch_policy<2, 4, 2, 4>(a, wa, wc, st);
    if (wc > 256) return launch_policy<4, 4, 2, 4>(a, wa, wc, st);
    return launch_policy<2, 4, 4, 4>(a, wa, wc, st);
  }
  const PolicyShape& ps = policy_shape_override();
  if (ps.set) {
#define MI_X(RA, NA, RV, NV, IP, W)                                                       \
  if (ps.rt_a == RA && ps.nb_a == NA && ps.rt_v == RV && ps.nb_v == NV &&                 \
      (ps.inplace != 0) == IP && ps.wps == W && (NA == 4 || wa <= 64))                    \
    return launch_policy<RA, NA, RV, NV, IP, W>(a, wa, wc, st);
    MI_POLICY_MENU(MI_X)
    MI_POLICY_MENU_EXTRA(MI_X)
#undef MI_X
    MI_REQUIRE(false, "mi_policy_fwd_bf16: MIPPO_POLICY_SHAPE names no built instantiation");
  }
  if (wa <= 64 && narrow_trunk_enabled()) return launch_policy<16, 1, 4, 4>(a, wa, wc, st);
  return launch_policy<4, 4, 4, 4>(a, wa, wc, st);
}

namespace {

// Fills a backward (dX) Chain from the C-ABI arrays (mi_mlp_bwd_dx_bf16,
// mi_policy_bwd_bf16).  Walks the layers backwards: step q handles layer
// l = L-1-q:  dz_{l-1} = dz_l . W_l^T.
int fill_bwd_chain(Chain& c, const char* who, const float* g_out, const void* aux_last,
                   int act_last, int64_t M, int64_t L, const void* const* w_bf,
                   const int64_t* dims, const int64_t* acts, const void* const* aux,
                   void* dz_last, void* const* dz_bf, float* g_in, int* maxw_out) {
  MI_REQUIRE(L >= 1 && L <= CH_MAXL, "%s: 1 <= L <= %d", who, CH_MAXL);
  MI_REQUIRE(w_bf && dims && acts && dz_last, "%s: null pointer", who);
  MI_REQUIRE(act_last >= MI_ACT_NONE && act_last <= MI_ACT_SWISH, "%s: bad act", who);
  MI_REQUIRE(act_last == MI_ACT_NONE || aux_last, "%s: aux_last needed", who);
  const int steps = (int)(g_in ? L : L - 1);
  c = {};
  c.x = g_out;
  c.x_tail = nullptr;
  c.M_head = M;
  c.M = M;
  c.L = steps;
  c.aux0 = act_last == MI_ACT_NONE ? nullptr : static_cast<const bf16_t*>(aux_last);
  c.ldaux0 = mippo::ceil_div(dims[L], 8) * 8;
  c.act0 = act_last;
  c.x_bf = static_cast<bf16_t*>(dz_last);
  c.ldx = mippo::ceil_div(dims[L], 8) * 8;
  c.out = g_in;
  MI_REQUIRE(al16(dz_last) && al16(aux_last), "%s: buffers must be 16-byte aligned", who);
  int maxw = (int)(mippo::ceil_div(dims[L], 32) * 32);
  for (int l = 0; l <= L; ++l)
    MI_REQUIRE(dims[l] >= 1 && dims[l] <= 512, "%s: widths must be in [1, 512]", who);
  MI_REQUIRE(steps >= 1, "%s: nothing to do (L == 1 without input gradient: use mi_cast_pad_bf16)",
             who);
  for (int q = 0; q < steps; ++q) {
    const int l = (int)L - 1 - q;
    const int64_t K = dims[l], N = dims[l + 1];  // layer l maps K -> N; here reduce N, emit K
    MI_REQUIRE(w_bf[l] && al16(w_bf[l]), "%s: weights must be 16-byte aligned", who);
    ChainLayer& ly = c.layer[q];
    ly.w = static_cast<const bf16_t*>(w_bf[l]);
    ly.K = (int)N;
    ly.N = (int)K;
    ly.ldo = mippo::ceil_div(K, 8) * 8;
    if (l > 0) {
      MI_REQUIRE(acts[l - 1] >= MI_ACT_NONE && acts[l - 1] <= MI_ACT_SWISH, "%s: bad act", who);
      ly.act = (int)acts[l - 1];
      ly.aux = (aux && ly.act != MI_ACT_NONE) ? static_cast<const bf16_t*>(aux[l - 1]) : nullptr;
      MI_REQUIRE(ly.act == MI_ACT_NONE || ly.aux, "%s: aux[%d] needed", who, l - 1);
      ly.out_bf = dz_bf ? static_cast<bf16_t*>(dz_bf[l - 1]) : nullptr;
      MI_REQUIRE(ly.out_bf, "%s: dz_bf[%d] needed", who, l - 1);
      MI_REQUIRE(al16(ly.out_bf) && al16(ly.aux), "%s: buffers must be 16-byte aligned", who);
    } else {
      ly.act = MI_ACT_NONE;  // input gradient: no activation upstream
    }
    const int w = (int)(mippo::ceil_div(K > N ? K : N, 32) * 32);
    if (w > maxw) maxw = w;
  }
  *maxw_out = maxw;
  return 0;
}

template <int RT_A, int NB_A, int RT_V, int NB_V, bool INPLACE = false, int WPS = 2>
int launch_policy_bwd(PolicyArgs& a, int wa, int wc, hipStream_t st) {
  constexpr int ROWS_A = 16 * RT_A, ROWS_V = 16 * RT_V;
  constexpr int NBUF = INPLACE ? 1 : 2;
  if (INPLACE)
    MI_REQUIRE(wa <= 64 * NB_A && wc <= 64 * NB_V,
               "policy_bwd_kernel: the in-place form needs single-pass layers (%d / %d columns)",
               64 * NB_A, 64 * NB_V);
  const size_t lds_a = (size_t)NBUF * ROWS_A * (wa + 8) * sizeof(bf16_t);
  const size_t lds_v = (size_t)NBUF * ROWS_V * (wc + 8) * sizeof(bf16_t);
  const size_t lds = lds_a > lds_v ? lds_a : lds_v;
  constexpr int kRowsMax = ROWS_A > ROWS_V ? ROWS_A : ROWS_V;
  constexpr int kWant = 2 * kRowsMax * (512 + 8) * (int)sizeof(bf16_t);
  constexpr int kCap = kWant < kLdsMax ? kWant : kLdsMax;
  static const hipError_t attr = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&policy_bwd_kernel<RT_A, NB_A, RT_V, NB_V, INPLACE, WPS>),
      hipFuncAttributeMaxDynamicSharedMemorySize, kCap);
  MI_REQUIRE(attr == hipSuccess, "policy_bwd_kernel: cannot raise the dynamic LDS limit: %s",
             hipGetErrorString(attr));
  MI_REQUIRE(lds <= (size_t)kCap, "policy_bwd_kernel: %zu bytes of LDS needed, %d available", lds,
             kCap);
  const int64_t ga = mippo::ceil_div(a.c[0].M, ROWS_A), gv = mippo::ceil_div(a.c[1].M, ROWS_V);
  hipLaunchKernelGGL((policy_bwd_kernel<RT_A, NB_A, RT_V, NB_V, INPLACE, WPS>),
                     dim3((unsigned)(ga > gv ? ga : gv), 2), dim3(kThreads), lds, st, a);
  return mippo::check_launch("mi_policy_bwd_bf16");
}

}  // namespace

extern "C" int mi_mlp_bwd_dx_bf16(const float* g_out, const void* aux_last, int act_last,
                                  int64_t M, int64_t L, const void* const* w_bf,
                                  const int64_t* dims, const int64_t* acts,
                                  const void* const* aux, void* dz_last, void* const* dz_bf,
                                  float* g_in, mi_stream_t stream) {
  MI_REQUIRE(M >= 0, "mi_mlp_bwd_dx_bf16: bad M");
  if (M == 0) return 0;
  MI_REQUIRE(g_out, "mi_mlp_bwd_dx_bf16: null pointer");
  Chain c;
  int maxw = 0;
  int rc = fill_bwd_chain(c, "mi_mlp_bwd_dx_bf16", g_out, aux_last, act_last, M, L, w_bf, dims,
                          acts, aux, dz_last, dz_bf, g_in, &maxw);
  if (rc) return rc;
  return launch_chain<true>(c, maxw, mippo::as_stream(stream));
}

extern "C" int mi_policy_bwd_bf16(
    const float* mean_and_std, const float* extras, const uint64_t* rng_state,
    uint64_t offset_add, const float* eps2, const float* g_loglik, float g_reg, float min_std,
    float std_scale, float entropy_weight, const float* g_value, int64_t M, int64_t La,
    const void* const* a_w, const int64_t* a_dims, const int64_t* a_acts,
    const void* const* a_aux, void* a_dz_last, void* const* a_dz_bf, int64_t Lc,
    const void* const* c_w, const int64_t* c_dims, const int64_t* c_acts,
    const void* const* c_aux, void* c_dz_last, void* const* c_dz_bf, mi_stream_t stream) {
  MI_REQUIRE(M >= 0, "mi_policy_bwd_bf16: bad M");
  if (M == 0) return 0;
  MI_REQUIRE(mean_and_std && extras && g_value && a_dims && c_dims && a_acts && c_acts,
             "mi_policy_bwd_bf16: null pointer");
  MI_REQUIRE(rng_state || eps2, "mi_policy_bwd_bf16: need rng_state or injected eps2");
  MI_REQUIRE(La >= 2 && Lc >= 2, "mi_policy_bwd_bf16: trunks of at least two layers");
  MI_REQUIRE(a_acts[La - 1] == MI_ACT_NONE && c_acts[Lc - 1] == MI_ACT_NONE,
             "mi_policy_bwd_bf16: the trunks' last layers must be linear");
  const int64_t A2 = a_dims[La];
  MI_REQUIRE(A2 >= 2 && A2 % 2 == 0 && A2 <= 128,
             "mi_policy_bwd_bf16: the action trunk must end in 2A <= 128 columns");
  PolicyArgs a;
  int wa = 0, wc = 0;
  // the action trunk's output gradient is produced in the kernel: c.x is unused there
  int rc = fill_bwd_chain(a.c[0], "mi_policy_bwd_bf16(action)", mean_and_std, nullptr,
                          MI_ACT_NONE, M, La, a_w, a_dims, a_acts, a_aux, a_dz_last, a_dz_bf,
                          nullptr, &wa);
  if (rc) return rc;
  rc = fill_bwd_chain(a.c[1], "mi_policy_bwd_bf16(value)", g_value, nullptr, MI_ACT_NONE, M, Lc,
                      c_w, c_dims, c_acts, c_aux, c_dz_last, c_dz_bf, nullptr, &wc);
  if (rc) return rc;
  a.c[0].arow = wa + 8;
  a.c[1].arow = wc + 8;
  a.px = {};
  a.px.sbwd = {mean_and_std, extras, {rng_state, offset_add, eps2, eps2}, g_loglik, g_reg,
               (int)(A2 / 2), min_std, std_scale, entropy_weight};
  const int maxw = wa > wc ? wa : wc;
  hipStream_t st = mippo::as_stream(stream);
  if (M <= 8192) return launch_policy_bwd<1, 4, 1, 4>(a, wa, wc, st);
  if (maxw > 256) {  // (as the forward: 32 rows per workgroup for a trunk wider than 256)
    if (wa > 256 && wc > 256) return launch_policy_bwd<2, 4, 2, 4>(a, wa, wc, st);
    if (wc > 256) return launch_policy_bwd<4, 4, 2, 4>(a, wa, wc, st);
    return launch_policy_bwd<2, 4, 4, 4>(a, wa, wc, st);
  }
  const PolicyShape& ps = policy_shape_override();
  if (ps.set) {
#define MI_X(RA, NA, RV, NV, IP, W)                                                       \
  if (ps.rt_a == RA && ps.nb_a == NA && ps.rt_v == RV && ps.nb_v == NV &&                 \
      (ps.inplace != 0) == IP && ps.wps == W && (NA == 4 || wa <= 64))                    \
    return launch_policy_bwd<RA, NA, RV, NV, IP, W>(a, wa, wc, st);
    MI_POLICY_MENU(MI_X)
    MI_POLICY_MENU_EXTRA(MI_X)
#undef MI_X
    MI_REQUIRE(false, "mi_policy_bwd_bf16: MIPPO_POLICY_SHAPE names no built instantiation");
  }
  if (wa <= 64 && narrow_trunk_enabled()) return launch_policy_bwd<16, 1, 4, 4>(a, wa, wc, st);
  return launch_policy_bwd<4, 4, 4, 4>(a, wa, wc, st);
}

#ifdef MIPPO_TRACE
extern "C" int mi_debug_trace(unsigned long long* host_out, int64_t n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_trace), (size_t)n * 8);
}
extern "C" int mi_debug_trace_clear() {
  void* p = nullptr;
  hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_trace));
  return e != hipSuccess ? (int)e : (int)hipMemset(p, 0, sizeof(g_trace));
}
#endif
