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
//     (M = T * minibatch = 30 720) — measured in tools/microbench_trunk.py.
// Forward:  v = act(acc + bias); training also copies each layer's output (and the
//           bf16 input) out of LDS in coalesced 16-byte rows for the backward.
// Backward: the same walk over the transposed problem: dz_{l-1} = (dz_l . W_l^T)
//           (.) act'_{l-1}(y_{l-1}); the y_{l-1} values a lane needs are loaded
//           global -> VGPR before the layer's last MFMAs; every dz_l is copied out
//           (bf16) for the grouped dW launch.
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

// `ly` = descriptor of layer s.l
__device__ inline IStep istep_next(const ChainLayer& ly, IStep s) {
  const int Kp = (ly.K + 31) / 32 * 32;
  s.kc += IF_KC;
  if (s.kc >= Kp) {
    s.kc = 0;
    s.p += 1;
    if (s.p * 256 >= ly.N) {
      s.p = 0;
      s.l += 1;
    }
  }
  return s;
}

struct BFrags {
  bf16x8 f[IF_KS][4];  // [k-step][column tile]
};

template <int RT, bool BWD, bool POLICY>
__device__ __forceinline__ void chain_body(const Chain& c, const PolicyExtra* px) {
  constexpr int ROWS = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int arow = c.arow;
  bf16_t* const act0 = reinterpret_cast<bf16_t*>(lds_raw);
  bf16_t* const act1 = act0 + ROWS * arow;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // wave-uniform by construction; telling the compiler so keeps the column-tile
  // arithmetic (and with it the weight addresses and the epilogue's tile tests) on
  // the scalar unit — the kernel is bound by VALU instruction issue, not by MFMA
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int64_t i0 = (int64_t)blockIdx.x * ROWS;
  if (i0 >= c.M) return;  // a launch covers the longer of two trunks (policy step)
  const int K0 = c.layer[0].K;
  const int K0p = (K0 + 31) / 32 * 32;

  // column tile b of this wave in pass p starts at column ((p*4 + b)*4 + wave) * 16
  auto load_frags = [&](const IStep& s, const ChainLayer& ly, BFrags& B) {
    const int KS = (ly.K + 31) / 32;        // k-steps of this layer
    const int NT = (ly.N + 15) / 16;        // column tiles of this layer
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int ct = (s.p * 4 + b) * 4 + wave;
#pragma unroll
      for (int ks = 0; ks < IF_KS; ++ks) {
        const int kg = s.kc / 32 + ks;
        u32x4 r = u32x4{0u, 0u, 0u, 0u};
        // one contiguous 1 KiB per wave-instruction (the image is zero padded)
        if (ct < NT && kg < KS)
          r = *reinterpret_cast<const u32x4*>(ly.w + (int64_t)(ct * KS + kg) * 512 + lane * 8);
        B.f[ks][b] = __builtin_bit_cast(bf16x8, r);
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
    if (cc < nch) {
      for (int row = tid >> sh; row < ROWS; row += kThreads >> sh) {
        const int64_t gi = i0 + row;
        if (gi < c.M)
          *reinterpret_cast<u32x4*>(dst + gi * ld + cc * 8) =
              *reinterpret_cast<const u32x4*>(buf + row * arow + cc * 8);
      }
    }
  };

  // Layer descriptors live in the kernel-argument segment; indexing them per step
  // costs a scalar-memory round trip in front of every step (PMC: ~100 s_load per
  // wave, WAIT_ANY 53 % of wave cycles).  The current and the next layer's
  // descriptors are kept in registers instead and rolled forward at layer changes,
  // so the load for layer l+2 is issued a whole layer before it is needed.
  ChainLayer Lc = c.layer[0];
  ChainLayer Ln = c.layer[c.L > 1 ? 1 : 0];
  IStep s = {0, 0, 0};
  BFrags B, Bn;
  load_frags(s, Lc, B);

  // stage 0: fp32 input tile (x act'(aux0) in the backward) -> bf16, zero padded
  bool from_sampler = false;
  if constexpr (POLICY && BWD) from_sampler = px->sbwd.A > 0 && blockIdx.y == 0;
  if constexpr (POLICY && BWD) {
    if (from_sampler) {
      // sampling_layers.py:82-147 differentiated: one thread per row writes the 2A
      // gradient columns (K0 == 2A); the others clear the pad columns
      for (int row = tid; row < ROWS; row += kThreads) {
        bf16_t* dst = act0 + row * arow;
        if (i0 + row < c.M) {
          mippo_sampler::bwd_row(i0 + row, px->sbwd, [dst](int j, float v) { dst[j] = (bf16_t)v; });
        } else {
          for (int k = 0; k < K0; ++k) dst[k] = (bf16_t)0.0f;
        }
      }
      for (int i = tid; i < ROWS * (K0p - K0); i += kThreads)
        act0[(i / (K0p - K0)) * arow + K0 + i % (K0p - K0)] = (bf16_t)0.0f;
    }
  }
  for (int row = tid >> 5; row < (from_sampler ? 0 : ROWS); row += kThreads >> 5)
  for (int k = tid & 31; k < K0p; k += 32) {
    const int64_t gi = i0 + row;
    float v = 0.0f;
    if (gi < c.M && k < K0)
      v = gi < c.M_head ? c.x[gi * K0 + k] : c.x_tail[(gi - c.M_head) * K0 + k];
    if constexpr (POLICY && !BWD) {
      // normalizer.py:76-81,92-96 — the same fp32 expression as normalize_fwd_kernel
      if (px->norm_mean && gi < c.M && k < K0) {
        const float cnt = *px->norm_count;
        const float sd = cnt > 0.0f ? sqrtf(fmaxf(px->norm_m2[k] / cnt, px->norm_eps)) : 10.0f;
        v = (v - px->norm_mean[k]) / sd;
      }
    }
    if (BWD && c.aux0 && gi < c.M && k < K0)
      v *= act_grad((float)c.aux0[gi * c.ldaux0 + k], c.act0);
    act0[row * arow + k] = (bf16_t)v;
  }
  __syncthreads();
  if (c.x_bf) flush(act0, c.x_bf, c.ldx);

  f32x4 acc[RT][4];   // acc[r][b][e]: row r*16 + li, column tile b, column 4*lq + e
  s16x4 auxr[RT][4];  // backward: the act' operands of the same elements
  // Unrolled epilogue of one column pass.  TRANS selects, at compile time, the
  // variant with transcendentals (tanh / swish) so that the common relu / none
  // variant stays a handful of VALU ops per element; the kernel keeps exactly ONE
  // instance of the step body (the two weight-fragment register sets are swapped by
  // register moves) — fully unrolled epilogues in several instances overflowed the
  // instruction cache and ran 5x slower than their MFMA + memory time.
  auto epilogue = [&](const IStep& st, const ChainLayer& ly, auto trans_tag) {
    constexpr bool TRANS = decltype(trans_tag)::value;
    bf16_t* const nbuf = (st.l & 1) ? act0 : act1;
    const bool last = st.l == c.L - 1;
    const bool keep = !last || ly.out_bf;
    const bool to_out = last && c.out;
    // action trunk of a policy step: the sampler reads the fp32 output row from LDS
    bool to_ms = false;
    float* ms_s = nullptr;
    if constexpr (POLICY && !BWD) {
      to_ms = last && px->samp.A > 0 && blockIdx.y == 0;
      ms_s = reinterpret_cast<float*>(lds_raw + px->ms_off);
    }
    const bool store_pre = !BWD && TRANS && ly.pre_bf;
    const bool use_aux = BWD && ly.aux && ly.act != MI_ACT_NONE;
    const int Np = (ly.N + 31) / 32 * 32;  // the next layer reduces over Np columns
    const int relu = ly.act == MI_ACT_RELU;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int ct = (st.p * 4 + b) * 4 + wave;  // scalar: the tests on it are uniform
      if (ct * 16 >= Np) continue;
      const int j0 = ct * 16 + 4 * lq;
      const bool full = ct * 16 + 16 <= ly.N;    // no pad column in this tile
      f32x4 bj = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!BWD && ly.bias) {
        if (full) {
          bj = *reinterpret_cast<const f32x4*>(ly.bias + j0);  // arena rows are 256-B aligned
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) bj[e] = j0 + e < ly.N ? ly.bias[j0 + e] : 0.0f;
        }
      }
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
          if (store_pre && gi < c.M && j0 < ly.ldo)
            *reinterpret_cast<bf16x4*>(ly.pre_bf + gi * ly.ldo + j0) = zo;
        }
        // fp32 results of the chain's last layer (a handful of columns): kept out
        // of the loop above so that the other layers do not pay for its predicates
        if (to_out) {
          const int64_t gi = i0 + row;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gi < c.M && j0 + e < ly.N) c.out[gi * ly.N + j0 + e] = v4[e];
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

  while (s.l < c.L) {
    const IStep st = s;
    const ChainLayer& ly = Lc;
    const int Kp = (ly.K + 31) / 32 * 32;
    const bf16_t* const cbuf = (st.l & 1) ? act1 : act0;
    const bool pass_done = st.kc + IF_KC >= Kp;  // this step completes the wave's columns
    const IStep sn = istep_next(ly, st);
    if (sn.l < c.L) {
      if (sn.l == st.l) load_frags(sn, Lc, Bn);
      else load_frags(sn, Ln, Bn);
    }
    if constexpr (BWD) {
      if (pass_done && ly.aux && ly.act != MI_ACT_NONE) {
        // act' operands of this pass: 8 bytes per (row, column tile), in flight
        // during the MFMAs below
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int j0 = ((st.p * 4 + b) * 4 + wave) * 16 + 4 * lq;
#pragma unroll
          for (int r = 0; r < RT; ++r) {
            const int64_t gi = i0 + r * 16 + li;
            s16x4 a = s16x4{0, 0, 0, 0};
            if (gi < c.M && j0 < ly.ldo)
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
        for (int b = 0; b < 4; ++b) acc[r][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < IF_KS; ++ks) {
      if (st.kc + ks * 32 < Kp) {
        bf16x8 af[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r)
          af[r] = *reinterpret_cast<const bf16x8*>(cbuf + (r * 16 + li) * arow + st.kc +
                                                   ks * 32 + 8 * lq);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (((st.p * 4 + b) * 4 + wave) * 16 < ly.N) {
#pragma unroll
            for (int r = 0; r < RT; ++r)  // transposed tile: weights are the A operand
              acc[r][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B.f[ks][b], af[r], acc[r][b],
                                                                  0, 0, 0);
          }
        }
      }
    }
    if (pass_done) {
      const bool trans = BWD ? ly.act == MI_ACT_SWISH : ly.act >= MI_ACT_TANH;
      if (trans) {
        epilogue(st, ly, std::true_type{});
      } else {
        epilogue(st, ly, std::false_type{});
      }
      if (sn.l != st.l) {  // layer finished: publish, copy out
        __syncthreads();
        if (ly.out_bf) flush((st.l & 1) ? act0 : act1, ly.out_bf, ly.ldo);
      }
    }
    // hand the prefetched fragments to the next step (register moves)
#pragma unroll
    for (int ks = 0; ks < IF_KS; ++ks)
#pragma unroll
      for (int b = 0; b < 4; ++b) B.f[ks][b] = Bn.f[ks][b];
    if (sn.l != st.l) {  // roll the descriptors: the load for layer l+2 starts now
      Lc = Ln;
      if (sn.l + 1 < c.L) Ln = c.layer[sn.l + 1];
    }
    s = sn;
  }
  if constexpr (POLICY && !BWD) {
    // sampling_layers.py:82-147 on this workgroup's rows (the last layer's barrier
    // has published ms_s); one thread per row, as in sampler_fwd_kernel
    if (px->samp.A > 0 && blockIdx.y == 0) {
      const float* ms_s = reinterpret_cast<const float*>(lds_raw + px->ms_off);
      for (int row = tid; row < ROWS; row += kThreads)
        if (i0 + row < c.M) mippo_sampler::fwd_row(ms_s + row * 2 * px->samp.A, i0 + row, px->samp);
    }
  }
}

template <int RT, bool BWD>
__global__ void __launch_bounds__(kThreads)
mlp_chain_kernel(Chain c) {
  chain_body<RT, BWD, false>(c, nullptr);
}

template <int RT>
__global__ void __launch_bounds__(kThreads)
policy_kernel(PolicyArgs a) {
  chain_body<RT, false, true>(a.c[blockIdx.y], &a.px);
}

template <int RT>
__global__ void __launch_bounds__(kThreads)
policy_bwd_kernel(PolicyArgs a) {
  chain_body<RT, true, true>(a.c[blockIdx.y], &a.px);
}

template <int RT, bool BWD>
int launch_rt(const Chain& c, hipStream_t st) {
  const size_t lds = (size_t)2 * 16 * RT * c.arow * sizeof(bf16_t);
  static const hipError_t attr = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&mlp_chain_kernel<RT, BWD>),
      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 16 * RT * (512 + 8) * (int)sizeof(bf16_t));
  MI_REQUIRE(attr == hipSuccess, "mlp_chain: cannot raise the dynamic LDS limit: %s",
             hipGetErrorString(attr));
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

template <int RT>
int launch_policy(PolicyArgs& a, int maxw, hipStream_t st) {
  constexpr int ROWS = 16 * RT;
  const int act_bytes = 2 * ROWS * (maxw + 8) * (int)sizeof(bf16_t);
  a.px.ms_off = act_bytes;
  const size_t lds = (size_t)act_bytes + (size_t)ROWS * 2 * a.px.samp.A * sizeof(float);
  constexpr int kWant = 2 * ROWS * (512 + 8) * (int)sizeof(bf16_t) + ROWS * 128 * (int)sizeof(float);
  constexpr int kCap = kWant < 160 * 1024 ? kWant : 160 * 1024;
  static const hipError_t attr = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&policy_kernel<RT>),
      hipFuncAttributeMaxDynamicSharedMemorySize, kCap);
  MI_REQUIRE(attr == hipSuccess, "policy_kernel: cannot raise the dynamic LDS limit: %s",
             hipGetErrorString(attr));
  MI_REQUIRE(lds <= (size_t)kCap, "policy_kernel: %zu bytes of LDS needed, %d available", lds, kCap);
  const int64_t m_max = a.c[0].M > a.c[1].M ? a.c[0].M : a.c[1].M;
  hipLaunchKernelGGL((policy_kernel<RT>), dim3((unsigned)mippo::ceil_div(m_max, ROWS), 2),
                     dim3(kThreads), lds, st, a);
  return mippo::check_launch("mi_policy_fwd_bf16");
}

}  // namespace

extern "C" int mi_mlp_fwd_bf16(const float* x, int64_t M, int64_t L, const void* const* wt_bf,
                               const float* const* bias, const int64_t* dims,
                               const int64_t* acts, float* out, void* const* y_bf,
                               void* const* pre_bf, void* x_bf, mi_stream_t stream) {
  MI_REQUIRE(M >= 0, "mi_mlp_fwd_bf16: bad M");
  if (M == 0) return 0;
  MI_REQUIRE(out, "mi_mlp_fwd_bf16: null pointer");
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
  if (M + M_tail <= 8192) return launch_policy<1>(a, maxw, st);
  MI_REQUIRE(maxw <= 256, "mi_policy_fwd_bf16: trunks wider than 256 take at most 8192 rows "
                          "(use mi_mlp_fwd_bf16 per trunk)");
  return launch_policy<4>(a, maxw, st);
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

template <int RT>
int launch_policy_bwd(PolicyArgs& a, int maxw, hipStream_t st) {
  constexpr int ROWS = 16 * RT;
  const size_t lds = (size_t)2 * ROWS * (maxw + 8) * sizeof(bf16_t);
  static const hipError_t attr = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&policy_bwd_kernel<RT>),
      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ROWS * (512 + 8) * (int)sizeof(bf16_t));
  MI_REQUIRE(attr == hipSuccess, "policy_bwd_kernel: cannot raise the dynamic LDS limit: %s",
             hipGetErrorString(attr));
  hipLaunchKernelGGL((policy_bwd_kernel<RT>), dim3((unsigned)mippo::ceil_div(a.c[0].M, ROWS), 2),
                     dim3(kThreads), lds, st, a);
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
  if (M <= 8192) return launch_policy_bwd<1>(a, maxw, st);
  MI_REQUIRE(maxw <= 256, "mi_policy_bwd_bf16: trunks wider than 256 take at most 8192 rows "
                          "(use mi_mlp_bwd_dx_bf16 per trunk)");
  return launch_policy_bwd<4>(a, maxw, st);
}
