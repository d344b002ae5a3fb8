// a9 (bf16 path, training sizes) — weights-STATIONARY whole-trunk kernels.
//
// mlp_bf16.hip gives every 64-row tile its own workgroup: each one fetches the trunk's
// weight fragments again (134 KB from L2 for the 5-256-256-1 critic), stages biases,
// warms L2, copies every layer's image out of LDS in a separate pass, and lives ~37 000
// cycles for ~4 000 cycles of MFMA (profiles/r01_trace_policy_v36.txt).  At training
// sizes (M = T x minibatch = 30 720 rows) there are ~2 tiles per CU, so all of that is
// paid twice per CU per launch.
//
// Here ONE 8-wave workgroup per CU keeps the whole trunk in REGISTERS — every wave holds
// the fragments of its own output-column tiles of every layer (the 256 x 256 layer is 64
// VGPRs per lane when split over 8 waves; two waves per SIMD, <= 256 registers each, so
// one wave's epilogue / copy-out overlaps its partner's MFMAs) — and walks row tiles in a
// loop: the input tile of the NEXT row tile is requested before the
// current one is computed, activations ping-pong through LDS as before (a wave needs every
// column of the previous layer), each layer's bf16 image is copied out of its LDS buffer in
// whole 16-byte-per-lane rows while the next layer multiplies, biases sit in registers.
// Per row tile nothing is loaded but the tile's own input.
//
// Shape class (checked on the host, everything else keeps the mlp_bf16.hip kernels):
//   K0 <= 32 inputs -> H -> (H -> H) x NH -> N_out <= 16, H in {64, 128, 256},
//   hidden activation relu, linear head, fragments fit the register file.
// That covers make_mlp_actor_critic's trunks with equal hidden sizes (BASELINE C2:
// actor 5-64-64-64-64-2 = <64, 3>, critic 5-256-256-1 = <256, 1>).
//
// Arithmetic is mlp_bf16.hip's, instruction for instruction where it matters: the same
// transposed 16x16x32 bf16 MFMA tiles accumulated over k-steps in the same order, the
// same bias + relu + round-to-bf16 epilogue — results are bit-identical to
// mi_mlp_fwd_bf16 (tests/test_trunk_ws_gpu.py).
#include <stdlib.h>

#include "bf16_common.h"
#include "sampler_math.h"

#include "comm_common.h"
#include "trunk_ws_fwd.h"

namespace {

// ---- one time step of the recurrent actor of make_gru_actor_critic ------------------------
// (nnx_ppo_amd/networks/factories.py: Dense(obs -> H, relu) -> GRU(H -> H) -> Dense(H -> 2A)
// -> sampler; the recurrent contract of networks/recurrent.py:89-161, cell of gru_mfma.hip.)
// At a rollout / evaluation step the "sequence" is one step long, so the whole actor is
// row-local like an MLP: normaliser, Dense, the GRU's input projection AND its recurrent
// product, the gate arithmetic, the output Dense and the sampler — seven launches of the
// generic containers — run here on one 16*RT-row tile with every weight in registers:
//   chain layer 0: obs -> H (relu); layer 1: the GRU's input projection H -> 3H (columns
//   r | z | n, bias b_i); layer 2: H -> N_out (the head);  `gx`: the recurrent kernel (fp32,
//   converted once per workgroup exactly as gru_fwd_mfma_kernel converts it), b_hn, the
//   carry in and out.
// Wave (wc, wr) owns unit tile wc: the r, z and n columns of its 16 units, so the gates of a
// (row, unit) meet in one lane and the fp32 carry never leaves registers — gru_mfma.hip's
// arrangement in the transposed MFMA form of this file.  Same operand roundings, same
// k-order, same expressions as the launches it replaces: bit-identical
// (tests/test_gru_policy_gpu.py).
struct GruStepExtra {
  const float* w_h;   // [H][3H]
  const float* b_hn;  // [H]
  const float* h_in;  // [M][H]
  float* h_out;       // [M][H]
};

template <int H, int RT>
__device__ __forceinline__ void ws_gru_step_body(const WsChain& c, const GruStepExtra& gx,
                                                 const int bid, const int nblk,
                                                 unsigned char* smem) {
#pragma clang fp contract(off)  // the gate expressions as written, as gru_fwd_mfma_kernel
  static_assert(H == 64 || H == 128, "GRU width: 64 or 128");
  using G = WsGeom<H>;
  constexpr int CW = G::CW, RW = G::RW;
  static_assert(G::TPW == 1 && RT % RW == 0, "one unit tile per wave");
  constexpr int RTW = RT / RW;
  constexpr int ROWS = 16 * RT;
  constexpr int KSH = H / 32;
  constexpr int UT = H / 16;
  constexpr int AROW = H + 8;
  constexpr int XROW = 32 + 8;
  using Lds = WsFwdLds<H, RT, true>;
  bf16_t* const bufX = reinterpret_cast<bf16_t*>(smem + Lds::bufX);
  bf16_t* const bufA = reinterpret_cast<bf16_t*>(smem + Lds::bufA);  // h, then h'
  bf16_t* const bufB = reinterpret_cast<bf16_t*>(smem + Lds::bufB);  // layer 0's output
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % CW, wr = wave / CW;
  const int li = lane & 15, lq = lane >> 4;
  const int64_t M = c.M;
  const int K0 = c.K0, N_out = c.N_out;
  const int64_t ntiles = (M + ROWS - 1) / ROWS;

  // ---- per-tile global operands: the observation tile, the carry tile (as the recurrent
  // product's operand) and this lane's own carry elements (for the blend) -----------------
  constexpr int IN_PT = (ROWS * 32 + kWsThreads - 1) / kWsThreads;
  constexpr int H_CH = ROWS * (H / 4);                               // 16-byte chunks of h
  constexpr int H_PT = (H_CH + kWsThreads - 1) / kWsThreads;
  const int nel = ROWS * K0;
  const float rcpK0 = 1.0f / (float)K0;
  float xin[IN_PT];
  f32x4 hin[H_PT];
  f32x4 hp[RTW];
  auto request_tile = [&](int64_t tile) {
    const int64_t i0 = tile * ROWS;
#pragma unroll
    for (int u = 0; u < IN_PT; ++u) {
      const int e = tid + u * kWsThreads;
      const int row = (int)(((float)e + 0.5f) * rcpK0);
      const int64_t gi = i0 + row;
      xin[u] = 0.0f;
      if (e < nel && gi < M) xin[u] = c.x[gi * K0 + (e - row * K0)];
    }
#pragma unroll
    for (int u = 0; u < H_PT; ++u) {
      const int ch = tid + u * kWsThreads;
      const int row = ch / (H / 4), q = ch % (H / 4);
      const int64_t gi = i0 + row;
      hin[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ch < H_CH && gi < M)
        hin[u] = *reinterpret_cast<const f32x4*>(gx.h_in + gi * H + 4 * q);
    }
#pragma unroll
    for (int r = 0; r < RTW; ++r) {
      const int64_t gi = i0 + (wr * RTW + r) * 16 + li;
      hp[r] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (gi < M) hp[r] = *reinterpret_cast<const f32x4*>(gx.h_in + gi * H + wc * 16 + 4 * lq);
    }
  };
  float* const s_mean = reinterpret_cast<float*>(smem + Lds::mean);
  float* const s_sd = reinterpret_cast<float*>(smem + Lds::sd);
  const bool norm = c.norm_mean != nullptr;
  if (norm && tid < K0) {
    const float cnt = *c.norm_count;
    s_mean[tid] = c.norm_mean[tid];
    s_sd[tid] = cnt > 0.0f ? sqrtf(fmaxf(c.norm_m2[tid] / cnt, c.norm_eps)) : 10.0f;
  }
  int64_t tile = bid;
  if (tile < ntiles) request_tile(tile);

  // ---- the actor, once ------------------------------------------------------------------
  auto bias4 = [&](const float* b, int col) {
    return b ? *reinterpret_cast<const f32x4*>(b + col) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  const bf16x8 W0 = ws_frag(c.layer[0].w, (unsigned)wc, 0, 1, lane);
  const f32x4 B0 = bias4(c.layer[0].bias, wc * 16 + 4 * lq);
  bf16x8 WI[3][KSH], WR[3][KSH];
  f32x4 BI[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
#pragma unroll
    for (int ks = 0; ks < KSH; ++ks) {
      WI[g][ks] = ws_frag(c.layer[1].w, (unsigned)(g * UT + wc), ks, KSH, lane);
      bf16x8 f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        f[i] = (bf16_t)gx.w_h[(int64_t)(ks * 32 + 8 * lq + i) * (3 * H) + g * H + wc * 16 + li];
      WR[g][ks] = f;
    }
    BI[g] = bias4(c.layer[1].bias, g * H + wc * 16 + 4 * lq);
  }
  const f32x4 BN = bias4(gx.b_hn, wc * 16 + 4 * lq);
  bf16x8 WO[KSH];
#pragma unroll
  for (int ks = 0; ks < KSH; ++ks) WO[ks] = ws_frag(c.layer[2].w, 0, ks, KSH, lane);
  f32x4 BO = f32x4{0.f, 0.f, 0.f, 0.f};
  if (c.layer[2].bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * lq + e < N_out) BO[e] = c.layer[2].bias[4 * lq + e];
  }
  for (int row = tid >> 5; row < ROWS; row += kWsThreads >> 5)
    for (int k = K0 + (tid & 31); k < 32; k += 32) bufX[row * XROW + k] = (bf16_t)0.0f;
  __syncthreads();  // s_mean / s_sd

  // the sampler's rows wait in LDS as in ws_fwd_body (one thread per row, many tiles at once)
  constexpr int kStashFloats = 4096;
  constexpr int kStashTilesMax = kWsThreads / ROWS;
  float* const ms_s = reinterpret_cast<float*>(smem + Lds::stash);
  int64_t stash_first = 0;
  int stash_n = 0;
  int stash_cap = kStashFloats / (ROWS * N_out);
  if (stash_cap > kStashTilesMax) stash_cap = kStashTilesMax;
  auto run_sampler = [&]() {
    const int slot = tid / ROWS, row = tid % ROWS;
    if (slot < stash_n) {
      const int64_t gi = (stash_first + (int64_t)slot * nblk) * ROWS + row;
      if (gi < M) mippo_sampler::fwd_row(ms_s + (slot * ROWS + row) * N_out, gi, c.samp);
    }
    __syncthreads();
  };

  for (; tile < ntiles; tile += nblk) {
    const int64_t i0 = tile * ROWS;
    // stage 0: observation tile -> bufX (normalised, bf16); carry tile -> bufA (bf16)
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
#pragma unroll
    for (int u = 0; u < H_PT; ++u) {
      const int ch = tid + u * kWsThreads;
      if (ch < H_CH) {
        const int row = ch / (H / 4), q = ch % (H / 4);
        bf16x4 hb;
#pragma unroll
        for (int e = 0; e < 4; ++e) hb[e] = (bf16_t)hin[u][e];
        *reinterpret_cast<bf16x4*>(bufA + row * AROW + 4 * q) = hb;
      }
    }
    f32x4 hprev[RTW];
#pragma unroll
    for (int r = 0; r < RTW; ++r) hprev[r] = hp[r];
    if (tile + nblk < ntiles) request_tile(tile + nblk);
    __syncthreads();
    // ---- Dense(obs -> H, relu) -----------------------------------------------------------
    {
#pragma unroll
      for (int r = 0; r < RTW; ++r) {
        const bf16x8 af =
            *reinterpret_cast<const bf16x8*>(bufX + ((wr * RTW + r) * 16 + li) * XROW + 8 * lq);
        const f32x4 a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            W0, af, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        bf16x4 vo;
#pragma unroll
        for (int e = 0; e < 4; ++e) vo[e] = (bf16_t)fmaxf(a0[e] + B0[e], 0.0f);
        *reinterpret_cast<bf16x4*>(bufB + ((wr * RTW + r) * 16 + li) * AROW + wc * 16 + 4 * lq) = vo;
      }
    }
    __syncthreads();
    // ---- the GRU cell: gi = a1 . W_i + b_i, gh = h . W_h, gates in registers ------------
    f32x4 hnew[RTW];
#pragma unroll
    for (int r = 0; r < RTW; ++r) {
      f32x4 ai[3], ah[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) ai[g] = ah[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSH; ++ks) {
        const int off = ((wr * RTW + r) * 16 + li) * AROW + ks * 32 + 8 * lq;
        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(bufB + off);
        const bf16x8 fh = *reinterpret_cast<const bf16x8*>(bufA + off);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          ai[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WI[g][ks], fa, ai[g], 0, 0, 0);
          ah[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WR[g][ks], fh, ah[g], 0, 0, 0);
        }
      }
      const int64_t gi_row = i0 + (wr * RTW + r) * 16 + li;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // gru_fwd_mfma_kernel's step, on gi = (a1 . W_i + b_i) as the projection launch
        // leaves it
        const float rg = fast_sigmoid((ai[0][e] + BI[0][e]) + ah[0][e]);
        const float zg = fast_sigmoid((ai[1][e] + BI[1][e]) + ah[1][e]);
        const float qn = ah[2][e] + BN[e];
        const float ng = fast_tanh((ai[2][e] + BI[2][e]) + rg * qn);
        hnew[r][e] = (1.0f - zg) * ng + zg * hprev[r][e];
      }
      if (gi_row < M)
        *reinterpret_cast<f32x4*>(gx.h_out + gi_row * H + wc * 16 + 4 * lq) = hnew[r];
    }
    __syncthreads();  // every wave has read h and a1: bufA takes h' now
#pragma unroll
    for (int r = 0; r < RTW; ++r) {
      bf16x4 vo;
#pragma unroll
      for (int e = 0; e < 4; ++e) vo[e] = (bf16_t)hnew[r][e];
      *reinterpret_cast<bf16x4*>(bufA + ((wr * RTW + r) * 16 + li) * AROW + wc * 16 + 4 * lq) = vo;
    }
    __syncthreads();
    // ---- Dense(H -> N_out) on h', wave w the row tile w; rows to the sampler stash -------
    if (wave < RT) {
      f32x4 ao = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSH; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(bufA + (wave * 16 + li) * AROW +
                                                          ks * 32 + 8 * lq);
        ao = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WO[ks], a, ao, 0, 0, 0);
      }
      const int row = wave * 16 + li;
      const int64_t gi = i0 + row;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (4 * lq + e < N_out) {
          const float v = ao[e] + BO[e];
          if (gi < M && c.out) c.out[gi * N_out + 4 * lq + e] = v;
          ms_s[(stash_n * ROWS + row) * N_out + 4 * lq + e] = v;
        }
      }
    }
    if (stash_n == 0) stash_first = tile;
    ++stash_n;
    if (stash_n == stash_cap || tile + nblk >= ntiles) {
      __syncthreads();
      run_sampler();
      stash_n = 0;
    }
    __syncthreads();  // the buffers are free for the next row tile
  }
}

// The recurrent policy's rollout step in one launch: value-trunk workgroups (ws_fwd_body)
// beside recurrent-actor workgroups, as policy_ws_dual_kernel.
template <int HV, int NHV, int H, int RT>
__global__ void __launch_bounds__(kWsThreads, 2)
gru_policy_ws_kernel(WsChain a, GruStepExtra gx, WsChain v, int n_value) {
  constexpr size_t nv = WsFwdLds<HV, RT, false>::bytes, na = WsFwdLds<H, RT, true>::bytes;
  __shared__ __attribute__((aligned(16))) unsigned char smem[nv > na ? nv : na];
  if ((int)blockIdx.x < n_value)
    ws_fwd_body<HV, NHV, RT, false>(v, (int)blockIdx.x, n_value, smem);
  else
    ws_gru_step_body<H, RT>(a, gx, (int)blockIdx.x - n_value, (int)gridDim.x - n_value, smem);
}

template <int H, int NH, int RT, bool SAMP>
__global__ void __launch_bounds__(kWsThreads, 2)
trunk_ws_fwd_kernel(WsChain c) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[WsFwdLds<H, RT, SAMP>::bytes];
  ws_fwd_body<H, NH, RT, SAMP>(c, (int)blockIdx.x, (int)gridDim.x, smem);
}

// The policy step at rollout / evaluation sizes (M <= 8192 rows): BOTH trunks in one
// launch, workgroups 0 .. n_value-1 the value trunk, the rest the action trunk + sampler,
// one 16*RT-row tile each when the launch fits the chip.  The tile kernels
// (mlp_bf16.hip) walk a trunk layer by layer with the weight fragments fetched from L2 one
// 64-deep step ahead — at 16 rows per workgroup that is a chain of dependent L2 round
// trips (~27 000 cycles for ~300 of MFMA); here all of a trunk's fragments are requested
// at once, up front, and the layers run out of registers.
template <int HV, int NHV, int HA, int NHA, int RT, bool FULL>
__global__ void __launch_bounds__(kWsThreads, 2)
policy_ws_dual_kernel(WsChain a, WsChain v, int n_value) {
  constexpr size_t nv = WsFwdLds<HV, RT, false>::bytes, na = WsFwdLds<HA, RT, true>::bytes;
  __shared__ __attribute__((aligned(16))) unsigned char smem[nv > na ? nv : na];
  if ((int)blockIdx.x < n_value)
    ws_fwd_body<HV, NHV, RT, false, FULL>(v, (int)blockIdx.x, n_value, smem);
  else
    ws_fwd_body<HA, NHA, RT, true, FULL>(a, (int)blockIdx.x - n_value,
                                         (int)gridDim.x - n_value, smem);
}

// ---- backward (dX chain) ------------------------------------------------------------------
// The same walk over the transposed problem (mlp_bf16.hip: chain_body<.., BWD>):
//   dz_{L-1} = bf16(g_out)                                  (or the sampler backward's rows)
//   dz_{l-1} = (dz_l . W_l^T) (.) relu'(y_{l-1}),  l = L-1 .. 1
// every dz_l is written out (bf16) for the grouped dW launch; no input gradient (the trunk
// reads observations).  Weights: the BACKWARD fragment-major images (columns = a layer's
// inputs, reduce = its outputs), stationary in registers as in the forward.
struct WsBwdLayer {
  const bf16_t* w;      // backward image of layer l (emits K_l columns, reduces N_l)
  const bf16_t* aux;    // y_{l-1} [M][ld]: relu' operand of the emitted gradient
  bf16_t* out_bf;       // dz_{l-1} [M][ld]
  int64_t ld;
  const unsigned char* mask;  // relu'(y_{l-1}) as WsLayer::mask_out wrote it (MASK kernels)
};
struct WsBwdChain {
  WsBwdLayer layer[WS_MAXL];  // [0] = head (reduce N_out), then the H x H layers, last first
  const float* g_out;         // [M][N_out] fp32, or null with the sampler backward
  bf16_t* dz_last;            // [M][ldx] bf16 image of the head's output gradient
  int64_t ldx;
  int64_t M;
  int N_out;
  mippo_sampler::BwdParams sbwd;  // sbwd.A == 0: gradient comes from g_out
};

template <int H, int RT, bool SAMP>
struct WsBwdLds {
  static constexpr size_t kRows = 16 * RT;
  static constexpr size_t kStashTiles = SAMP ? kWsThreads / kRows : 1;
  static constexpr size_t bufX = 0;
  static constexpr size_t bufA = bufX + kStashTiles * kRows * (32 + 8) * 2;
  static constexpr size_t bufB = bufA + kRows * (H + 8) * 2;
  static constexpr size_t bytes = bufB + kRows * (H + 8) * 2;
};

// ---- GAE: the reverse scan, the advantage statistics and the loss terms INSIDE this launch ----
// (a13 + a14, ppo.py:351-394, 447-458, 477-503; arithmetic of gae_loss.hip expression for
// expression).  mi_gae_ppo_loss_f32 is a launch of 16 workgroups between the replay forward
// and this kernel — 15 us of launch, round-trip and exchange latency per gradient step for
// 0.4 MB of operands.  Here every workgroup scans the 64 envs of each of its OWN row tiles
// (row tile = 64 rows = (step t, env group g): the group's scan from T-1 down to t, one wave
// per tile) and keeps the tile's advantages (and values) in LDS; d loss / d value needs the
// raw advantage only.  The advantage STATISTICS are the one global quantity: action-trunk
// workgroup g < B / 64 also scans env group g whole and publishes its (sum, sum of squares);
// every action-trunk workgroup waits for the B / 64 partials (a counter and a bounded spin:
// the launch has at most one workgroup per CU, all co-resident) and sums them in
// gae_loss_kernel's order — same per-env accumulation, same shuffle trees, the same bits.
// (A first form had every action-trunk workgroup scan all of [T, B] itself, no exchange: 2 x
// 13 000 cycles of fp64 accumulation and LDS transposes per workgroup, slower than the launch
// it replaced.)  d loss / d log-likelihood and d loss / d value are evaluated by the threads
// that consume them (the sampler backward's row thread, the head-gradient stage), and the four
// loss scalars are summed from per-tile fp64 partials by the last workgroup to finish.
// B % 64 == 0.
struct WsGae {
  const float *rewards, *values, *last_value;  // [T][B], [T][B], [B]
  const unsigned char *done, *trunc;           // [T][B]
  const float *ll_new, *ll_old, *reg;          // [T][B]; reg nullable
  float* loss_out;                             // [4] actor, critic, regularisation, clip fraction
  double* part;                                // workspace: [ntiles][4] partial sums
  double* sp;                                  // workspace: [groups][2] statistics partials
  unsigned int *ticket, *arrive;               // workspace header (zero between launches);
                                               // arrive[1]: sticky count of timed-out waits;
                                               // arrive[2], [3]: test hook (spin limit in us,
                                               // extra arrivals), zero in production
  int T, B;
  float gamma, lambda, clip, critic_weight;
  int normalize;
  // Env-sharded run (world > 1; SURVEY 8e (2): the normalisation of `ppo.py:477-480` is over
  // the GLOBAL minibatch): a publishing workgroup writes its group's partial into slot
  // (parity, rank) of EVERY rank's one-shot region (comm.hip) and raises that group's flag
  // there; the consumers wait for world x B/64 flags of their own region and sum the partials
  // in (rank, group) order — every rank the same bits.  The launch counts as one collective
  // of the communicator (comm_finish), so the two slot parities alternate with the other
  // exchanges of the gradient step.
  mippo_comm::CommDev cd;
  int world;
};
// -DGAE_W_FIRST=1: every trunk's stationary fragments requested before the scan (measured:
// the 256-wide value trunk then spills 29 registers to scratch, 66.4 -> 64.5 M env-steps/s)
#ifndef GAE_W_FIRST
#define GAE_W_FIRST 0
#endif
constexpr int kGaeMaxT = 32;
constexpr int kGaeMaxQ = 16;    // row tiles a workgroup may own
constexpr int kGaeMaxGroups = 32;  // B <= 2048
constexpr int kGaeHeaderBytes = 64 + kGaeMaxGroups * 2 * 8;  // [ticket | arrive | pad][sp]
constexpr unsigned long long kGaeSpinTicks = 200000000ull;  // 2 s of the 100 MHz wall clock
constexpr size_t kGaeStageBytes = kGaeMaxT * 64 * 5;  // per wave: 32 x 64 floats + 32 x 64 bytes
struct WsGaeLds {
  static constexpr size_t adv = 0;                                   // float [kGaeMaxQ][64]
  static constexpr size_t val = adv + kGaeMaxQ * 64 * 4;              // float [kGaeMaxQ][64]
  static constexpr size_t norm = val + kGaeMaxQ * 64 * 4;             // float [2]
  static constexpr size_t last = norm + 8;                            // int
  static constexpr size_t bytes = last + 8;
};

// The reverse scan of env group `grp` (one env per lane) from step T-1 down to t_stop;
// emit(t, advantage, value) at every step, (sum, sum of squares) of the lane's advantages
// returned through s / s2.  Wave-uniform arguments.
// Operand loads: what bounds this phase is the NUMBER of vector-memory instructions a CU
// issues (~22 cycles each, L2-warm), not bytes — one dword per lane and step was 128
// instructions per wave and 10 us per 8 groups.  So every instruction moves 1 KiB (16 bytes per
// lane: 4 steps x 64 floats, or 16 steps x 64 flag bytes), 20 instructions per group, all
// requested before the first is used, and the [step][env] blocks are turned into per-lane
// columns through a wave-private LDS staging area (sf: 32 x 64 floats, sb: 32 x 64 bytes).
// Rows start 16-byte aligned (B % 64 == 0, bases checked on the host); steps past T re-read
// step T-1 (branch-free: behind a per-step predicate the compiler paired each step's loads
// with the step before's arithmetic — 30 dependent round trips).
template <bool STATS, typename Emit>
__device__ __forceinline__ void ws_gae_scan_group(const WsGae& g, const int grp, const int t_stop,
                                                  const int lane, float* sf, unsigned char* sb,
                                                  double& s, double& s2, Emit emit) {
#pragma clang fp contract(off)
  const int T = g.T;
  const int64_t B = g.B;
  const int64_t e0 = (int64_t)grp * 64;
  u32x4 qr[kGaeMaxT / 4], qv[kGaeMaxT / 4], qd[kGaeMaxT / 16], qt[kGaeMaxT / 16];
#pragma unroll
  for (int k = 0; k < kGaeMaxT / 4; ++k) {
    const int t = 4 * k + (lane >> 4);
    const int64_t o = (int64_t)(t < T ? t : T - 1) * B + e0 + 4 * (lane & 15);
    qr[k] = *reinterpret_cast<const u32x4*>(g.rewards + o);
    qv[k] = *reinterpret_cast<const u32x4*>(g.values + o);
  }
#pragma unroll
  for (int k = 0; k < kGaeMaxT / 16; ++k) {
    const int t = 16 * k + (lane >> 2);
    const int64_t o = (int64_t)(t < T ? t : T - 1) * B + e0 + 16 * (lane & 3);
    qd[k] = *reinterpret_cast<const u32x4*>(g.done + o);
    qt[k] = *reinterpret_cast<const u32x4*>(g.trunc + o);
  }
  float next_v = g.last_value[e0 + lane], next_a = 0.0f;
  __builtin_amdgcn_sched_barrier(0);
  float r[kGaeMaxT], v[kGaeMaxT];
  unsigned dm = 0u, tm = 0u;  // bit t: done / truncated at step t (one register each, not 32)
  u32x4* const sf4 = reinterpret_cast<u32x4*>(sf) + lane;  // (4 k + lane / 16) * 64 + 4 (lane % 16)
  u32x4* const sb4 = reinterpret_cast<u32x4*>(sb) + lane;  // (16 k + lane / 4) * 64 + 16 (lane % 4)
#pragma unroll
  for (int k = 0; k < kGaeMaxT / 4; ++k) sf4[64 * k] = qr[k];
#pragma unroll
  for (int t = 0; t < kGaeMaxT; ++t) r[t] = sf[t * 64 + lane];
#pragma unroll
  for (int k = 0; k < kGaeMaxT / 4; ++k) sf4[64 * k] = qv[k];
#pragma unroll
  for (int t = 0; t < kGaeMaxT; ++t) v[t] = sf[t * 64 + lane];
#pragma unroll
  for (int k = 0; k < kGaeMaxT / 16; ++k) sb4[64 * k] = qd[k];
#pragma unroll
  for (int t = 0; t < kGaeMaxT; ++t) dm |= (sb[t * 64 + lane] ? 1u : 0u) << t;
#pragma unroll
  for (int k = 0; k < kGaeMaxT / 16; ++k) sb4[64 * k] = qt[k];
#pragma unroll
  for (int t = 0; t < kGaeMaxT; ++t) tm |= (sb[t * 64 + lane] ? 1u : 0u) << t;
  s = 0.0;
  s2 = 0.0;
#pragma unroll
  for (int t = kGaeMaxT - 1; t >= 0; --t) {
    if (t < T && t >= t_stop) {
      const float vt = v[t];
      const bool dn = (dm >> t) & 1u;
      const float nv = dn ? 0.0f : next_v;
      float delta = (r[t] + g.gamma * nv) - vt;
      delta = ((tm >> t) & 1u) ? 0.0f : delta;
      const float keep = dn ? 0.0f : 1.0f;
      const float av = delta + ((keep * g.gamma) * g.lambda) * next_a;
      if constexpr (STATS) {
        // gae_loss.hip: s2 += (double)av * (double)av — the product of two fp32 values is exact
        // in fp64, so the fused form rounds the same sum once: the same bits, one DP op fewer
        const double ad = (double)av;
        s += ad;
        s2 = __builtin_fma(ad, ad, s2);
      }
      next_a = av;
      next_v = vt;
      emit(t, av, vt);
    }
  }
}

// d loss / d value of one element and its squared error (gae_loss.hip phase 2, critic terms)
__device__ __forceinline__ float ws_gae_value_grad(const WsGae& g, const float a_raw,
                                                   const float v, double& sq) {
#pragma clang fp contract(off)
  const float inv_n = 1.0f / (float)((int64_t)g.T * (int64_t)g.B);
  const float target = v + a_raw;  // ppo.py:456-458
  const float diff = v - target;
  sq = (double)(diff * diff);
  return g.critic_weight * diff * inv_n;
}

// d loss / d log-likelihood of one element and its three actor-side loss terms
// (gae_loss.hip phase 2: clipped surrogate, regulariser row, clipping indicator)
__device__ __forceinline__ float ws_gae_actor_grad(const WsGae& g, const float a_raw,
                                                   const float mean, const float denom,
                                                   const float lln, const float llo,
                                                   const float rg, double (&q)[3]) {
#pragma clang fp contract(off)
  const float inv_n = 1.0f / (float)((int64_t)g.T * (int64_t)g.B);
  const float lo = 1.0f - g.clip, hi = 1.0f + g.clip;
  const float an = g.normalize ? (a_raw - mean) / denom : a_raw;
  const float rt = expf(lln - llo);
  const float c1 = rt * an;
  const float c2 = fminf(fmaxf(rt, lo), hi) * an;
  q[0] = (double)fminf(c1, c2);
  q[1] = (double)rg;
  q[2] = fabsf(rt - 1.0f) > g.clip ? 1.0 : 0.0;
  return c1 <= c2 ? -(an * rt) * inv_n : 0.0f;
}

// MASK: relu' comes from the forward's masks — one byte per lane and 16 x 16 tile instead of
// 8 bytes of the bf16 image; the images are 48 % of this kernel's HBM bytes at BASELINE C2
// and the kernel waits on exactly those loads (profiles/r02_trace_ws_bwd.txt).
template <int H, int NH, int RT, bool SAMP, bool MASK = false, bool GAE = false>
__device__ __forceinline__ void ws_bwd_body(const WsBwdChain& c, const int bid, const int nblk,
                                            unsigned char* smem, const WsGae* gp = nullptr,
                                            unsigned char* gsm = nullptr) {
  static_assert(!GAE || (RT == 4 && MASK), "GAE: 64-row tiles, masks");
  using G = WsGeom<H>;
  constexpr int CW = G::CW, RW = G::RW, TPW = G::TPW;
  constexpr int RTW = RT / RW;
  constexpr int ROWS = 16 * RT;
  constexpr int KSH = H / 32;
  constexpr int AROW = H + 8;
  constexpr int XROW = 32 + 8;
  constexpr int kStashTiles = SAMP ? kWsThreads / ROWS : 1;  // one row per thread
  using Lds = WsBwdLds<H, RT, SAMP>;
  bf16_t* const bufX = reinterpret_cast<bf16_t*>(smem + Lds::bufX);  // [kStashTiles][ROWS][XROW]
  bf16_t* const bufA = reinterpret_cast<bf16_t*>(smem + Lds::bufA);  // [ROWS][AROW]
  bf16_t* const bufB = reinterpret_cast<bf16_t*>(smem + Lds::bufB);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % CW, wr = wave / CW;
  const int li = lane & 15, lq = lane >> 4;
  const int64_t M = c.M;
  const int N_out = c.N_out;
  const int64_t ntiles = (M + ROWS - 1) / ROWS;
#ifdef MIPPO_TRACE
  int ev_ = 0;
#endif
  WS_TR();  // 0: start

  // ---- the transposed trunk, once -------------------------------------------------------
  bf16x8 WO[TPW];
  bf16x8 WH[NH > 0 ? NH : 1][TPW][KSH];
  auto load_weights = [&]() {
#pragma unroll
    for (int b = 0; b < TPW; ++b) {
      const unsigned ct = (unsigned)(wc + CW * b);
      WO[b] = ws_frag(c.layer[0].w, ct, 0, 1, lane);
#pragma unroll
      for (int l = 0; l < NH; ++l)
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks)
          WH[l][b][ks] = ws_frag(c.layer[1 + l].w, ct, ks, KSH, lane);
    }
  };
  // GAE: the scan holds ~130 registers of operands in flight; the value trunk's stationary
  // fragments (72 registers) are requested after it (with both live the allocator spilled to
  // scratch), a narrow action trunk's (12) before
  if constexpr (!GAE || GAE_W_FIRST || (SAMP && H <= 128)) load_weights();
  // pad columns N_out..31 of the head-gradient rows stay zero for the whole kernel
  auto zero_pad = [&]() {
    for (int i = tid; i < kStashTiles * ROWS * 32; i += kWsThreads) {
      const int row = i >> 5, k = i & 31;
      if (k >= N_out) bufX[row * XROW + k] = (bf16_t)0.0f;
    }
  };
  if constexpr (!GAE) zero_pad();  // GAE: the scan's staging area lies over these buffers

  // ---- GAE: the scan, before anything needs a gradient row (see WsGae) ----------------------
  float* const s_adv = reinterpret_cast<float*>(gsm + WsGaeLds::adv);
  float* const s_val = reinterpret_cast<float*>(gsm + WsGaeLds::val);
  float g_mean = 0.0f, g_denom = 1.0f;
  float pre_lln = 0.0f, pre_llo = 0.0f, pre_rg = 0.0f;  // SAMP: the first stash round's rows
  if constexpr (GAE) {
    const WsGae& g = *gp;
    const int GB = g.B >> 6;  // env groups = row tiles per step
    // sharded: this launch is collective number seq + 1 of the communicator (every workgroup
    // reads the old count: it moves only once all of them have finished, comm_finish)
    unsigned int x_seq = 0;
    int x_parity = 0;
    if (g.world > 1) {
      const unsigned long long s64 =
          reinterpret_cast<const mippo_comm::CommHeader*>(g.cd.peer[g.cd.rank])->seq + 1;
      x_seq = (unsigned int)s64;
      x_parity = (int)(s64 & 1);
    }
    float* const sf = reinterpret_cast<float*>(smem + (size_t)wave * kGaeStageBytes);
    unsigned char* const sb = smem + (size_t)wave * kGaeStageBytes + kGaeMaxT * 64 * 4;
    if constexpr (SAMP) {
      const int64_t t0 = (int64_t)bid + (int64_t)wave * nblk;  // slot = wave, row = lane
      if (t0 < ntiles) {
        const int64_t gi = t0 * ROWS + lane;
        pre_lln = g.ll_new[gi];
        pre_llo = g.ll_old[gi];
        if (g.reg) pre_rg = g.reg[gi];
      }
      // the statistics: action-trunk workgroup `bid` < GB scans env group `bid` whole (its
      // last wave, beside the others' own-tile scans) and publishes the group's partial
      if (g.normalize && wave == kWsThreads / 64 - 1 && bid < GB) {
        double s1, s2;
        ws_gae_scan_group<true>(g, bid, 0, lane, sf, sb, s1, s2, [](int, float, float) {});
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          s1 += __shfl_down(s1, off, 64);
          s2 += __shfl_down(s2, off, 64);
        }
        if (g.world > 1) {
          // lane q hands the partial to rank q (its own region included): payload, system
          // release, then the group's flag (comm.hip: push_chunk)
          const double b1 = __shfl(s1, 0, 64), b2 = __shfl(s2, 0, 64);
          if (lane < g.world) {
            using namespace mippo_comm;
            char* region = g.cd.peer[lane];
            double* dst = reinterpret_cast<double*>(
                region + slot_off(g.cd.chunks, g.cd.world, g.cd.slot_bytes, x_parity, g.cd.rank) +
                (int64_t)bid * 16);
            __hip_atomic_store(dst, b1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(dst + 1, b2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            unsigned int* flag = reinterpret_cast<unsigned int*>(
                region + flags_off(g.cd.chunks, g.cd.world, x_parity, g.cd.rank)) + bid;
            __hip_atomic_store(flag, x_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        } else if (lane == 0) {
          // write-through (sc1) stores, drained, then the arrival (gae_loss.hip's hand-over)
          __hip_atomic_store(&g.sp[2 * bid], s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&g.sp[2 * bid + 1], s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_fetch_add(g.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        WS_TR();  // GAE (publishing workgroups only): group partial out
      }
    }
    // the advantages (and values) of this workgroup's own row tiles: tile q = (step t_q, group),
    // the group's scan from T-1 down to t_q, one wave per tile
    for (int q = wave; q < kGaeMaxQ; q += 8) {
      const int64_t tq = (int64_t)bid + (int64_t)q * nblk;
      if (tq < ntiles) {
        const int t_q = (int)(tq / GB);
        const int grp = (int)(tq - (int64_t)t_q * GB);
        double s1, s2;
        ws_gae_scan_group<false>(g, grp, t_q, lane, sf, sb, s1, s2,
                                 [&](int t, float av, float vt) {
                                   if (t == t_q) {
                                     s_adv[q * 64 + lane] = av;
                                     if (!SAMP) s_val[q * 64 + lane] = vt;
                                   }
                                 });
      }
    }
    WS_TR();  // GAE: own row tiles scanned
    if (SAMP && g.normalize) {
      float* const s_norm = reinterpret_cast<float*>(gsm + WsGaeLds::norm);
      if (wave == 0) {
        bool timed_out = false;
        double t1 = 0.0, t2 = 0.0;
        int n_ranks = 1;
        if (g.world > 1) {
          using namespace mippo_comm;
          n_ranks = g.world;
          char* own = g.cd.peer[g.cd.rank];
          CommHeader* hdr = reinterpret_cast<CommHeader*>(own);
          const unsigned long long limit = hdr->timeout;
          const int total = g.world * GB;
          bool late = hdr->errors != 0;  // a lost peer earlier: nothing is trusted any more
          for (int idx = lane; idx < total && !late; idx += 64) {
            const int q = idx / GB, gg = idx - q * GB;
            const unsigned int* flag = reinterpret_cast<const unsigned int*>(
                own + flags_off(g.cd.chunks, g.cd.world, x_parity, q)) + gg;
            const unsigned long long t0 = wall_clock64();
            while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) -
                         x_seq) < 0) {
              __builtin_amdgcn_s_sleep(2);
              if (wall_clock64() - t0 > limit) {
                atomicAdd(&hdr->errors, 1u);
                if (g.cd.error_word) atomicAdd(g.cd.error_word, 1u);
                __hip_atomic_fetch_add(g.arrive + 1, 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                late = true;
                break;
              }
            }
          }
          timed_out = __ballot(late) != 0ull;
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: written by other devices
          // the same tree over world x GB partials, (rank, group) order: lane-strided, then
          // lane order — identical on every rank
          for (int idx = lane; idx < total; idx += 64) {
            const int q = idx / GB, gg = idx - q * GB;
            const double* sp = reinterpret_cast<const double*>(
                own + slot_off(g.cd.chunks, g.cd.world, g.cd.slot_bytes, x_parity, q) +
                (int64_t)gg * 16);
            t1 += __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            t2 += __hip_atomic_load(sp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        } else {
        if (lane == 0) {
          // header words 3, 4 (zero in production): a spin limit in microseconds and a number
          // of arrivals to wait for beyond the real ones — the test hook that forces the
          // timeout branch (ops.set_handover_test_hook); requested with the first poll
          const unsigned int hook_us = __builtin_nontemporal_load(g.arrive + 2);
          const unsigned int hook_extra = __builtin_nontemporal_load(g.arrive + 3);
          const unsigned long long t0 = wall_clock64();
          while (__hip_atomic_load(g.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <
                 (unsigned)GB + hook_extra) {
            __builtin_amdgcn_s_sleep(1);
            const unsigned long long limit =
                hook_us ? (unsigned long long)hook_us * 100ull : kGaeSpinTicks;
            if (wall_clock64() - t0 > limit) {  // never in a healthy launch
              // sticky word: it travels to the host with the iteration's metrics
              // (ops.health_words -> loop.MetricPack; IterationRunner.collect raises in the
              // iteration it happens) and is read again by loop.health_check
              __hip_atomic_fetch_add(g.arrive + 1, 1u, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
              timed_out = true;
              break;
            }
          }
        }
        // acquire only: the partials may sit stale in this XCD's L2 from the previous launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // gae_loss_kernel's tree over the group partials: lane-strided, then lane order
        for (int gg = lane; gg < GB; gg += 64) {
          t1 += __hip_atomic_load(&g.sp[2 * gg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          t2 += __hip_atomic_load(&g.sp[2 * gg + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          t1 += __shfl_down(t1, off, 64);
          t2 += __shfl_down(t2, off, 64);
        }
        if (lane == 0) {
          const double cnt = (double)g.T * (double)g.B * (double)n_ranks;
          const double m = t1 / cnt;
          double var = t2 / cnt - m * m;
          if (var < 0.0) var = 0.0;
          // a hand-over that ran out has incomplete partials: poison the normalisation so
          // that a missed host check cannot train on wrong statistics silently
          s_norm[0] = (float)m;
          // (a zero denominator, not a NaN mean: the surrogate's `c1 <= c2` select would turn a
          // NaN advantage into a ZERO gradient; +-inf advantages go through it and reach the
          // parameters as NaN)
          s_norm[1] = timed_out ? 0.0f : (float)sqrt(var) + 1e-8f;
        }
      }
      __syncthreads();
      g_mean = s_norm[0];
      g_denom = s_norm[1];
    } else {
      __syncthreads();
    }
    WS_TR();  // GAE: prologue done
  }

  if constexpr (GAE) {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(GAE_W_FIRST || (SAMP && H <= 128))) load_weights();
    zero_pad();  // after the prologue's last barrier: the staging area is free
  }

  constexpr int IN_PT = (ROWS * 16 + kWsThreads - 1) / kWsThreads;  // N_out <= 16
  const int nel = ROWS * N_out;
  const float rcpN = 1.0f / (float)N_out;
  float gin[IN_PT];
  int q_next = 0;  // GAE: ordinal (among this workgroup's) of the tile requested next
  auto request_input = [&](int64_t tile) {
    if (SAMP) return;
    if constexpr (GAE) {  // N_out == 1: row e of the tile, wave 0
#pragma unroll
      for (int u = 0; u < IN_PT; ++u) gin[u] = 0.0f;
      if (tid < 64) {
        double sq;
        gin[0] = ws_gae_value_grad(*gp, s_adv[q_next * 64 + tid], s_val[q_next * 64 + tid], sq);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
        if (tid == 0)
          __hip_atomic_store(&gp->part[4 * tile + 1], sq, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
      ++q_next;
      return;
    }
    const int64_t g0 = tile * ROWS * N_out;
#pragma unroll
    for (int u = 0; u < IN_PT; ++u) {
      const int e = tid + u * kWsThreads;
      gin[u] = (e < nel && g0 + e < M * N_out) ? c.g_out[g0 + e] : 0.0f;
    }
  };

  WS_TR();  // 1: weights requested
  int64_t tile = bid;
  int stash_n = 0, stash_i = 0;  // SAMP: head-gradient rows of `stash_n` row tiles wait in bufX
  int stash_round = 0;           // GAE: rounds of kStashTiles row tiles done
  if (tile < ntiles) request_input(tile);
  for (; tile < ntiles; tile += nblk) {
    const int64_t i0 = tile * ROWS;
    const bf16_t* xin = bufX;
    WS_TR();  // tile + 0
    if constexpr (SAMP) {
      // sampling_layers.py:82-147 differentiated, one thread per row — a long chain of
      // transcendentals whatever the number of rows, so the rows of up to kStashTiles of
      // this workgroup's row tiles are produced at once
      if (stash_i == stash_n) {
        const int slot = tid / ROWS, row = tid % ROWS;
        const int64_t t = tile + (int64_t)slot * nblk;
        if (t < ntiles) {
          bf16_t* dst = bufX + (slot * ROWS + row) * XROW;
          const int64_t gi = t * ROWS + row;
          if constexpr (GAE) {  // whole tiles (M = T B, B % 64 == 0): slot = wave, row = lane
            const WsGae& g = *gp;
            float lln = pre_lln, llo = pre_llo, rg = pre_rg;
            if (stash_round > 0) {
              lln = g.ll_new[gi];
              llo = g.ll_old[gi];
              rg = g.reg ? g.reg[gi] : 0.0f;
            }
            double q3[3];
            const float gl = ws_gae_actor_grad(
                g, s_adv[(stash_round * kStashTiles + slot) * 64 + row], g_mean, g_denom, lln, llo,
                rg, q3);
            mippo_sampler::bwd_row_gl(gi, c.sbwd, gl, [dst](int j, float v) { dst[j] = (bf16_t)v; });
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
#pragma unroll
              for (int k = 0; k < 3; ++k) q3[k] += __shfl_down(q3[k], off, 64);
            if (row == 0) {
              __hip_atomic_store(&g.part[4 * t + 0], q3[0], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(&g.part[4 * t + 2], q3[1], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(&g.part[4 * t + 3], q3[2], __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
            }
          } else if (gi < M) {
            mippo_sampler::bwd_row(gi, c.sbwd, [dst](int j, float v) { dst[j] = (bf16_t)v; });
          } else {
            for (int k = 0; k < N_out; ++k) dst[k] = (bf16_t)0.0f;
          }
        }
        stash_n = 0;
        for (int q = 0; q < kStashTiles; ++q)
          if (tile + (int64_t)q * nblk < ntiles) ++stash_n;
        stash_i = 0;
        ++stash_round;
      }
      xin = bufX + stash_i * ROWS * XROW;
      ++stash_i;
    } else {
#pragma unroll
      for (int u = 0; u < IN_PT; ++u) {
        const int e = tid + u * kWsThreads;
        if (e < nel) {
          const int row = (int)(((float)e + 0.5f) * rcpN);
          bufX[row * XROW + (e - row * N_out)] = (bf16_t)gin[u];
        }
      }
      if (tile + nblk < ntiles) request_input(tile + nblk);
    }
    // relu' operands of this row tile, requested now: 8 bytes of the bf16 image per (row,
    // column tile) — or, MASK, the lane's byte of the forward's mask
    s16x4 auxr[NH + 1][MASK ? 1 : RTW][MASK ? 1 : TPW];
    unsigned auxm[NH + 1][MASK ? TPW : 1];  // byte r: this wave's row tile r
#pragma unroll
    for (int l = 0; l <= NH; ++l)
#pragma unroll
      for (int b = 0; b < TPW; ++b) {
        const int col = (wc + CW * b) * 16 + 4 * lq;
        if constexpr (MASK)
          auxm[l][b] = ws_mask_load<RTW>(c.layer[l].mask, (i0 >> 4) + wr * RTW, H / 16,
                                         wc + CW * b, lane);
#pragma unroll
        for (int r = 0; r < RTW; ++r) {
          if constexpr (!MASK) {
            const int64_t gi = i0 + (wr * RTW + r) * 16 + li;
            s16x4 a = s16x4{0, 0, 0, 0};
            if (gi < M)
              a = *reinterpret_cast<const s16x4*>(c.layer[l].aux + gi * c.layer[l].ld + col);
            auxr[l][r][b] = a;
          }
        }
      }
    WS_TR();  // tile + 1: head gradient staged, aux requested
    __syncthreads();
    WS_TR();  // tile + 2
    if (c.dz_last && tid < ROWS && i0 + tid < M) {  // bf16 image of the head's gradient
      const int64_t ldx = c.ldx;
      for (int k = 0; k < (int)ldx; k += 8)
        *reinterpret_cast<u32x4*>(c.dz_last + (i0 + tid) * ldx + k) =
            *reinterpret_cast<const u32x4*>(xin + tid * XROW + k);
    }
    f32x4 acc[RTW][TPW];
    auto epilogue = [&](const s16x4(&aux)[MASK ? 1 : RTW][MASK ? 1 : TPW],
                        const unsigned(&am)[MASK ? TPW : 1], bf16_t* nbuf) {
#pragma unroll
      for (int b = 0; b < TPW; ++b) {
        const int col = (wc + CW * b) * 16 + 4 * lq;
#pragma unroll
        for (int r = 0; r < RTW; ++r) {
          bf16x4 vo;
          if constexpr (MASK) {
            const unsigned mk = am[b] >> (8 * r);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              vo[e] = (bf16_t)(acc[r][b][e] * (((mk >> e) & 1u) ? 1.0f : 0.0f));
          } else {
            const bf16x4 a4 = __builtin_bit_cast(bf16x4, aux[r][b]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              vo[e] = (bf16_t)(acc[r][b][e] * ((float)a4[e] > 0.0f ? 1.0f : 0.0f));
          }
          *reinterpret_cast<bf16x4*>(nbuf + ((wr * RTW + r) * 16 + li) * AROW + col) = vo;
        }
      }
    };
    auto copy_out = [&](const bf16_t* buf, bf16_t* dst, int64_t ld) {
      constexpr int CPR = H / 8;
      constexpr int RPP = kWsThreads / CPR;
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
        if (row < ROWS && i0 + row < M)
          *reinterpret_cast<u32x4*>(dst + (i0 + row) * ld + cc * 8) = v[p];
      }
    };
    // ---- head, transposed: reduce over N_out (one k-step), emit H columns ----------------
    {
      bf16x8 af[RTW];
#pragma unroll
      for (int r = 0; r < RTW; ++r)
        af[r] = *reinterpret_cast<const bf16x8*>(xin + ((wr * RTW + r) * 16 + li) * XROW + 8 * lq);
#pragma unroll
      for (int b = 0; b < TPW; ++b)
#pragma unroll
        for (int r = 0; r < RTW; ++r)
          acc[r][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              WO[b], af[r], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    WS_TR();  // tile + 3: head multiplied
    epilogue(auxr[0], auxm[0], bufA);
    WS_TR();  // tile + 4: epilogue (waits for the aux loads)
    __syncthreads();
    WS_TR();  // tile + 5
    copy_out(bufA, c.layer[0].out_bf, c.layer[0].ld);
    WS_TR();  // tile + 6: copy-out issued
    bf16_t* cur = bufA;
    bf16_t* nxt = bufB;
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
      epilogue(auxr[1 + l], auxm[1 + l], nxt);
      WS_TR();  // hidden: epilogue
      __syncthreads();
      WS_TR();
      copy_out(nxt, c.layer[1 + l].out_bf, c.layer[1 + l].ld);
      WS_TR();  // hidden: copy-out issued
      bf16_t* t = cur;
      cur = nxt;
      nxt = t;
    }
    __syncthreads();  // the buffers are free for the next row tile
    WS_TR();  // tile end
  }
}

template <int H, int NH, int RT, bool SAMP>
__global__ void __launch_bounds__(kWsThreads, 2)
trunk_ws_bwd_kernel(WsBwdChain c) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[WsBwdLds<H, RT, SAMP>::bytes];
  ws_bwd_body<H, NH, RT, SAMP>(c, (int)blockIdx.x, (int)gridDim.x, smem);
}

// Sampler backward + both dX chains in one launch: workgroups 0 .. n_value-1 the value
// trunk, the rest the action trunk (the CUs split between them as in policy_ws_dual_kernel).
template <int HV, int NHV, int HA, int NHA, int RT, bool MASK>
__global__ void __launch_bounds__(kWsThreads, 2)
policy_ws_bwd_dual_kernel(WsBwdChain a, WsBwdChain v, int n_value) {
  constexpr size_t nv = WsBwdLds<HV, RT, false>::bytes, na = WsBwdLds<HA, RT, true>::bytes;
  __shared__ __attribute__((aligned(16))) unsigned char smem[nv > na ? nv : na];
  if ((int)blockIdx.x < n_value)
    ws_bwd_body<HV, NHV, RT, false, MASK>(v, (int)blockIdx.x, n_value, smem);
  else
    ws_bwd_body<HA, NHA, RT, true, MASK>(a, (int)blockIdx.x - n_value,
                                         (int)gridDim.x - n_value, smem);
}

// The same launch with the GAE scan and the loss terms inside (WsGae): no g_loglik / g_value
// operands, the four loss scalars summed from the per-tile partials by the last workgroup.
template <int HV, int NHV, int HA, int NHA>
__global__ void __launch_bounds__(kWsThreads, 2)
policy_ws_bwd_gae_kernel(WsBwdChain a, WsBwdChain v, WsGae g, int n_value) {
  constexpr size_t nv = WsBwdLds<HV, 4, false>::bytes, na = WsBwdLds<HA, 4, true>::bytes;
  constexpr size_t nst = (kWsThreads / 64) * kGaeStageBytes;  // the scan's staging, all waves
  constexpr size_t nb0 = nv > na ? nv : na;
  constexpr size_t nb = ((nb0 > nst ? nb0 : nst) + 15) / 16 * 16;
  __shared__ __attribute__((aligned(16))) unsigned char smem[nb + WsGaeLds::bytes];
  unsigned char* const gsm = smem + nb;
  if ((int)blockIdx.x < n_value)
    ws_bwd_body<HV, NHV, 4, false, true, true>(v, (int)blockIdx.x, n_value, smem, &g, gsm);
  else
    ws_bwd_body<HA, NHA, 4, true, true, true>(a, (int)blockIdx.x - n_value,
                                              (int)gridDim.x - n_value, smem, &g, gsm);
  // sharded: the launch counts as one collective once every workgroup has finished
  if (g.world > 1)
    mippo_comm::comm_finish(reinterpret_cast<mippo_comm::CommHeader*>(g.cd.peer[g.cd.rank]),
                            reinterpret_cast<const mippo_comm::CommHeader*>(
                                g.cd.peer[g.cd.rank])->seq + 1);
  // this workgroup's partials are out (write-through stores, drained) before it takes a ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* const s_last = reinterpret_cast<int*>(gsm + WsGaeLds::last);
  const int tid = threadIdx.x;
  if (tid == 0)
    *s_last = __hip_atomic_fetch_add(g.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
              gridDim.x - 1;
  __syncthreads();
  if (*s_last && !g.loss_out) {
    // deferred: the per-tile partials stay where they are for mi_policy_loss_finalize_f32 (the
    // sum below, with its acquire and its load round trip, is 2.5-3 us at the END of the
    // launch's critical path — for four scalars nobody reads before the iteration's metrics);
    // every workgroup is past the statistics spin: re-arm both counters
    if (tid == 0) {
      __hip_atomic_store(g.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(g.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (*s_last && tid < 64) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int64_t ntiles = a.M / 64;
    double z[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t t = tid; t < ntiles; t += 64)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        z[k] += __hip_atomic_load(&g.part[4 * t + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int k = 0; k < 4; ++k) z[k] += __shfl_down(z[k], off, 64);
    if (tid == 0) {
      const double dn = (double)g.T * (double)g.B;
      g.loss_out[0] = (float)(-z[0] / dn);
      g.loss_out[1] = (float)(0.5 * z[1] / dn);
      g.loss_out[2] = (float)(z[2] / dn);
      g.loss_out[3] = (float)(z[3] / dn);
      // every workgroup is past the statistics spin (it finished): re-arm both counters
      __hip_atomic_store(g.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(g.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The four loss scalars of up to kLossFinalMax deferred launches from their per-tile partials:
// one wave per launch, the in-kernel sum's order (lane-strided, then the shuffle tree) — the
// same bits.
constexpr int kLossFinalMax = 32;
struct LossFinal {
  const double* part[kLossFinalMax];
  float* out[kLossFinalMax];
  long long ntiles[kLossFinalMax];
  double dn[kLossFinalMax];
};
__global__ void __launch_bounds__(64) policy_loss_finalize_kernel(LossFinal f) {
  const int s = blockIdx.x, tid = threadIdx.x;
  const double* part = f.part[s];
  double z[4] = {0.0, 0.0, 0.0, 0.0};
  for (long long t = tid; t < f.ntiles[s]; t += 64)
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] += part[4 * t + k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int k = 0; k < 4; ++k) z[k] += __shfl_down(z[k], off, 64);
  if (tid == 0) {
    const double dn = f.dn[s];
    f.out[s][0] = (float)(-z[0] / dn);
    f.out[s][1] = (float)(0.5 * z[1] / dn);
    f.out[s][2] = (float)(z[2] / dn);
    f.out[s][3] = (float)(z[3] / dn);
  }
}

template <int H, int NH, int RT>
int ws_launch(const WsChain& c, hipStream_t st) {
  const int64_t ntiles = mippo::ceil_div(c.M, 16 * RT);
  const dim3 grid((unsigned)ws_grid(ntiles));
  if (c.samp.A > 0) {
    MI_REQUIRE(16 * RT * c.N_out <= 4096, "weights-stationary trunk: 2A = %d is too wide for "
               "the sampler stash", c.N_out);
    hipLaunchKernelGGL((trunk_ws_fwd_kernel<H, NH, RT, true>), grid, dim3(kWsThreads), 0, st, c);
  } else {
    hipLaunchKernelGGL((trunk_ws_fwd_kernel<H, NH, RT, false>), grid, dim3(kWsThreads), 0, st, c);
  }
  return mippo::check_launch("mi_mlp_ws_fwd_bf16");
}

}  // namespace

#ifdef MIPPO_TRACE
extern "C" int mi_debug_ws_trace(unsigned long long* host_out, int64_t n) {
  int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ws_trace), (size_t)n * 8);
  void* p = nullptr;
  if (rc == 0 && hipGetSymbolAddress(&p, HIP_SYMBOL(g_ws_trace)) == hipSuccess)
    rc = (int)hipMemset(p, 0, sizeof(g_ws_trace));
  return rc;
}
#endif

// 1 if (dims, acts) is a trunk the weights-stationary kernels take.
extern "C" int mi_mlp_ws_supported(int64_t L, const int64_t* dims, const int64_t* acts) {
  if (!dims || !acts || L < 2 || L > WS_MAXL) return 0;
  const int64_t H = dims[1];
  if (dims[0] < 1 || dims[0] > 32 || dims[L] < 1 || dims[L] > 16) return 0;
  for (int64_t l = 1; l < L; ++l)
    if (dims[l] != H) return 0;
  for (int64_t l = 0; l + 1 < L; ++l)
    if (acts[l] != MI_ACT_RELU) return 0;
  if (acts[L - 1] != MI_ACT_NONE) return 0;
  const int64_t NH = L - 2;
  return (H == 256 && NH <= 1) || (H == 128 && NH <= 2) || (H == 64 && NH <= 3);
}

namespace {

int ws_fill(WsChain& c, const char* who, const float* x, int64_t M, int64_t L,
            const void* const* wt_bf, const float* const* bias, const int64_t* dims,
            const int64_t* acts, float* out, void* const* y_bf, void* x_bf,
            void* const* relu_mask = nullptr) {
  MI_REQUIRE(x && wt_bf && dims && acts && out, "%s: null pointer", who);
  MI_REQUIRE(mi_mlp_ws_supported(L, dims, acts),
             "%s: trunk outside the weights-stationary shape class", who);
  c = {};
  c.x = x;
  c.M_head = M;
  c.x_bf = static_cast<bf16_t*>(x_bf);
  c.ldx = mippo::ceil_div(dims[0], 8) * 8;
  c.out = out;
  c.M = M;
  c.K0 = (int)dims[0];
  c.N_out = (int)dims[L];
  for (int64_t l = 0; l < L; ++l) {
    MI_REQUIRE(wt_bf[l] && al16(wt_bf[l]), "%s: weights must be 16-byte aligned", who);
    c.layer[l].w = static_cast<const bf16_t*>(wt_bf[l]);
    c.layer[l].bias = bias ? bias[l] : nullptr;
    MI_REQUIRE(!c.layer[l].bias || (reinterpret_cast<uintptr_t>(c.layer[l].bias) & 15) == 0 ||
                   l == L - 1,
               "%s: hidden biases must be 16-byte aligned", who);
    c.layer[l].out_bf = (y_bf && l + 1 < L) ? static_cast<bf16_t*>(y_bf[l]) : nullptr;
    c.layer[l].mask_out =
        (relu_mask && l + 1 < L) ? static_cast<unsigned char*>(relu_mask[l]) : nullptr;
    c.layer[l].ldo = mippo::ceil_div(dims[l + 1], 8) * 8;
    MI_REQUIRE(al16(c.layer[l].out_bf), "%s: outputs must be 16-byte aligned", who);
  }
  return 0;
}

int ws_dispatch(const WsChain& c, int64_t H, int64_t NH, hipStream_t st) {
  // RT = 4 (64-row tiles) once every CU has at least one of them, else 32-row tiles
  // (MIPPO_WS_RT=2|4 overrides: tuning aid)
  static const int rt_override = [] {
    const char* e = getenv("MIPPO_WS_RT");
    return e ? atoi(e) : 0;
  }();
  const bool big = rt_override ? rt_override == 4 : c.M >= 64 * (int64_t)ws_grid(1 << 30);
#define WS_CASE(h, nh)                                                        \
  if (H == h && NH == nh)                                                     \
    return big ? ws_launch<h, nh, 4>(c, st) : ws_launch<h, nh, 2>(c, st);
  WS_CASE(256, 0)
  WS_CASE(256, 1)
  WS_CASE(128, 0)
  WS_CASE(128, 1)
  WS_CASE(128, 2)
  WS_CASE(64, 0)
  WS_CASE(64, 1)
  WS_CASE(64, 2)
  WS_CASE(64, 3)
#undef WS_CASE
  MI_REQUIRE(false, "weights-stationary trunk: no instantiation for H=%lld NH=%lld",
             (long long)H, (long long)NH);
}

}  // namespace

extern "C" int mi_mlp_ws_fwd_bf16(const float* x, int64_t M, int64_t L, const void* const* wt_bf,
                                  const float* const* bias, const int64_t* dims,
                                  const int64_t* acts, float* out, void* const* y_bf,
                                  void* x_bf, mi_stream_t stream) {
  MI_REQUIRE(M >= 0, "mi_mlp_ws_fwd_bf16: bad M");
  if (M == 0) return 0;
  WsChain c;
  int rc = ws_fill(c, "mi_mlp_ws_fwd_bf16", x, M, L, wt_bf, bias, dims, acts, out, y_bf, x_bf);
  if (rc) return rc;
  return ws_dispatch(c, dims[1], L - 2, mippo::as_stream(stream));
}

namespace {

template <int H, int NH, int RT>
int ws_bwd_launch(const WsBwdChain& c, hipStream_t st) {
  const int64_t ntiles = mippo::ceil_div(c.M, 16 * RT);
  const dim3 grid((unsigned)ws_grid(ntiles));
  if (c.sbwd.A > 0)
    hipLaunchKernelGGL((trunk_ws_bwd_kernel<H, NH, RT, true>), grid, dim3(kWsThreads), 0, st, c);
  else
    hipLaunchKernelGGL((trunk_ws_bwd_kernel<H, NH, RT, false>), grid, dim3(kWsThreads), 0, st, c);
  return mippo::check_launch("mi_mlp_ws_bwd_dx_bf16");
}

int ws_bwd_dispatch(const WsBwdChain& c, int64_t H, int64_t NH, hipStream_t st) {
  static const int rt_override = [] {
    const char* e = getenv("MIPPO_WS_RT");
    return e ? atoi(e) : 0;
  }();
  const bool big = rt_override ? rt_override == 4 : c.M >= 64 * (int64_t)ws_grid(1 << 30);
#define WS_CASE(h, nh)                                                        \
  if (H == h && NH == nh)                                                     \
    return big ? ws_bwd_launch<h, nh, 4>(c, st) : ws_bwd_launch<h, nh, 2>(c, st);
  WS_CASE(256, 0)
  WS_CASE(256, 1)
  WS_CASE(128, 0)
  WS_CASE(128, 1)
  WS_CASE(128, 2)
  WS_CASE(64, 0)
  WS_CASE(64, 1)
  WS_CASE(64, 2)
  WS_CASE(64, 3)
#undef WS_CASE
  MI_REQUIRE(false, "weights-stationary trunk backward: no instantiation for H=%lld NH=%lld",
             (long long)H, (long long)NH);
}

// Fills the transposed chain from the C-ABI arrays of mi_mlp_bwd_dx_bf16 (w_bf[l]: backward
// image of layer l; aux[l] = y_l, dz_bf[l] = dz_l for l < L - 1; dz_last = dz_{L-1}).
int ws_bwd_fill(WsBwdChain& c, const char* who, const float* g_out, int64_t M, int64_t L,
                const void* const* w_bf, const int64_t* dims, const int64_t* acts,
                const void* const* aux, void* dz_last, void* const* dz_bf,
                const void* const* relu_mask = nullptr) {
  MI_REQUIRE(w_bf && dims && acts && aux && dz_last && dz_bf, "%s: null pointer", who);
  MI_REQUIRE(mi_mlp_ws_supported(L, dims, acts),
             "%s: trunk outside the weights-stationary shape class", who);
  c = {};
  c.g_out = g_out;
  c.dz_last = static_cast<bf16_t*>(dz_last);
  c.ldx = mippo::ceil_div(dims[L], 8) * 8;
  c.M = M;
  c.N_out = (int)dims[L];
  MI_REQUIRE(al16(dz_last), "%s: buffers must be 16-byte aligned", who);
  // chain position q handles layer l = L-1-q: emits dz_{l-1}, masks by relu'(y_{l-1})
  for (int64_t q = 0; q + 1 < L; ++q) {
    const int64_t l = L - 1 - q;
    MI_REQUIRE(w_bf[l] && al16(w_bf[l]) && aux[l - 1] && dz_bf[l - 1] && al16(aux[l - 1]) &&
                   al16(dz_bf[l - 1]),
               "%s: layer %lld operands missing or misaligned", who, (long long)l);
    c.layer[q].w = static_cast<const bf16_t*>(w_bf[l]);
    c.layer[q].aux = static_cast<const bf16_t*>(aux[l - 1]);
    c.layer[q].out_bf = static_cast<bf16_t*>(dz_bf[l - 1]);
    c.layer[q].ld = mippo::ceil_div(dims[l], 8) * 8;
    c.layer[q].mask = relu_mask ? static_cast<const unsigned char*>(relu_mask[l - 1]) : nullptr;
    MI_REQUIRE(!relu_mask || c.layer[q].mask, "%s: mask of layer %lld missing", who,
               (long long)(l - 1));
  }
  return 0;
}

}  // namespace

// Weights-stationary form of mi_mlp_bwd_dx_bf16 without an input gradient (act_last none):
// same operands, same results bit for bit.
extern "C" int mi_mlp_ws_bwd_dx_bf16(const float* g_out, int64_t M, int64_t L,
                                     const void* const* w_bf, const int64_t* dims,
                                     const int64_t* acts, const void* const* aux,
                                     void* dz_last, void* const* dz_bf, mi_stream_t stream) {
  MI_REQUIRE(M >= 0, "mi_mlp_ws_bwd_dx_bf16: bad M");
  if (M == 0) return 0;
  MI_REQUIRE(g_out, "mi_mlp_ws_bwd_dx_bf16: null pointer");
  WsBwdChain c;
  int rc = ws_bwd_fill(c, "mi_mlp_ws_bwd_dx_bf16", g_out, M, L, w_bf, dims, acts, aux, dz_last,
                       dz_bf);
  if (rc) return rc;
  return ws_bwd_dispatch(c, dims[1], L - 2, mippo::as_stream(stream));
}

// mi_policy_bwd_bf16 on the weights-stationary kernels: the action trunk's transposed chain
// fed by the sampler backward, and the value trunk's fed by g_value — two launches.
namespace {

// The CUs of a two-trunk launch, split in proportion to the tile counts (all of them when
// both trunks' tiles fit the chip).  MIPPO_WS_DUAL_SPLIT = percent for the value trunk.
void ws_dual_split(int64_t tv, int64_t ta, int64_t* nv, int64_t* na, int pct = 0) {
  static const int env_pct = [] {
    const char* e = getenv("MIPPO_WS_DUAL_SPLIT");
    return e ? atoi(e) : 0;
  }();
  const int split_pct = pct > 0 ? pct : env_pct;
  const int64_t cus = ws_grid(1 << 30);
  *nv = tv;
  *na = ta;
  if (tv + ta > cus) {
    *nv = split_pct > 0 ? cus * split_pct / 100 : cus * tv / (tv + ta);
    if (*nv < 1) *nv = 1;
    if (*nv > cus - 1) *nv = cus - 1;
    *na = cus - *nv;
    if (*nv > tv) *nv = tv;
    if (*na > ta) *na = ta;
  }
}

template <int RT>
int ws_bwd_dual_launch_rt(const WsBwdChain& a, const WsBwdChain& v, int64_t hv, int64_t nhv,
                          int64_t ha, int64_t nha, hipStream_t st) {
  int64_t nv, na;
  ws_dual_split(mippo::ceil_div(v.M, 16 * RT), mippo::ceil_div(a.M, 16 * RT), &nv, &na);
#define X(p, q, r, s_)                                                                       \
  if (hv == p && nhv == q && ha == r && nha == s_) {                                         \
    if (a.layer[0].mask)                                                                     \
      hipLaunchKernelGGL((policy_ws_bwd_dual_kernel<p, q, r, s_, RT, true>),                 \
                         dim3((unsigned)(nv + na)), dim3(kWsThreads), 0, st, a, v, (int)nv); \
    else                                                                                     \
      hipLaunchKernelGGL((policy_ws_bwd_dual_kernel<p, q, r, s_, RT, false>),                \
                         dim3((unsigned)(nv + na)), dim3(kWsThreads), 0, st, a, v, (int)nv); \
    return mippo::check_launch("mi_policy_ws_bwd_bf16(one launch)");                         \
  }
  WS_DUAL_MENU(X)
#undef X
  MI_REQUIRE(false, "mi_policy_ws_bwd_bf16: no one-launch instantiation for these trunks");
}

}  // namespace

extern "C" int mi_policy_ws_bwd_bf16(
    const float* mean_and_std, const float* extras, const uint64_t* rng_state,
    uint64_t offset_add, const float* eps2, const float* g_loglik, float g_reg, float min_std,
    float std_scale, float entropy_weight, const float* g_value, int64_t M, int64_t La,
    const void* const* a_w, const int64_t* a_dims, const int64_t* a_acts,
    const void* const* a_aux, void* a_dz_last, void* const* a_dz_bf, int64_t Lc,
    const void* const* c_w, const int64_t* c_dims, const int64_t* c_acts,
    const void* const* c_aux, void* c_dz_last, void* const* c_dz_bf,
    const void* const* a_mask, const void* const* c_mask, mi_stream_t stream) {
  MI_REQUIRE(M >= 0, "mi_policy_ws_bwd_bf16: bad M");
  MI_REQUIRE((a_mask == nullptr) == (c_mask == nullptr),
             "mi_policy_ws_bwd_bf16: masks for both trunks or for neither");
  if (M == 0) return 0;
  MI_REQUIRE(mean_and_std && extras && g_value && a_dims && c_dims && a_acts && c_acts,
             "mi_policy_ws_bwd_bf16: null pointer");
  MI_REQUIRE(rng_state || eps2, "mi_policy_ws_bwd_bf16: need rng_state or injected eps2");
  MI_REQUIRE(mi_policy_ws_supported(La, a_dims, a_acts, Lc, c_dims, c_acts),
             "mi_policy_ws_bwd_bf16: trunks outside the weights-stationary shape class");
  hipStream_t st = mippo::as_stream(stream);
  WsBwdChain a;
  int rc = ws_bwd_fill(a, "mi_policy_ws_bwd_bf16(action)", nullptr, M, La, a_w, a_dims, a_acts,
                       a_aux, a_dz_last, a_dz_bf, a_mask);
  if (rc) return rc;
  const int64_t A2 = a_dims[La];
  a.sbwd = {mean_and_std, extras, {rng_state, offset_add, eps2, eps2}, g_loglik, g_reg,
            (int)(A2 / 2), min_std, std_scale, entropy_weight};
  WsBwdChain v;
  rc = ws_bwd_fill(v, "mi_policy_ws_bwd_bf16(value)", g_value, M, Lc, c_w, c_dims, c_acts, c_aux,
                   c_dz_last, c_dz_bf, c_mask);
  if (rc) return rc;
  static const bool two_launches = [] {  // MIPPO_WS_BWD_DUAL=0: one launch per trunk (A/B)
    const char* e = getenv("MIPPO_WS_BWD_DUAL");
    return e && e[0] == '0';
  }();
  if (!two_launches && ws_dual_has(c_dims[1], Lc - 2, a_dims[1], La - 2)) {
    const bool one_each = 2 * mippo::ceil_div(M, 32) <= ws_grid(1 << 30);
    return one_each ? ws_bwd_dual_launch_rt<2>(a, v, c_dims[1], Lc - 2, a_dims[1], La - 2, st)
                    : ws_bwd_dual_launch_rt<4>(a, v, c_dims[1], Lc - 2, a_dims[1], La - 2, st);
  }
  rc = ws_bwd_dispatch(a, a_dims[1], La - 2, st);
  if (rc) return rc;
  return ws_bwd_dispatch(v, c_dims[1], Lc - 2, st);
}

// ---- mi_policy_ws_bwd_gae_bf16: mi_gae_ppo_loss_f32 + mi_policy_ws_bwd_bf16 in one launch ----
namespace {
// percent of the CUs for the value trunk in the launch with the GAE inside (0: in proportion
// to the tile counts, as the plain launches); MIPPO_WS_GAE_SPLIT: tuning aid
int ws_gae_split_pct() {
  static const int pct = [] {
    const char* e = getenv("MIPPO_WS_GAE_SPLIT");
    return e ? atoi(e) : 0;
  }();
  return pct;
}
}  // namespace

extern "C" int64_t mi_policy_ws_bwd_gae_workspace_bytes(int64_t M) {
  return kGaeHeaderBytes + mippo::ceil_div(M, 64) * 4 * (int64_t)sizeof(double);
}

// 1 if a [T, B] minibatch on these trunks can take the launch with the GAE inside: the
// one-launch menu at 64-row tiles, a scalar value head, whole (step, env group) row tiles, and
// no workgroup with more than kGaeMaxQ of them.
extern "C" int mi_policy_ws_bwd_gae_supported(int64_t T, int64_t B, int64_t La,
                                              const int64_t* a_dims, const int64_t* a_acts,
                                              int64_t Lc, const int64_t* c_dims,
                                              const int64_t* c_acts) {
  if (!a_dims || !a_acts || !c_dims || !c_acts || La < 2 || Lc < 2) return 0;
  if (T < 1 || T > kGaeMaxT || B < 64 || B % 64 || B > 64 * kGaeMaxGroups) return 0;
  if (!mi_policy_ws_supported(La, a_dims, a_acts, Lc, c_dims, c_acts)) return 0;
  if (c_dims[Lc] != 1 || !ws_dual_has(c_dims[1], Lc - 2, a_dims[1], La - 2)) return 0;
  const int64_t M = T * B;
  if (2 * mippo::ceil_div(M, 32) <= ws_grid(1 << 30)) return 0;  // 32-row-tile sizes
  int64_t nv, na;
  ws_dual_split(M / 64, M / 64, &nv, &na, ws_gae_split_pct());
  return na >= B / 64 &&  // one publishing workgroup per env group
         mippo::ceil_div(M / 64, nv) <= kGaeMaxQ && mippo::ceil_div(M / 64, na) <= kGaeMaxQ;
}

extern "C" int mi_policy_ws_bwd_gae_bf16(
    const float* mean_and_std, const float* extras, const uint64_t* rng_state,
    uint64_t offset_add, const float* eps2, float g_reg, float min_std, float std_scale,
    float entropy_weight, const float* rewards, const float* values, const float* last_value,
    const uint8_t* done, const uint8_t* truncated, const float* ll_new, const float* ll_old,
    const float* reg, float gamma, float lambda, int normalize, float clip_range,
    float critic_weight, float* loss_out, double* partials_out, void* workspace, int64_t T,
    int64_t B, int64_t La,
    const void* const* a_w, const int64_t* a_dims, const int64_t* a_acts,
    const void* const* a_aux, void* a_dz_last, void* const* a_dz_bf, int64_t Lc,
    const void* const* c_w, const int64_t* c_dims, const int64_t* c_acts,
    const void* const* c_aux, void* c_dz_last, void* const* c_dz_bf,
    const void* const* a_mask, const void* const* c_mask, void* comm, mi_stream_t stream) {
  MI_REQUIRE(mean_and_std && extras && rewards && values && last_value && done && truncated &&
                 ll_new && ll_old && workspace && a_mask && c_mask,
             "mi_policy_ws_bwd_gae_bf16: null pointer");
  mippo_comm::CommDev cd = {};
  int world = 1;
  if (comm) {  // env-sharded: the advantage statistics cross the ranks inside the launch
    MI_REQUIRE(mippo_comm::dev_view_of(comm, &cd),
               "mi_policy_ws_bwd_gae_bf16: communicator not connected (mi_comm_connect)");
    world = cd.world;
    MI_REQUIRE(world >= 1 && world <= mippo_comm::kMaxWorld && B / 64 <= cd.chunks &&
                   (B / 64) * 16 <= cd.slot_bytes,
               "mi_policy_ws_bwd_gae_bf16: %lld env groups do not fit the communicator's slots",
               (long long)(B / 64));
  }
  MI_REQUIRE((loss_out != nullptr) != (partials_out != nullptr),
             "mi_policy_ws_bwd_gae_bf16: exactly one of loss_out (summed in the launch) and "
             "partials_out (deferred: mi_policy_loss_finalize_f32)");
  MI_REQUIRE(rng_state || eps2, "mi_policy_ws_bwd_gae_bf16: need rng_state or injected eps2");
  MI_REQUIRE(mi_policy_ws_bwd_gae_supported(T, B, La, a_dims, a_acts, Lc, c_dims, c_acts),
             "mi_policy_ws_bwd_gae_bf16: [T=%lld, B=%lld] on these trunks is outside the fused "
             "class (mi_policy_ws_bwd_gae_supported)", (long long)T, (long long)B);
  MI_REQUIRE(al16(rewards) && al16(values) && al16(done) && al16(truncated),
             "mi_policy_ws_bwd_gae_bf16: rewards / values / done / truncated must be 16-byte "
             "aligned");
  const int64_t M = T * B;
  hipStream_t st = mippo::as_stream(stream);
  WsBwdChain a;
  int rc = ws_bwd_fill(a, "mi_policy_ws_bwd_gae_bf16(action)", nullptr, M, La, a_w, a_dims,
                       a_acts, a_aux, a_dz_last, a_dz_bf, a_mask);
  if (rc) return rc;
  const int64_t A2 = a_dims[La];
  a.sbwd = {mean_and_std, extras, {rng_state, offset_add, eps2, eps2}, nullptr, g_reg,
            (int)(A2 / 2), min_std, std_scale, entropy_weight};
  WsBwdChain v;
  rc = ws_bwd_fill(v, "mi_policy_ws_bwd_gae_bf16(value)", nullptr, M, Lc, c_w, c_dims, c_acts,
                   c_aux, c_dz_last, c_dz_bf, c_mask);
  if (rc) return rc;
  WsGae g = {rewards, values, last_value, done, truncated, ll_new, ll_old, reg, loss_out,
             partials_out ? partials_out
                          : reinterpret_cast<double*>(static_cast<char*>(workspace) +
                                                      kGaeHeaderBytes),
             reinterpret_cast<double*>(static_cast<char*>(workspace) + 64),
             static_cast<unsigned int*>(workspace), static_cast<unsigned int*>(workspace) + 1,
             (int)T, (int)B, gamma, lambda, clip_range, critic_weight, normalize, cd, world};
  int64_t nv, na;
  ws_dual_split(M / 64, M / 64, &nv, &na, ws_gae_split_pct());
  const int64_t hv = c_dims[1], nhv = Lc - 2, ha = a_dims[1], nha = La - 2;
#define X(p, q, r, s_)                                                                     \
  if (hv == p && nhv == q && ha == r && nha == s_) {                                       \
    hipLaunchKernelGGL((policy_ws_bwd_gae_kernel<p, q, r, s_>), dim3((unsigned)(nv + na)), \
                       dim3(kWsThreads), 0, st, a, v, g, (int)nv);                         \
    return mippo::check_launch("mi_policy_ws_bwd_gae_bf16");                               \
  }
  WS_DUAL_MENU(X)
#undef X
  MI_REQUIRE(false, "mi_policy_ws_bwd_gae_bf16: no instantiation for these trunks");
}

extern "C" int mi_policy_loss_finalize_f32(int64_t n, const void* const* partials,
                                           const int64_t* n_partials, const int64_t* n_elements,
                                           float* const* loss_out, mi_stream_t stream) {
  MI_REQUIRE(n >= 0 && n <= kLossFinalMax, "mi_policy_loss_finalize_f32: 0 <= n <= %d",
             kLossFinalMax);
  if (n == 0) return 0;
  MI_REQUIRE(partials && n_partials && n_elements && loss_out,
             "mi_policy_loss_finalize_f32: null pointer");
  LossFinal f = {};
  for (int64_t s = 0; s < n; ++s) {
    MI_REQUIRE(partials[s] && loss_out[s] && n_partials[s] >= 1 && n_elements[s] >= 1,
               "mi_policy_loss_finalize_f32: bad entry %lld", (long long)s);
    f.part[s] = static_cast<const double*>(partials[s]);
    f.out[s] = loss_out[s];
    f.ntiles[s] = n_partials[s];
    f.dn[s] = (double)n_elements[s];
  }
  hipLaunchKernelGGL(policy_loss_finalize_kernel, dim3((unsigned)n), dim3(64), 0,
                     mippo::as_stream(stream), f);
  return mippo::check_launch("mi_policy_loss_finalize_f32");
}

// 1 if both trunks of a policy step are in the weights-stationary shape class (and the
// sampler's rows fit the LDS stash: 2A <= 64).
extern "C" int mi_policy_ws_supported(int64_t La, const int64_t* a_dims, const int64_t* a_acts,
                                      int64_t Lc, const int64_t* c_dims, const int64_t* c_acts) {
  return mi_mlp_ws_supported(La, a_dims, a_acts) && mi_mlp_ws_supported(Lc, c_dims, c_acts) &&
         a_dims[0] == c_dims[0] && a_dims[La] % 2 == 0;
}

namespace {

// Both trunks in one launch.  When the tiles of both fit the chip every workgroup has
// exactly one (32-row tiles: rollout sizes); otherwise the CUs are split in proportion to
// the tile counts (MIPPO_WS_DUAL_SPLIT = percent of the CUs for the value trunk: tuning aid)
// and a workgroup walks its trunk's tiles (64-row tiles unless MIPPO_WS_DUAL_RT says 2).
template <int RT>
int ws_dual_launch_rt(const WsChain& a, const WsChain& v, int64_t hv, int64_t nhv, int64_t ha,
                      int64_t nha, hipStream_t st) {
  MI_REQUIRE(16 * RT * a.N_out <= 4096, "mi_policy_ws_fwd_bf16: 2A = %d is too wide for the "
             "sampler stash", a.N_out);
  int64_t nv, na;
  ws_dual_split(mippo::ceil_div(v.M, 16 * RT), mippo::ceil_div(a.M, 16 * RT), &nv, &na);
  // every row tile whole, every image and mask asked for: the guard-free instantiation
  auto whole = [](const WsChain& c, int L) {
    if (c.M % (16 * RT) || c.M_head % (16 * RT) || !c.x_bf) return false;
    for (int l = 0; l + 1 < L; ++l)
      if (!c.layer[l].out_bf || !c.layer[l].mask_out) return false;
    return true;
  };
  const bool full = RT == 4 && whole(a, (int)nha + 2) && whole(v, (int)nhv + 2);
#define X(p, q, r, s_)                                                                      \
  if (hv == p && nhv == q && ha == r && nha == s_) {                                        \
    if (full)                                                                               \
      hipLaunchKernelGGL((policy_ws_dual_kernel<p, q, r, s_, RT, RT == 4>),                 \
                         dim3((unsigned)(nv + na)), dim3(kWsThreads), 0, st, a, v, (int)nv); \
    else                                                                                    \
      hipLaunchKernelGGL((policy_ws_dual_kernel<p, q, r, s_, RT, false>),                   \
                         dim3((unsigned)(nv + na)), dim3(kWsThreads), 0, st, a, v, (int)nv); \
    return mippo::check_launch("mi_policy_ws_fwd_bf16(one launch)");                        \
  }
  WS_DUAL_MENU(X)
#undef X
  MI_REQUIRE(false, "mi_policy_ws_fwd_bf16: no one-launch instantiation for these trunks");
}

int ws_dual_launch(const WsChain& a, const WsChain& v, int64_t hv, int64_t nhv, int64_t ha,
                   int64_t nha, hipStream_t st) {
  static const int rt_override = [] {
    const char* e = getenv("MIPPO_WS_DUAL_RT");
    return e ? atoi(e) : 0;
  }();
  const int64_t cus = ws_grid(1 << 30);
  const bool one_each = mippo::ceil_div(v.M, 32) + mippo::ceil_div(a.M, 32) <= cus;
  const int rt = rt_override == 2 || rt_override == 4 ? rt_override : (one_each ? 2 : 4);
  if (rt == 2 || 64 * a.N_out > 4096) return ws_dual_launch_rt<2>(a, v, hv, nhv, ha, nha, st);
  return ws_dual_launch_rt<4>(a, v, hv, nhv, ha, nha, st);
}

}  // namespace

// ---- the recurrent policy's rollout step ---------------------------------------------------
namespace {

// (value trunk) x (GRU width) pairs the one-launch recurrent step is instantiated for
#define GRU_STEP_MENU(X) \
  X(256, 1, 64) X(256, 1, 128) X(256, 0, 64) X(128, 1, 64) X(128, 1, 128) X(64, 1, 64)

bool gru_step_has(int64_t hv, int64_t nhv, int64_t h) {
#define X(a, b, c) if (hv == a && nhv == b && h == c) return true;
  GRU_STEP_MENU(X)
#undef X
  return false;
}

}  // namespace

// 1 if mi_gru_policy_step_bf16 takes this network: observation width <= 32, GRU width 64 or
// 128 (Dense_in: K0 -> H relu; head: H -> 2A <= 16 columns), value trunk in the
// weights-stationary shape class, and the pair instantiated.
extern "C" int mi_gru_policy_step_supported(int64_t K0, int64_t H, int64_t A2, int64_t Lc,
                                            const int64_t* c_dims, const int64_t* c_acts) {
  if (K0 < 1 || K0 > 32 || A2 < 2 || A2 > 16 || (A2 & 1)) return 0;
  if (!mi_mlp_ws_supported(Lc, c_dims, c_acts) || c_dims[0] != K0) return 0;
  return gru_step_has(c_dims[1], Lc - 2, H) ? 1 : 0;
}

// One rollout / evaluation step of make_gru_actor_critic's network (networks/factories.py;
// recurrent contract networks/recurrent.py:89-161): normaliser -> Dense(K0 -> H, relu) ->
// GRU(H -> H) -> Dense(H -> 2A) -> NormalTanhSampler beside the value trunk, ONE launch
// (gru_policy_ws_kernel).  w_in / w_proj / w_out: forward fragment-major images of the three
// Dense-like kernels (the GRU's input projection [H, 3H] with columns r | z | n); w_h: the
// fp32 recurrent kernel [H, 3H]; h_in / h_out: the carry [M, H] (h_out is also the GRU's
// output).  Sampler arguments and outputs as mi_policy_fwd_bf16 (mean_and_std optional).
// Bit-identical to the generic containers' seven launches.
extern "C" int mi_gru_policy_step_bf16(
    const float* obs, int64_t M, int64_t K0, int64_t H, int64_t A2, const float* norm_mean,
    const float* norm_m2, const float* norm_count, float norm_eps, const void* w_in,
    const float* b_in, const void* w_proj, const float* b_proj, const float* w_h,
    const float* b_hn, const void* w_out, const float* b_out, const float* h_in, float* h_out,
    int64_t Lc, const void* const* c_w, const float* const* c_bias, const int64_t* c_dims,
    const int64_t* c_acts, const uint64_t* rng_state, uint64_t offset_add, const float* eps,
    const float* eps2, float min_std, float std_scale, float entropy_weight, int deterministic,
    float* mean_and_std, float* raw_out, float* action, float* loglik, float* reg,
    float* mu_out, float* sigma_out, float* value, mi_stream_t stream) {
  const char* who = "mi_gru_policy_step_bf16";
  MI_REQUIRE(M >= 0, "%s: bad M", who);
  if (M == 0) return 0;
  MI_REQUIRE(obs && w_in && w_proj && w_h && w_out && h_in && h_out && value && c_dims && c_acts,
             "%s: null pointer", who);
  MI_REQUIRE(mi_gru_policy_step_supported(K0, H, A2, Lc, c_dims, c_acts),
             "%s: network outside the supported class", who);
  MI_REQUIRE(!norm_mean || (norm_m2 && norm_count), "%s: incomplete normaliser", who);
  MI_REQUIRE(rng_state || (eps && eps2), "%s: need rng_state or both injected noises", who);
  MI_REQUIRE(al16(w_in) && al16(w_proj) && al16(w_out) && al16(h_in) && al16(h_out) &&
                 al16(b_in) && al16(b_proj) && al16(b_hn),
             "%s: buffers must be 16-byte aligned", who);
  hipStream_t st = mippo::as_stream(stream);
  WsChain a = {};
  a.x = obs;
  a.M = a.M_head = M;
  a.K0 = (int)K0;
  a.N_out = (int)A2;
  a.out = mean_and_std;
  a.layer[0].w = static_cast<const bf16_t*>(w_in);
  a.layer[0].bias = b_in;
  a.layer[1].w = static_cast<const bf16_t*>(w_proj);
  a.layer[1].bias = b_proj;
  a.layer[2].w = static_cast<const bf16_t*>(w_out);
  a.layer[2].bias = b_out;
  a.norm_mean = norm_mean;
  a.norm_m2 = norm_m2;
  a.norm_count = norm_count;
  a.norm_eps = norm_eps;
  a.samp = {nullptr, {rng_state, offset_add, eps, eps2}, raw_out, action, mu_out, sigma_out,
            loglik, reg, (int)(A2 / 2), min_std, std_scale, entropy_weight, deterministic};
  const GruStepExtra gx = {w_h, b_hn, h_in, h_out};
  WsChain v;
  int rc = ws_fill(v, who, obs, M, Lc, c_w, c_bias, c_dims, c_acts, value, nullptr, nullptr);
  if (rc) return rc;
  v.norm_mean = norm_mean;
  v.norm_m2 = norm_m2;
  v.norm_count = norm_count;
  v.norm_eps = norm_eps;
  constexpr int RT = 2;
  MI_REQUIRE(16 * RT * A2 <= 4096, "%s: 2A too wide for the sampler stash", who);
  int64_t nv, na;
  ws_dual_split(mippo::ceil_div(M, 16 * RT), mippo::ceil_div(M, 16 * RT), &nv, &na);
  const int64_t hv = c_dims[1], nhv = Lc - 2;
#define X(p, q, r)                                                                          \
  if (hv == p && nhv == q && H == r) {                                                      \
    hipLaunchKernelGGL((gru_policy_ws_kernel<p, q, r, RT>), dim3((unsigned)(nv + na)),      \
                       dim3(kWsThreads), 0, st, a, gx, v, (int)nv);                         \
    return mippo::check_launch(who);                                                        \
  }
  GRU_STEP_MENU(X)
#undef X
  MI_REQUIRE(false, "%s: no instantiation", who);
}

// 1 if mi_policy_ws_fwd_bf16 runs these two trunks as ONE launch (at any size; required
// below 8192 rows, where the one-launch-per-trunk form is not used).
extern "C" int mi_policy_ws_dual_supported(int64_t La, const int64_t* a_dims,
                                           const int64_t* a_acts, int64_t Lc,
                                           const int64_t* c_dims, const int64_t* c_acts) {
  if (!mi_policy_ws_supported(La, a_dims, a_acts, Lc, c_dims, c_acts)) return 0;
  return ws_dual_has(c_dims[1], Lc - 2, a_dims[1], La - 2) ? 1 : 0;
}

// mi_policy_fwd_bf16 on the weights-stationary kernels: the action trunk (normaliser in the
// input stage, sampler on its head rows) and the value trunk (normaliser, bootstrap tail
// rows) — one launch with the CUs shared between the trunks for the instantiated pairs
// (policy_ws_dual_kernel), else two launches of trunk_ws_fwd_kernel.  Same arguments, same
// results bit for bit.
extern "C" int mi_policy_ws_fwd_bf16(
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
  // relu trunks keep no pre-activations: a_pre_bf[l] / c_pre_bf[l], when given, receive the
  // relu' MASK of layer l's output instead (uint8 [ceil(rows / 64)][N_l / 16][64][4],
  // WsLayer::mask_out) — what mi_policy_ws_bwd_bf16 takes as a_mask / c_mask
  MI_REQUIRE(M >= 0 && M_tail >= 0 && (M_tail == 0 || value_tail_obs),
             "mi_policy_ws_fwd_bf16: bad M / tail");
  if (M == 0) return 0;
  MI_REQUIRE(obs && value && a_dims && c_dims && mean_and_std,
             "mi_policy_ws_fwd_bf16: null pointer (mean_and_std is required here)");
  MI_REQUIRE(mi_policy_ws_supported(La, a_dims, a_acts, Lc, c_dims, c_acts),
             "mi_policy_ws_fwd_bf16: trunks outside the weights-stationary shape class");
  MI_REQUIRE(!norm_mean || (norm_m2 && norm_count), "mi_policy_ws_fwd_bf16: incomplete normaliser");
  MI_REQUIRE(rng_state || (eps && eps2),
             "mi_policy_ws_fwd_bf16: need rng_state or both injected noises");
  hipStream_t st = mippo::as_stream(stream);
  WsChain a;
  int rc = ws_fill(a, "mi_policy_ws_fwd_bf16(action)", obs, M, La, a_w, a_bias, a_dims, a_acts,
                   mean_and_std, a_y_bf, a_x_bf, a_pre_bf);
  if (rc) return rc;
  const int64_t A2 = a_dims[La];
  a.norm_mean = norm_mean;
  a.norm_m2 = norm_m2;
  a.norm_count = norm_count;
  a.norm_eps = norm_eps;
  a.samp = {extras, {rng_state, offset_add, eps, eps2}, raw_out, action, mu_out, sigma_out,
            loglik, reg, (int)(A2 / 2), min_std, std_scale, entropy_weight, deterministic};
  WsChain v;
  rc = ws_fill(v, "mi_policy_ws_fwd_bf16(value)", obs, M + M_tail, Lc, c_w, c_bias, c_dims,
               c_acts, value, c_y_bf, c_x_bf, c_pre_bf);
  if (rc) return rc;
  v.x_tail = value_tail_obs;
  v.M_head = M;
  v.norm_mean = norm_mean;
  v.norm_m2 = norm_m2;
  v.norm_count = norm_count;
  v.norm_eps = norm_eps;
  // one launch for both trunks whenever the pair is instantiated (measured at C2's replay,
  // 31 744 + 30 720 rows: 2.24 -> 2.15 ms per iteration against one launch per trunk — the
  // second prologue, launch and tail run beside the other trunk's tiles);
  // MIPPO_WS_DUAL_MAX_ROWS caps it (A/B)
  static const int64_t dual_max_rows = [] {
    const char* e = getenv("MIPPO_WS_DUAL_MAX_ROWS");
    return e ? (int64_t)atoll(e) : (int64_t)1 << 40;
  }();
  if (M + M_tail <= dual_max_rows && ws_dual_has(c_dims[1], Lc - 2, a_dims[1], La - 2))
    return ws_dual_launch(a, v, c_dims[1], Lc - 2, a_dims[1], La - 2, st);
  rc = ws_dispatch(a, a_dims[1], La - 2, st);
  if (rc) return rc;
  return ws_dispatch(v, c_dims[1], Lc - 2, st);
}
