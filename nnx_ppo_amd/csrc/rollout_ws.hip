// a6 / a7 / a18 — the WHOLE rollout of a device-steppable env in ONE launch.
//
// `unroll_env` (nnx_ppo/algorithms/rollout.py:48-73) scans `single_transition` (rollout.py:11-45)
// T times: network forward -> env.step -> record the Transition -> reset-on-done select of env
// state and carry.  The stepwise form of this package is two launches per step (policy step:
// trunk_ws.hip policy_ws_dual_kernel; env step + EpisodeWrapper + reset select:
// misc.hip episode_select_kernel), i.e. 60 dependent launches of 5-7 us for T = 30 — 0.38 ms,
// a fifth of BASELINE C2's iteration — plus one launch for all T x N reset states and one to
// stack the per-step leaves.  But envs are row-independent (rollout.py:21,39 vmaps over them),
// so nothing in a rollout crosses rows: a workgroup can own a 32-row tile of envs for ALL T
// steps.  Here
//   * the trunk's weight fragments are loaded ONCE and stay in registers for the T steps
//     (the stepwise launch spends ~40 % of its life on that prologue, every step);
//   * the env state of the tile (MockEnv key + step count, the wrapper's step counter, the
//     observation) lives in LDS / registers between steps; the step, the wrapper's counter /
//     truncation / done arithmetic and the reset-on-done select are evaluated by the threads
//     that stage the next observation — between the input stage and the first barrier of the
//     trunk, i.e. beside the MFMAs, with no barrier of their own;
//   * the reset state of (step t, env n) is a function of the reset key alone
//     (rollout.py:57-59: keys[t][n] = split(reset_key, (T, N))[t][n]); it is derived where a
//     done flag asks for it instead of for all T x N up front;
//   * every Transition leaf is written straight into its [T, N, ...] buffer (rollout.py:61-66).
// As in policy_ws_dual_kernel, value-trunk workgroups and action-trunk (+ sampler) workgroups
// run side by side; both step the env of their tile (integer hashing: cheaper than handing the
// observation over), the action-trunk workgroup writes the env's Transition leaves and the
// final state.  That needs an env whose step does not read the action — true of the synthetic
// benchmark env (test_dummies/mock_env.py:25-63 ignores it), which is the only env with a
// device-side step so far.
//
// Arithmetic: ws_fwd_body's, MFMA for MFMA (same tiles, same k order, same epilogues, the same
// normaliser expression and sampler row function with the step's noise offset), and the env /
// wrapper / key expressions of keys.hip / misc.hip — every leaf is BIT-IDENTICAL to the stepwise
// rollout (tests/test_rollout_fused_gpu.py), so the dispatch in algorithms/rollout.py is
// invisible.
#include "keys_common.h"
#include "trunk_ws_fwd.h"

namespace {

using mippo_keys::kGolden;
using mippo_keys::kM2;
using mippo_keys::mix;

// EpisodeWrapper(MockEnv) — wrappers/episode_wrapper.py, envs/synthetic.py
struct RolloutEnv {
  // state at step 0 (read only)
  const int64_t* key;      // [N] MockEnv data["key"]
  const int64_t* count;    // [N] MockEnv data["step_count"]
  const int64_t* counter;  // [N] wrapper info["step_counter"]
  const float* obs;        // [N][K0]
  const int64_t* reset_key;  // device scalar: rng_key_for_env_reset (rollout.py:54)
  int64_t max_steps, max_len;
  int T;
  int64_t N;
  // Transition leaves (written by the action-trunk workgroups)
  float* obs_seq;       // [T][N][K0]
  float* next_obs_seq;  // [T][N][K0]  (before the reset select, rollout.py:33)
  float* reward_seq;    // [T][N]
  uint8_t* done_seq;    // [T][N]
  uint8_t* trunc_seq;   // [T][N]
  // state after T steps, reset select applied (rollout.py:41-44)
  int64_t* key_out;
  int64_t* count_out;
  int64_t* counter_out;
  float* obs_out;     // [N][K0]
  float* reward_out;  // [N] select(done, reset.reward = 0, stepped.reward = 1)
};

template <int H, int RT, bool SAMP>
struct RolloutLds {
  using F = WsFwdLds<H, RT, SAMP>;
  static constexpr size_t key = (F::bytes + 15) / 16 * 16;
  static constexpr size_t count = key + 16 * RT * 8;
  static constexpr size_t counter = count + 16 * RT * 8;
  static constexpr size_t bytes = counter + 16 * RT * 8;
};

// One trunk of the policy for all T steps of the row tiles bid, bid + nblk, ...
// `c`: the chain as mi_policy_ws_fwd_bf16 fills it for ONE step, except that `c.x` is unused,
// `c.out` (value trunk: [T][N][N_out], may be null) and the sampler's output pointers
// ([T][N][A] / [T][N]) address step 0 and advance by N rows per step; `c.M` = N.
template <int H, int NH, int RT, bool SAMP>
__device__ __forceinline__ void ws_rollout_body(const WsChain& c, const RolloutEnv& env,
                                                const int bid, const int nblk,
                                                unsigned char* smem) {
  using G = WsGeom<H>;
  constexpr int CW = G::CW, RW = G::RW, TPW = G::TPW;
  static_assert(RT % RW == 0 && RT <= 8, "row tiles must split over the row groups");
  constexpr int RTW = RT / RW;
  constexpr int ROWS = 16 * RT;
  constexpr int KSH = H / 32;
  constexpr int AROW = H + 8;
  constexpr int XROW = 32 + 8;
  using Lds = WsFwdLds<H, RT, SAMP>;
  using RL = RolloutLds<H, RT, SAMP>;
  bf16_t* const bufX = reinterpret_cast<bf16_t*>(smem + Lds::bufX);
  bf16_t* const bufA = reinterpret_cast<bf16_t*>(smem + Lds::bufA);
  bf16_t* const bufB = reinterpret_cast<bf16_t*>(smem + Lds::bufB);
  int64_t* const s_key = reinterpret_cast<int64_t*>(smem + RL::key);        // [ROWS]
  int64_t* const s_count = reinterpret_cast<int64_t*>(smem + RL::count);
  int64_t* const s_counter = reinterpret_cast<int64_t*>(smem + RL::counter);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % CW, wr = wave / CW;
  const int li = lane & 15, lq = lane >> 4;
  const int64_t N = env.N;
  const int T = env.T;
  const int K0 = c.K0, N_out = c.N_out;
  const int64_t ntiles = (N + ROWS - 1) / ROWS;

  constexpr int IN_PT = (ROWS * 32 + kWsThreads - 1) / kWsThreads;  // K0 <= 32
  const int nel = ROWS * K0;
  const float rcpK0 = 1.0f / (float)K0;

  float* const s_mean = reinterpret_cast<float*>(smem + Lds::mean);
  float* const s_sd = reinterpret_cast<float*>(smem + Lds::sd);
  const bool norm = c.norm_mean != nullptr;
  if (norm && tid < K0) {  // frozen for the whole rollout (ppo.py:336 updates them afterwards)
    const float cnt = *c.norm_count;
    s_mean[tid] = c.norm_mean[tid];
    s_sd[tid] = cnt > 0.0f ? sqrtf(fmaxf(c.norm_m2[tid] / cnt, c.norm_eps)) : 10.0f;
  }
  const uint64_t reset_key = (uint64_t)*env.reset_key;
  const uint64_t span = env.max_len / 2 > 0 ? (uint64_t)(env.max_len / 2) : 0;

  // ---- the trunk, once for all tiles and steps (ws_fwd_body's prologue) --------------------
  bf16x8 W0[TPW];
  bf16x8 WH[NH > 0 ? NH : 1][TPW][KSH];
  f32x4 B0[TPW], BH[NH > 0 ? NH : 1][TPW], BO;
  bf16_t* const wo_s = reinterpret_cast<bf16_t*>(smem + Lds::wo);
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
  for (int row = tid >> 5; row < ROWS; row += kWsThreads >> 5)
    for (int k = K0 + (tid & 31); k < 32; k += 32) bufX[row * XROW + k] = (bf16_t)0.0f;

  float* const ms_s = reinterpret_cast<float*>(smem + Lds::stash);  // [SB][ROWS][N_out]
  // steps whose head rows share one sampler pass: one row per thread, 4096 stash floats
  int SB = kWsThreads / ROWS;
  if (SAMP && SB * ROWS * N_out > 4096) SB = 4096 / (ROWS * N_out);
  if (SB < 1) SB = 1;

  for (int64_t tile = bid; tile < ntiles; tile += nblk) {
    const int64_t i0 = tile * ROWS;
    // ---- the tile's env state at step 0 ---------------------------------------------------
    if (tid < ROWS) {
      const int64_t gi = i0 + tid < N ? i0 + tid : N - 1;  // clamped rows are never written out
      s_key[tid] = env.key[gi];
      s_count[tid] = env.count[gi];
      s_counter[tid] = env.counter[gi];
    }
    float xin[IN_PT];  // the raw observation elements this thread stages (element e = tid + 512 u)
#pragma unroll
    for (int u = 0; u < IN_PT; ++u) {
      const int e = tid + u * kWsThreads;
      const int row = (int)(((float)e + 0.5f) * rcpK0);
      const int64_t gi = i0 + row;
      xin[u] = (e < nel && gi < N) ? env.obs[gi * K0 + (e - row * K0)] : 0.0f;
    }
    __syncthreads();  // s_* (and, first tile, s_mean / s_sd and the pad columns)

    for (int t = 0; t < T; ++t) {
      const int64_t tN = (int64_t)t * N;
      // stage 0: the observation -> bf16 in bufX (normalizer.py:76-96); its raw value is the
      // Transition's `obs` (and the normaliser's rollout extras)
      // env step of the element's row, in registers (nothing of step t + 1 is published before
      // the barrier below: every thread still reads the state of step t)
      float xnext[IN_PT];
      int64_t nkey = 0, ncount = 0, ncounter = 0;
      bool owner = false, m_row = false;
      int owner_row = 0;
#pragma unroll
      for (int u = 0; u < IN_PT; ++u) {
        const int e = tid + u * kWsThreads;
        xnext[u] = 0.0f;
        if (e < nel) {
          const int row = (int)(((float)e + 0.5f) * rcpK0);
          const int k = e - row * K0;
          const int64_t gi = i0 + row;
          const bool live = gi < N;
          float v = xin[u];
          if (SAMP && live) env.obs_seq[(tN + gi) * K0 + k] = v;
          if (norm && live) v = (v - s_mean[k]) / s_sd[k];
          bufX[row * XROW + k] = (bf16_t)v;
          if (live) {
            const int64_t key = s_key[row];
            const int64_t step = s_count[row] + 1;               // MockEnv.step
            const bool d = step >= env.max_steps;
            const int64_t cw = s_counter[row] + 1;               // EpisodeWrapper.step
            const bool tr = cw >= env.max_len;
            const bool m = d || tr;
            float nobs = mippo_keys::mock_obs(key, step, k);
            if (SAMP) {
              env.next_obs_seq[(tN + gi) * K0 + k] = nobs;
              if (k == 0) {
                env.reward_seq[tN + gi] = 1.0f;
                env.done_seq[tN + gi] = m ? 1 : 0;
                env.trunc_seq[tN + gi] = tr ? 1 : 0;
              }
            }
            int64_t k2 = key, s2 = step, c2 = cw;
            if (m) {
              // env.reset(keys[t][gi]) — rollout.py:41-44,57-59; EpisodeWrapper.reset:
              // (base, counter key) = split(rng); MockEnv.reset(base); counter =
              // randint(counter key, 0, max_len // 2)
              const uint64_t rk = mix(reset_key + (uint64_t)(tN + gi + 1) * kGolden);
              const uint64_t base = mix(rk + kGolden);
              k2 = (int64_t)base;
              s2 = 0;
              nobs = mippo_keys::mock_obs(k2, 0, k);
              if (k == 0) {
                const uint64_t ck = mix(rk + 2 * kGolden);
                const uint64_t bits = mix(mix(ck) ^ kM2);
                c2 = span ? (int64_t)((bits >> 1) % span) : 0;
              }
            }
            xnext[u] = nobs;
            if (k == 0) {
              owner = true;
              owner_row = row;
              nkey = k2;
              ncount = s2;
              ncounter = c2;
              m_row = m;
            }
          }
        }
      }
      __syncthreads();  // bufX staged; every read of the step-t state is done
      if (owner) {
        s_key[owner_row] = nkey;
        s_count[owner_row] = ncount;
        s_counter[owner_row] = ncounter;
      }

      f32x4 acc[RTW][TPW];
      // ---- layer 0 ---------------------------------------------------------------------
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
      auto epilogue_hidden = [&](const f32x4(&bias)[TPW], bf16_t* nbuf) {
#pragma unroll
        for (int b = 0; b < TPW; ++b) {
          const int col = (wc + CW * b) * 16 + 4 * lq;
#pragma unroll
          for (int r = 0; r < RTW; ++r) {
            bf16x4 vo;
#pragma unroll
            for (int e = 0; e < 4; ++e) vo[e] = (bf16_t)fmaxf(acc[r][b][e] + bias[b][e], 0.0f);
            *reinterpret_cast<bf16x4*>(nbuf + ((wr * RTW + r) * 16 + li) * AROW + col) = vo;
          }
        }
      };
      epilogue_hidden(B0, bufB);
      __syncthreads();
      // ---- hidden layers ---------------------------------------------------------------
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
        epilogue_hidden(BH[l], nxt);
        __syncthreads();
        bf16_t* tmp = cur;
        cur = nxt;
        nxt = tmp;
      }
      // ---- head: wave w takes row tile w ------------------------------------------------
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
            if (!SAMP && gi < N) c.out[(tN + gi) * N_out + 4 * lq + e] = v;
            if (SAMP) ms_s[((t % SB) * ROWS + row) * N_out + 4 * lq + e] = v;
          }
        }
      }
      if (SAMP && ((t + 1) % SB == 0 || t == T - 1)) {
        // The sampler row function is a ~2 500-cycle chain of transcendentals for ONE thread
        // per row: run once per step it kept 32 of the 512 threads busy for a third of the
        // step.  Nothing downstream waits for an action (the env of this launch does not read
        // it), so the head's rows of SB steps wait in the stash and are sampled together.
        __syncthreads();  // the head's rows of this step are in the stash
        const int slot = tid / ROWS, row = tid % ROWS;
        const int ts = t - (t % SB) + slot;  // the step whose rows sit in `slot`
        if (slot < SB && ts <= t && i0 + row < N) {
          // the step's own noise offset and [N]-row output blocks (sampling_layers.py:82-147:
          // one `_next_offset()` per call of the stepwise rollout)
          const int64_t sN = (int64_t)ts * N;
          mippo_sampler::FwdParams p = c.samp;
          const int A = p.A;
          p.noise.offset_add += (uint64_t)ts;
          if (p.raw_out) p.raw_out += sN * A;
          if (p.action) p.action += sN * A;
          if (p.mu_out) p.mu_out += sN * A;
          if (p.sigma_out) p.sigma_out += sN * A;
          if (p.ll) p.ll += sN;
          if (p.reg) p.reg += sN;
          mippo_sampler::fwd_row(ms_s + (slot * ROWS + row) * N_out, i0 + row, p);
        }
      }
      __syncthreads();  // bufA / bufB / bufX (and a sampled stash) are free for the next step
#pragma unroll
      for (int u = 0; u < IN_PT; ++u) xin[u] = xnext[u];
      // ---- after the last step: the carried state (reset select applied) ----------------
      if (SAMP && t == T - 1) {
#pragma unroll
        for (int u = 0; u < IN_PT; ++u) {
          const int e = tid + u * kWsThreads;
          if (e < nel) {
            const int row = (int)(((float)e + 0.5f) * rcpK0);
            const int64_t gi = i0 + row;
            if (gi < N) env.obs_out[gi * K0 + (e - row * K0)] = xin[u];
          }
        }
        if (owner) {
          const int64_t gi = i0 + owner_row;
          env.key_out[gi] = nkey;
          env.count_out[gi] = ncount;
          env.counter_out[gi] = ncounter;
          env.reward_out[gi] = m_row ? 0.0f : 1.0f;
        }
      }
    }
  }
}

template <int HV, int NHV, int HA, int NHA, int RT>
__global__ void __launch_bounds__(kWsThreads, 2)
rollout_ws_kernel(WsChain a, WsChain v, RolloutEnv env, int n_value) {
  constexpr size_t nv = RolloutLds<HV, RT, false>::bytes, na = RolloutLds<HA, RT, true>::bytes;
  __shared__ __attribute__((aligned(16))) unsigned char smem[nv > na ? nv : na];
  if ((int)blockIdx.x < n_value)
    ws_rollout_body<HV, NHV, RT, false>(v, env, (int)blockIdx.x, n_value, smem);
  else
    ws_rollout_body<HA, NHA, RT, true>(a, env, (int)blockIdx.x - n_value,
                                       (int)gridDim.x - n_value, smem);
}

// ---- the recurrent actor of make_gru_actor_critic for all T steps -----------------------------
// trunk_ws.hip's ws_gru_step_body (normaliser -> Dense(K0 -> H, relu) -> GRU(H -> H) ->
// Dense(H -> 2A) -> sampler on one row tile, every weight in registers) with the time loop
// around it: the carry of the tile's envs stays in registers (fp32, the lane that owns a
// (row, unit) pair) and in LDS (its bf16 image, the recurrent product's operand) between
// steps, and the reset-on-done select of the carry (rollout.py:41-44 with GRU.reset_state:
// zeros) is applied where the env step's done flag says so.  Same operand roundings, k order
// and gate expressions as the stepwise launch: bit-identical.
struct GruRolloutExtra {
  const float* w_h;   // [H][3H] fp32 recurrent kernel
  const float* b_hn;  // [H]
  const float* h_in;  // [N][H] carry at step 0
  float* h_out;       // [N][H] carry after T steps (reset select applied)
};

template <int H, int RT>
struct GruRolloutLds {
  using F = WsFwdLds<H, RT, true>;
  static constexpr size_t key = (F::bytes + 15) / 16 * 16;
  static constexpr size_t count = key + 16 * RT * 8;
  static constexpr size_t counter = count + 16 * RT * 8;
  static constexpr size_t done = counter + 16 * RT * 8;
  static constexpr size_t bytes = done + 16 * RT * 4;
};

template <int H, int RT>
__device__ __forceinline__ void ws_gru_rollout_body(const WsChain& c, const GruRolloutExtra& gx,
                                                    const RolloutEnv& env, const int bid,
                                                    const int nblk, unsigned char* smem) {
#pragma clang fp contract(off)  // the gate expressions as written (gru_fwd_mfma_kernel)
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
  using RL = GruRolloutLds<H, RT>;
  bf16_t* const bufX = reinterpret_cast<bf16_t*>(smem + Lds::bufX);
  bf16_t* const bufA = reinterpret_cast<bf16_t*>(smem + Lds::bufA);  // h, then h'
  bf16_t* const bufB = reinterpret_cast<bf16_t*>(smem + Lds::bufB);  // Dense-in output
  int64_t* const s_key = reinterpret_cast<int64_t*>(smem + RL::key);
  int64_t* const s_count = reinterpret_cast<int64_t*>(smem + RL::count);
  int64_t* const s_counter = reinterpret_cast<int64_t*>(smem + RL::counter);
  int* const s_done = reinterpret_cast<int*>(smem + RL::done);       // [ROWS] this step's flag
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % CW, wr = wave / CW;
  const int li = lane & 15, lq = lane >> 4;
  const int64_t N = env.N;
  const int T = env.T;
  const int K0 = c.K0, N_out = c.N_out;
  const int64_t ntiles = (N + ROWS - 1) / ROWS;
  constexpr int IN_PT = (ROWS * 32 + kWsThreads - 1) / kWsThreads;
  constexpr int H_CH = ROWS * (H / 4);
  constexpr int H_PT = (H_CH + kWsThreads - 1) / kWsThreads;
  const int nel = ROWS * K0;
  const float rcpK0 = 1.0f / (float)K0;

  float* const s_mean = reinterpret_cast<float*>(smem + Lds::mean);
  float* const s_sd = reinterpret_cast<float*>(smem + Lds::sd);
  const bool norm = c.norm_mean != nullptr;
  if (norm && tid < K0) {
    const float cnt = *c.norm_count;
    s_mean[tid] = c.norm_mean[tid];
    s_sd[tid] = cnt > 0.0f ? sqrtf(fmaxf(c.norm_m2[tid] / cnt, c.norm_eps)) : 10.0f;
  }
  const uint64_t reset_key = (uint64_t)*env.reset_key;
  const uint64_t span = env.max_len / 2 > 0 ? (uint64_t)(env.max_len / 2) : 0;

  // ---- the actor, once (ws_gru_step_body's prologue) ------------------------------------------
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
  float* const ms_s = reinterpret_cast<float*>(smem + Lds::stash);  // [SB][ROWS][N_out]
  int SB = kWsThreads / ROWS;
  if (SB * ROWS * N_out > 4096) SB = 4096 / (ROWS * N_out);
  if (SB < 1) SB = 1;

  for (int64_t tile = bid; tile < ntiles; tile += nblk) {
    const int64_t i0 = tile * ROWS;
    if (tid < ROWS) {
      const int64_t gi = i0 + tid < N ? i0 + tid : N - 1;
      s_key[tid] = env.key[gi];
      s_count[tid] = env.count[gi];
      s_counter[tid] = env.counter[gi];
      s_done[tid] = 0;  // rows beyond N never get an owner
    }
    float xin[IN_PT];
#pragma unroll
    for (int u = 0; u < IN_PT; ++u) {
      const int e = tid + u * kWsThreads;
      const int row = (int)(((float)e + 0.5f) * rcpK0);
      const int64_t gi = i0 + row;
      xin[u] = (e < nel && gi < N) ? env.obs[gi * K0 + (e - row * K0)] : 0.0f;
    }
    // the carry: its bf16 image in bufA (what the stepwise launch stages from h_in), and the
    // fp32 elements of this lane's own (row, unit) pairs
#pragma unroll
    for (int u = 0; u < H_PT; ++u) {
      const int ch = tid + u * kWsThreads;
      if (ch < H_CH) {
        const int row = ch / (H / 4), q = ch % (H / 4);
        const int64_t gi = i0 + row;
        f32x4 hv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (gi < N) hv = *reinterpret_cast<const f32x4*>(gx.h_in + gi * H + 4 * q);
        bf16x4 hb;
#pragma unroll
        for (int e = 0; e < 4; ++e) hb[e] = (bf16_t)hv[e];
        *reinterpret_cast<bf16x4*>(bufA + row * AROW + 4 * q) = hb;
      }
    }
    f32x4 hprev[RTW];
#pragma unroll
    for (int r = 0; r < RTW; ++r) {
      const int64_t gi = i0 + (wr * RTW + r) * 16 + li;
      hprev[r] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (gi < N) hprev[r] = *reinterpret_cast<const f32x4*>(gx.h_in + gi * H + wc * 16 + 4 * lq);
    }
    __syncthreads();

    for (int t = 0; t < T; ++t) {
      const int64_t tN = (int64_t)t * N;
      float xnext[IN_PT];
      int64_t nkey = 0, ncount = 0, ncounter = 0;
      bool owner = false, m_row = false;
      int owner_row = 0;
#pragma unroll
      for (int u = 0; u < IN_PT; ++u) {
        const int e = tid + u * kWsThreads;
        xnext[u] = 0.0f;
        if (e < nel) {
          const int row = (int)(((float)e + 0.5f) * rcpK0);
          const int k = e - row * K0;
          const int64_t gi = i0 + row;
          const bool live = gi < N;
          float v = xin[u];
          if (live) env.obs_seq[(tN + gi) * K0 + k] = v;
          if (norm && live) v = (v - s_mean[k]) / s_sd[k];
          bufX[row * XROW + k] = (bf16_t)v;
          if (live) {
            const int64_t key = s_key[row];
            const int64_t step = s_count[row] + 1;
            const bool d = step >= env.max_steps;
            const int64_t cw = s_counter[row] + 1;
            const bool tr = cw >= env.max_len;
            const bool m = d || tr;
            float nobs = mippo_keys::mock_obs(key, step, k);
            env.next_obs_seq[(tN + gi) * K0 + k] = nobs;
            if (k == 0) {
              env.reward_seq[tN + gi] = 1.0f;
              env.done_seq[tN + gi] = m ? 1 : 0;
              env.trunc_seq[tN + gi] = tr ? 1 : 0;
            }
            int64_t k2 = key, s2 = step, c2 = cw;
            if (m) {
              const uint64_t rk = mix(reset_key + (uint64_t)(tN + gi + 1) * kGolden);
              const uint64_t base = mix(rk + kGolden);
              k2 = (int64_t)base;
              s2 = 0;
              nobs = mippo_keys::mock_obs(k2, 0, k);
              if (k == 0) {
                const uint64_t ck = mix(rk + 2 * kGolden);
                const uint64_t bits = mix(mix(ck) ^ kM2);
                c2 = span ? (int64_t)((bits >> 1) % span) : 0;
              }
            }
            xnext[u] = nobs;
            if (k == 0) {
              owner = true;
              owner_row = row;
              nkey = k2;
              ncount = s2;
              ncounter = c2;
              m_row = m;
            }
          }
        }
      }
      __syncthreads();  // bufX and (first step) bufA staged; the step-t state has been read
      if (owner) {
        s_key[owner_row] = nkey;
        s_count[owner_row] = ncount;
        s_counter[owner_row] = ncounter;
        s_done[owner_row] = m_row ? 1 : 0;
      }
      // ---- Dense(obs -> H, relu) -------------------------------------------------------------
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
      __syncthreads();
      // ---- the GRU cell ------------------------------------------------------------------------
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
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float rg = fast_sigmoid((ai[0][e] + BI[0][e]) + ah[0][e]);
          const float zg = fast_sigmoid((ai[1][e] + BI[1][e]) + ah[1][e]);
          const float qn = ah[2][e] + BN[e];
          const float ng = fast_tanh((ai[2][e] + BI[2][e]) + rg * qn);
          hnew[r][e] = (1.0f - zg) * ng + zg * hprev[r][e];
        }
      }
      __syncthreads();  // every wave has read h and the Dense-in output: bufA takes h' now
#pragma unroll
      for (int r = 0; r < RTW; ++r) {
        bf16x4 vo;
#pragma unroll
        for (int e = 0; e < 4; ++e) vo[e] = (bf16_t)hnew[r][e];
        *reinterpret_cast<bf16x4*>(bufA + ((wr * RTW + r) * 16 + li) * AROW + wc * 16 + 4 * lq) = vo;
      }
      __syncthreads();
      // ---- Dense(H -> N_out) on h', wave w the row tile w; rows to the sampler stash ----------
      if (wave < RT) {
        f32x4 ao = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSH; ++ks) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(bufA + (wave * 16 + li) * AROW +
                                                            ks * 32 + 8 * lq);
          ao = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WO[ks], a, ao, 0, 0, 0);
        }
        const int row = wave * 16 + li;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * lq + e < N_out)
            ms_s[((t % SB) * ROWS + row) * N_out + 4 * lq + e] = ao[e] + BO[e];
      }
      __syncthreads();  // the head's rows are in the stash; every wave has read h'
      if ((t + 1) % SB == 0 || t == T - 1) {  // SB steps' rows sampled together (see above)
        const int slot = tid / ROWS, row = tid % ROWS;
        const int ts = t - (t % SB) + slot;
        if (slot < SB && ts <= t && i0 + row < N) {
          const int64_t sN = (int64_t)ts * N;
          mippo_sampler::FwdParams p = c.samp;
          const int A = p.A;
          p.noise.offset_add += (uint64_t)ts;
          if (p.raw_out) p.raw_out += sN * A;
          if (p.action) p.action += sN * A;
          if (p.mu_out) p.mu_out += sN * A;
          if (p.sigma_out) p.sigma_out += sN * A;
          if (p.ll) p.ll += sN;
          if (p.reg) p.reg += sN;
          mippo_sampler::fwd_row(ms_s + (slot * ROWS + row) * N_out, i0 + row, p);
        }
      }
      // the carry of the next step: reset-on-done (rollout.py:41-44; GRU.reset_state = zeros)
#pragma unroll
      for (int r = 0; r < RTW; ++r) {
        const int row = (wr * RTW + r) * 16 + li;
        if (s_done[row]) {
          hnew[r] = f32x4{0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<bf16x4*>(bufA + row * AROW + wc * 16 + 4 * lq) =
              bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
        }
        hprev[r] = hnew[r];
      }
      __syncthreads();  // bufA holds the next step's carry; bufX / bufB / the stash are free
#pragma unroll
      for (int u = 0; u < IN_PT; ++u) xin[u] = xnext[u];
      if (t == T - 1) {
#pragma unroll
        for (int u = 0; u < IN_PT; ++u) {
          const int e = tid + u * kWsThreads;
          if (e < nel) {
            const int row = (int)(((float)e + 0.5f) * rcpK0);
            const int64_t gi = i0 + row;
            if (gi < N) env.obs_out[gi * K0 + (e - row * K0)] = xin[u];
          }
        }
        if (owner) {
          const int64_t gi = i0 + owner_row;
          env.key_out[gi] = nkey;
          env.count_out[gi] = ncount;
          env.counter_out[gi] = ncounter;
          env.reward_out[gi] = m_row ? 0.0f : 1.0f;
        }
#pragma unroll
        for (int r = 0; r < RTW; ++r) {
          const int64_t gi = i0 + (wr * RTW + r) * 16 + li;
          if (gi < N) *reinterpret_cast<f32x4*>(gx.h_out + gi * H + wc * 16 + 4 * lq) = hprev[r];
        }
      }
    }
  }
}

template <int HV, int NHV, int H, int RT>
__global__ void __launch_bounds__(kWsThreads, 2)
rollout_gru_ws_kernel(WsChain a, GruRolloutExtra gx, WsChain v, RolloutEnv env, int n_value) {
  constexpr size_t nv = RolloutLds<HV, RT, false>::bytes, na = GruRolloutLds<H, RT>::bytes;
  __shared__ __attribute__((aligned(16))) unsigned char smem[nv > na ? nv : na];
  if ((int)blockIdx.x < n_value)
    ws_rollout_body<HV, NHV, RT, false>(v, env, (int)blockIdx.x, n_value, smem);
  else
    ws_gru_rollout_body<H, RT>(a, gx, env, (int)blockIdx.x - n_value,
                               (int)gridDim.x - n_value, smem);
}

// (value trunk) x (GRU width) pairs: trunk_ws.hip's GRU_STEP_MENU
#define GRU_ROLLOUT_MENU(X) \
  X(256, 1, 64) X(256, 1, 128) X(256, 0, 64) X(128, 1, 64) X(128, 1, 128) X(64, 1, 64)

int fill_chain(WsChain& c, const char* who, int64_t N, int64_t L, const void* const* w,
               const float* const* bias, const int64_t* dims, float* out) {
  c = {};
  c.M = N;
  c.M_head = N;
  c.K0 = (int)dims[0];
  c.N_out = (int)dims[L];
  c.out = out;
  for (int64_t l = 0; l < L; ++l) {
    MI_REQUIRE(w[l] && al16(w[l]), "%s: weight images must be 16-byte aligned", who);
    c.layer[l].w = static_cast<const bf16_t*>(w[l]);
    c.layer[l].bias = bias ? bias[l] : nullptr;
    MI_REQUIRE(!c.layer[l].bias || (reinterpret_cast<uintptr_t>(c.layer[l].bias) & 15) == 0 ||
                   l == L - 1,
               "%s: hidden biases must be 16-byte aligned", who);
  }
  return 0;
}

}  // namespace

// 1 if mi_rollout_mock_ws_bf16 takes these trunks: the pairs the one-launch policy step is
// instantiated for (mi_policy_ws_dual_supported) with a sampler row that fits the stash.
extern "C" int mi_rollout_mock_ws_supported(int64_t La, const int64_t* a_dims,
                                            const int64_t* a_acts, int64_t Lc,
                                            const int64_t* c_dims, const int64_t* c_acts) {
  if (!a_dims || !a_acts || !c_dims || !c_acts || La < 2 || Lc < 2) return 0;
  if (!mi_policy_ws_dual_supported(La, a_dims, a_acts, Lc, c_dims, c_acts)) return 0;
  return 32 * a_dims[La] <= 4096 && a_dims[La] <= 16 && c_dims[Lc] <= 16;
}

extern "C" int mi_rollout_mock_ws_bf16(
    const int64_t* env_key, const int64_t* env_step_count, const int64_t* wrap_step_counter,
    const float* obs0, const int64_t* reset_key, int64_t max_steps, int64_t max_len, int64_t T,
    int64_t N, const float* norm_mean, const float* norm_m2, const float* norm_count,
    float norm_eps, int64_t La, const void* const* a_w, const float* const* a_bias,
    const int64_t* a_dims, const int64_t* a_acts, int64_t Lc, const void* const* c_w,
    const float* const* c_bias, const int64_t* c_dims, const int64_t* c_acts,
    const uint64_t* rng_state, uint64_t offset_add, float min_std, float std_scale,
    float entropy_weight, int deterministic, float* obs_seq, float* next_obs_seq,
    float* reward_seq, uint8_t* done_seq, uint8_t* trunc_seq, float* raw_seq, float* action_seq,
    float* loglik_seq, float* mu_seq, float* sigma_seq, float* value_seq, int64_t* env_key_out,
    int64_t* env_step_count_out, int64_t* wrap_step_counter_out, float* obs_out,
    float* reward_out, mi_stream_t stream) {
  MI_REQUIRE(T >= 1 && T <= (1 << 20) && N >= 1, "mi_rollout_mock_ws_bf16: bad T / N");
  MI_REQUIRE(env_key && env_step_count && wrap_step_counter && obs0 && reset_key && rng_state,
             "mi_rollout_mock_ws_bf16: null state pointer");
  MI_REQUIRE(obs_seq && next_obs_seq && reward_seq && done_seq && trunc_seq && raw_seq &&
                 action_seq && loglik_seq && value_seq,
             "mi_rollout_mock_ws_bf16: null Transition pointer");
  MI_REQUIRE(env_key_out && env_step_count_out && wrap_step_counter_out && obs_out && reward_out,
             "mi_rollout_mock_ws_bf16: null final-state pointer");
  // a value-trunk workgroup may still be reading the step-0 state of a tile whose
  // action-trunk workgroup has already finished: the final state needs its own buffers
  MI_REQUIRE(env_key_out != env_key && env_step_count_out != env_step_count &&
                 wrap_step_counter_out != wrap_step_counter && obs_out != obs0,
             "mi_rollout_mock_ws_bf16: the final state must not alias the initial state");
  MI_REQUIRE(mi_rollout_mock_ws_supported(La, a_dims, a_acts, Lc, c_dims, c_acts),
             "mi_rollout_mock_ws_bf16: trunks outside the one-launch rollout's class "
             "(mi_rollout_mock_ws_supported)");
  MI_REQUIRE(!norm_mean || (norm_m2 && norm_count), "mi_rollout_mock_ws_bf16: incomplete normaliser");
  MI_REQUIRE(max_steps >= 0 && max_len >= 0, "mi_rollout_mock_ws_bf16: bad episode lengths");
  WsChain a, v;
  int rc = fill_chain(a, "mi_rollout_mock_ws_bf16(action)", N, La, a_w, a_bias, a_dims, nullptr);
  if (rc) return rc;
  rc = fill_chain(v, "mi_rollout_mock_ws_bf16(value)", N, Lc, c_w, c_bias, c_dims, value_seq);
  if (rc) return rc;
  a.norm_mean = v.norm_mean = norm_mean;
  a.norm_m2 = v.norm_m2 = norm_m2;
  a.norm_count = v.norm_count = norm_count;
  a.norm_eps = v.norm_eps = norm_eps;
  const int64_t A2 = a_dims[La];
  a.samp = {nullptr, {rng_state, offset_add, nullptr, nullptr}, raw_seq, action_seq, mu_seq,
            sigma_seq, loglik_seq, nullptr, (int)(A2 / 2), min_std, std_scale, entropy_weight,
            deterministic};
  RolloutEnv env = {env_key, env_step_count, wrap_step_counter, obs0, reset_key, max_steps,
                    max_len, (int)T, N, obs_seq, next_obs_seq, reward_seq, done_seq, trunc_seq,
                    env_key_out, env_step_count_out, wrap_step_counter_out, obs_out, reward_out};
  constexpr int RT = 2;  // 32-row tiles: 4096 envs = one tile per workgroup and trunk
  const int64_t ntiles = mippo::ceil_div(N, 16 * RT);
  const int64_t cus = ws_grid(1 << 30);
  int64_t nv = ntiles, na = ntiles;
  if (nv + na > cus) {
    nv = cus / 2;
    na = cus - nv;
    if (nv > ntiles) nv = ntiles;
    if (na > ntiles) na = ntiles;
  }
  const int64_t hv = c_dims[1], nhv = Lc - 2, ha = a_dims[1], nha = La - 2;
  hipStream_t st = mippo::as_stream(stream);
#define X(p, q, r, s_)                                                                    \
  if (hv == p && nhv == q && ha == r && nha == s_) {                                      \
    hipLaunchKernelGGL((rollout_ws_kernel<p, q, r, s_, RT>), dim3((unsigned)(nv + na)),   \
                       dim3(kWsThreads), 0, st, a, v, env, (int)nv);                      \
    return mippo::check_launch("mi_rollout_mock_ws_bf16");                                \
  }
  WS_DUAL_MENU(X)
#undef X
  MI_REQUIRE(false, "mi_rollout_mock_ws_bf16: no instantiation for these trunks");
}

// `unroll_env` for EpisodeWrapper(MockEnv) under make_gru_actor_critic's network (recurrent
// contract `recurrent.py:89-161`; factories.py): mi_rollout_mock_ws_bf16 with the recurrent
// actor of mi_gru_policy_step_bf16 in the action-trunk workgroups — the carry h [N][H] rides
// in registers / LDS for the T steps and is reset to zeros where a step ends an episode
// (`rollout.py:41-44`).  Network arguments as mi_gru_policy_step_bf16, env / Transition /
// final-state arguments as mi_rollout_mock_ws_bf16; h_out must not alias h_in.  Bit-identical
// to T stepwise launches of mi_gru_policy_step_bf16 + mi_mock_episode_step_select +
// mi_select_rows_multi.
extern "C" int mi_rollout_mock_gru_ws_bf16(
    const int64_t* env_key, const int64_t* env_step_count, const int64_t* wrap_step_counter,
    const float* obs0, const int64_t* reset_key, int64_t max_steps, int64_t max_len, int64_t T,
    int64_t N, int64_t K0, int64_t H, int64_t A2, const float* norm_mean, const float* norm_m2,
    const float* norm_count, float norm_eps, const void* w_in, const float* b_in,
    const void* w_proj, const float* b_proj, const float* w_h, const float* b_hn,
    const void* w_out, const float* b_out, const float* h_in, float* h_out, int64_t Lc,
    const void* const* c_w, const float* const* c_bias, const int64_t* c_dims,
    const int64_t* c_acts, const uint64_t* rng_state, uint64_t offset_add, float min_std,
    float std_scale, float entropy_weight, int deterministic, float* obs_seq,
    float* next_obs_seq, float* reward_seq, uint8_t* done_seq, uint8_t* trunc_seq,
    float* raw_seq, float* action_seq, float* loglik_seq, float* mu_seq, float* sigma_seq,
    float* value_seq, int64_t* env_key_out, int64_t* env_step_count_out,
    int64_t* wrap_step_counter_out, float* obs_out, float* reward_out, mi_stream_t stream) {
  const char* who = "mi_rollout_mock_gru_ws_bf16";
  MI_REQUIRE(T >= 1 && T <= (1 << 20) && N >= 1, "%s: bad T / N", who);
  MI_REQUIRE(env_key && env_step_count && wrap_step_counter && obs0 && reset_key && rng_state &&
                 w_in && w_proj && w_h && w_out && h_in && h_out && c_dims && c_acts,
             "%s: null input pointer", who);
  MI_REQUIRE(obs_seq && next_obs_seq && reward_seq && done_seq && trunc_seq && raw_seq &&
                 action_seq && loglik_seq && value_seq && env_key_out && env_step_count_out &&
                 wrap_step_counter_out && obs_out && reward_out,
             "%s: null output pointer", who);
  MI_REQUIRE(env_key_out != env_key && env_step_count_out != env_step_count &&
                 wrap_step_counter_out != wrap_step_counter && obs_out != obs0 && h_out != h_in,
             "%s: the final state must not alias the initial state", who);
  MI_REQUIRE(mi_gru_policy_step_supported(K0, H, A2, Lc, c_dims, c_acts),
             "%s: network outside the supported class (mi_gru_policy_step_supported)", who);
  MI_REQUIRE(!norm_mean || (norm_m2 && norm_count), "%s: incomplete normaliser", who);
  MI_REQUIRE(al16(w_in) && al16(w_proj) && al16(w_out) && al16(h_in) && al16(h_out) &&
                 al16(b_in) && al16(b_proj) && al16(b_hn),
             "%s: buffers must be 16-byte aligned", who);
  MI_REQUIRE(32 * A2 <= 4096 && max_steps >= 0 && max_len >= 0, "%s: bad sizes", who);
  WsChain a = {};
  a.M = a.M_head = N;
  a.K0 = (int)K0;
  a.N_out = (int)A2;
  a.layer[0].w = static_cast<const bf16_t*>(w_in);
  a.layer[0].bias = b_in;
  a.layer[1].w = static_cast<const bf16_t*>(w_proj);
  a.layer[1].bias = b_proj;
  a.layer[2].w = static_cast<const bf16_t*>(w_out);
  a.layer[2].bias = b_out;
  WsChain v;
  int rc = fill_chain(v, who, N, Lc, c_w, c_bias, c_dims, value_seq);
  if (rc) return rc;
  a.norm_mean = v.norm_mean = norm_mean;
  a.norm_m2 = v.norm_m2 = norm_m2;
  a.norm_count = v.norm_count = norm_count;
  a.norm_eps = v.norm_eps = norm_eps;
  a.samp = {nullptr, {rng_state, offset_add, nullptr, nullptr}, raw_seq, action_seq, mu_seq,
            sigma_seq, loglik_seq, nullptr, (int)(A2 / 2), min_std, std_scale, entropy_weight,
            deterministic};
  const GruRolloutExtra gx = {w_h, b_hn, h_in, h_out};
  RolloutEnv env = {env_key, env_step_count, wrap_step_counter, obs0, reset_key, max_steps,
                    max_len, (int)T, N, obs_seq, next_obs_seq, reward_seq, done_seq, trunc_seq,
                    env_key_out, env_step_count_out, wrap_step_counter_out, obs_out, reward_out};
  constexpr int RT = 2;
  const int64_t ntiles = mippo::ceil_div(N, 16 * RT);
  const int64_t cus = ws_grid(1 << 30);
  int64_t nv = ntiles, na = ntiles;
  if (nv + na > cus) {
    nv = cus / 2;
    na = cus - nv;
    if (nv > ntiles) nv = ntiles;
    if (na > ntiles) na = ntiles;
  }
  const int64_t hv = c_dims[1], nhv = Lc - 2;
  hipStream_t st = mippo::as_stream(stream);
#define X(p, q, r)                                                                           \
  if (hv == p && nhv == q && H == r) {                                                       \
    hipLaunchKernelGGL((rollout_gru_ws_kernel<p, q, r, RT>), dim3((unsigned)(nv + na)),      \
                       dim3(kWsThreads), 0, st, a, gx, v, env, (int)nv);                     \
    return mippo::check_launch(who);                                                         \
  }
  GRU_ROLLOUT_MENU(X)
#undef X
  MI_REQUIRE(false, "%s: no instantiation", who);
}
