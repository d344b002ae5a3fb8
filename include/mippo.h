/*
 * mippo.h — C ABI of libmippo.so: the MI355X (gfx950) kernels behind the
 * nnx-ppo `ppo_step` hot path.
 *
 * The reference (emiwar/nnx-ppo) is pure Python on JAX and has NO FFI / plugin
 * registry; every entry point below replaces an *implicit* XLA op sequence of
 * the reference, cited per function as `file:line` relative to the reference
 * root.  The Python host side (`nnx_ppo_amd/`) binds these with ctypes; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - `extern "C"`, plain device pointers + explicit int64 sizes, scalars by
 *     value, the HIP stream as an opaque `mi_stream_t` (a `hipStream_t`).
 *   - The caller owns every buffer (inputs, outputs, workspaces).  Nothing here
 *     allocates, frees, synchronises or throws; kernels are only enqueued on
 *     `stream`, so every call is HIP-graph-capturable.
 *   - Return 0 on success or a negative errno value (-EINVAL bad shape or null
 *     pointer, -EIO launch failure).  `mi_last_error()` returns a description
 *     of the last failure on the calling thread.
 *   - All tensors are dense row-major; time-major rollouts are `[T, N, *feat]`
 *     as in the reference (`nnx_ppo/algorithms/rollout.py:61-66`).
 *   - Flags (`done`, `truncated`) are uint8 (0/1) — torch.bool storage.
 */
#ifndef MIPPO_H
#define MIPPO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mi_stream_t; /* hipStream_t */

/* Activation codes shared by the dense / MLP entry points
 * (`nnx_ppo/networks/factories.py:107-110`: relu | swish | tanh; 0 = none). */
enum { MI_ACT_NONE = 0, MI_ACT_RELU = 1, MI_ACT_TANH = 2, MI_ACT_SWISH = 3,
       MI_ACT_SIGMOID = 4 /* LSTM gate functions only */ };

/* ---- library ---------------------------------------------------------- */

/* ABI version of this header (bumped when a signature changes). */
int mi_abi_version(void);

/* Description of the last error on this thread ("" if none). */
const char* mi_last_error(void);

/* ---- a13: GAE reverse scan -------------------------------------------- */

/* Generalised advantage estimation, `nnx_ppo/algorithms/ppo.py:351-394`:
 *   nv    = done[t] ? 0 : V[t+1]            (V[T] = last_value)
 *   delta = r[t] + gamma*nv - V[t];  delta = truncated[t] ? 0 : delta
 *   A[t]  = delta + (1-done[t]) * gamma * lambda * A[t+1],   A[T] = 0
 * rewards, values, done, truncated, advantages: [T, N]; last_value: [N].
 * `targets` (nullable) receives V + A (`ppo.py:456-458`).
 * One thread per env, time loop in registers, rows coalesced. */
int mi_gae_f32(const float* rewards, const float* values,
               const float* last_value, const uint8_t* done,
               const uint8_t* truncated, float* advantages, float* targets,
               int64_t T, int64_t N, float gamma, float lambda,
               mi_stream_t stream);

/* The same scan, also reducing adv_stats[3] = (sum A, sum A^2, T*N) in fp64 in
 * the same launch — the operands of the advantage normalisation that follows
 * (`ppo.py:477-480`; see mi_adv_stats_f32).  workspace:
 * mi_gae_stats_workspace_bytes(N) bytes whose first 16 bytes (a ticket counter)
 * are ZERO before the first call; the call leaves them zero. */
int64_t mi_gae_stats_workspace_bytes(int64_t N);
int mi_gae_stats_f32(const float* rewards, const float* values,
                     const float* last_value, const uint8_t* done,
                     const uint8_t* truncated, float* advantages, float* targets,
                     int64_t T, int64_t N, float gamma, float lambda,
                     double* adv_stats, void* workspace, mi_stream_t stream);

/* ---- a8 / a16: running observation normaliser -------------------------- */

/* Normalizer.__call__, `nnx_ppo/networks/normalizer.py:63-96`:
 *   std = counter > 0 ? sqrt(max(M2 / counter, epsilon)) : 10 ; out = (x - mean) / std
 * x, out: [M, F]; mean, m2: [F]; counter: device scalar (fp32, as the reference). */
int mi_normalize_fwd_f32(const float* x, const float* mean, const float* m2,
                         const float* counter, float epsilon, float* out,
                         int64_t M, int64_t F, mi_stream_t stream);
/* The same over [x ; x_tail] into one out [M + M_tail, F]: a replay's T x B observation rows and
 * the bootstrap observation behind them (`ppo.py:433-437`) in one launch. */
int mi_normalize_fwd_tail_f32(const float* x, const float* x_tail, const float* mean,
                              const float* m2, const float* counter, float epsilon, float* out,
                              int64_t M, int64_t M_tail, int64_t F, mi_stream_t stream);

/* Gradient of the above w.r.t. x (statistics are constants): g_x = g_out / std. */
int mi_normalize_bwd_f32(const float* g_out, const float* m2, const float* counter,
                         float epsilon, float* g_x, int64_t M, int64_t F,
                         mi_stream_t stream);

/* Normalizer.update_statistics, `normalizer.py:98-136`, in two halves so that a
 * multi-GPU run can combine per-shard batch statistics in between:
 *   (1) batch_stats[3][F] = (n, mean, sum of squared deviations) of x[M, F]
 *       (the reference's batch_mean / batch_M2, computed in one pass);
 *   (2) merge into the running (mean, M2, counter) with the reference's formula.
 * `advance_counter` != 0 also does counter += n (a PyTree normaliser shares one
 * counter across leaves, so only the last leaf's merge advances it). */
int64_t mi_welford_workspace_bytes(int64_t M, int64_t F);
int mi_welford_batch_stats_f32(const float* x, float* batch_stats, void* workspace,
                               int64_t M, int64_t F, mi_stream_t stream);
int mi_welford_merge_f32(float* mean, float* m2, float* counter,
                         const float* batch_stats, int64_t F, int advance_counter,
                         mi_stream_t stream);

/* a17 — `compute_metrics` / `_log_metric`, `nnx_ppo/algorithms/metrics.py:17-100`, for the
 * LOSSES family: x is row-major [R, C] (one row per gradient step, one column per loss
 * term); out[0][c] = mean_r(scale * x[r][c]), out[1][c] = population std (as `jp.std`).
 * fp64 accumulation in row order.  `scale` = 1 / world_size after a sum all-reduce. */
int mi_col_mean_std_f32(const float* x, int64_t R, int64_t C, float scale, float* out,
                        mi_stream_t stream);

/* ---- a10: tanh-Gaussian sampler ---------------------------------------- */

/* NormalTanhSampler.__call__, `nnx_ppo/networks/sampling_layers.py:82-147`.
 * mean_and_std: [B, 2A] = [mu | s];  sigma = (softplus(s) + min_std) * std_scale.
 * extras == NULL (ROLLOUT / INFERENCE): z = deterministic ? mu : mu + sigma*eps;
 * extras != NULL (LOSS_REPLAY): z = extras (the stored raw action).
 * Outputs (each nullable): raw_out[B,A] = z, action[B,A] = tanh z,
 * loglik[B], reg[B] = -entropy_weight * H_hat (one-sample entropy estimate with
 * noise eps2), mu_out / sigma_out [B,A].
 * Noise: Philox4x32-10 on (seed, offset + offset_add, element) read from the
 * device-resident rng_state = {seed, offset} (uint64[2]), unless eps / eps2
 * ([B,A], nullable) are injected. */
int mi_tanh_gauss_fwd_f32(const float* mean_and_std, const float* extras,
                          const uint64_t* rng_state, uint64_t offset_add,
                          const float* eps, const float* eps2, float* raw_out,
                          float* action, float* loglik, float* reg, float* mu_out,
                          float* sigma_out, int64_t B, int64_t A, float min_std,
                          float std_scale, float entropy_weight, int deterministic,
                          mi_stream_t stream);

/* Backward of the replay forward (what nnx.grad derives, `ppo.py:301-312`):
 * g_mean_and_std[B,2A] from g_loglik[B] (nullable = 0) and the uniform
 * regulariser gradient g_reg (d loss / d reg element).  The entropy noise is
 * regenerated from the same (rng_state, offset_add) or taken from eps2. */
int mi_tanh_gauss_bwd_f32(const float* mean_and_std, const float* extras,
                          const uint64_t* rng_state, uint64_t offset_add,
                          const float* eps2, const float* g_loglik, float g_reg,
                          float* g_mean_and_std, int64_t B, int64_t A, float min_std,
                          float std_scale, float entropy_weight, mi_stream_t stream);

/* The sampler's noise streams on their own (eps, eps2: [n], nullable), and the
 * offset bump that ends an iteration (replaces `self.rng()` advancing,
 * `sampling_layers.py:96,144`). */
int mi_philox_normal_f32(const uint64_t* rng_state, uint64_t offset_add, float* eps,
                         float* eps2, int64_t n, mi_stream_t stream);
int mi_rng_advance(uint64_t* rng_state, uint64_t n, mi_stream_t stream);

/* ---- a9: Dense layer ---------------------------------------------------- */

/* Dense.__call__, `nnx_ppo/networks/feedforward.py:42-51`: y = act(x @ w + bias).
 * x: [M, K]; w: [K, N] (flax kernel layout); bias: [N] (nullable); y: [M, N];
 * preact (nullable): [M, N] pre-activation, needed by the swish backward.
 * fp32 MFMA (exact fp32 products, fp32 accumulate). */
int mi_dense_fwd_f32(const float* x, const float* w, const float* bias, float* y,
                     float* preact, int64_t M, int64_t K, int64_t N, int act,
                     mi_stream_t stream);

/* g_x[M,K] = (g_y ⊙ act'(aux)) @ w^T.  aux = y for relu / tanh, the
 * pre-activation for swish, ignored (nullable) for MI_ACT_NONE. */
int mi_dense_bwd_dx_f32(const float* g_y, const float* aux, const float* w, float* g_x,
                        int64_t M, int64_t K, int64_t N, int act, mi_stream_t stream);

/* g_w[K,N] (+)= x^T @ (g_y ⊙ act'(aux)), g_b[N] (+)= column sums (g_b nullable).
 * Split over M into slabs in `workspace`, reduced in fixed order. */
int64_t mi_dense_bwd_dw_workspace_bytes(int64_t M, int64_t K, int64_t N);
int mi_dense_bwd_dw_f32(const float* x, const float* g_y, const float* aux, float* g_w,
                        float* g_b, void* workspace, int64_t M, int64_t K, int64_t N,
                        int act, int accumulate, mi_stream_t stream);

/* ---- a9 (bf16 path): Dense layer on bf16 MFMA, fp32 accumulate --------- */

/* Operand conventions: every bf16 operand is a dense ROW-MAJOR matrix whose
 * leading dimension `ld*` is a multiple of 8 elements (16 bytes) with the
 * padding ZERO-filled (kernels that produce a bf16 matrix write the padding),
 * 16-byte-aligned base.  fp32 master weights stay in the optimiser arena;
 * `mi_weights_to_bf16` refreshes the two bf16 shadows (w_bf [K][ldw] for dX,
 * wt_bf [N][ldwt] = W^T for the forward).  Activations are kept once, as bf16
 * [M][pad8 N]; the dW kernel transposes on the fly with ds_read_b64_tr_b16. */

/* fp32 x[M][F] (times act'(aux_bf[M][ldaux]) if act != MI_ACT_NONE) -> bf16
 * out[M][ld] (zero padded). */
int mi_cast_pad_bf16(const float* x, const void* aux_bf, int64_t ldaux, int act, void* out,
                     int64_t ld, int64_t M, int64_t F, mi_stream_t stream);

int mi_weights_to_bf16(const float* w, void* w_bf, int64_t ldw, void* wt_bf, int64_t ldwt,
                       int64_t K, int64_t N, mi_stream_t stream);

/* The same for up to 16 layers in one launch (ld = pad8 of N / K as above), plus
 * (arrays nullable) the FRAGMENT-MAJOR images the whole-trunk kernels stream:
 * frag_fwd[l] (ceil(N/16) * ceil(K/32) * 512 bf16; columns = outputs, reduce = K)
 * and frag_bwd[l] (ceil(K/16) * ceil(N/32) * 512 bf16; columns = K, reduce = N).
 * Block (ct, ks) of an image is the 64-lane MFMA 16x16x32 B-operand fragment of
 * column tile ct, k-step ks (lane l: column ct*16 + (l & 15), reduce elements
 * ks*32 + 8*(l >> 4) .. +7), so a wave reads it as one contiguous 1 KiB. */
int mi_weights_to_bf16_multi(int64_t n_layers, const float* const* w, void* const* w_bf,
                             void* const* wt_bf, void* const* frag_fwd, void* const* frag_bwd,
                             const int64_t* K, const int64_t* N, mi_stream_t stream);

/* y = act(x @ w + bias) (`feedforward.py:42-51`).  Outputs (each nullable, at
 * least one of y_f32 / y_bf): y_f32 [M][N], y_bf [M][ldy], preact_bf [M][ldy]
 * (pre-activation, for the swish backward). */
int mi_dense_fwd_bf16(const void* x_bf, int64_t ldx, const void* wt_bf, int64_t ldwt,
                      const float* bias, float* y_f32, void* y_bf, int64_t ldy,
                      void* preact_bf, int64_t M, int64_t K, int64_t N, int act,
                      mi_stream_t stream);

/* gx = (dz @ w^T) (.) prev_act'(prev): the gradient w.r.t. the PREVIOUS layer's
 * pre-activation when `prev_bf` is that layer's output (its pre-activation for
 * swish), or the plain input gradient with prev_act = MI_ACT_NONE.
 * Outputs (nullable): gx_f32 [M][K], gx_bf [M][ldgx]. */
int mi_dense_bwd_dx_bf16(const void* dz_bf, int64_t lddz, const void* w_bf, int64_t ldw,
                         const void* prev_bf, int64_t ldprev, int prev_act, float* gx_f32,
                         void* gx_bf, int64_t ldgx, int64_t M, int64_t K, int64_t N,
                         mi_stream_t stream);

/* g_w[K][N] (+)= x^T dz, g_b[N] (+)= column sums of dz, from the row-major
 * x_bf [M][ldx] and dz_bf [M][lddz]; split over M into fp32 slabs in
 * `workspace`, reduced in fixed order. */
int64_t mi_dense_bwd_dw_bf16_workspace_bytes(int64_t M, int64_t K, int64_t N);
int mi_dense_bwd_dw_bf16(const void* x_bf, int64_t ldx, const void* dz_bf, int64_t lddz,
                         float* g_w, float* g_b, void* workspace, int64_t M, int64_t K,
                         int64_t N, int accumulate, mi_stream_t stream);

/* The dW / db of up to 8 layers that share M (an MLP trunk) in at most three
 * launches (one per tile class) plus one grouped reduction: the per-layer dW
 * GEMMs are independent once every dz exists, and side by side they fill the
 * chip.  x_bf[l] [M][pad8 K_l], dz_bf[l] [M][pad8 N_l]; g_b / its entries nullable. */
int64_t mi_dense_bwd_dw_grouped_bf16_workspace_bytes(int64_t n, const int64_t* K,
                                                     const int64_t* N, int64_t M);
int mi_dense_bwd_dw_grouped_bf16(int64_t n, const void* const* x_bf, const void* const* dz_bf,
                                 float* const* g_w, float* const* g_b, const int64_t* K,
                                 const int64_t* N, int64_t M, void* workspace, int accumulate,
                                 mi_stream_t stream);

/* A whole MLP trunk (L <= 8 Dense layers, widths <= 512) in ONE launch; only
 * weights stream (`wt_bf[l]` = the frag_fwd image of layer l, mi_weights_to_bf16_multi).
 * dims[L+1] = (K_0, N_0 = K_1, ..., N_{L-1}); acts[L]; bias[l] nullable.
 * out: fp32 [M][N_{L-1}].  A workgroup walks 16 (M <= 8192) or 64 rows through
 * every layer with the activations resident in LDS, output columns split over
 * the waves, weight fragments global -> VGPR, one barrier per layer.  Training
 * stores (arrays nullable, entries nullable): y_bf[l] [M][pad8 N_l], pre_bf[l]
 * [M][pad8 N_l]; x_bf [M][pad8 K_0] = bf16 copy of the input.  `out` may be null when
 * y_bf[L-1] is given (a caller that reads the last layer's bf16 image only). */
int mi_mlp_fwd_bf16(const float* x, int64_t M, int64_t L, const void* const* wt_bf,
                    const float* const* bias, const int64_t* dims, const int64_t* acts,
                    float* out, void* const* y_bf, void* const* pre_bf, void* x_bf,
                    mi_stream_t stream);

/* One policy evaluation in ONE launch — the rollout / replay forward of
 * `make_mlp_actor_critic`'s network (`nnx_ppo/networks/factories.py:88-146`):
 *   x  = (obs - mean) / std                  normalizer.py:63-96   (norm_mean nullable)
 *   ms = action trunk(x); sampler(ms)        feedforward.py:42-51, sampling_layers.py:82-147
 *   v  = value trunk(x)                      adapter.py:75-117
 * obs [M, K0] fp32.  Trunk arguments as mi_mlp_fwd_bf16 (a_* action, c_* value;
 * a_dims[0] == c_dims[0] == K0, a_dims[La] == 2A).  Sampler arguments and outputs
 * as mi_tanh_gauss_fwd_f32 (extras != null scores stored raw actions: replay).
 * mean_and_std [M, 2A] (nullable) receives the action trunk's fp32 output (the
 * sampler backward reads it); value [M, c_dims[Lc]].  The *_y_bf / *_pre_bf /
 * *_x_bf arrays (nullable) receive the training images of mi_mlp_fwd_bf16.
 * value_tail_obs (nullable) [M_tail, K0]: extra rows for the VALUE trunk only — the
 * bootstrap observation of `ppo.py:433-437` rides along with the replay; value and the
 * c_* images then have M + M_tail rows (rows M.. belong to the tail). */
int mi_policy_fwd_bf16(
    const float* obs, int64_t M, const float* norm_mean, const float* norm_m2,
    const float* norm_count, float norm_eps, int64_t La, const void* const* a_w,
    const float* const* a_bias, const int64_t* a_dims, const int64_t* a_acts, int64_t Lc,
    const void* const* c_w, const float* const* c_bias, const int64_t* c_dims,
    const int64_t* c_acts, const float* extras, const uint64_t* rng_state, uint64_t offset_add,
    const float* eps, const float* eps2, float min_std, float std_scale, float entropy_weight,
    int deterministic, float* mean_and_std, float* raw_out, float* action, float* loglik,
    float* reg, float* mu_out, float* sigma_out, float* value, void* const* a_y_bf,
    void* const* a_pre_bf, void* a_x_bf, void* const* c_y_bf, void* const* c_pre_bf,
    void* c_x_bf, const float* value_tail_obs, int64_t M_tail, mi_stream_t stream);

/* Backward of mi_policy_fwd_bf16's replay form in ONE launch: the sampler backward
 * (arguments as mi_tanh_gauss_bwd_f32) produces the action trunk's output gradient
 * in the kernel's input stage, g_value [M, c_dims[Lc]] is the value trunk's; both dX
 * chains then run as mi_mlp_bwd_dx_bf16 (a_* / c_* as there, no input gradient).
 * Every dz (bf16) is written for mi_dense_bwd_dw_grouped_bf16. */
int mi_policy_bwd_bf16(
    const float* mean_and_std, const float* extras, const uint64_t* rng_state,
    uint64_t offset_add, const float* eps2, const float* g_loglik, float g_reg, float min_std,
    float std_scale, float entropy_weight, const float* g_value, int64_t M, int64_t La,
    const void* const* a_w, const int64_t* a_dims, const int64_t* a_acts,
    const void* const* a_aux, void* a_dz_last, void* const* a_dz_bf, int64_t Lc,
    const void* const* c_w, const int64_t* c_dims, const int64_t* c_acts,
    const void* const* c_aux, void* c_dz_last, void* const* c_dz_bf, mi_stream_t stream);

/* The dX chain of the same trunk in ONE launch (the backward twin of
 * mi_mlp_fwd_bf16): from g_out [M][N_{L-1}] (fp32; times act'_{L-1}(aux_last) if
 * act_last != MI_ACT_NONE) it produces dz_last [M][pad8 N_{L-1}] (bf16) and, walking
 * the layers backwards, dz_bf[l-1] = (dz_l . W_l^T) (.) act'_{l-1}(aux[l-1]) for
 * l = L-1..1 (each [M][pad8 K_l], bf16) — every operand the grouped dW launch
 * needs — and optionally the fp32 input gradient g_in [M][K_0].  w_bf[l] = the
 * frag_bwd image of layer l; aux[l] = layer l's output (pre-activation for swish);
 * acts[l] as in the forward.  dz_bf has L-1 entries, aux L-1 entries. */
int mi_mlp_bwd_dx_bf16(const float* g_out, const void* aux_last, int act_last, int64_t M,
                       int64_t L, const void* const* w_bf, const int64_t* dims,
                       const int64_t* acts, const void* const* aux, void* dz_last,
                       void* const* dz_bf, float* g_in, mi_stream_t stream);

/* ---- a20: GRU carry (persistent T-loop) ------------------------------------ */

/* GRU over a sequence with reset-on-done.  The reference has no GRU; the module
 * contract is the LSTM wrapper's (`nnx_ppo/networks/recurrent.py:89-161`: zeros
 * init, zeros-like reset, output = new hidden state) and the cell is flax's
 * GRUCell (r, z, n gates; h' = (1-z) n + z h; hidden-side bias on n only).
 * gi [T,B,3H] = x W_i + b_i (computed by the dense kernels); w_h [H,3H];
 * b_hn [H]; h0 [B,H]; done [T,B] uint8 (nullable = never).  Outputs: h_out
 * [T,B,H] (pre-reset h'), h_final [B,H] (post-reset carry); for training also
 * h_prev_out [T,B,H] (state entering each step) and gates_out [T,B,4H]
 * (r, z, n, gh_n + b_hn), both nullable. */
int mi_gru_seq_fwd_f32(const float* gi, const float* w_h, const float* b_hn, const float* h0,
                       const uint8_t* done, float* h_out, float* h_prev_out, float* gates_out,
                       float* h_final, int64_t T, int64_t B, int64_t H, mi_stream_t stream);

/* BPTT of the above: from g_h [T,B,H] (gradient w.r.t. h_out) to dgi [T,B,3H]
 * (gradient w.r.t. gi) and dgh [T,B,3H] (gradient w.r.t. h W_h); dh0 [B,H]
 * nullable.  dW_h = h_prev^T dgh, db_hn, dW_i, db_i, dx follow as time-batched
 * GEMMs (dense kernels). */
int mi_gru_seq_bwd_f32(const float* g_h, const float* gates, const float* h_prev,
                       const float* w_h, const uint8_t* done, float* dgi, float* dgh, float* dh0,
                       int64_t T, int64_t B, int64_t H, mi_stream_t stream);

/* The same recurrence and BPTT with h W_h (dgh W_h^T) on the bf16 matrix cores
 * (operands rounded to bf16, fp32 accumulation, fp32 cell arithmetic and carry):
 * the arguments of mi_gru_seq_fwd_f32 / mi_gru_seq_bwd_f32; H in {32, 64, 96, 128};
 * h_prev_out and gates_out are both null (inference) or both given (training).  Two more,
 * nullable: h_prev_bf [T*B, H] / dgh_bf [T*B, 3H] receive the bf16 images of h_prev / dgh —
 * the operands of the recurrent kernel's dW launch, which otherwise cost a cast launch
 * each; with dgh_bf the fp32 dgh may be null. */
int mi_gru_seq_fwd_bf16(const float* gi, const float* w_h, const float* b_hn, const float* h0,
                        const uint8_t* done, float* h_out, float* h_prev_out, float* gates_out,
                        float* h_final, void* h_prev_bf, int64_t T, int64_t B, int64_t H,
                        mi_stream_t stream);
/* mi_gru_seq_bwd_bf16 with the backward of those two layers IN FRONT of the BPTT inside the
 * launch: the sampler's backward (`mi_tanh_gauss_bwd_f32`'s operands, rows t*B + env) and the
 * head's dX (w_out_bwd = its backward fragment-major image); dz_out_bf [T*B, pad8(N_out)]
 * receives the head's output-gradient image (its dW operand).  dgi and the bf16 image of dgh
 * out as mi_gru_seq_bwd_bf16.  Bit-identical to mi_tanh_gauss_bwd_f32 + mi_mlp_bwd_dx_bf16 +
 * mi_gru_seq_bwd_bf16. */
int mi_gru_seq_bwd_tail_supported(int64_t T, int64_t H, int64_t N_out);
int mi_gru_seq_bwd_tail_bf16(
    const float* gates, const float* h_prev, const float* w_h, const uint8_t* done, float* dgi,
    float* dh0, void* dgh_bf, const void* w_out_bwd, int64_t N_out, const float* mean_and_std,
    const float* extras, const uint64_t* rng_state, uint64_t offset_add, const float* eps2,
    const float* g_loglik, float g_reg, float min_std, float std_scale, float entropy_weight,
    void* dz_out_bf, int64_t T, int64_t B, int64_t H, mi_stream_t stream);

/* The tail forms with the input projection inside as well (`recurrent.py:89-117`: gi = x W_i + b_i
 * is the first line of the cell): forward, gi_t is evaluated per step from y_bf [T*B, ldy], the
 * bf16 image of the GRU's input (the x operand of W_i's dW launch), w_i = forward fragment-major
 * image of W_i, b_i [3H] — fp32 gi [T, B, 3H] is neither written by the Dense chain in front nor
 * read here; backward, dgi leaves as its bf16 image dgi_bf [T*B, 3H] and the gradient of the relu
 * layer in front as dz0_bf [T*B, H] = bf16((dgi_bf . W_i^T) . relu'(y)) (w_i_bwd = backward
 * fragment-major image), the dz operands of the two dW problems — the chain's backward launch
 * goes.  Class: mi_gru_seq_proj_supported (the tail forms' class, K_in == H, ldy == H, and a
 * batch that runs in the 4-rows-per-workgroup form).  Bit-identical to the launches replaced. */
int mi_gru_seq_proj_supported(int64_t T, int64_t B, int64_t H, int64_t K_in, int64_t N_out);
int mi_gru_seq_fwd_proj_tail_bf16(
    const void* y_bf, int64_t ldy, const void* w_i, const float* b_i, const float* w_h,
    const float* b_hn, const float* h0, const uint8_t* done, float* h_out, float* h_prev_out,
    float* gates_out, float* h_final, void* h_prev_bf, const void* w_out, const float* b_out,
    int64_t N_out, float* ms_out, void* h_bf_out, const float* extras, const uint64_t* rng_state,
    uint64_t offset_add, const float* eps2, float min_std, float std_scale, float entropy_weight,
    float* loglik, float* reg, int64_t T, int64_t B, int64_t H, mi_stream_t stream);
/* ... and with the relu Dense(K0 <= 8 -> H) in FRONT of the GRU inside the forward launch as well
 * (`feedforward.py:42-51`): x [T*B, K0] fp32 its input, w_0 the forward fragment-major image of
 * its kernel, b_0 [H]; out x_bf_out [T*B, 8] and y_bf_out [T*B, H], the bf16 images of its input
 * and output (x operands of its own and of W_i's dW; y_bf_out is what
 * mi_gru_seq_bwd_proj_tail_bf16 reads).  Bit-identical to mi_mlp_fwd_bf16 on that layer +
 * mi_gru_seq_fwd_proj_tail_bf16. */
int mi_gru_seq_front_supported(int64_t T, int64_t B, int64_t H, int64_t K0, int64_t N_out);
int mi_gru_seq_fwd_front_proj_tail_bf16(
    const float* x, int64_t K0, const void* w_0, const float* b_0, void* x_bf_out, void* y_bf_out,
    const void* w_i, const float* b_i, const float* w_h, const float* b_hn, const float* h0,
    const uint8_t* done, float* h_out, float* h_prev_out, float* gates_out, float* h_final,
    void* h_prev_bf, const void* w_out, const float* b_out, int64_t N_out, float* ms_out,
    void* h_bf_out, const float* extras, const uint64_t* rng_state, uint64_t offset_add,
    const float* eps2, float min_std, float std_scale, float entropy_weight, float* loglik,
    float* reg, int64_t T, int64_t B, int64_t H, mi_stream_t stream);
int mi_gru_seq_bwd_proj_tail_bf16(
    const void* y_bf, int64_t ldy, const void* w_i_bwd, void* dgi_bf, void* dz0_bf,
    const float* gates, const float* h_prev, const float* w_h, const uint8_t* done, float* dh0,
    void* dgh_bf, const void* w_out_bwd, int64_t N_out, const float* mean_and_std,
    const float* extras, const uint64_t* rng_state, uint64_t offset_add, const float* eps2,
    const float* g_loglik, float g_reg, float min_std, float std_scale, float entropy_weight,
    void* dz_out_bf, int64_t T, int64_t B, int64_t H, mi_stream_t stream);

/* mi_gru_seq_fwd_bf16 (training form) with the layers BEHIND the recurrence of
 * make_gru_actor_critic's actor in the same launch: Dense(H -> N_out = 2A) (`feedforward.py:42-51`;
 * w_out = its forward fragment-major image) and NormalTanhSampler in replay mode
 * (`sampling_layers.py:82-147`: the stored raw actions `extras` [T*B, A] scored -> loglik, reg
 * [T*B]).  ms_out [T*B, N_out]: the head's fp32 rows (the sampler backward's input);
 * h_bf_out [T*B, H]: bf16 image of h_out (the head's dW operand).  Bit-identical to
 * mi_gru_seq_fwd_bf16 + mi_mlp_fwd_bf16 + mi_tanh_gauss_fwd_f32.  Class:
 * mi_gru_seq_fwd_tail_supported (the T x 16-row history must fit 96 KB of LDS). */
int mi_gru_seq_fwd_tail_supported(int64_t T, int64_t H, int64_t N_out);
int mi_gru_seq_fwd_tail_bf16(
    const float* gi, const float* w_h, const float* b_hn, const float* h0, const uint8_t* done,
    float* h_out, float* h_prev_out, float* gates_out, float* h_final, void* h_prev_bf,
    const void* w_out, const float* b_out, int64_t N_out, float* ms_out, void* h_bf_out,
    const float* extras, const uint64_t* rng_state, uint64_t offset_add, const float* eps2,
    float min_std, float std_scale, float entropy_weight, float* loglik, float* reg, int64_t T,
    int64_t B, int64_t H, mi_stream_t stream);
int mi_gru_seq_bwd_bf16(const float* g_h, const float* gates, const float* h_prev,
                        const float* w_h, const uint8_t* done, float* dgi, float* dgh, float* dh0,
                        void* dgh_bf, int64_t T, int64_t B, int64_t H, mi_stream_t stream);

/* ---- f2: LSTM carry (`nnx_ppo/networks/recurrent.py:16-161`) -------------- */

/* The recurrence of the reference's LSTM layer over a whole sequence with
 * reset-on-done (`ppo.py:411-413`): a = gi[t] + h W_h (gate order i, f, g, o),
 * c' = f c + i g, h' = o act(c'); carry <- done[t] ? (h_init, c_init) : (h', c').
 * gi [T,B,4H] = x W_i + b_h (a time-batched dense launch); w_h [H,4H]; h0, c0 [B,H];
 * done [T,B] nullable.  h_init, c_init [H]: the learnable initial state of
 * `recurrent.py:85-88,143-161` (both null: zeros).  gate_act (i, f, o) and cell_act (g and
 * the new cell state) are MI_ACT_NONE / RELU / TANH / SIGMOID — `gate_fn` /
 * `activation_fn` of `recurrent.py:36-37`; the reference's defaults are SIGMOID / TANH.
 * Outputs: h_out [T,B,H]; training (nullable): h_prev_out, c_prev_out [T,B,H] (the carry
 * entering each step) and gates_out [T,B,5H] = (i, f, g, o, act(c')); h_final, c_final
 * [B,H].  H <= 1024. */
int mi_lstm_seq_fwd_f32(const float* gi, const float* w_h, const float* h0, const float* c0,
                        const uint8_t* done, float* h_out, float* h_prev_out, float* c_prev_out,
                        float* gates_out, float* h_final, float* c_final, const float* h_init,
                        const float* c_init, int gate_act, int cell_act, int64_t T, int64_t B,
                        int64_t H, mi_stream_t stream);

/* BPTT of the above: from g_h [T,B,H] to d_gates [T,B,4H] (gradient w.r.t. the gate
 * pre-activations, i.e. w.r.t. gi and w.r.t. h W_h); dh0, dc0 [B,H] nullable.
 * dW_h = h_prev^T d_gates, db_h = colsum(d_gates), dW_i, dx follow as time-batched
 * GEMMs (dense kernels).  dinit_part (nullable) [mi_lstm_seq_bwd_blocks(B, H)][2][H]:
 * per-workgroup sums of the carried (dh, dc) over the rows whose carry was reset to the
 * learnable initial state (done[t]); summing the blocks gives d h_init, d c_init. */
int64_t mi_lstm_seq_bwd_blocks(int64_t B, int64_t H);
int mi_lstm_seq_bwd_f32(const float* g_h, const float* gates, const float* c_prev,
                        const float* w_h, const uint8_t* done, float* d_gates, float* dh0,
                        float* dc0, float* dinit_part, int gate_act, int cell_act, int64_t T,
                        int64_t B, int64_t H, mi_stream_t stream);

/* The LSTM recurrence and BPTT with h W_h (d_gates W_h^T) on the bf16 matrix cores
 * (operands rounded to bf16, fp32 accumulation, fp32 cell arithmetic and carries):
 * the arguments of mi_lstm_seq_fwd_f32 / mi_lstm_seq_bwd_f32 without the initial-state and
 * gate-function options (sigmoid / tanh, zero reset); H in {32, 64, 96, 128};
 * h_prev_out, c_prev_out and gates_out are all null (inference) or all given. */
int mi_lstm_seq_fwd_bf16(const float* gi, const float* w_h, const float* h0, const float* c0,
                         const uint8_t* done, float* h_out, float* h_prev_out,
                         float* c_prev_out, float* gates_out, float* h_final, float* c_final,
                         int64_t T, int64_t B, int64_t H, mi_stream_t stream);
int mi_lstm_seq_bwd_bf16(const float* g_h, const float* gates, const float* c_prev,
                         const float* w_h, const uint8_t* done, float* d_gates, float* dh0,
                         float* dc0, int64_t T, int64_t B, int64_t H, mi_stream_t stream);

/* ---- a14: loss terms ---------------------------------------------------- */

/* Advantage statistics for `ppo.py:477-480`: stats[3] = (sum, sum of squares,
 * count) in fp64 — a multi-GPU run all-reduces this triple before the loss.
 * workspace (both calls): mi_ppo_loss_workspace_bytes(n) bytes whose first 16
 * bytes (a ticket counter) are ZERO before the first call; calls leave them zero. */
int64_t mi_ppo_loss_workspace_bytes(int64_t n);
int mi_adv_stats_f32(const float* adv, int64_t n, double* stats, void* workspace,
                     mi_stream_t stream);

/* Clipped surrogate + value loss + regulariser mean, `ppo.py:456-531`, over a
 * flattened [T*mb] minibatch.  adv is the raw GAE output; adv_stats (nullable)
 * turns on normalisation; reg (nullable) is the per-element regulariser.
 * PyTree rewards / value heads / log-likelihoods (`ppo.py:440-474,494-510`) are one
 * call per leaf: the (ll_new, ll_old, g_ll) group or the (values, g_v) group may be
 * null — an actor term on a summed advantage has no critic part, a value head whose
 * key has no policy term has no actor part; the absent losses come back as 0.
 * Outputs: g_ll = d total / d ll_new, g_v = d total / d values (includes
 * critic_weight), loss_out[4] = (actor, critic, regularization,
 * clipping_fraction). */
int mi_ppo_loss_f32(const float* ll_new, const float* ll_old, const float* adv,
                    const float* values, const float* reg, const double* adv_stats,
                    float clip_range, float critic_weight, float* g_ll, float* g_v,
                    float* loss_out, void* workspace, int64_t n, mi_stream_t stream);

/* ---- a15: optimiser ------------------------------------------------------ */

/* Start of a gradient step: grads[n] = 0 and *step += 1 (step nullable). */
int mi_begin_grad_step_f32(float* grads, int64_t n, int64_t* step, mi_stream_t stream);

/* norm_out[0] = ||grads||_2 (fp64 accumulation) — `ppo.py:313-315` GRAD_NORM and
 * optax.clip_by_global_norm. */
int64_t mi_global_norm_workspace_bytes(int64_t n);
int mi_global_norm_f32(const float* grads, int64_t n, float* norm_out, void* workspace,
                       mi_stream_t stream);

/* optax.chain([clip_by_global_norm(max_norm)]?, adam | adamw) over flat arenas
 * (`ppo.py:555-569`).  grad_norm (nullable): device scalar from mi_global_norm_f32,
 * enables clipping; weight_decay = 0 gives plain adam.
 * begin_next_ticket == null: step holds t, already advanced by mi_begin_grad_step_f32.
 * begin_next_ticket != null (16 bytes, ZERO before the first call, left zero): step
 * holds the number of COMPLETED steps; the launch uses t = step + 1, stores it, and
 * zeroes grads after reading them — it is the next step's mi_begin_grad_step_f32.
 * n_shadows (<= 16) Dense kernels W [K, N] stored at params + shadow_begin[l] also
 * receive their new values in their bf16 images (layouts of mi_weights_to_bf16_multi;
 * the images' zero padding is left untouched), so that launch is not needed after an
 * update. */
int mi_adam_step_f32(float* params, float* grads, float* m, float* v, int64_t n, float lr,
                     float b1, float b2, float eps, float weight_decay, int64_t* step,
                     const float* grad_norm, float max_norm, void* begin_next_ticket,
                     int64_t n_shadows, const int64_t* shadow_begin, const int64_t* shadow_K,
                     const int64_t* shadow_N, void* const* w_bf, void* const* wt_bf,
                     void* const* frag_fwd, void* const* frag_bwd, mi_stream_t stream);

/* ---- a5 / a7: data movement ---------------------------------------------- */

/* Minibatch gather `x[:, inds]`, `ppo.py:297-300`: dst[t, j, :] = src[t, idx[j], :]
 * for time-major src [T, N, row_bytes], idx int64 [L]. */
int mi_gather_cols(const void* src, const int64_t* idx, void* dst, int64_t T, int64_t N,
                   int64_t L, int64_t row_bytes, mi_stream_t stream);

/* tree_where leaf, `rollout.py:270-279`: out[b,:] = mask[b] ? on_true[b,:] : on_false[b,:].
 * true_row_stride_bytes = row_bytes, or 0 to broadcast one on_true row. */
int mi_select_rows(const uint8_t* mask, const void* on_true, int64_t true_row_stride_bytes,
                   const void* on_false, void* out, int64_t B, int64_t row_bytes,
                   mi_stream_t stream);

/* Several tree_where leaves in one launch (n_leaves <= 16): host arrays of
 * per-leaf pointers / strides / row sizes. */
int mi_select_rows_multi(const uint8_t* mask, const void* const* on_true,
                         const int64_t* true_row_stride_bytes, const void* const* on_false,
                         void* const* out, const int64_t* row_bytes, int64_t n_leaves, int64_t B,
                         mi_stream_t stream);

/* dst[l][0..nbytes[l]) = src[l][...] for n_leaves <= 16 contiguous buffers in one
 * launch: the hand-over of the new env / carry / key tensors into the training
 * state's buffers at the end of an iteration (`ppo.py:343-348` returns a new
 * TrainingState; a captured HIP graph needs the same buffers every replay). */
int mi_copy_multi(const void* const* src, void* const* dst, const int64_t* nbytes,
                  int64_t n_leaves, mi_stream_t stream);

/* The minibatch gather for several leaves in one launch (n_leaves <= 16); leaf l
 * is time-major [T[l], N, row_bytes[l]].  The L indices are L / group_len consecutive
 * groups (minibatches, `ppo.py:284-300`): dst[l] is [L / group_len][T[l]][group_len][row],
 * so every group is a contiguous time-major block; group_len = L is one group. */
int mi_gather_cols_multi(const void* const* src, void* const* dst, const int64_t* T,
                         const int64_t* row_bytes, int64_t n_leaves, const int64_t* idx,
                         int64_t N, int64_t L, int64_t group_len, mi_stream_t stream);

/* ---- a4 / a18: integer keys and episode bookkeeping ----------------------- */

/* Key expansion, the integer scheme of nnx_ppo_amd/random.py (splitmix64) in
 * one launch: keys[n] -> out[n*m].  mode: 0 split (int64 children,
 * `jax.random.split`), 1 bits (int64), 2 randint in [minval, maxval) (int64,
 * `episode_wrapper.py:28-30`), 3 uniform [0,1) (fp32, 24 exact bits),
 * 4 zero-mean unit-variance uniform (fp32).  child_major != 0 lays the output out
 * as [m][n] (each child set contiguous) instead of [n][m].  fold (nullable,
 * int64[n]): expand mi_key_fold(keys, fold) instead of keys, in the same launch
 * (`test_dummies/mock_env.py:41-52` folds the step count into the obs key). */
int mi_key_expand(const int64_t* keys, const int64_t* fold, void* out, int64_t n, int64_t m,
                  int mode, int64_t minval, int64_t maxval, int child_major,
                  mi_stream_t stream);

/* The minibatch permutations of `ppo.py:284-294` in one launch: out[e][:] =
 * permutation(fold_in(key, e), n) for e < n_perm (argsort of the n hashes of the
 * folded key; (hash, index) order = torch's stable argsort).  key: one int64;
 * out: int64 [n_perm][n]; n <= 8192. */
int mi_key_permutations(const int64_t* key, int64_t* out, int64_t n_perm, int64_t n,
                        mi_stream_t stream);

/* out[i] = mix(a[i] ^ mix(b[i] + GOLDEN)): fold a per-env integer into a key. */
int mi_key_fold(const int64_t* a, const int64_t* b, int64_t* out, int64_t n,
                mi_stream_t stream);

/* EpisodeWrapper.step, `nnx_ppo/wrappers/episode_wrapper.py:12-22`:
 * counter' = counter + 1; truncated = inner_truncated | counter' >= max_len;
 * done = float(inner_done | truncated).  inner_done is uint8/bool or fp32
 * (done_is_float); inner_truncated nullable.  done_flag_out (nullable, uint8[n])
 * receives the same flag as a byte (`rollout.py:24,41-44` selects on it). */
int mi_episode_step(const int64_t* counter, const void* inner_done, int done_is_float,
                    const uint8_t* inner_truncated, int64_t max_len, int64_t* counter_out,
                    uint8_t* truncated_out, float* done_out, uint8_t* done_flag_out, int64_t n,
                    mi_stream_t stream);

/* The scan's output stacking (`rollout.py:61-66`: every Transition leaf of T steps becomes
 * one `[T, ...]` array) for n_leaves leaves in one launch: dst[l] + t * nbytes[l] <-
 * src[l * T + t][0 .. nbytes[l]).  n_leaves <= 16 and n_leaves * T <= 448. */
int mi_stack_multi(const void* const* src, void* const* dst, const int64_t* nbytes,
                   int64_t n_leaves, int64_t T, mi_stream_t stream);

/* EpisodeWrapper.step (`nnx_ppo/wrappers/episode_wrapper.py:12-22`) AND the rollout's
 * reset-on-done select of the env state (`rollout.py:41-44`, `tree_where` 270-279) in one
 * launch.  The first nine arguments are `mi_episode_step`'s; mask[b] = done flag of row b.
 * The three leaves the wrapper itself produces are selected against their reset values
 * (`*_sel[b] = mask[b] ? reset_*[b] : *_out[b]`), every other leaf l < n_leaves (<= 16) as
 * `mi_select_rows_multi` does.  B rows. */
int mi_episode_step_select(const int64_t* counter, const void* inner_done, int done_is_float,
                           const uint8_t* inner_truncated, int64_t max_len,
                           int64_t* counter_out, uint8_t* truncated_out, float* done_out,
                           uint8_t* done_flag_out, const int64_t* reset_counter,
                           const uint8_t* reset_truncated, const float* reset_done,
                           int64_t* counter_sel, uint8_t* truncated_sel, float* done_sel,
                           const void* const* on_true, const int64_t* true_row_stride_bytes,
                           const void* const* on_false, void* const* out,
                           const int64_t* row_bytes, int64_t n_leaves, int64_t B,
                           mi_stream_t stream);

/* mi_episode_step_select with the synthetic env's own step (mi_mock_env_step) inside the
 * launch: inner done = mock_count + 1 >= mock_max_steps, and leaves marked in `produced`
 * (1: float32 observation columns from produced_col0[l] of the env's flat draw; 2: the
 * int64 step counter) are computed — stored to on_false[l] as the stepped state and
 * selected against on_true[l] — instead of read.  envs/synthetic.py MockEnv under
 * wrappers/episode_wrapper.py; bit-identical to the two launches it replaces. */
int mi_mock_episode_step_select(
    const int64_t* mock_key, const int64_t* mock_count, int64_t mock_max_steps,
    const int64_t* produced, const int64_t* produced_col0, const int64_t* counter,
    const uint8_t* inner_truncated, int64_t max_len, int64_t* counter_out,
    uint8_t* truncated_out, float* done_out, uint8_t* done_flag_out,
    const int64_t* reset_counter, const uint8_t* reset_truncated, const float* reset_done,
    int64_t* counter_sel, uint8_t* truncated_sel, float* done_sel, const void* const* on_true,
    const int64_t* true_row_stride_bytes, const void* const* on_false, void* const* out,
    const int64_t* row_bytes, int64_t n_leaves, int64_t B, mi_stream_t stream);

/* Weights-stationary form of mi_mlp_fwd_bf16 for training sizes (csrc/trunk_ws.hip): one
 * workgroup per CU keeps the whole trunk in registers and loops over row tiles.  Shape
 * class (mi_mlp_ws_supported): dims = [K0 <= 32, H, ..., H, N_out <= 16] with
 * H in {64, 128, 256} and at most 3 / 2 / 1 H x H layers, relu on every hidden layer, a
 * linear head.  Same operands as mi_mlp_fwd_bf16 (fragment-major forward images, fp32
 * biases, fp32 input); y_bf[l] (l < L - 1) and x_bf are the bf16 images kept for the
 * backward (nullable: inference).  Bit-identical to mi_mlp_fwd_bf16. */
int mi_mlp_ws_supported(int64_t L, const int64_t* dims, const int64_t* acts);
int mi_mlp_ws_fwd_bf16(const float* x, int64_t M, int64_t L, const void* const* wt_bf,
                       const float* const* bias, const int64_t* dims, const int64_t* acts,
                       float* out, void* const* y_bf, void* x_bf, mi_stream_t stream);

/* mi_policy_fwd_bf16 on the weights-stationary kernels (action trunk + sampler, value
 * trunk + bootstrap tail rows; one launch or two, see below): same arguments (mean_and_std
 * is required; relu trunks keep no pre-activations: a_pre_bf[l] / c_pre_bf[l], when given,
 * receive the relu' mask of layer l's output instead — see mi_policy_ws_bwd_bf16), same
 * results bit for bit.
 * mi_policy_ws_supported: both trunks in the shape class of mi_mlp_ws_supported, 2A <= 16. */
int mi_policy_ws_supported(int64_t La, const int64_t* a_dims, const int64_t* a_acts, int64_t Lc,
                           const int64_t* c_dims, const int64_t* c_acts);
/* For the trunk pairs it is instantiated for, mi_policy_ws_fwd_bf16 runs both trunks in ONE
 * launch (value-trunk workgroups beside action-trunk workgroups: one 32-row tile each when
 * the launch fits the chip — rollout sizes — else the CUs are split between the trunks);
 * mi_policy_ws_dual_supported says whether a pair is one of them.  Below 8192 rows only
 * this form is used (the per-trunk launches pay off from ~2 tiles per CU). */
int mi_policy_ws_dual_supported(int64_t La, const int64_t* a_dims, const int64_t* a_acts,
                                int64_t Lc, const int64_t* c_dims, const int64_t* c_acts);
int mi_policy_ws_fwd_bf16(
    const float* obs, int64_t M, const float* norm_mean, const float* norm_m2,
    const float* norm_count, float norm_eps, int64_t La, const void* const* a_w,
    const float* const* a_bias, const int64_t* a_dims, const int64_t* a_acts, int64_t Lc,
    const void* const* c_w, const float* const* c_bias, const int64_t* c_dims,
    const int64_t* c_acts, const float* extras, const uint64_t* rng_state, uint64_t offset_add,
    const float* eps, const float* eps2, float min_std, float std_scale, float entropy_weight,
    int deterministic, float* mean_and_std, float* raw_out, float* action, float* loglik,
    float* reg, float* mu_out, float* sigma_out, float* value, void* const* a_y_bf,
    void* const* a_pre_bf, void* a_x_bf, void* const* c_y_bf, void* const* c_pre_bf,
    void* c_x_bf, const float* value_tail_obs, int64_t M_tail, mi_stream_t stream);

/* The two halves of mi_dense_bwd_dw_grouped_bf16 on their own: the dW launch that leaves
 * the split-M slabs in `workspace` (slab_ptr_out[l] / n_slabs_out[l]: where, how many;
 * slab s of problem l is [K_l * N_l + N_l] floats: dW then db), and their fixed-order
 * reduction into the gradients.  mi_adam_step_slabs_f32 is mi_adam_step_f32 that sums the
 * slabs itself while it reads the gradient arena (gw_offset / gb_offset: arena index of
 * each problem's kernel / bias gradient, gb < 0: no bias) — bit-identical to reducing
 * first, one launch fewer per gradient step.  gb_first (nullable: all zero): first bias column
 * of problem l that has a home in the arena — column j >= gb_first[l] goes to arena element
 * gb_offset[l] + j - gb_first[l], the columns below are dropped (a GRU's recurrent kernel:
 * only the n gate's third of the 3H column sums of dgh is a parameter's gradient). */
int mi_dense_bwd_dw_grouped_slabs_bf16(int64_t n, const void* const* x_bf,
                                       const void* const* dz_bf, const int64_t* K,
                                       const int64_t* N, int64_t M, void* workspace,
                                       const void** slab_ptr_out, int64_t* n_slabs_out,
                                       mi_stream_t stream);
int mi_reduce_slabs_grouped_f32(int64_t n, const void* const* slab_ptr, const int64_t* n_slabs,
                                const int64_t* K, const int64_t* N, float* const* g_w,
                                float* const* g_b, int accumulate, mi_stream_t stream);
int mi_adam_step_slabs_f32(
    float* params, float* grads, float* m, float* v, int64_t n, float lr, float b1, float b2,
    float eps, float weight_decay, int64_t* step, const float* grad_norm, float max_norm,
    void* begin_next_ticket, int64_t n_shadows, const int64_t* shadow_begin,
    const int64_t* shadow_K, const int64_t* shadow_N, void* const* w_bf, void* const* wt_bf,
    void* const* frag_fwd, void* const* frag_bwd, int64_t n_slab_leaves,
    const void* const* slab_ptr, const int64_t* n_slabs, const int64_t* slab_K,
    const int64_t* slab_N, const int64_t* gw_offset, const int64_t* gb_offset,
    const int64_t* gb_first, mi_stream_t stream);

/* One rollout / evaluation step of the recurrent actor-critic of make_gru_actor_critic
 * (normaliser -> Dense(K0 -> H, relu) -> GRU(H -> H) -> Dense(H -> 2A) -> NormalTanhSampler,
 * beside an MLP value trunk) in ONE launch; the recurrent contract is the reference's
 * networks/recurrent.py:89-161, the composition recurrent_test.py:245-261.  w_in / w_proj /
 * w_out: forward fragment-major images (w_proj: the GRU's input projection [H, 3H], columns
 * r | z | n); w_h: fp32 [H, 3H]; h_in / h_out: the carry [M, H].  Sampler arguments and
 * outputs as mi_policy_fwd_bf16.  Bit-identical to the generic containers' launches.
 * mi_gru_policy_step_supported: K0 <= 32, H in {64, 128}, 2A <= 16, value trunk in the
 * class of mi_mlp_ws_supported and the (value trunk, H) pair instantiated. */
int mi_gru_policy_step_supported(int64_t K0, int64_t H, int64_t A2, int64_t Lc,
                                 const int64_t* c_dims, const int64_t* c_acts);
int mi_gru_policy_step_bf16(
    const float* obs, int64_t M, int64_t K0, int64_t H, int64_t A2, const float* norm_mean,
    const float* norm_m2, const float* norm_count, float norm_eps, const void* w_in,
    const float* b_in, const void* w_proj, const float* b_proj, const float* w_h,
    const float* b_hn, const void* w_out, const float* b_out, const float* h_in, float* h_out,
    int64_t Lc, const void* const* c_w, const float* const* c_bias, const int64_t* c_dims,
    const int64_t* c_acts, const uint64_t* rng_state, uint64_t offset_add, const float* eps,
    const float* eps2, float min_std, float std_scale, float entropy_weight, int deterministic,
    float* mean_and_std, float* raw_out, float* action, float* loglik, float* reg,
    float* mu_out, float* sigma_out, float* value, mi_stream_t stream);

/* Weights-stationary forms of mi_mlp_bwd_dx_bf16 (no input gradient, linear last layer)
 * and mi_policy_bwd_bf16 for training sizes: same operands (w_bf: the BACKWARD
 * fragment-major images; aux[l] = y_l, dz_bf[l] = dz_l for l < L - 1; dz_last = dz_{L-1}),
 * same results bit for bit.  a_mask / c_mask (both or neither; entry l for l < L - 1): the
 * relu' masks that mi_policy_ws_fwd_bf16 wrote through its a_pre_bf / c_pre_bf arrays —
 * uint8 [ceil(rows / 64)][N_l / 16][64][4]: 16 x 16 tile (rt, ct) has the byte at
 * [rt >> 2][ct][lane][rt & 3], bit e = y_l[16 rt + (lane & 15)][16 ct + 4 (lane >> 4) + e] > 0.
 * With them the backward reads one byte per lane and tile instead of 8 bytes of the bf16
 * image (half its HBM bytes). */
int mi_mlp_ws_bwd_dx_bf16(const float* g_out, int64_t M, int64_t L, const void* const* w_bf,
                          const int64_t* dims, const int64_t* acts, const void* const* aux,
                          void* dz_last, void* const* dz_bf, mi_stream_t stream);
int mi_policy_ws_bwd_bf16(
    const float* mean_and_std, const float* extras, const uint64_t* rng_state,
    uint64_t offset_add, const float* eps2, const float* g_loglik, float g_reg, float min_std,
    float std_scale, float entropy_weight, const float* g_value, int64_t M, int64_t La,
    const void* const* a_w, const int64_t* a_dims, const int64_t* a_acts,
    const void* const* a_aux, void* a_dz_last, void* const* a_dz_bf, int64_t Lc,
    const void* const* c_w, const int64_t* c_dims, const int64_t* c_acts,
    const void* const* c_aux, void* c_dz_last, void* const* c_dz_bf,
    const void* const* a_mask, const void* const* c_mask, mi_stream_t stream);

/* mi_gae_ppo_loss_f32 + mi_policy_ws_bwd_bf16 in ONE launch (replaces the GAE / loss launch
 * between the replay forward and the backward of `nnx_ppo/algorithms/ppo.py:433-503` for the
 * fused MLP actor-critic): the [T, B] operands of mi_gae_ppo_loss_f32 come in instead of
 * g_loglik / g_value.  Every workgroup scans the env groups of its own row tiles; the
 * advantage statistics are summed from per-group partials that B / 64 action-trunk
 * workgroups publish (same accumulation order and shuffle trees as gae_loss_kernel: the same
 * bits); d loss / d log-likelihood and d loss / d value are evaluated where they are
 * consumed; loss_out[4] (actor, critic, regularisation, clipping fraction) is summed from
 * per-tile fp64 partials by the last workgroup.  dz images bit-identical to the two-launch
 * path; the four scalars differ from it in fp64 summation order only.
 * Supported: T <= 32, B % 64 == 0, B <= 2048, scalar value head, the one-launch trunk menu at
 * 64-row tiles, masks given.  workspace: mi_policy_ws_bwd_gae_workspace_bytes(T * B) bytes,
 * zeroed once (the launch re-arms it).
 * `comm` (nullable; section e): an env-sharded run's one-shot communicator.  The advantage
 * statistics are then those of the GLOBAL minibatch (`ppo.py:477-480` over world x B envs;
 * SURVEY 8e (2)): a publishing workgroup hands its group's partial to every rank's region,
 * the consumers sum world x B/64 partials in (rank, group) order — the same bits on every
 * rank — and the launch counts as one collective of the communicator.  A sharded rank then
 * launches what a single GPU launches (no GAE / exchange / loss launches of their own). */
int64_t mi_policy_ws_bwd_gae_workspace_bytes(int64_t M);
int mi_policy_ws_bwd_gae_supported(int64_t T, int64_t B, int64_t La, const int64_t* a_dims,
                                   const int64_t* a_acts, int64_t Lc, const int64_t* c_dims,
                                   const int64_t* c_acts);
int mi_policy_ws_bwd_gae_bf16(
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
    const void* const* a_mask, const void* const* c_mask, void* comm, mi_stream_t stream);
/* Exactly one of loss_out / partials_out above is given.  With partials_out (double
 * [T * B / 64][4], caller-owned) the launch leaves its per-tile fp64 partials there and does
 * not sum them — the sum at the tail of the launch is 2.5-3 us on its critical path for four
 * scalars that are read at the end of the iteration; mi_policy_loss_finalize_f32 sums the
 * partials of up to 32 deferred launches in one launch, in the in-kernel order (the same
 * bits): launch s left n_partials[s] rows of 4 doubles (T * B / 64 here; N / 64 for
 * mi_gae_ppo_loss_f32's partials_out) over n_elements[s] = T * B elements. */
int mi_policy_loss_finalize_f32(int64_t n, const void* const* partials,
                                const int64_t* n_partials, const int64_t* n_elements,
                                float* const* loss_out, mi_stream_t stream);

/* The synthetic benchmark env's whole step in one launch (`nnx_ppo_amd/envs/synthetic.py`
 * MockEnv, restating `nnx_ppo/test_dummies/mock_env.py:25-63`): step' = step + 1,
 * done = step' >= max_steps, obs = unit-variance noise from fold(key, step') written to
 * n_leaves <= 8 observation leaves [n, leaf_width[l]] (a PyTree observation is the flat
 * draw cut at the leaf widths).  Bit-identical to mi_key_expand (unit-uniform, folded)
 * followed by mi_episode_step. */
int mi_mock_env_step(const int64_t* key, const int64_t* step_count, int64_t max_steps,
                     int64_t* step_count_out, uint8_t* done_out, float* const* obs_leaf,
                     const int64_t* leaf_width, int64_t n_leaves, int64_t n,
                     mi_stream_t stream);

/* a13 + a14 in ONE launch for a single reward key at minibatch size (T <= 32, N <= 16384;
 * mi_gae_ppo_loss_supported): `gae` (ppo.py:351-394), the advantage statistics and
 * normalisation (ppo.py:477-480, `normalize` != 0) and the loss terms with their gradients
 * (ppo.py:456-503) — what mi_gae_stats_f32 followed by mi_ppo_loss_f32 compute, with the
 * advantages kept in registers in between.  advantages [T,N] and adv_stats [3] are
 * nullable outputs (diagnostics); reg [T,N] is nullable.  Advantages, statistics and both
 * gradients are bit-identical to the two-launch path; the four loss scalars agree to fp64
 * summation order.  workspace: mi_gae_ppo_loss_workspace_bytes(), zero-initialised once
 * (the kernel re-arms it). */
int64_t mi_gae_ppo_loss_workspace_bytes(void);
int mi_gae_ppo_loss_supported(int64_t T, int64_t N);
int mi_gae_ppo_loss_f32(const float* rewards, const float* values, const float* last_value,
                        const uint8_t* done, const uint8_t* truncated, const float* ll_new,
                        const float* ll_old, const float* reg, float gamma, float lambda,
                        int normalize, float clip_range, float critic_weight,
                        float* advantages, double* adv_stats, float* g_ll, float* g_v,
                        float* loss_out, double* partials_out, void* workspace, int64_t T,
                        int64_t N, mi_stream_t stream);
/* Exactly one of loss_out / partials_out: with partials_out (double [ceil(N / 64)][4],
 * caller-owned) the launch leaves its per-workgroup partials there instead of summing them
 * at its tail; mi_policy_loss_finalize_f32 (n_partials = ceil(N / 64), n_elements = T * N)
 * sums them later, in the same order. */

/* ---- a6 / a7 / a18: the whole rollout of a device-steppable env in ONE launch ---------- */

/* `unroll_env` (`rollout.py:48-73`) for EpisodeWrapper(MockEnv) (`episode_wrapper.py:7-39`
 * around `test_dummies/mock_env.py:25-63`) and the fused MLP actor-critic of
 * `factories.py:88-146`: T x {`single_transition` (`rollout.py:11-45`): network forward
 * (normaliser `normalizer.py:63-96`, Dense trunks `feedforward.py:42-51`, sampler
 * `sampling_layers.py:82-147`, adapter `adapter.py:75-117`) -> env.step -> EpisodeWrapper
 * counter / truncation / done -> record -> reset-on-done select (`tree_where`,
 * `rollout.py:270-279`, with env.reset(split(reset_key, (T, N))[t][n]), `rollout.py:57-59`)}.
 * A workgroup owns a 32-env tile for all T steps (envs never interact: `rollout.py:21,39`
 * vmaps over them); weights stay in registers, env state in LDS, every Transition leaf goes
 * straight into its time-major [T, N, ...] buffer (`rollout.py:61-66`).  Replaces 2 T + 2
 * launches of the stepwise form (mi_policy_ws_fwd_bf16 + mi_mock_episode_step_select per
 * step, the batched reset, mi_stack_multi) and is BIT-IDENTICAL to it in every leaf.
 *   env state in (read only; [N] int64 / [N][K0] fp32): MockEnv key and step count, the
 *     wrapper's step counter, the current observation; reset_key: DEVICE scalar;
 *   network: as mi_policy_ws_fwd_bf16 (trunk pairs of mi_rollout_mock_ws_supported);
 *     step t draws its noise at offset_add + t;
 *   Transition out: obs / next_obs [T][N][K0], reward [T][N], done / truncated [T][N] (0/1
 *     bytes), raw action / action / mu / sigma [T][N][A] (mu, sigma nullable), log-likelihood
 *     [T][N], value [T][N][N_value];
 *   final state out (reset select applied; must NOT alias the inputs): key, step count,
 *     step counter, obs, reward (0 where the last step ended an episode, else 1). */
int mi_rollout_mock_ws_supported(int64_t La, const int64_t* a_dims, const int64_t* a_acts,
                                 int64_t Lc, const int64_t* c_dims, const int64_t* c_acts);
int mi_rollout_mock_ws_bf16(
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
    float* reward_out, mi_stream_t stream);

/* The same rollout under make_gru_actor_critic's network (`recurrent.py:89-161` contract;
 * the recurrent actor of mi_gru_policy_step_bf16 in the action-trunk workgroups): the carry
 * h [N][H] stays in registers / LDS for the T steps and is reset to zeros where a step ends an
 * episode (`rollout.py:41-44`, GRU.reset_state).  Network arguments as
 * mi_gru_policy_step_bf16 (class: mi_gru_policy_step_supported), the rest as
 * mi_rollout_mock_ws_bf16; h_out (the carry after the last reset select) must not alias h_in.
 * Bit-identical to the stepwise launches. */
int mi_rollout_mock_gru_ws_bf16(
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
    int64_t* wrap_step_counter_out, float* obs_out, float* reward_out, mi_stream_t stream);

/* ---- e: one-shot peer exchange (new; the reference is single-device) -------- */

/* Env-sharded data parallelism (BASELINE.json north_star): one process per
 * GPU; the only exchanges are the gradient arena once per gradient step (the loss of
 * `ppo.py:301-316` is a mean over envs, so the global-minibatch gradient is the mean of the
 * shards' gradients), the advantage-statistics triple of `ppo.py:477-480`, the
 * normaliser's batch statistics (`normalizer.py:98-136`) and the logged loss rows.
 *
 * Transport: each rank owns one fine-grained device buffer, exported through HIP IPC and
 * mapped by every peer.  A collective is ONE plain kernel launch per rank: write this
 * rank's contribution into its slot on every peer (one xGMI hop, all links at once),
 * raise per-chunk flags behind a system-scope release, wait (bounded) for the peers'
 * flags, reduce the `world` slots in rank order — every rank gets bit-identical results,
 * and the launch is HIP-graph capturable.
 *
 * Host protocol (the only entry points that allocate or synchronise; not capturable):
 *   mi_comm_create   allocate the local region (2 parities x world slots of slot_bytes
 *                    each, slot_bytes a multiple of 4096) and return its IPC handle
 *                    (mi_comm_handle_bytes() bytes) for the caller to all-gather by any
 *                    host channel (torch.distributed, MPI, a file);
 *   mi_comm_connect  map the peers' regions from the gathered handles
 *                    (world x mi_comm_handle_bytes() bytes, in rank order);
 *   mi_comm_status   synchronise and report (collectives completed, spins that timed out);
 *   mi_comm_destroy  unmap and free.
 * `timeout_seconds` bounds every wait inside a kernel: a peer that never arrives makes
 * the kernel count an error and finish (with NaN / no update, see mi_comm_set_error_word)
 * instead of hanging the device. */
int64_t mi_comm_handle_bytes(void);
int mi_comm_create(int rank, int world, int64_t slot_bytes, double timeout_seconds,
                   void** comm_out, void* handle_out);
int mi_comm_connect(void* comm, const void* all_handles);
int mi_comm_status(void* comm, int64_t* seq_out, int64_t* errors_out);
/* Mirror the sticky timeout count into a caller-owned, zero-initialised 4-byte device word
 * (NULL: stop mirroring): every kernel launched afterwards adds to it whenever it adds to
 * the region's own error count, so the caller can copy it to the host together with its
 * other per-iteration scalars instead of synchronising on mi_comm_status — the training
 * loop's one host sync per iteration (`ppo.py:209`) then also carries "a peer did not
 * arrive".  After a timeout the collectives never hand back a partial result: the
 * all-reduce / all-gather write NaN for the chunk, mi_adam_step_allreduce_f32 applies no
 * update (this launch and every later one: the count is sticky). */
int mi_comm_set_error_word(void* comm, void* device_word);
int64_t mi_comm_slot_bytes(void* comm);
int mi_comm_destroy(void* comm);

/* buf[i] <- scale * sum_r buf_r[i], summed in rank order (in place, n * size <= slot). */
int mi_allreduce_oneshot_f32(void* comm, float* buf, int64_t n, float scale,
                             mi_stream_t stream);
int mi_allreduce_oneshot_f64(void* comm, double* buf, int64_t n, double scale,
                             mi_stream_t stream);

/* dst[r][0..nbytes) <- rank r's src (dst: world x nbytes bytes; nbytes <= slot). */
int mi_allgather_oneshot(void* comm, const void* src, int64_t nbytes, void* dst,
                         mi_stream_t stream);

/* mi_adam_step_f32 (no clipping) with the gradient all-reduce-MEAN inside the launch:
 * chunk by chunk, exchange, reduce in rank order, scale by 1 / world, Adam.  One launch
 * per gradient step for exchange + optimiser + bf16 images + gradient zeroing. */
int mi_adam_step_allreduce_f32(void* comm, float* params, float* grads, float* m, float* v,
                               int64_t n, float lr, float b1, float b2, float eps,
                               float weight_decay, int64_t* step, void* begin_next_ticket,
                               int64_t n_shadows, const int64_t* shadow_begin,
                               const int64_t* shadow_K, const int64_t* shadow_N,
                               void* const* w_bf, void* const* wt_bf, void* const* frag_fwd,
                               void* const* frag_bwd, int64_t n_slab_leaves,
                               const void* const* slab_ptr, const int64_t* n_slabs,
                               const int64_t* slab_K, const int64_t* slab_N,
                               const int64_t* gw_offset, const int64_t* gb_offset,
                               const int64_t* gb_first, mi_stream_t stream);
/* (n_slab_leaves > 0: the pending split-M slabs of mi_dense_bwd_dw_grouped_slabs_bf16 are
 * summed into this rank's gradient chunk before it is pushed — mi_adam_step_slabs_f32's sums
 * in the same order — so a sharded gradient step is forward, backward, dW and this launch.) */

#ifdef __cplusplus
}
#endif
#endif /* MIPPO_H */
