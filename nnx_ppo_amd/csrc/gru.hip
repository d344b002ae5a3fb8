// a20 — GRU carry: persistent T-loop forward and BPTT backward.
//
// The reference has no GRU (its recurrent layer is an LSTM wrapper,
// nnx_ppo/networks/recurrent.py:16-161); BASELINE.json config 4 asks for a GRU
// with the same StatefulModule contract (zeros init, zeros-like reset on done,
// output = new hidden state).  Cell arithmetic follows flax's GRUCell
// (third-party, not in the reference tree: PARITY UNPINNED):
//     r = sigmoid(gi_r + gh_r)          gi = x W_i + b_i   (time-batched GEMM, dense kernels)
//     z = sigmoid(gi_z + gh_z)          gh = h W_h         (inside this kernel)
//     n = tanh(gi_n + r * (gh_n + b_hn))
//     h' = (1 - z) n + z h ;  carry <- done ? 0 : h'      (reset-on-done, ppo.py:411-413)
//
// One workgroup owns 4..16 envs for ALL T steps: the hidden state tile lives in LDS
// across the time loop (and W_h too when it fits), so a sequence costs one launch
// and no per-step HBM round trip of the carry.  fp32 throughout.
// Thread map: 256 threads = 4 row-lanes x 64 unit-lanes; a thread owns rows
// RPT*rl..RPT*rl+RPT-1 and units ul, ul+64, ...
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int MAXU = 4;         // units per thread: H <= 256
// Rows (envs) per workgroup = 4 row-lanes x RPT rows per thread.  The recurrence is
// T dependent steps of VALU work per workgroup, so the only parallelism is across
// workgroups: RPT is chosen so that a launch has >= ~256 of them (RPT = 1 at a
// 1024-env minibatch, 4 at a 4096-env rollout) — fewer rows per thread also means
// fewer serial FMAs per step.

__device__ inline float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// LDS layout: hs[ROWS][H] (carry), then W[H][3H] if w_in_lds.
template <int RPT>
__global__ void __launch_bounds__(kThreads)
gru_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ w_h,
               const float* __restrict__ b_hn, const float* __restrict__ h0,
               const uint8_t* __restrict__ done, float* __restrict__ h_out,
               float* __restrict__ h_prev_out, float* __restrict__ gates_out,
               float* __restrict__ h_final, int64_t T, int64_t B, int H, int w_in_lds) {
  constexpr int ROWS = 4 * RPT;
  extern __shared__ float lds[];
  float* hs = lds;                 // [ROWS][H]
  float* hn = lds + ROWS * H;      // [ROWS][H] next carry
  float* wl = lds + 2 * ROWS * H;  // [H][3H]
  const int tid = threadIdx.x;
  const int ul = tid & 63, rl = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int H3 = 3 * H;
  if (w_in_lds) {
    for (int i = tid; i < H * H3; i += kThreads) wl[i] = w_h[i];
  }
  for (int i = tid; i < ROWS * H; i += kThreads) {
    const int64_t r = row0 + i / H;
    hs[i] = r < B ? h0[r * H + (i % H)] : 0.0f;
  }
  __syncthreads();
  const float* W = w_in_lds ? wl : w_h;
  const int nu = (H + 63) / 64;
  for (int64_t t = 0; t < T; ++t) {
#pragma unroll 1
    for (int ui = 0; ui < nu; ++ui) {
      const int u = ul + 64 * ui;
      if (u >= H) continue;
      float ar[RPT] = {}, az[RPT] = {}, an[RPT] = {};
      for (int k = 0; k < H; ++k) {
        const float wr = W[k * H3 + u], wz = W[k * H3 + H + u], wn = W[k * H3 + 2 * H + u];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
          const float hk = hs[(rl * RPT + q) * H + k];
          ar[q] += hk * wr;
          az[q] += hk * wz;
          an[q] += hk * wn;
        }
      }
      const float bn = b_hn[u];
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const int lr = rl * RPT + q;
        const int64_t row = row0 + lr;
        if (row >= B) continue;
        const float* g = gi + (t * B + row) * H3;
        const float hp = hs[lr * H + u];
        const float r = sigm(g[u] + ar[q]);
        const float z = sigm(g[H + u] + az[q]);
        const float qn = an[q] + bn;
        const float n = tanhf(g[2 * H + u] + r * qn);
        const float hnew = (1.0f - z) * n + z * hp;
        const int64_t o = (t * B + row) * H + u;
        h_out[o] = hnew;
        if (h_prev_out) h_prev_out[o] = hp;
        if (gates_out) {
          float* go = gates_out + (t * B + row) * 4 * H;
          go[u] = r;
          go[H + u] = z;
          go[2 * H + u] = n;
          go[3 * H + u] = qn;
        }
        const bool d = done ? done[t * B + row] != 0 : false;
        hn[lr * H + u] = d ? 0.0f : hnew;
      }
    }
    __syncthreads();
    float* tmp = hs;
    hs = hn;
    hn = tmp;
  }
  for (int i = tid; i < ROWS * H; i += kThreads) {
    const int64_t r = row0 + i / H;
    if (r < B) h_final[r * H + (i % H)] = hs[i];
  }
}

// BPTT.  Carries dh (gradient w.r.t. the post-reset state entering step t+1) in
// LDS; per step:
//   dh_tot = g_h[t] + (done[t] ? 0 : dh)            (reset cuts the carry gradient)
//   dn = dh_tot (1-z) ; dz = dh_tot (h_prev - n) ; dh_prev = dh_tot z
//   da_n = dn (1-n^2) ; dr = da_n q ; da_z = dz z(1-z) ; da_r = dr r(1-r)
//   dgi = [da_r, da_z, da_n] ; dgh = [da_r, da_z, da_n r]
//   dh = dh_prev + dgh W_h^T
// dgi / dgh are written out; dW_h = h_prev^T dgh, db_hn = colsum(dgh_n), dW_i,
// db_i, dx are time-batched GEMMs done by the dense kernels afterwards.
// LDS: dh[ROWS][H], dgh tile [ROWS][3H], W[H][3H] if it fits.
template <int RPT>
__global__ void __launch_bounds__(kThreads)
gru_bwd_kernel(const float* __restrict__ g_h, const float* __restrict__ gates,
               const float* __restrict__ h_prev, const float* __restrict__ w_h,
               const uint8_t* __restrict__ done, float* __restrict__ dgi,
               float* __restrict__ dgh, float* __restrict__ dh0, int64_t T, int64_t B, int H,
               int w_in_lds) {
  constexpr int ROWS = 4 * RPT;
  extern __shared__ float lds[];
  float* dh = lds;                    // [ROWS][H]
  float* dg = lds + ROWS * H;         // [ROWS][3H]
  float* wl = lds + ROWS * H * 4;     // [H][3H + 1]
  const int tid = threadIdx.x;
  const int ul = tid & 63, rl = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int H3 = 3 * H;
  // Phase 2 reads W[k][j] with k = the lane's unit: with the natural row stride 3H
  // (a multiple of 64 words for H = 64) all 64 lanes hit one LDS bank — a 64-way
  // conflict that made this kernel 8x slower than the forward.  Rows are padded by
  // one word in LDS so that consecutive k fall in consecutive banks.
  const int WS = w_in_lds ? H3 + 1 : H3;
  if (w_in_lds) {
    for (int i = tid; i < H * H3; i += kThreads) wl[(i / H3) * WS + i % H3] = w_h[i];
  }
  for (int i = tid; i < ROWS * H; i += kThreads) dh[i] = 0.0f;
  __syncthreads();
  const float* W = w_in_lds ? wl : w_h;
  const int nu = (H + 63) / 64;
  for (int64_t t = T - 1; t >= 0; --t) {
    // phase 1: gate gradients for owned (row, unit); dh <- dh_prev part
#pragma unroll 1
    for (int ui = 0; ui < nu; ++ui) {
      const int u = ul + 64 * ui;
      if (u >= H) continue;
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const int lr = rl * RPT + q;
        const int64_t row = row0 + lr;
        float da_r = 0.f, da_z = 0.f, da_n = 0.f, dgn = 0.f, dhp = 0.f;
        if (row < B) {
          const int64_t o = (t * B + row) * H + u;
          const float* go = gates + (t * B + row) * 4 * H;
          const float r = go[u], z = go[H + u], n = go[2 * H + u], qn = go[3 * H + u];
          const bool d = done ? done[t * B + row] != 0 : false;
          const float dht = g_h[o] + (d ? 0.0f : dh[lr * H + u]);
          const float hp = h_prev[o];
          const float dn = dht * (1.0f - z);
          const float dz = dht * (hp - n);
          dhp = dht * z;
          da_n = dn * (1.0f - n * n);
          const float dr = da_n * qn;
          da_z = dz * z * (1.0f - z);
          da_r = dr * r * (1.0f - r);
          dgn = da_n * r;
          float* gi_o = dgi + (t * B + row) * H3;
          gi_o[u] = da_r;
          gi_o[H + u] = da_z;
          gi_o[2 * H + u] = da_n;
          float* gh_o = dgh + (t * B + row) * H3;
          gh_o[u] = da_r;
          gh_o[H + u] = da_z;
          gh_o[2 * H + u] = dgn;
        }
        dg[lr * H3 + u] = da_r;
        dg[lr * H3 + H + u] = da_z;
        dg[lr * H3 + 2 * H + u] = dgn;
        dh[lr * H + u] = dhp;  // own element only: no race
      }
    }
    __syncthreads();
    // phase 2: dh[row][k] += sum_j dg[row][j] * W[k][j]   (k = owned unit index)
#pragma unroll 1
    for (int ui = 0; ui < nu; ++ui) {
      const int k = ul + 64 * ui;
      if (k >= H) continue;
      float acc[RPT] = {};
      for (int j = 0; j < H3; ++j) {
        const float w = W[k * WS + j];
#pragma unroll
        for (int q = 0; q < RPT; ++q) acc[q] += dg[(rl * RPT + q) * H3 + j] * w;
      }
#pragma unroll
      for (int q = 0; q < RPT; ++q) dh[(rl * RPT + q) * H + k] += acc[q];
    }
    __syncthreads();
  }
  if (dh0) {
    for (int i = tid; i < ROWS * H; i += kThreads) {
      const int64_t r = row0 + i / H;
      if (r < B) dh0[r * H + (i % H)] = dh[i];
    }
  }
}

}  // namespace

static size_t gru_lds_bytes(int rows, int H, int bwd, int* w_in_lds) {
  const size_t base = (size_t)(bwd ? 4 : 2) * rows * H * sizeof(float);
  const size_t w = (size_t)H * (3 * H + (bwd ? 1 : 0)) * sizeof(float);
  *w_in_lds = base + w <= 150 * 1024 ? 1 : 0;
  return base + (*w_in_lds ? w : 0);
}

static int gru_rpt(int64_t B) {
  if (B >= 16 * 256) return 4;
  if (B >= 8 * 256) return 2;
  return 1;
}

template <int RPT>
static int launch_gru_fwd(const float* gi, const float* w_h, const float* b_hn, const float* h0,
                          const uint8_t* done, float* h_out, float* h_prev_out,
                          float* gates_out, float* h_final, int64_t T, int64_t B, int H,
                          hipStream_t st) {
  int w_in_lds = 0;
  const size_t lds = gru_lds_bytes(4 * RPT, H, 0, &w_in_lds);
  static const hipError_t attr =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_fwd_kernel<RPT>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  MI_REQUIRE(attr == hipSuccess, "mi_gru_seq_fwd_f32: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL((gru_fwd_kernel<RPT>), dim3((unsigned)mippo::ceil_div(B, 4 * RPT)),
                     dim3(kThreads), lds, st, gi, w_h, b_hn, h0, done, h_out, h_prev_out,
                     gates_out, h_final, T, B, H, w_in_lds);
  return mippo::check_launch("mi_gru_seq_fwd_f32");
}

template <int RPT>
static int launch_gru_bwd(const float* g_h, const float* gates, const float* h_prev,
                          const float* w_h, const uint8_t* done, float* dgi, float* dgh,
                          float* dh0, int64_t T, int64_t B, int H, hipStream_t st) {
  int w_in_lds = 0;
  const size_t lds = gru_lds_bytes(4 * RPT, H, 1, &w_in_lds);
  static const hipError_t attr =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_bwd_kernel<RPT>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  MI_REQUIRE(attr == hipSuccess, "mi_gru_seq_bwd_f32: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL((gru_bwd_kernel<RPT>), dim3((unsigned)mippo::ceil_div(B, 4 * RPT)),
                     dim3(kThreads), lds, st, g_h, gates, h_prev, w_h, done, dgi, dgh, dh0, T, B,
                     H, w_in_lds);
  return mippo::check_launch("mi_gru_seq_bwd_f32");
}

extern "C" int mi_gru_seq_fwd_f32(const float* gi, const float* w_h, const float* b_hn,
                                  const float* h0, const uint8_t* done, float* h_out,
                                  float* h_prev_out, float* gates_out, float* h_final, int64_t T,
                                  int64_t B, int64_t H, mi_stream_t stream) {
  MI_REQUIRE(T >= 0 && B >= 0 && H >= 1 && H <= 64 * MAXU,
             "mi_gru_seq_fwd_f32: bad shape T=%lld B=%lld H=%lld (H <= %d)", (long long)T,
             (long long)B, (long long)H, 64 * MAXU);
  if (B == 0) return 0;
  MI_REQUIRE(gi || T == 0, "mi_gru_seq_fwd_f32: null gi");
  MI_REQUIRE(w_h && b_hn && h0 && h_final && (h_out || T == 0), "mi_gru_seq_fwd_f32: null pointer");
  hipStream_t st = mippo::as_stream(stream);
  switch (gru_rpt(B)) {
    case 4: return launch_gru_fwd<4>(gi, w_h, b_hn, h0, done, h_out, h_prev_out, gates_out, h_final, T, B, (int)H, st);
    case 2: return launch_gru_fwd<2>(gi, w_h, b_hn, h0, done, h_out, h_prev_out, gates_out, h_final, T, B, (int)H, st);
    default: return launch_gru_fwd<1>(gi, w_h, b_hn, h0, done, h_out, h_prev_out, gates_out, h_final, T, B, (int)H, st);
  }
}

extern "C" int mi_gru_seq_bwd_f32(const float* g_h, const float* gates, const float* h_prev,
                                  const float* w_h, const uint8_t* done, float* dgi, float* dgh,
                                  float* dh0, int64_t T, int64_t B, int64_t H,
                                  mi_stream_t stream) {
  MI_REQUIRE(T >= 1 && B >= 1 && H >= 1 && H <= 64 * MAXU, "mi_gru_seq_bwd_f32: bad shape");
  MI_REQUIRE(g_h && gates && h_prev && w_h && dgi && dgh, "mi_gru_seq_bwd_f32: null pointer");
  hipStream_t st = mippo::as_stream(stream);
  switch (gru_rpt(B)) {
    case 4: return launch_gru_bwd<4>(g_h, gates, h_prev, w_h, done, dgi, dgh, dh0, T, B, (int)H, st);
    case 2: return launch_gru_bwd<2>(g_h, gates, h_prev, w_h, done, dgi, dgh, dh0, T, B, (int)H, st);
    default: return launch_gru_bwd<1>(g_h, gates, h_prev, w_h, done, dgi, dgh, dh0, T, B, (int)H, st);
  }
}
