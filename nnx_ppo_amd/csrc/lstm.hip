// f2 — LSTM carry: persistent T-loop forward and BPTT backward
// (reference: nnx_ppo/networks/recurrent.py:16-161, a wrapper of flax.nnx.LSTMCell /
// OptimizedLSTMCell).  flax is third-party and not in the reference tree: the cell
// arithmetic below is its published formula and is PARITY UNPINNED:
//     a = x W_i + h W_h + b_h            (gate order i, f, g, o; bias on the hidden side)
//     i = sigmoid(a_i)  f = sigmoid(a_f)  g = tanh(a_g)  o = sigmoid(a_o)
//     c' = f c + i g ;  h' = o tanh(c')
//     carry <- done ? (h_init, c_init) : (h', c')         (reset-on-done, ppo.py:411-413;
//              h_init = c_init = 0, or the learnable initial state of
//              recurrent.py:85-88,143-161 broadcast over the batch)
// `gate_fn` (i, f, o) and `activation_fn` (g and the new cell state) default to sigmoid /
// tanh (recurrent.py:36-37); identity, relu, tanh and sigmoid are available for both —
// the ones whose derivative is a function of the OUTPUT, which is what BPTT keeps.
// gi = x W_i + b_h is one time-batched GEMM (dense kernels); this kernel adds h W_h.
// Same organisation as gru.hip: one workgroup owns 4..16 envs for ALL T steps, the
// hidden tile (and W_h when it fits) lives in LDS across the time loop; a thread owns
// RPT rows x the units ul, ul+64, ...; the cell state element of a (row, unit) is only
// ever touched by its owner.  fp32 throughout.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int MAXU = 16;  // units per thread: H <= 1024 (W_h stays in L2 once it outgrows LDS)

__device__ inline float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// MI_ACT_NONE / RELU / TANH / SIGMOID applied to a pre-activation, and the derivative
// expressed through the output value y = act(x)
__device__ inline float act_apply(int code, float x) {
  switch (code) {
    case MI_ACT_SIGMOID: return sigm(x);
    case MI_ACT_TANH: return tanhf(x);
    case MI_ACT_RELU: return x > 0.0f ? x : 0.0f;
    default: return x;
  }
}
__device__ inline float act_dfromy(int code, float y) {
  switch (code) {
    case MI_ACT_SIGMOID: return y * (1.0f - y);
    case MI_ACT_TANH: return 1.0f - y * y;
    case MI_ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
    default: return 1.0f;
  }
}

// LDS: hs[ROWS][H], hn[ROWS][H], cs[ROWS][H], then W[H][4H] if w_in_lds.
template <int RPT>
__global__ void __launch_bounds__(kThreads)
lstm_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ w_h,
                const float* __restrict__ h0, const float* __restrict__ c0,
                const uint8_t* __restrict__ done, float* __restrict__ h_out,
                float* __restrict__ h_prev_out, float* __restrict__ c_prev_out,
                float* __restrict__ gates_out,
                float* __restrict__ h_final, float* __restrict__ c_final, int64_t T, int64_t B,
                int H, int w_in_lds, const float* __restrict__ h_init,
                const float* __restrict__ c_init, int gate_act, int cell_act) {
  constexpr int ROWS = 4 * RPT;
  extern __shared__ float lds[];
  float* hs = lds;
  float* hn = lds + ROWS * H;
  float* cs = lds + 2 * ROWS * H;
  float* wl = lds + 3 * ROWS * H;
  const int tid = threadIdx.x;
  const int ul = tid & 63, rl = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int H4 = 4 * H;
  if (w_in_lds) {
    for (int i = tid; i < H * H4; i += kThreads) wl[i] = w_h[i];
  }
  for (int i = tid; i < ROWS * H; i += kThreads) {
    const int64_t r = row0 + i / H;
    hs[i] = r < B ? h0[r * H + (i % H)] : 0.0f;
    cs[i] = r < B ? c0[r * H + (i % H)] : 0.0f;
  }
  __syncthreads();
  const float* W = w_in_lds ? wl : w_h;
  const int nu = (H + 63) / 64;
  for (int64_t t = 0; t < T; ++t) {
#pragma unroll 1
    for (int ui = 0; ui < nu; ++ui) {
      const int u = ul + 64 * ui;
      if (u >= H) continue;
      float ai[RPT] = {}, af[RPT] = {}, ag[RPT] = {}, ao[RPT] = {};
      for (int k = 0; k < H; ++k) {
        const float* wk = W + k * H4 + u;
        const float wi = wk[0], wf = wk[H], wg = wk[2 * H], wo = wk[3 * H];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
          const float hk = hs[(rl * RPT + q) * H + k];
          ai[q] += hk * wi;
          af[q] += hk * wf;
          ag[q] += hk * wg;
          ao[q] += hk * wo;
        }
      }
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const int lr = rl * RPT + q;
        const int64_t row = row0 + lr;
        if (row >= B) continue;
        const float* g = gi + (t * B + row) * H4;
        const float cp = cs[lr * H + u];
        const float i_ = act_apply(gate_act, g[u] + ai[q]);
        const float f_ = act_apply(gate_act, g[H + u] + af[q]);
        const float g_ = act_apply(cell_act, g[2 * H + u] + ag[q]);
        const float o_ = act_apply(gate_act, g[3 * H + u] + ao[q]);
        const float cn = f_ * cp + i_ * g_;
        const float tc = act_apply(cell_act, cn);
        const float hnew = o_ * tc;
        const int64_t o = (t * B + row) * H + u;
        h_out[o] = hnew;
        if (h_prev_out) h_prev_out[o] = hs[lr * H + u];
        if (c_prev_out) c_prev_out[o] = cp;
        if (gates_out) {
          float* go = gates_out + (t * B + row) * 5 * H;
          go[u] = i_;
          go[H + u] = f_;
          go[2 * H + u] = g_;
          go[3 * H + u] = o_;
          go[4 * H + u] = tc;
        }
        const bool d = done ? done[t * B + row] != 0 : false;
        hn[lr * H + u] = d ? (h_init ? h_init[u] : 0.0f) : hnew;
        cs[lr * H + u] = d ? (c_init ? c_init[u] : 0.0f) : cn;  // own element only
      }
    }
    __syncthreads();
    float* tmp = hs;
    hs = hn;
    hn = tmp;
  }
  for (int i = tid; i < ROWS * H; i += kThreads) {
    const int64_t r = row0 + i / H;
    if (r < B) {
      h_final[r * H + (i % H)] = hs[i];
      c_final[r * H + (i % H)] = cs[i];
    }
  }
}

// BPTT.  Carries dh, dc (gradients w.r.t. the post-reset carry entering step t+1);
// per step, with tc = tanh(c'):
//   dh_tot = g_h[t] + (done[t] ? 0 : dh)
//   dc_tot = (done[t] ? 0 : dc) + dh_tot o (1 - tc^2)
//   da_o = dh_tot tc o(1-o) ; da_i = dc_tot g i(1-i) ; da_g = dc_tot i (1-g^2)
//   da_f = dc_tot c_prev f(1-f) ; dc = dc_tot f ; dh = [da_i da_f da_g da_o] W_h^T
// (written for sigmoid / tanh; in general y(1-y) and 1-y^2 are act_dfromy of the gate /
// cell function).  With a learnable initial state the carry entering step t+1 is
// (h_init, c_init) wherever done[t]: the carried (dh, dc) of those rows are what the
// initial state receives — summed per workgroup into dinit_part[block][2][H] (fixed
// order; the host adds the blocks).
// The gate gradients da (= d/d gi = d/d (h W_h)) are written out; dW_h = h_prev^T da,
// db_h = colsum(da), dW_i and dx are time-batched GEMMs done by the dense kernels.
// LDS: dh[ROWS][H], dc[ROWS][H], da tile [ROWS][4H], W[H][4H + 1] if it fits (rows
// padded by one word: phase 2 reads W[k][j] with k = the lane's unit, see gru.hip).
template <int RPT>
__global__ void __launch_bounds__(kThreads)
lstm_bwd_kernel(const float* __restrict__ g_h, const float* __restrict__ gates,
                const float* __restrict__ c_prev, const float* __restrict__ w_h,
                const uint8_t* __restrict__ done, float* __restrict__ da_out,
                float* __restrict__ dh0, float* __restrict__ dc0, int64_t T, int64_t B, int H,
                int w_in_lds, float* __restrict__ dinit_part, int gate_act, int cell_act) {
  constexpr int ROWS = 4 * RPT;
  extern __shared__ float lds[];
  float* dh = lds;
  float* dc = lds + ROWS * H;
  float* dg = lds + 2 * ROWS * H;  // [ROWS][4H]
  float* wl = lds + 6 * ROWS * H;
  const int tid = threadIdx.x;
  const int ul = tid & 63, rl = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int H4 = 4 * H;
  const int WS = w_in_lds ? H4 + 1 : H4;
  if (w_in_lds) {
    for (int i = tid; i < H * H4; i += kThreads) wl[(i / H4) * WS + i % H4] = w_h[i];
  }
  for (int i = tid; i < ROWS * H; i += kThreads) {
    dh[i] = 0.0f;
    dc[i] = 0.0f;
  }
  __syncthreads();
  const float* W = w_in_lds ? wl : w_h;
  const int nu = (H + 63) / 64;
  float init_h[MAXU] = {}, init_c[MAXU] = {};  // this thread's rows, units ul + 64 ui
  for (int64_t t = T - 1; t >= 0; --t) {
#pragma unroll 1
    for (int ui = 0; ui < nu; ++ui) {
      const int u = ul + 64 * ui;
      if (u >= H) continue;
#pragma unroll
      for (int q = 0; q < RPT; ++q) {
        const int lr = rl * RPT + q;
        const int64_t row = row0 + lr;
        float da_i = 0.f, da_f = 0.f, da_g = 0.f, da_o = 0.f, dcp = 0.f;
        if (row < B) {
          const int64_t o = (t * B + row) * H + u;
          const float* go = gates + (t * B + row) * 5 * H;
          const float i_ = go[u], f_ = go[H + u], g_ = go[2 * H + u], o_ = go[3 * H + u],
                      tc = go[4 * H + u];
          const bool d = done ? done[t * B + row] != 0 : false;
          if (d && dinit_part) {  // the reset carry of step t+1 is the learnable initial state
#pragma unroll
            for (int k = 0; k < MAXU; ++k)
              if (k == ui) {
                init_h[k] += dh[lr * H + u];
                init_c[k] += dc[lr * H + u];
              }
          }
          const float dht = g_h[o] + (d ? 0.0f : dh[lr * H + u]);
          const float dct = (d ? 0.0f : dc[lr * H + u]) + dht * o_ * act_dfromy(cell_act, tc);
          da_o = dht * tc * act_dfromy(gate_act, o_);
          da_i = dct * g_ * act_dfromy(gate_act, i_);
          da_g = dct * i_ * act_dfromy(cell_act, g_);
          da_f = dct * c_prev[o] * act_dfromy(gate_act, f_);
          dcp = dct * f_;
          float* ao = da_out + (t * B + row) * H4;
          ao[u] = da_i;
          ao[H + u] = da_f;
          ao[2 * H + u] = da_g;
          ao[3 * H + u] = da_o;
        }
        dg[lr * H4 + u] = da_i;
        dg[lr * H4 + H + u] = da_f;
        dg[lr * H4 + 2 * H + u] = da_g;
        dg[lr * H4 + 3 * H + u] = da_o;
        dc[lr * H + u] = dcp;  // own element only: no race
      }
    }
    __syncthreads();
    // dh[row][k] = sum_j da[row][j] * W[k][j]   (k = owned unit index)
#pragma unroll 1
    for (int ui = 0; ui < nu; ++ui) {
      const int k = ul + 64 * ui;
      if (k >= H) continue;
      float acc[RPT] = {};
      for (int j = 0; j < H4; ++j) {
        const float w = W[k * WS + j];
#pragma unroll
        for (int q = 0; q < RPT; ++q) acc[q] += dg[(rl * RPT + q) * H4 + j] * w;
      }
#pragma unroll
      for (int q = 0; q < RPT; ++q) dh[(rl * RPT + q) * H + k] = acc[q];
    }
    __syncthreads();
  }
  for (int i = tid; i < ROWS * H; i += kThreads) {
    const int64_t r = row0 + i / H;
    if (r < B) {
      if (dh0) dh0[r * H + (i % H)] = dh[i];
      if (dc0) dc0[r * H + (i % H)] = dc[i];
    }
  }
  if (dinit_part) {
    // sum the four row groups (waves) in wave order through LDS: dg is free now
    __syncthreads();
    for (int ui = 0; ui < nu; ++ui) {
      const int u = ul + 64 * ui;
      if (u >= H) continue;
#pragma unroll
      for (int k = 0; k < MAXU; ++k)
        if (k == ui) {
          dg[rl * 2 * H + u] = init_h[k];
          dg[rl * 2 * H + H + u] = init_c[k];
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * H; i += kThreads)
      dinit_part[(int64_t)blockIdx.x * 2 * H + i] =
          ((dg[i] + dg[2 * H + i]) + dg[4 * H + i]) + dg[6 * H + i];
  }
}

size_t lstm_lds_bytes(int rows, int H, int bwd, int* w_in_lds) {
  const size_t base = (size_t)(bwd ? 6 : 3) * rows * H * sizeof(float);
  const size_t w = (size_t)H * (4 * H + (bwd ? 1 : 0)) * sizeof(float);
  *w_in_lds = base + w <= 150 * 1024 ? 1 : 0;
  return base + (*w_in_lds ? w : 0);
}

int lstm_rpt(int64_t B, int H) {
  // rows per thread by batch size (>= 256 workgroups), capped so the carry / gate tiles
  // of the backward (6 x ROWS x H floats) stay inside LDS for wide cells
  int rpt = B >= 16 * 256 ? 4 : (B >= 8 * 256 ? 2 : 1);
  while (rpt > 1 && (size_t)6 * 4 * rpt * H * sizeof(float) > 96 * 1024) rpt >>= 1;
  return rpt;
}

template <int RPT>
int launch_fwd(const float* gi, const float* w_h, const float* h0, const float* c0,
               const uint8_t* done, float* h_out, float* h_prev_out, float* c_prev_out,
               float* gates_out, float* h_final, float* c_final, int64_t T, int64_t B, int H,
               const float* h_init, const float* c_init, int gate_act, int cell_act,
               hipStream_t st) {
  int w_in_lds = 0;
  const size_t lds = lstm_lds_bytes(4 * RPT, H, 0, &w_in_lds);
  static const hipError_t attr =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_kernel<RPT>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  MI_REQUIRE(attr == hipSuccess, "mi_lstm_seq_fwd_f32: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL((lstm_fwd_kernel<RPT>), dim3((unsigned)mippo::ceil_div(B, 4 * RPT)),
                     dim3(kThreads), lds, st, gi, w_h, h0, c0, done, h_out, h_prev_out, c_prev_out,
                     gates_out, h_final, c_final, T, B, H, w_in_lds, h_init, c_init, gate_act,
                     cell_act);
  return mippo::check_launch("mi_lstm_seq_fwd_f32");
}

template <int RPT>
int launch_bwd(const float* g_h, const float* gates, const float* c_prev, const float* w_h,
               const uint8_t* done, float* da, float* dh0, float* dc0, int64_t T, int64_t B,
               int H, float* dinit_part, int gate_act, int cell_act, hipStream_t st) {
  int w_in_lds = 0;
  const size_t lds = lstm_lds_bytes(4 * RPT, H, 1, &w_in_lds);
  static const hipError_t attr =
      hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_kernel<RPT>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  MI_REQUIRE(attr == hipSuccess, "mi_lstm_seq_bwd_f32: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL((lstm_bwd_kernel<RPT>), dim3((unsigned)mippo::ceil_div(B, 4 * RPT)),
                     dim3(kThreads), lds, st, g_h, gates, c_prev, w_h, done, da, dh0, dc0, T, B, H,
                     w_in_lds, dinit_part, gate_act, cell_act);
  return mippo::check_launch("mi_lstm_seq_bwd_f32");
}

}  // namespace

namespace {
bool lstm_act_ok(int a) {
  return a == MI_ACT_NONE || a == MI_ACT_RELU || a == MI_ACT_TANH || a == MI_ACT_SIGMOID;
}
}  // namespace

extern "C" int64_t mi_lstm_seq_bwd_blocks(int64_t B, int64_t H) {
  if (B < 1 || H < 1 || H > 64 * MAXU) return -EINVAL;
  return mippo::ceil_div(B, 4 * lstm_rpt(B, (int)H));
}

extern "C" int mi_lstm_seq_fwd_f32(const float* gi, const float* w_h, const float* h0,
                                   const float* c0, const uint8_t* done, float* h_out,
                                   float* h_prev_out, float* c_prev_out, float* gates_out,
                                   float* h_final, float* c_final, const float* h_init,
                                   const float* c_init, int gate_act, int cell_act, int64_t T,
                                   int64_t B, int64_t H, mi_stream_t stream) {
  MI_REQUIRE(T >= 0 && B >= 0 && H >= 1 && H <= 64 * MAXU,
             "mi_lstm_seq_fwd_f32: bad shape T=%lld B=%lld H=%lld (H <= %d)", (long long)T,
             (long long)B, (long long)H, 64 * MAXU);
  if (B == 0) return 0;
  MI_REQUIRE(gi || T == 0, "mi_lstm_seq_fwd_f32: null gi");
  MI_REQUIRE(w_h && h0 && c0 && h_final && c_final && (h_out || T == 0),
             "mi_lstm_seq_fwd_f32: null pointer");
  MI_REQUIRE((h_init == nullptr) == (c_init == nullptr),
             "mi_lstm_seq_fwd_f32: h_init and c_init go together");
  MI_REQUIRE(lstm_act_ok(gate_act) && lstm_act_ok(cell_act),
             "mi_lstm_seq_fwd_f32: gate / cell functions: none, relu, tanh or sigmoid");
  hipStream_t st = mippo::as_stream(stream);
  switch (lstm_rpt(B, (int)H)) {
    case 4:
      return launch_fwd<4>(gi, w_h, h0, c0, done, h_out, h_prev_out, c_prev_out, gates_out,
                           h_final, c_final, T, B, (int)H, h_init, c_init, gate_act, cell_act, st);
    case 2:
      return launch_fwd<2>(gi, w_h, h0, c0, done, h_out, h_prev_out, c_prev_out, gates_out,
                           h_final, c_final, T, B, (int)H, h_init, c_init, gate_act, cell_act, st);
    default:
      return launch_fwd<1>(gi, w_h, h0, c0, done, h_out, h_prev_out, c_prev_out, gates_out,
                           h_final, c_final, T, B, (int)H, h_init, c_init, gate_act, cell_act, st);
  }
}

extern "C" int mi_lstm_seq_bwd_f32(const float* g_h, const float* gates, const float* c_prev,
                                   const float* w_h, const uint8_t* done, float* d_gates,
                                   float* dh0, float* dc0, float* dinit_part, int gate_act,
                                   int cell_act, int64_t T, int64_t B, int64_t H,
                                   mi_stream_t stream) {
  MI_REQUIRE(T >= 1 && B >= 1 && H >= 1 && H <= 64 * MAXU, "mi_lstm_seq_bwd_f32: bad shape");
  MI_REQUIRE(g_h && gates && c_prev && w_h && d_gates, "mi_lstm_seq_bwd_f32: null pointer");
  MI_REQUIRE(lstm_act_ok(gate_act) && lstm_act_ok(cell_act),
             "mi_lstm_seq_bwd_f32: gate / cell functions: none, relu, tanh or sigmoid");
  hipStream_t st = mippo::as_stream(stream);
  switch (lstm_rpt(B, (int)H)) {
    case 4:
      return launch_bwd<4>(g_h, gates, c_prev, w_h, done, d_gates, dh0, dc0, T, B, (int)H,
                           dinit_part, gate_act, cell_act, st);
    case 2:
      return launch_bwd<2>(g_h, gates, c_prev, w_h, done, d_gates, dh0, dc0, T, B, (int)H,
                           dinit_part, gate_act, cell_act, st);
    default:
      return launch_bwd<1>(g_h, gates, c_prev, w_h, done, d_gates, dh0, dc0, T, B, (int)H,
                           dinit_part, gate_act, cell_act, st);
  }
}
