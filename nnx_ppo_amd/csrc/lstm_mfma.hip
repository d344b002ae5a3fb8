// f2 (bf16 path) — the LSTM recurrence on the matrix cores: lstm.hip's cell
// (nnx_ppo/networks/recurrent.py:16-161, flax LSTMCell; PARITY UNPINNED) with h W_h and
// d_gates W_h^T as v_mfma_f32_16x16x32_bf16 products (operands rounded to bf16, fp32
// accumulation, fp32 cell arithmetic and carries).  Organisation as gru_mfma.hip: a
// workgroup owns 16 envs for all T steps; wave w owns the units of tiles w, w+4, ... and,
// for each, the i, f, g, o gate columns — the four gates of a (row, unit) land in one
// lane, the fp32 carries h, c stay in registers; W_h fragments stay in VGPRs for the whole
// sequence; per-step operands come from a register ring; no per-element branches.
#include "bf16_common.h"

namespace {

using namespace mippo_bf16;

constexpr int LROWS = 16;

template <bool TRAIN, int UTW>
__global__ void __launch_bounds__(kThreads)
lstm_fwd_mfma_kernel(const float* __restrict__ gi, const float* __restrict__ w_h,
                     const float* __restrict__ h0, const float* __restrict__ c0,
                     const uint8_t* __restrict__ done, float* __restrict__ h_out,
                     float* __restrict__ h_prev_out, float* __restrict__ c_prev_out,
                     float* __restrict__ gates_out, float* __restrict__ h_final,
                     float* __restrict__ c_final, int64_t T, int64_t B, int H) {
  constexpr int PFW = UTW == 1 ? 4 : 1;  // steps of look-ahead that fit the registers
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int HROW = H + 8;
  bf16_t* hb0 = reinterpret_cast<bf16_t*>(lds_raw);  // [2][16][H + 8]
  bf16_t* hb1 = hb0 + LROWS * HROW;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * LROWS;
  const int H4 = 4 * H;
  const int KS = H / 32;
  const int UT = H / 16;

  bf16x8 wf[UTW][4][4];  // B operand = W_h[k][gate*H + unit]
#pragma unroll
  for (int ui = 0; ui < UTW; ++ui) {
    const int ut = wave + 4 * ui;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = ks * 32 + 8 * lq + i;
          f[i] = (ut < UT && ks < KS) ? (bf16_t)w_h[(int64_t)k * H4 + g * H + ut * 16 + li]
                                      : (bf16_t)0.0f;
        }
        wf[ui][g][ks] = f;
      }
  }
  bool valid[4];
  unsigned rowc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t row = row0 + 4 * lq + e;
    valid[e] = row < B;
    rowc[e] = (unsigned)(valid[e] ? row : B - 1);
  }
  float h[UTW][4], c[UTW][4];
  unsigned unit[UTW];
#pragma unroll
  for (int ui = 0; ui < UTW; ++ui) {
    const int ut = wave + 4 * ui;
    const bool on = ut < UT;
    unit[ui] = (unsigned)((on ? ut : 0) * 16 + li);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const unsigned o = rowc[e] * (unsigned)H + unit[ui];
      h[ui][e] = (on && valid[e]) ? h0[o] : 0.0f;
      c[ui][e] = (on && valid[e]) ? c0[o] : 0.0f;
      if (on) hb0[(4 * lq + e) * HROW + unit[ui]] = (bf16_t)h[ui][e];
    }
  }
  const int64_t last_t = T - 1;
  float gq[PFW][UTW][4][4];
  float dq[PFW][4];
  auto load_step = [&](int64_t t, float (&dst)[UTW][4][4], float (&dn)[4]) {
    const int64_t tc = t < last_t ? t : last_t;
    const float* gt = gi + tc * B * H4;
    const uint8_t* dt = done ? done + tc * B : nullptr;
#pragma unroll
    for (int e = 0; e < 4; ++e) dn[e] = (dt && dt[rowc[e]] != 0) ? 1.0f : 0.0f;
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          dst[ui][g][e] = gt[rowc[e] * (unsigned)H4 + (unsigned)(g * H) + unit[ui]];
  };
#pragma unroll
  for (int d = 0; d < PFW; ++d) load_step(d, gq[d], dq[d]);
  __syncthreads();
  bf16_t* hb = hb0;
  bf16_t* hbn = hb1;
  auto step = [&](int64_t t, float (&gcur)[UTW][4][4], float (&dcur)[4]) {
    f32x4 acc[UTW][4];
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[ui][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < KS) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(hb + li * HROW + ks * 32 + 8 * lq);
#pragma unroll
        for (int ui = 0; ui < UTW; ++ui)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            acc[ui][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf[ui][g][ks], acc[ui][g],
                                                                0, 0, 0);
      }
    }
    float* ho = h_out + t * B * H;
    float* hpo = TRAIN ? h_prev_out + t * B * H : nullptr;
    float* cpo = TRAIN ? c_prev_out + t * B * H : nullptr;
    float* gto = TRAIN ? gates_out + t * B * 5 * H : nullptr;
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui) {
      if (wave + 4 * ui >= UT) continue;  // wave-uniform
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float hp = h[ui][e], cp = c[ui][e];
        const float i_ = fast_sigmoid(gcur[ui][0][e] + acc[ui][0][e]);
        const float f_ = fast_sigmoid(gcur[ui][1][e] + acc[ui][1][e]);
        const float g_ = fast_tanh(gcur[ui][2][e] + acc[ui][2][e]);
        const float o_ = fast_sigmoid(gcur[ui][3][e] + acc[ui][3][e]);
        const float cn = f_ * cp + i_ * g_;
        const float tcn = fast_tanh(cn);
        const float hnew = o_ * tcn;
        if (valid[e]) {
          const unsigned o = rowc[e] * (unsigned)H + unit[ui];
          ho[o] = hnew;
          if constexpr (TRAIN) {
            hpo[o] = hp;
            cpo[o] = cp;
            const unsigned og = rowc[e] * (unsigned)(5 * H) + unit[ui];
            gto[og] = i_;
            gto[og + (unsigned)H] = f_;
            gto[og + (unsigned)(2 * H)] = g_;
            gto[og + (unsigned)(3 * H)] = o_;
            gto[og + (unsigned)(4 * H)] = tcn;
          }
        }
        const bool d = dcur[e] != 0.0f;
        h[ui][e] = d ? 0.0f : hnew;
        c[ui][e] = d ? 0.0f : cn;
        hbn[(4 * lq + e) * HROW + unit[ui]] = (bf16_t)h[ui][e];
      }
    }
    __syncthreads();
    bf16_t* tmp = hb;
    hb = hbn;
    hbn = tmp;
    load_step(t + PFW, gcur, dcur);
  };
  for (int64_t t0 = 0; t0 < T; t0 += PFW) {
#pragma unroll
    for (int d = 0; d < PFW; ++d)
      if (t0 + d < T) step(t0 + d, gq[d], dq[d]);
  }
#pragma unroll
  for (int ui = 0; ui < UTW; ++ui) {
    if (wave + 4 * ui >= UT) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (valid[e]) {
        h_final[rowc[e] * (unsigned)H + unit[ui]] = h[ui][e];
        c_final[rowc[e] * (unsigned)H + unit[ui]] = c[ui][e];
      }
  }
}

// BPTT (formulas in lstm.hip).  dh, dc carries in registers; the gate-gradient tile
// (bf16, LDS) is the A operand of dh = d_gates . W_h^T.
template <int UTW>
__global__ void __launch_bounds__(kThreads)
lstm_bwd_mfma_kernel(const float* __restrict__ g_h, const float* __restrict__ gates,
                     const float* __restrict__ c_prev, const float* __restrict__ w_h,
                     const uint8_t* __restrict__ done, float* __restrict__ da_out,
                     float* __restrict__ dh0, float* __restrict__ dc0, int64_t T, int64_t B,
                     int H) {
  constexpr int PFW = UTW == 1 ? 4 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int H4 = 4 * H;
  const int GROW = H4 + 8;
  bf16_t* dg0 = reinterpret_cast<bf16_t*>(lds_raw);  // [2][16][4H + 8]
  bf16_t* dg1 = dg0 + LROWS * GROW;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * LROWS;
  const int KS = H4 / 32;  // <= 16
  const int UT = H / 16;

  bf16x8 wf[UTW][16];  // B[k_red = j][col = unit] = W_h[unit][j]
#pragma unroll
  for (int ui = 0; ui < UTW; ++ui) {
    const int ut = wave + 4 * ui;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      bf16x8 f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        f[i] = (ut < UT && ks < KS)
                   ? (bf16_t)w_h[(int64_t)(ut * 16 + li) * H4 + ks * 32 + 8 * lq + i]
                   : (bf16_t)0.0f;
      wf[ui][ks] = f;
    }
  }
  bool valid[4];
  unsigned rowc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t row = row0 + 4 * lq + e;
    valid[e] = row < B;
    rowc[e] = (unsigned)(valid[e] ? row : B - 1);
  }
  unsigned unit[UTW];
  float dh[UTW][4], dc[UTW][4];
#pragma unroll
  for (int ui = 0; ui < UTW; ++ui) {
    const int ut = wave + 4 * ui;
    unit[ui] = (unsigned)((ut < UT ? ut : 0) * 16 + li);
#pragma unroll
    for (int e = 0; e < 4; ++e) dh[ui][e] = dc[ui][e] = 0.0f;
  }
  struct In {
    float v[UTW][4][7];  // i, f, g, o, tanh(c'), g_h, c_prev
    float dn[4];
  };
  In inq[PFW];
  auto load_in = [&](int64_t t, In& dst) {
    const int64_t tc = t > 0 ? t : 0;
    const float* gt = gates + tc * B * 5 * H;
    const float* ght = g_h + tc * B * H;
    const float* cpt = c_prev + tc * B * H;
    const uint8_t* dt = done ? done + tc * B : nullptr;
#pragma unroll
    for (int e = 0; e < 4; ++e) dst.dn[e] = (dt && dt[rowc[e]] != 0) ? 1.0f : 0.0f;
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned og = rowc[e] * (unsigned)(5 * H) + unit[ui];
        const unsigned o = rowc[e] * (unsigned)H + unit[ui];
#pragma unroll
        for (int g = 0; g < 5; ++g) dst.v[ui][e][g] = gt[og + (unsigned)(g * H)];
        dst.v[ui][e][5] = ght[o];
        dst.v[ui][e][6] = cpt[o];
      }
  };
#pragma unroll
  for (int d = 0; d < PFW; ++d) load_in(T - 1 - d, inq[d]);
  bf16_t* dg = dg0;
  bf16_t* dgn_buf = dg1;
  auto step = [&](int64_t t, In& in) {
    float* ao = da_out + t * B * H4;
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui) {
      if (wave + 4 * ui >= UT) continue;  // wave-uniform
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int lr = 4 * lq + e;
        const float i_ = in.v[ui][e][0], f_ = in.v[ui][e][1], g_ = in.v[ui][e][2],
                    o_ = in.v[ui][e][3], tc = in.v[ui][e][4];
        const bool d = in.dn[e] != 0.0f;
        const float dht = in.v[ui][e][5] + (d ? 0.0f : dh[ui][e]);
        const float dct = (d ? 0.0f : dc[ui][e]) + dht * o_ * (1.0f - tc * tc);
        float da_o = dht * tc * o_ * (1.0f - o_);
        float da_i = dct * g_ * i_ * (1.0f - i_);
        float da_g = dct * i_ * (1.0f - g_ * g_);
        float da_f = dct * in.v[ui][e][6] * f_ * (1.0f - f_);
        float dcp = dct * f_;
        if (valid[e]) {
          const unsigned o4 = rowc[e] * (unsigned)H4 + unit[ui];
          ao[o4] = da_i;
          ao[o4 + (unsigned)H] = da_f;
          ao[o4 + (unsigned)(2 * H)] = da_g;
          ao[o4 + (unsigned)(3 * H)] = da_o;
        } else {
          da_i = da_f = da_g = da_o = dcp = 0.0f;
        }
        dg[lr * GROW + unit[ui]] = (bf16_t)da_i;
        dg[lr * GROW + H + unit[ui]] = (bf16_t)da_f;
        dg[lr * GROW + 2 * H + unit[ui]] = (bf16_t)da_g;
        dg[lr * GROW + 3 * H + unit[ui]] = (bf16_t)da_o;
        dc[ui][e] = dcp;
      }
    }
    load_in(t - PFW, in);
    __syncthreads();
    f32x4 acc[UTW];
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui) acc[ui] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks < KS) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(dg + li * GROW + ks * 32 + 8 * lq);
#pragma unroll
        for (int ui = 0; ui < UTW; ++ui)
          acc[ui] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf[ui][ks], acc[ui], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui)
#pragma unroll
      for (int e = 0; e < 4; ++e) dh[ui][e] = acc[ui][e];
    bf16_t* tmp = dg;
    dg = dgn_buf;
    dgn_buf = tmp;
  };
  for (int64_t t0 = T - 1; t0 >= 0; t0 -= PFW) {
#pragma unroll
    for (int d = 0; d < PFW; ++d)
      if (t0 - d >= 0) step(t0 - d, inq[d]);
  }
#pragma unroll
  for (int ui = 0; ui < UTW; ++ui) {
    if (wave + 4 * ui >= UT) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (valid[e]) {
        if (dh0) dh0[rowc[e] * (unsigned)H + unit[ui]] = dh[ui][e];
        if (dc0) dc0[rowc[e] * (unsigned)H + unit[ui]] = dc[ui][e];
      }
  }
}

bool mfma_shape_ok(int64_t H) { return H >= 32 && H <= 128 && H % 32 == 0; }

}  // namespace

extern "C" int mi_lstm_seq_fwd_bf16(const float* gi, const float* w_h, const float* h0,
                                    const float* c0, const uint8_t* done, float* h_out,
                                    float* h_prev_out, float* c_prev_out, float* gates_out,
                                    float* h_final, float* c_final, int64_t T, int64_t B,
                                    int64_t H, mi_stream_t stream) {
  MI_REQUIRE(T >= 0 && B >= 0 && mfma_shape_ok(H) && B * 5 * H < (1LL << 31),
             "mi_lstm_seq_fwd_bf16: bad shape T=%lld B=%lld H=%lld (H in {32, 64, 96, 128})",
             (long long)T, (long long)B, (long long)H);
  if (B == 0) return 0;
  MI_REQUIRE(gi || T == 0, "mi_lstm_seq_fwd_bf16: null gi");
  MI_REQUIRE(w_h && h0 && c0 && h_final && c_final && (h_out || T == 0),
             "mi_lstm_seq_fwd_bf16: null pointer");
  const bool train = h_prev_out != nullptr;
  MI_REQUIRE(train == (c_prev_out != nullptr) && train == (gates_out != nullptr),
             "mi_lstm_seq_fwd_bf16: h_prev_out, c_prev_out and gates_out go together");
  const size_t lds = (size_t)2 * LROWS * (H + 8) * sizeof(bf16_t);
  const dim3 grid((unsigned)mippo::ceil_div(B, LROWS));
  hipStream_t st = mippo::as_stream(stream);
#define MI_LSTM_FWD(TRAIN, UTW)                                                                \
  hipLaunchKernelGGL((lstm_fwd_mfma_kernel<TRAIN, UTW>), grid, dim3(kThreads), lds, st, gi,    \
                     w_h, h0, c0, done, h_out, h_prev_out, c_prev_out, gates_out, h_final,     \
                     c_final, T, B, (int)H)
  if (train) {
    if (H <= 64) MI_LSTM_FWD(true, 1); else MI_LSTM_FWD(true, 2);
  } else {
    if (H <= 64) MI_LSTM_FWD(false, 1); else MI_LSTM_FWD(false, 2);
  }
#undef MI_LSTM_FWD
  return mippo::check_launch("mi_lstm_seq_fwd_bf16");
}

extern "C" int mi_lstm_seq_bwd_bf16(const float* g_h, const float* gates, const float* c_prev,
                                    const float* w_h, const uint8_t* done, float* d_gates,
                                    float* dh0, float* dc0, int64_t T, int64_t B, int64_t H,
                                    mi_stream_t stream) {
  MI_REQUIRE(T >= 1 && B >= 1 && mfma_shape_ok(H) && B * 5 * H < (1LL << 31),
             "mi_lstm_seq_bwd_bf16: bad shape");
  MI_REQUIRE(g_h && gates && c_prev && w_h && d_gates, "mi_lstm_seq_bwd_bf16: null pointer");
  const size_t lds = (size_t)2 * LROWS * (4 * H + 8) * sizeof(bf16_t);
  const dim3 grid((unsigned)mippo::ceil_div(B, LROWS));
  hipStream_t st = mippo::as_stream(stream);
  if (H <= 64) {
    hipLaunchKernelGGL((lstm_bwd_mfma_kernel<1>), grid, dim3(kThreads), lds, st, g_h, gates,
                       c_prev, w_h, done, d_gates, dh0, dc0, T, B, (int)H);
  } else {
    hipLaunchKernelGGL((lstm_bwd_mfma_kernel<2>), grid, dim3(kThreads), lds, st, g_h, gates,
                       c_prev, w_h, done, d_gates, dh0, dc0, T, B, (int)H);
  }
  return mippo::check_launch("mi_lstm_seq_bwd_bf16");
}
