// a20 (bf16 path) — the GRU recurrence on the matrix cores.
//
// gru.hip evaluates h W_h with fp32 VALU FMAs: 3.7 us per time step at H = 64, so a
// 30-step sequence costs 110 us forward and more backward, and BASELINE config 4
// spends two thirds of its iteration there.  Here the per-step product is a handful
// of v_mfma_f32_16x16x32_bf16 (operands rounded to bf16, fp32 accumulation — the same
// contract as the Dense layers' bf16 path); everything else stays fp32.
//
//   * a workgroup owns 16 envs (one MFMA row tile) for ALL T steps;
//   * wave w owns the hidden units of tiles w, w+4, ... and, for each, the r, z and n
//     gate columns: the three gates of a (row, unit) land in the same lane, so the cell
//     arithmetic is lane-local, and the fp32 carry h[row][unit] never leaves registers;
//   * W_h fragments are converted once and kept in VGPRs for the whole sequence
//     (H <= 128: at most 96 registers);
//   * the bf16 image of h (the next step's A operand) ping-pongs between two LDS tiles:
//     ONE barrier per step;  gi[t+1] is in flight while step t computes.
// Backward: the same layout with the dh carry in registers; the gate-gradient tile is
// the A operand of dh = dgh . W_h^T.
// Cell arithmetic and the reset-on-done rule as in gru.hip (flax GRUCell: PARITY UNPINNED).
#include "bf16_common.h"

namespace {

using namespace mippo_bf16;

constexpr int GROWS = 16;
constexpr int MAXUT = 2;  // unit tiles per wave: H <= 128

__device__ inline float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

template <bool TRAIN>
__global__ void __launch_bounds__(kThreads)
gru_fwd_mfma_kernel(const float* __restrict__ gi, const float* __restrict__ w_h,
                    const float* __restrict__ b_hn, const float* __restrict__ h0,
                    const uint8_t* __restrict__ done, float* __restrict__ h_out,
                    float* __restrict__ h_prev_out, float* __restrict__ gates_out,
                    float* __restrict__ h_final, int64_t T, int64_t B, int H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int HROW = H + 8;
  bf16_t* hb0 = reinterpret_cast<bf16_t*>(lds_raw);  // [2][16][H + 8]
  bf16_t* hb1 = hb0 + GROWS * HROW;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * GROWS;
  const int H3 = 3 * H;
  const int KS = H / 32;
  const int UT = H / 16;

  // W_h fragments of this wave's units: B operand = W_h[k][gate*H + unit]
  bf16x8 wf[MAXUT][3][4];
#pragma unroll
  for (int ui = 0; ui < MAXUT; ++ui) {
    const int ut = wave + 4 * ui;
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = ks * 32 + 8 * lq + i;
          f[i] = (ut < UT && ks < KS) ? (bf16_t)w_h[(int64_t)k * H3 + g * H + ut * 16 + li]
                                      : (bf16_t)0.0f;
        }
        wf[ui][g][ks] = f;
      }
  }
  // carry of (row 4*lq + e, unit ut*16 + li), e = 0..3
  float h[MAXUT][4];
  float bn[MAXUT];
#pragma unroll
  for (int ui = 0; ui < MAXUT; ++ui) {
    const int ut = wave + 4 * ui;
    bn[ui] = ut < UT ? b_hn[ut * 16 + li] : 0.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t row = row0 + 4 * lq + e;
      const float v = (ut < UT && row < B) ? h0[row * H + ut * 16 + li] : 0.0f;
      h[ui][e] = v;
      if (ut < UT) hb0[(4 * lq + e) * HROW + ut * 16 + li] = (bf16_t)v;
    }
  }
  // gi of step t for the owned elements, prefetched one step ahead
  float gcur[MAXUT][3][4], gnxt[MAXUT][3][4];
  auto load_gi = [&](int64_t t, float (&dst)[MAXUT][3][4]) {
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui) {
      const int ut = wave + 4 * ui;
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t row = row0 + 4 * lq + e;
          dst[ui][g][e] = (ut < UT && row < B && t < T)
                              ? gi[(t * B + row) * H3 + g * H + ut * 16 + li]
                              : 0.0f;
        }
    }
  };
  load_gi(0, gcur);
  __syncthreads();
  bf16_t* hb = hb0;
  bf16_t* hbn = hb1;
  for (int64_t t = 0; t < T; ++t) {
    load_gi(t + 1, gnxt);
    f32x4 acc[MAXUT][3];
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui)
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[ui][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < KS) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(hb + li * HROW + ks * 32 + 8 * lq);
#pragma unroll
        for (int ui = 0; ui < MAXUT; ++ui) {
          if (wave + 4 * ui < UT) {
#pragma unroll
            for (int g = 0; g < 3; ++g)  // D[row = 4*lq + e][col = li]
              acc[ui][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf[ui][g][ks], acc[ui][g],
                                                                  0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui) {
      const int ut = wave + 4 * ui;
      if (ut >= UT) continue;
      const int u = ut * 16 + li;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int lr = 4 * lq + e;
        const int64_t row = row0 + lr;
        const float hp = h[ui][e];
        const float r = sigm(gcur[ui][0][e] + acc[ui][0][e]);
        const float z = sigm(gcur[ui][1][e] + acc[ui][1][e]);
        const float qn = acc[ui][2][e] + bn[ui];
        const float n = tanhf(gcur[ui][2][e] + r * qn);
        const float hnew = (1.0f - z) * n + z * hp;
        bool d = false;
        if (row < B) {
          const int64_t o = (t * B + row) * H + u;
          h_out[o] = hnew;
          if constexpr (TRAIN) {
            h_prev_out[o] = hp;
            float* go = gates_out + (t * B + row) * 4 * H;
            go[u] = r;
            go[H + u] = z;
            go[2 * H + u] = n;
            go[3 * H + u] = qn;
          }
          d = done ? done[t * B + row] != 0 : false;
        }
        const float hc = d ? 0.0f : hnew;
        h[ui][e] = hc;
        hbn[lr * HROW + u] = (bf16_t)hc;
      }
    }
    __syncthreads();
    bf16_t* tmp = hb;
    hb = hbn;
    hbn = tmp;
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui)
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) gcur[ui][g][e] = gnxt[ui][g][e];
  }
#pragma unroll
  for (int ui = 0; ui < MAXUT; ++ui) {
    const int ut = wave + 4 * ui;
    if (ut >= UT) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t row = row0 + 4 * lq + e;
      if (row < B) h_final[row * H + ut * 16 + li] = h[ui][e];
    }
  }
}

// BPTT (formulas in gru.hip).  dh carry in registers; dgh tile (bf16) in LDS is the A
// operand of dh_prev += dgh . W_h^T; B operand = W_h[unit][j] rows (contiguous in j).
__global__ void __launch_bounds__(kThreads)
gru_bwd_mfma_kernel(const float* __restrict__ g_h, const float* __restrict__ gates,
                    const float* __restrict__ h_prev, const float* __restrict__ w_h,
                    const uint8_t* __restrict__ done, float* __restrict__ dgi,
                    float* __restrict__ dgh, float* __restrict__ dh0, int64_t T, int64_t B,
                    int H) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int H3 = 3 * H;
  const int GROW = H3 + 8;
  bf16_t* dg0 = reinterpret_cast<bf16_t*>(lds_raw);  // [2][16][3H + 8]
  bf16_t* dg1 = dg0 + GROWS * GROW;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * GROWS;
  const int KS = H3 / 32;  // <= 12
  const int UT = H / 16;

  // W_h^T fragments: B[k_red = j][col = unit] = W_h[unit][j]
  bf16x8 wf[MAXUT][12];
#pragma unroll
  for (int ui = 0; ui < MAXUT; ++ui) {
    const int ut = wave + 4 * ui;
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) {
      bf16x8 f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        f[i] = (ut < UT && ks < KS)
                   ? (bf16_t)w_h[(int64_t)(ut * 16 + li) * H3 + ks * 32 + 8 * lq + i]
                   : (bf16_t)0.0f;
      wf[ui][ks] = f;
    }
  }
  float dh[MAXUT][4];
#pragma unroll
  for (int ui = 0; ui < MAXUT; ++ui)
#pragma unroll
    for (int e = 0; e < 4; ++e) dh[ui][e] = 0.0f;
  bf16_t* dg = dg0;
  bf16_t* dgn_buf = dg1;
  for (int64_t t = T - 1; t >= 0; --t) {
    float dhp[MAXUT][4];
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui) {
      const int ut = wave + 4 * ui;
      if (ut >= UT) continue;
      const int u = ut * 16 + li;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int lr = 4 * lq + e;
        const int64_t row = row0 + lr;
        float da_r = 0.f, da_z = 0.f, da_n = 0.f, dgn = 0.f, dp = 0.f;
        if (row < B) {
          const int64_t o = (t * B + row) * H + u;
          const float* go = gates + (t * B + row) * 4 * H;
          const float r = go[u], z = go[H + u], n = go[2 * H + u], qn = go[3 * H + u];
          const bool d = done ? done[t * B + row] != 0 : false;
          const float dht = g_h[o] + (d ? 0.0f : dh[ui][e]);
          const float hp = h_prev[o];
          const float dn = dht * (1.0f - z);
          const float dz = dht * (hp - n);
          dp = dht * z;
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
        dg[lr * GROW + u] = (bf16_t)da_r;
        dg[lr * GROW + H + u] = (bf16_t)da_z;
        dg[lr * GROW + 2 * H + u] = (bf16_t)dgn;
        dhp[ui][e] = dp;
      }
    }
    __syncthreads();
    f32x4 acc[MAXUT];
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui) acc[ui] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) {
      if (ks < KS) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(dg + li * GROW + ks * 32 + 8 * lq);
#pragma unroll
        for (int ui = 0; ui < MAXUT; ++ui)
          if (wave + 4 * ui < UT)
            acc[ui] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, wf[ui][ks], acc[ui], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui)
#pragma unroll
      for (int e = 0; e < 4; ++e) dh[ui][e] = dhp[ui][e] + acc[ui][e];
    bf16_t* tmp = dg;  // the next step writes the other tile: one barrier per step
    dg = dgn_buf;
    dgn_buf = tmp;
  }
  if (dh0) {
#pragma unroll
    for (int ui = 0; ui < MAXUT; ++ui) {
      const int ut = wave + 4 * ui;
      if (ut >= UT) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int64_t row = row0 + 4 * lq + e;
        if (row < B) dh0[row * H + ut * 16 + li] = dh[ui][e];
      }
    }
  }
}

bool mfma_shape_ok(int64_t H) { return H >= 32 && H <= 128 && H % 32 == 0; }

}  // namespace

extern "C" int mi_gru_seq_fwd_bf16(const float* gi, const float* w_h, const float* b_hn,
                                   const float* h0, const uint8_t* done, float* h_out,
                                   float* h_prev_out, float* gates_out, float* h_final,
                                   int64_t T, int64_t B, int64_t H, mi_stream_t stream) {
  MI_REQUIRE(T >= 0 && B >= 0 && mfma_shape_ok(H),
             "mi_gru_seq_fwd_bf16: bad shape T=%lld B=%lld H=%lld (H in {32, 64, 96, 128})",
             (long long)T, (long long)B, (long long)H);
  if (B == 0) return 0;
  MI_REQUIRE(gi || T == 0, "mi_gru_seq_fwd_bf16: null gi");
  MI_REQUIRE(w_h && b_hn && h0 && h_final && (h_out || T == 0),
             "mi_gru_seq_fwd_bf16: null pointer");
  MI_REQUIRE((h_prev_out == nullptr) == (gates_out == nullptr),
             "mi_gru_seq_fwd_bf16: h_prev_out and gates_out go together");
  const size_t lds = (size_t)2 * GROWS * (H + 8) * sizeof(bf16_t);
  const dim3 grid((unsigned)mippo::ceil_div(B, GROWS));
  hipStream_t st = mippo::as_stream(stream);
  if (h_prev_out) {
    hipLaunchKernelGGL((gru_fwd_mfma_kernel<true>), grid, dim3(kThreads), lds, st, gi, w_h, b_hn,
                       h0, done, h_out, h_prev_out, gates_out, h_final, T, B, (int)H);
  } else {
    hipLaunchKernelGGL((gru_fwd_mfma_kernel<false>), grid, dim3(kThreads), lds, st, gi, w_h, b_hn,
                       h0, done, h_out, h_prev_out, gates_out, h_final, T, B, (int)H);
  }
  return mippo::check_launch("mi_gru_seq_fwd_bf16");
}

extern "C" int mi_gru_seq_bwd_bf16(const float* g_h, const float* gates, const float* h_prev,
                                   const float* w_h, const uint8_t* done, float* dgi, float* dgh,
                                   float* dh0, int64_t T, int64_t B, int64_t H,
                                   mi_stream_t stream) {
  MI_REQUIRE(T >= 1 && B >= 1 && mfma_shape_ok(H), "mi_gru_seq_bwd_bf16: bad shape");
  MI_REQUIRE(g_h && gates && h_prev && w_h && dgi && dgh, "mi_gru_seq_bwd_bf16: null pointer");
  const size_t lds = (size_t)2 * GROWS * (3 * H + 8) * sizeof(bf16_t);
  hipLaunchKernelGGL(gru_bwd_mfma_kernel, dim3((unsigned)mippo::ceil_div(B, GROWS)),
                     dim3(kThreads), lds, mippo::as_stream(stream), g_h, gates, h_prev, w_h, done,
                     dgi, dgh, dh0, T, B, (int)H);
  return mippo::check_launch("mi_gru_seq_bwd_bf16");
}
