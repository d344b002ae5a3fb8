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
#include "sampler_math.h"

namespace {

using namespace mippo_bf16;

constexpr int GROWS = 16;
// Look-ahead for the per-step global operands (gi forward; gates, g_h, h_prev backward).
// They do not depend on the recurrence, and a step's arithmetic (~0.3 us) is far shorter
// than an HBM round trip (~2 us): the time loop is unrolled PFW times over a ring of register
// slots (PFW per kernel, see there: it is bounded by the 63 memory instructions vmcnt can
// leave in flight).


// Addressing: a lane's element offsets (row * stride + unit) are fixed for the whole
// sequence and fit 32 bits; per step only a wave-uniform base pointer moves.  Rows past
// B (last workgroup) are clamped for loads and masked for stores, so the step body has
// no per-element branches — with one wave per SIMD every instruction's latency is
// exposed and the first version of this kernel spent ~2000 instructions per step on
// predicates and 64-bit address arithmetic.
// H (32 / 64 / 96 / 128) is a template parameter, GUARD (rows past B exist: B % 16 != 0) and
// BF (write the bf16 image of h_prev) too: every data-dependent or pointer-dependent branch
// around a memory instruction makes the compiler's wait counting take the path with the
// FEWEST younger instructions, i.e. wait for almost everything — the steady state below
// must be straight-line code.
// TAIL (loss replay of make_gru_actor_critic's actor): the layers BEHIND the recurrence —
// Dense(H -> 2A) and the tanh-Gaussian sampler scoring the stored actions
// (`feedforward.py:42-51`, `sampling_layers.py:82-147`) — ride in this launch.  They do not
// depend on the recurrence beyond h_t, and one thread per row of a transcendental chain has
// no place inside the time loop: the bf16 image of every h_t stays in LDS (T x 16 rows), and
// after the loop the workgroup multiplies its T row tiles by the head's weights and samples
// its T x 16 rows in parallel.  Two launches (a 1024-row-tile chain walk and the sampler)
// leave the critical path of every gradient step; same MFMA tiles, same k order, same row
// function: bit-identical to them.
struct GruTail {
  const bf16_t* wo;   // forward fragment-major image of the head's kernel [H -> N_out <= 16]
  const float* bo;    // [N_out] or null
  float* ms_out;      // [T * B][N_out] fp32: the head's rows (the sampler backward's input)
  bf16_t* h_bf;       // [T * B][H]: bf16 image of h_out — the x operand of the head's dW
  mippo_sampler::FwdParams samp;  // rows are t * B + env
  int N_out;
};

template <bool TRAIN, int H, bool GUARD, bool BF, bool TAIL = false>
__global__ void __launch_bounds__(kThreads)
gru_fwd_mfma_kernel(const float* __restrict__ gi, const float* __restrict__ w_h,
                    const float* __restrict__ b_hn, const float* __restrict__ h0,
                    const uint8_t* __restrict__ done, float* __restrict__ h_out,
                    float* __restrict__ h_prev_out, float* __restrict__ gates_out,
                    float* __restrict__ h_final, bf16_t* __restrict__ h_prev_bf, int64_t T,
                    int64_t B, GruTail tail) {
  // (the gate expressions are evaluated as written — no fma contraction — in every
  // instantiation and in the one-launch rollout step of trunk_ws.hip: bit-identical carries)
#pragma clang fp contract(off)
  constexpr int UT = H / 16;          // unit tiles
  constexpr int UTW = (UT + 3) / 4;   // per wave
  constexpr int KS = H / 32 + (H % 32 ? 1 : 0);
  // Look-ahead of the per-step operands.  vmcnt counts loads AND stores, in order, up to 63:
  // the wait for the operands requested PFW steps ago can leave at most 63 younger memory
  // instructions in flight.  With the tile in its natural orientation (a lane = 4 rows x 1
  // unit: 12 + 4 scalar loads and up to 28 scalar stores per step) that window was barely
  // one step, so every step waited for the PREVIOUS step's stores to reach memory (~2 us;
  // 62 us for T = 30 whatever the look-ahead).  In the TRANSPOSED orientation (weights as
  // the A operand: a lane = 1 row x 4 consecutive units) a step is 3 + 1 loads and <= 7
  // stores of 16 bytes, and PFW = 5 steps of them fit the window.
  constexpr int PFW = UTW == 1 ? 5 : 3;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  constexpr int HROW = H + 8;
  bf16_t* hb0 = reinterpret_cast<bf16_t*>(lds_raw);  // [2][16][H + 8]
  bf16_t* hb1 = hb0 + GROWS * HROW;
  bf16_t* const hist = hb1 + GROWS * HROW;            // TAIL: [T][16][H + 8], bf16(h_t)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * GROWS;
  constexpr int H3 = 3 * H;

  // W_h fragments of this wave's units: W_h[k][gate*H + unit], lane (li, lq) the column
  // `unit tile * 16 + li`, reduce elements 8 lq .. 8 lq + 7 of the k-step
  // (loads unconditional — a wave without a unit tile re-reads tile 0 and never uses it — so
  // that all of them are in flight at once: behind a per-element branch each one was waited
  // for before the next was issued)
  bf16x8 wf[UTW][3][KS];
#pragma unroll
  for (int ui = 0; ui < UTW; ++ui) {
    const int ut = wave + 4 * ui < UT ? wave + 4 * ui : 0;
    float wv[3][KS][8];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          wv[g][ks][i] = w_h[(ks * 32 + 8 * lq + i) * H3 + g * H + ut * 16 + li];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 f;
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (bf16_t)wv[g][ks][i];
        wf[ui][g][ks] = f;
      }
  }
  // this lane: row `row0 + li`, units (wave + 4 ui) * 16 + 4 lq + e  (e = 0..3)
  const int64_t row = row0 + li;
  const bool valid = !GUARD || row < B;
  const unsigned rowc = (unsigned)(valid ? row : B - 1);  // clamped: loads stay inside
  f32x4 h[UTW], bn[UTW];
  unsigned ucol[UTW];
#pragma unroll
  for (int ui = 0; ui < UTW; ++ui) {
    const int ut = wave + 4 * ui;
    const bool on = ut < UT;
    ucol[ui] = (unsigned)((on ? ut : 0) * 16 + 4 * lq);
    bn[ui] = on ? *reinterpret_cast<const f32x4*>(b_hn + ucol[ui]) : f32x4{0.f, 0.f, 0.f, 0.f};
    h[ui] = (on && valid) ? *reinterpret_cast<const f32x4*>(h0 + rowc * (unsigned)H + ucol[ui])
                          : f32x4{0.f, 0.f, 0.f, 0.f};
    if (on) {
      bf16x4 hb4;
#pragma unroll
      for (int e = 0; e < 4; ++e) hb4[e] = (bf16_t)h[ui][e];
      *reinterpret_cast<bf16x4*>(hb0 + li * HROW + ucol[ui]) = hb4;
    }
  }
  const int64_t last_t = T - 1;
  // gi / done of steps t .. t+PFW-1 for the owned elements (ring of PFW register slots)
  f32x4 gq[PFW][UTW][3];
  // (the done byte stays RAW in its slot: testing it here would make every load_step wait
  // for its own load — and, vmcnt being in order, for every store before it: one full
  // memory round trip per time step, which is what this kernel used to cost)
  unsigned dq[PFW];
  const uint8_t* const done_c = done ? done : reinterpret_cast<const uint8_t*>(w_h);
  auto load_step = [&](int64_t t, f32x4 (&dst)[UTW][3], unsigned& dn) {
    const int64_t tc = t < last_t ? t : last_t;  // past the end: reload the last step
    const float* gt = gi + tc * B * H3;
    dn = done_c[done ? tc * B + rowc : 0];
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui)
#pragma unroll
      for (int g = 0; g < 3; ++g)
        dst[ui][g] = *reinterpret_cast<const f32x4*>(gt + rowc * (unsigned)H3 +
                                                     (unsigned)(g * H) + ucol[ui]);
  };
#pragma unroll
  for (int d = 0; d < PFW; ++d) load_step(d, gq[d], dq[d]);
  __syncthreads();
  bf16_t* hb = hb0;
  bf16_t* hbn = hb1;
  auto step = [&](int64_t t, f32x4 (&gcur)[UTW][3], unsigned& dcur) {
    const bool reset = done != nullptr && dcur != 0;
    f32x4 acc[UTW][3];
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui)
#pragma unroll
      for (int g = 0; g < 3; ++g) acc[ui][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(hb + li * HROW + ks * 32 + 8 * lq);
#pragma unroll
      for (int ui = 0; ui < UTW; ++ui)
#pragma unroll
        for (int g = 0; g < 3; ++g)  // D[unit = 4*lq + e][row = li]
          acc[ui][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ui][g][ks], af, acc[ui][g],
                                                              0, 0, 0);
    }
    float* ho = h_out + t * B * H;
    float* hpo = TRAIN ? h_prev_out + t * B * H : nullptr;
    float* gto = TRAIN ? gates_out + t * B * 4 * H : nullptr;
    // bf16 image of h_prev [T*B][H]: the x operand of the recurrent kernel's dW launch
    bf16_t* hpb = (TRAIN && BF) ? h_prev_bf + t * B * H : nullptr;
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui) {
      if (wave + 4 * ui >= UT) continue;  // wave-uniform
      const f32x4 hp = h[ui];
      f32x4 r, z, n, qn, hnew;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r[e] = fast_sigmoid(gcur[ui][0][e] + acc[ui][0][e]);
        z[e] = fast_sigmoid(gcur[ui][1][e] + acc[ui][1][e]);
        qn[e] = acc[ui][2][e] + bn[ui][e];
        n[e] = fast_tanh(gcur[ui][2][e] + r[e] * qn[e]);
        hnew[e] = (1.0f - z[e]) * n[e] + z[e] * hp[e];
      }
      if (valid) {
        const unsigned o = rowc * (unsigned)H + ucol[ui];
        *reinterpret_cast<f32x4*>(ho + o) = hnew;
        if constexpr (TRAIN) {
          *reinterpret_cast<f32x4*>(hpo + o) = hp;
          if constexpr (BF) {
            bf16x4 pb;
#pragma unroll
            for (int e = 0; e < 4; ++e) pb[e] = (bf16_t)hp[e];
            *reinterpret_cast<bf16x4*>(hpb + o) = pb;
          }
          const unsigned og = rowc * (unsigned)(4 * H) + ucol[ui];
          *reinterpret_cast<f32x4*>(gto + og) = r;
          *reinterpret_cast<f32x4*>(gto + og + (unsigned)H) = z;
          *reinterpret_cast<f32x4*>(gto + og + (unsigned)(2 * H)) = n;
          *reinterpret_cast<f32x4*>(gto + og + (unsigned)(3 * H)) = qn;
        }
      }
      if constexpr (TAIL) {  // the head's operand: h_t before the reset select
        bf16x4 hr;
#pragma unroll
        for (int e = 0; e < 4; ++e) hr[e] = (bf16_t)hnew[e];
        *reinterpret_cast<bf16x4*>(hist + ((int)t * GROWS + li) * HROW + ucol[ui]) = hr;
        if (valid)
          *reinterpret_cast<bf16x4*>(tail.h_bf + ((int64_t)t * B + rowc) * H + ucol[ui]) = hr;
      }
      bf16x4 hb4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float hc = reset ? 0.0f : hnew[e];
        h[ui][e] = hc;
        hb4[e] = (bf16_t)hc;
      }
      *reinterpret_cast<bf16x4*>(hbn + li * HROW + ucol[ui]) = hb4;
    }
    __syncthreads();
    bf16_t* tmp = hb;
    hb = hbn;
    hbn = tmp;
    load_step(t + PFW, gcur, dcur);  // refill this slot: consumed PFW steps from now
  };
  // whole groups of PFW steps: straight-line; then the remainder
  int64_t t0 = 0;
  for (; t0 + PFW <= T; t0 += PFW) {
#pragma unroll
    for (int d = 0; d < PFW; ++d) step(t0 + d, gq[d], dq[d]);
  }
#pragma unroll
  for (int d = 0; d < PFW; ++d)
    if (t0 + d < T) step(t0 + d, gq[d], dq[d]);
#pragma unroll
  for (int ui = 0; ui < UTW; ++ui) {
    if (wave + 4 * ui >= UT) continue;
    if (valid) *reinterpret_cast<f32x4*>(h_final + rowc * (unsigned)H + ucol[ui]) = h[ui];
  }
  if constexpr (TAIL) {
    // ---- the head on the workgroup's T row tiles, then the sampler on its T x 16 rows ------
    float* const ms_s = reinterpret_cast<float*>(hist + (size_t)T * GROWS * HROW);  // [T*16][N_out]
    const int N_out = tail.N_out;
    bf16x8 wo[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wo[ks] = *reinterpret_cast<const bf16x8*>(tail.wo + ((size_t)ks << 9) + lane * 8);
    f32x4 bo = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tail.bo) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * lq + e < N_out) bo[e] = tail.bo[4 * lq + e];
    }
    __syncthreads();  // every h_t image is in `hist` (the last step's barrier covers the rest)
    for (int t = wave; t < (int)T; t += kThreads / 64) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(hist + (t * GROWS + li) * HROW +
                                                          ks * 32 + 8 * lq);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wo[ks], a, acc, 0, 0, 0);  // D[n][row]
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = 4 * lq + e;
        if (n < N_out) {
          const float v = acc[e] + bo[e];
          ms_s[(t * GROWS + li) * N_out + n] = v;
          if (valid) tail.ms_out[((int64_t)t * B + rowc) * N_out + n] = v;
        }
      }
    }
    __syncthreads();
    for (int idx = tid; idx < (int)T * GROWS; idx += kThreads) {
      const int t = idx / GROWS, r = idx % GROWS;
      if (row0 + r < B)
        mippo_sampler::fwd_row(ms_s + idx * N_out, (int64_t)t * B + row0 + r, tail.samp);
    }
  }
}

// BPTT (formulas in gru.hip), same arrangement as the forward: transposed tile (a lane =
// one row x 4 consecutive units), 16-byte loads and stores, raw done bytes in the ring,
// H / GUARD / output flags as template parameters, straight-line groups of PFW steps.
// dh carry in registers; the dgh tile (bf16) in LDS is the B operand of
// dh_prev^T += W_h . dgh^T; A operand = W_h[unit][j] rows (contiguous in j).
// F32: write dgh as fp32; BF: write its bf16 image (the dW operand).
template <int H, bool GUARD, bool F32, bool BF>
__global__ void __launch_bounds__(kThreads)
gru_bwd_mfma_kernel(const float* __restrict__ g_h, const float* __restrict__ gates,
                    const float* __restrict__ h_prev, const float* __restrict__ w_h,
                    const uint8_t* __restrict__ done, float* __restrict__ dgi,
                    float* __restrict__ dgh, float* __restrict__ dh0,
                    bf16_t* __restrict__ dgh_bf, int64_t T, int64_t B) {
#pragma clang fp contract(off)
  constexpr int UT = H / 16;
  constexpr int UTW = (UT + 3) / 4;
  constexpr int H3 = 3 * H;
  constexpr int KS = H3 / 32;  // <= 12
  constexpr int GROW = H3 + 8;
  constexpr int PFW = UTW == 1 ? 4 : 2;  // 7 loads + 6 stores per step: PFW steps < 63
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  bf16_t* dg0 = reinterpret_cast<bf16_t*>(lds_raw);  // [2][16][3H + 8]
  bf16_t* dg1 = dg0 + GROWS * GROW;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * GROWS;

  // W_h rows of this wave's units: A[i = unit tile * 16 + li][k = j] = W_h[unit][j]
  bf16x8 wf[UTW][KS];
#pragma unroll
  for (int ui = 0; ui < UTW; ++ui) {
    const int ut = wave + 4 * ui < UT ? wave + 4 * ui : 0;
    f32x4 wv[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float* src = w_h + (ut * 16 + li) * H3 + ks * 32 + 8 * lq;
      wv[ks][0] = *reinterpret_cast<const f32x4*>(src);
      wv[ks][1] = *reinterpret_cast<const f32x4*>(src + 4);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f[i] = (bf16_t)wv[ks][0][i];
        f[4 + i] = (bf16_t)wv[ks][1][i];
      }
      wf[ui][ks] = f;
    }
  }
  const int64_t row = row0 + li;
  const bool valid = !GUARD || row < B;
  const unsigned rowc = (unsigned)(valid ? row : B - 1);
  unsigned ucol[UTW];
  f32x4 dh[UTW];
#pragma unroll
  for (int ui = 0; ui < UTW; ++ui) {
    const int ut = wave + 4 * ui < UT ? wave + 4 * ui : 0;
    ucol[ui] = (unsigned)(ut * 16 + 4 * lq);
    dh[ui] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // per-step operands of the owned elements: (r, z, n, qn, g_h, h_prev), done; ring of PFW
  struct In {
    f32x4 v[UTW][6];
    unsigned dn;
  };
  In inq[PFW];
  const uint8_t* const done_c = done ? done : reinterpret_cast<const uint8_t*>(w_h);
  auto load_in = [&](int64_t t, In& dst) {
    const int64_t tc = t > 0 ? t : 0;  // before the start: reload step 0
    const float* gt = gates + tc * B * 4 * H;
    const float* ght = g_h + tc * B * H;
    const float* hpt = h_prev + tc * B * H;
    dst.dn = done_c[done ? tc * B + rowc : 0];  // raw: tested where it is used
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui) {
      const unsigned og = rowc * (unsigned)(4 * H) + ucol[ui];
      const unsigned o = rowc * (unsigned)H + ucol[ui];
      dst.v[ui][0] = *reinterpret_cast<const f32x4*>(gt + og);
      dst.v[ui][1] = *reinterpret_cast<const f32x4*>(gt + og + (unsigned)H);
      dst.v[ui][2] = *reinterpret_cast<const f32x4*>(gt + og + (unsigned)(2 * H));
      dst.v[ui][3] = *reinterpret_cast<const f32x4*>(gt + og + (unsigned)(3 * H));
      dst.v[ui][4] = *reinterpret_cast<const f32x4*>(ght + o);
      dst.v[ui][5] = *reinterpret_cast<const f32x4*>(hpt + o);
    }
  };
#pragma unroll
  for (int d = 0; d < PFW; ++d) load_in(T - 1 - d, inq[d]);
  bf16_t* dg = dg0;
  bf16_t* dgn_buf = dg1;
  auto step = [&](int64_t t, In& in) {
    const bool reset = done != nullptr && in.dn != 0;
    f32x4 dhp[UTW];
    float* gio = dgi + t * B * H3;
    float* gho = F32 ? dgh + t * B * H3 : nullptr;
    // bf16 image of dgh [T*B][3H]: the dz operand of the recurrent kernel's dW launch
    bf16_t* ghb = BF ? dgh_bf + t * B * H3 : nullptr;
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui) {
      if (wave + 4 * ui >= UT) continue;  // wave-uniform (never taken when UT % 4 == 0)
      f32x4 a_r, a_z, a_n, g_n;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float r = in.v[ui][0][e], z = in.v[ui][1][e], n = in.v[ui][2][e],
                    qn = in.v[ui][3][e];
        const float dht = in.v[ui][4][e] + (reset ? 0.0f : dh[ui][e]);
        const float hp = in.v[ui][5][e];
        const float dn = dht * (1.0f - z);
        const float dz = dht * (hp - n);
        const float dp = dht * z;
        const float da_n = dn * (1.0f - n * n);
        const float dr = da_n * qn;
        const float da_z = dz * z * (1.0f - z);
        const float da_r = dr * r * (1.0f - r);
        const float dgn = da_n * r;
        // rows past B (GUARD) contribute nothing
        a_r[e] = valid ? da_r : 0.0f;
        a_z[e] = valid ? da_z : 0.0f;
        a_n[e] = valid ? da_n : 0.0f;
        g_n[e] = valid ? dgn : 0.0f;
        dhp[ui][e] = valid ? dp : 0.0f;
      }
      if (valid) {
        const unsigned o3 = rowc * (unsigned)H3 + ucol[ui];
        *reinterpret_cast<f32x4*>(gio + o3) = a_r;
        *reinterpret_cast<f32x4*>(gio + o3 + (unsigned)H) = a_z;
        *reinterpret_cast<f32x4*>(gio + o3 + (unsigned)(2 * H)) = a_n;
        if constexpr (F32) {
          *reinterpret_cast<f32x4*>(gho + o3) = a_r;
          *reinterpret_cast<f32x4*>(gho + o3 + (unsigned)H) = a_z;
          *reinterpret_cast<f32x4*>(gho + o3 + (unsigned)(2 * H)) = g_n;
        }
      }
      bf16x4 b_r, b_z, b_n;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        b_r[e] = (bf16_t)a_r[e];
        b_z[e] = (bf16_t)a_z[e];
        b_n[e] = (bf16_t)g_n[e];
      }
      if constexpr (BF) {
        if (valid) {
          const unsigned o3 = rowc * (unsigned)H3 + ucol[ui];
          *reinterpret_cast<bf16x4*>(ghb + o3) = b_r;
          *reinterpret_cast<bf16x4*>(ghb + o3 + (unsigned)H) = b_z;
          *reinterpret_cast<bf16x4*>(ghb + o3 + (unsigned)(2 * H)) = b_n;
        }
      }
      *reinterpret_cast<bf16x4*>(dg + li * GROW + ucol[ui]) = b_r;
      *reinterpret_cast<bf16x4*>(dg + li * GROW + H + ucol[ui]) = b_z;
      *reinterpret_cast<bf16x4*>(dg + li * GROW + 2 * H + ucol[ui]) = b_n;
    }
    load_in(t - PFW, in);  // refill this slot: consumed PFW steps from now
    __syncthreads();
    f32x4 acc[UTW];
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui) acc[ui] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(dg + li * GROW + ks * 32 + 8 * lq);
#pragma unroll
      for (int ui = 0; ui < UTW; ++ui)  // D[unit = 4*lq + e][row = li]
        acc[ui] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ui][ks], af, acc[ui], 0, 0, 0);
    }
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui)
#pragma unroll
      for (int e = 0; e < 4; ++e) dh[ui][e] = dhp[ui][e] + acc[ui][e];
    bf16_t* tmp = dg;  // the next step writes the other tile: one barrier per step
    dg = dgn_buf;
    dgn_buf = tmp;
  };
  int64_t t0 = T - 1;
  for (; t0 - (PFW - 1) >= 0; t0 -= PFW) {
#pragma unroll
    for (int d = 0; d < PFW; ++d) step(t0 - d, inq[d]);
  }
#pragma unroll
  for (int d = 0; d < PFW; ++d)
    if (t0 - d >= 0) step(t0 - d, inq[d]);
  if (dh0) {
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui) {
      if (wave + 4 * ui >= UT) continue;
      if (valid) *reinterpret_cast<f32x4*>(dh0 + rowc * (unsigned)H + ucol[ui]) = dh[ui];
    }
  }
}

bool mfma_shape_ok(int64_t H) { return H >= 32 && H <= 128 && H % 32 == 0; }

}  // namespace

extern "C" int mi_gru_seq_fwd_bf16(const float* gi, const float* w_h, const float* b_hn,
                                   const float* h0, const uint8_t* done, float* h_out,
                                   float* h_prev_out, float* gates_out, float* h_final,
                                   void* h_prev_bf, int64_t T, int64_t B, int64_t H,
                                   mi_stream_t stream) {
  MI_REQUIRE(T >= 0 && B >= 0 && mfma_shape_ok(H) && B * 4 * H < (1LL << 31),
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
  const bool guard = B % GROWS != 0;
  bf16_t* hpb = static_cast<bf16_t*>(h_prev_bf);
#define MI_GRU_FWD(TRAIN, HH, GUARD, BF)                                                       \
  hipLaunchKernelGGL((gru_fwd_mfma_kernel<TRAIN, HH, GUARD, BF>), grid, dim3(kThreads), lds, st, \
                     gi, w_h, b_hn, h0, done, h_out, h_prev_out, gates_out, h_final, hpb, T, B,  \
                     GruTail{})
#define MI_GRU_FWD_H(HH)                                                       \
  if (H == HH) {                                                               \
    if (!h_prev_out) {                                                         \
      if (guard) MI_GRU_FWD(false, HH, true, false);                           \
      else MI_GRU_FWD(false, HH, false, false);                                \
    } else if (hpb) {                                                          \
      if (guard) MI_GRU_FWD(true, HH, true, true);                             \
      else MI_GRU_FWD(true, HH, false, true);                                  \
    } else {                                                                   \
      if (guard) MI_GRU_FWD(true, HH, true, false);                            \
      else MI_GRU_FWD(true, HH, false, false);                                 \
    }                                                                          \
  }
  MI_GRU_FWD_H(32)
  MI_GRU_FWD_H(64)
  MI_GRU_FWD_H(96)
  MI_GRU_FWD_H(128)
#undef MI_GRU_FWD_H
#undef MI_GRU_FWD
  return mippo::check_launch("mi_gru_seq_fwd_bf16");
}

// LDS of the TAIL form: the two carry tiles + the T x 16-row history + the head's rows
static size_t gru_tail_lds(int64_t T, int64_t H, int64_t N_out) {
  return (size_t)(2 + T) * GROWS * (H + 8) * sizeof(bf16_t) + (size_t)T * GROWS * N_out * 4;
}

extern "C" int mi_gru_seq_fwd_tail_supported(int64_t T, int64_t H, int64_t N_out) {
  return T >= 1 && mfma_shape_ok(H) && N_out >= 2 && N_out <= 16 && N_out % 2 == 0 &&
         gru_tail_lds(T, H, N_out) <= 96 * 1024;
}

// mi_gru_seq_fwd_bf16 (training form, bf16 image of h_prev) with the head Dense(H -> N_out)
// and the tanh-Gaussian sampler in replay mode (stored raw actions `extras` scored: log-
// likelihood and entropy regulariser out) inside the launch — see GruTail.  w_out: forward
// fragment-major image of the head's kernel; ms_out [T*B, N_out] the head's fp32 rows;
// h_bf_out [T*B, H] the bf16 image of h_out (the head's dW operand).
extern "C" int mi_gru_seq_fwd_tail_bf16(
    const float* gi, const float* w_h, const float* b_hn, const float* h0, const uint8_t* done,
    float* h_out, float* h_prev_out, float* gates_out, float* h_final, void* h_prev_bf,
    const void* w_out, const float* b_out, int64_t N_out, float* ms_out, void* h_bf_out,
    const float* extras, const uint64_t* rng_state, uint64_t offset_add, const float* eps2,
    float min_std, float std_scale, float entropy_weight, float* loglik, float* reg, int64_t T,
    int64_t B, int64_t H, mi_stream_t stream) {
  const char* who = "mi_gru_seq_fwd_tail_bf16";
  MI_REQUIRE(B >= 1 && B * 4 * H < (1LL << 31) && mi_gru_seq_fwd_tail_supported(T, H, N_out),
             "%s: T=%lld H=%lld N_out=%lld outside the supported class", who, (long long)T,
             (long long)H, (long long)N_out);
  MI_REQUIRE(gi && w_h && b_hn && h0 && h_out && h_prev_out && gates_out && h_final &&
                 h_prev_bf && w_out && ms_out && h_bf_out && extras && loglik && reg,
             "%s: null pointer", who);
  MI_REQUIRE(rng_state || eps2, "%s: need rng_state or injected entropy noise", who);
  MI_REQUIRE(al16(w_out) && al16(h_bf_out), "%s: buffers must be 16-byte aligned", who);
  GruTail tail = {static_cast<const bf16_t*>(w_out), b_out, ms_out,
                  static_cast<bf16_t*>(h_bf_out),
                  {extras, {rng_state, offset_add, eps2, eps2}, nullptr, nullptr, nullptr, nullptr,
                   loglik, reg, (int)(N_out / 2), min_std, std_scale, entropy_weight, 0},
                  (int)N_out};
  const size_t lds = gru_tail_lds(T, H, N_out);
  const dim3 grid((unsigned)mippo::ceil_div(B, GROWS));
  hipStream_t st = mippo::as_stream(stream);
  const bool guard = B % GROWS != 0;
  bf16_t* hpb = static_cast<bf16_t*>(h_prev_bf);
#define MI_GRU_TAIL(HH, GUARD)                                                                   \
  {                                                                                              \
    static const hipError_t attr = hipFuncSetAttribute(                                          \
        reinterpret_cast<const void*>(&gru_fwd_mfma_kernel<true, HH, GUARD, true, true>),        \
        hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                                  \
    MI_REQUIRE(attr == hipSuccess, "%s: cannot raise the LDS limit", who);                       \
    hipLaunchKernelGGL((gru_fwd_mfma_kernel<true, HH, GUARD, true, true>), grid, dim3(kThreads), \
                       lds, st, gi, w_h, b_hn, h0, done, h_out, h_prev_out, gates_out, h_final,  \
                       hpb, T, B, tail);                                                         \
  }
#define MI_GRU_TAIL_H(HH)            \
  if (H == HH) {                     \
    if (guard) MI_GRU_TAIL(HH, true) \
    else MI_GRU_TAIL(HH, false)      \
  }
  MI_GRU_TAIL_H(32)
  MI_GRU_TAIL_H(64)
  MI_GRU_TAIL_H(96)
  MI_GRU_TAIL_H(128)
#undef MI_GRU_TAIL_H
#undef MI_GRU_TAIL
  return mippo::check_launch(who);
}

extern "C" int mi_gru_seq_bwd_bf16(const float* g_h, const float* gates, const float* h_prev,
                                   const float* w_h, const uint8_t* done, float* dgi, float* dgh,
                                   float* dh0, void* dgh_bf, int64_t T, int64_t B, int64_t H,
                                   mi_stream_t stream) {
  MI_REQUIRE(T >= 1 && B >= 1 && mfma_shape_ok(H) && B * 4 * H < (1LL << 31),
             "mi_gru_seq_bwd_bf16: bad shape");
  MI_REQUIRE(g_h && gates && h_prev && w_h && dgi && (dgh || dgh_bf),
             "mi_gru_seq_bwd_bf16: null pointer (dgh or its bf16 image is required)");
  const size_t lds = (size_t)2 * GROWS * (3 * H + 8) * sizeof(bf16_t);
  const dim3 grid((unsigned)mippo::ceil_div(B, GROWS));
  hipStream_t st = mippo::as_stream(stream);
  const bool guard = B % GROWS != 0;
  bf16_t* gb = static_cast<bf16_t*>(dgh_bf);
#define MI_GRU_BWD(HH, GUARD, F32, BF)                                                       \
  hipLaunchKernelGGL((gru_bwd_mfma_kernel<HH, GUARD, F32, BF>), grid, dim3(kThreads), lds, st, \
                     g_h, gates, h_prev, w_h, done, dgi, dgh, dh0, gb, T, B)
#define MI_GRU_BWD_H(HH)                                      \
  if (H == HH) {                                              \
    if (dgh && gb) {                                          \
      if (guard) MI_GRU_BWD(HH, true, true, true);            \
      else MI_GRU_BWD(HH, false, true, true);                 \
    } else if (gb) {                                          \
      if (guard) MI_GRU_BWD(HH, true, false, true);           \
      else MI_GRU_BWD(HH, false, false, true);                \
    } else {                                                  \
      if (guard) MI_GRU_BWD(HH, true, true, false);           \
      else MI_GRU_BWD(HH, false, true, false);                \
    }                                                         \
  }
  MI_GRU_BWD_H(32)
  MI_GRU_BWD_H(64)
  MI_GRU_BWD_H(96)
  MI_GRU_BWD_H(128)
#undef MI_GRU_BWD_H
#undef MI_GRU_BWD
  return mippo::check_launch("mi_gru_seq_bwd_bf16");
}
