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
//
// Forms of the two kernels (template parameters; every form returns the same bits):
//   * full tiles (16 rows per workgroup, batches that fill the chip): results stored from a
//     row-major lane arrangement (RowLanes);
//   * PACK = 4 (small batches, `rows_per_group`): 4 rows per workgroup, ONE element per lane
//     (spread4), results through per-row LDS records swept with whole stores (PackedStores);
//   * TAIL: the head Dense + tanh-Gaussian sampler behind the recurrence (forward), their
//     backward in front of the BPTT (GruTail / GruBwdTail);
//   * PROJ (with TAIL, PACK): the input projection inside, forward and backward (GruProj /
//     GruBwdProj); PROJ = 2 (forward): the relu Dense in front of it as well.
// DESIGN.md section 3 ("The GRU actor's loss replay in two sequence launches") has the
// measurements behind each of them.
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


// Stores of the transposed tile.  A lane (li, lq) of the MFMA result holds 16 bytes of ROW li:
// lanes 0..15 are 16 different rows, so a store straight from that arrangement reaches memory
// as 64 separate 16-byte requests — 7 such stores per step were 22 of the training forward's
// 39 us at T = 30, B = 1024, H = 64 (17.5 us with the stores compiled out).  One ds_bpermute
// per dword moves the data to the arrangement lane' = 4 * row + chunk: 4 consecutive lanes then
// hold the 64 contiguous bytes a wave owns of a row, 16 requests per store.
// A lane whose row is not there (past B, or past the rows the workgroup fills) repeats the
// store of the tile's row 0 — same address, same data, same instruction — instead of being
// switched off: a predicated store is a branch around it, and with a path that skips the stores
// the compiler's s_waitcnt vmcnt(N) for the ring of per-step operands may only count the LOADS
// issued since (N ~ 12: every step waited for the previous step's stores to land).
struct RowLanes {
  int src4;        // 4 * (the lane whose data this lane stores)
  unsigned rowc;   // the row this lane stores, its 16-byte chunk
  unsigned chunk;
  __device__ inline RowLanes(int lane, int64_t row0, int64_t B, int rpw, bool guard) {
    int r = lane >> 2;
    chunk = (unsigned)(lane & 3);
    if (guard && !(r < rpw && row0 + r < B)) r = 0;  // row0 itself always exists
    src4 = 4 * (r + 16 * (int)chunk);
    rowc = (unsigned)(row0 + r);
  }
  __device__ inline f32x4 operator()(const f32x4 v) const {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float f = v[e];  // (a bit_cast of the element expression itself reads element 0)
      o[e] = __int_as_float(__builtin_amdgcn_ds_bpermute(src4, __float_as_int(f)));
    }
    return o;
  }
};
__device__ inline bf16x4 to_bf16x4(const f32x4 v) {
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (bf16_t)v[e];
  return o;
}

// The 4 elements a lane (li < 4) of the transposed tile holds, one each to lanes li, li + 4,
// li + 8, li + 12 of its row of 16 (three DPP row shifts): see the PACK form of the kernels.
__device__ inline float spread4(const f32x4 a) {
  const float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
  int v = __float_as_int(a0);
  v = __builtin_amdgcn_update_dpp(v, __float_as_int(a1), 0x114, 0xF, 0x2, false);  // row_shr:4
  v = __builtin_amdgcn_update_dpp(v, __float_as_int(a2), 0x118, 0xF, 0x4, false);  // row_shr:8
  v = __builtin_amdgcn_update_dpp(v, __float_as_int(a3), 0x11C, 0xF, 0x8, false);  // row_shr:12
  return __int_as_float(v);
}

// The done flags of the workgroup's rows, a window of DONE_WIN steps at a time in LDS.  They used
// to ride in the register ring with the other per-step operands; the compiler widens a loaded
// byte (v_and 0xff) at the END of each unrolled group of PFW steps, and that touch waits for
// loads issued a few instructions earlier — one memory round trip per group, a third of the
// kernels' time.  From LDS the flag is a ds_read_u8 where it is used.
constexpr int DONE_WIN = 240;  // steps; a multiple of every PFW, so a group never straddles
struct DoneWindow {
  uint8_t* s;  // [DONE_WIN][GROWS]
  int64_t lo;  // first step held
  // all threads of the workgroup (barriers inside): hold steps lo_ .. lo_ + DONE_WIN - 1
  __device__ inline void stage(const uint8_t* done, int64_t lo_, int64_t T, int64_t B,
                               int64_t row0, int rpw, int tid) {
    lo = lo_;
    __syncthreads();  // nobody still reads the previous window
    const int64_t first = lo_ > 0 ? lo_ : 0;
    const int64_t end = lo_ + DONE_WIN < T ? lo_ + DONE_WIN : T;
    for (int i = (int)(first - lo_) * GROWS + tid; i < (int)(end - lo_) * GROWS; i += kThreads) {
      const int r = i % GROWS;
      s[i] = (r < rpw && row0 + r < B) ? done[(lo_ + i / GROWS) * B + row0 + r] : 0;
    }
    __syncthreads();
  }
  __device__ inline bool holds(int64_t t_lo, int64_t t_hi) const {
    return t_lo >= lo && t_hi < lo + DONE_WIN;
  }
  __device__ inline bool at(int64_t t, int row) const { return s[(int)(t - lo) * GROWS + row] != 0; }
};

// PACK rows per workgroup (a small batch spread over the chip, `rows_per_group`): the step's
// outputs leave through LDS.  What a store costs here is the INSTRUCTION, not its bytes — at
// T = 30, B = 1024, H = 64 every one of the training forward's 7 stores per wave and step added
// 2.0-2.6 us to the launch whether it carried 16 rows or 4 — and a tile with 4 live rows fills a
// quarter of each.  So every wave drops its pieces into a per-row record in LDS (all output
// arrays of the row back to back), and after the step's barrier the workgroup sweeps the PACK
// records with whole 64-lane, 16-bytes-per-lane stores: 8 store instructions per workgroup and
// step instead of 28-32 (forward), 5 instead of 24 (BPTT).  Two record buffers alternate, so the
// sweep of step t runs beside the arithmetic of step t + 1.
struct OutArray {
  void* base;     // [T][B][row_bytes]
  int rec_off;    // where the row's bytes sit in the record
  int row_bytes;
};
template <int NA, int RECB, int PACK>
struct PackedStores {
  static constexpr int REC = RECB + 16;  // + 16: rows land in different LDS banks
  static constexpr int CPR = RECB / 16;  // 16-byte chunks per row
  static constexpr int NCH = PACK * CPR;
  static constexpr int J = (NCH + kThreads - 1) / kThreads;
  static constexpr int LDS_BYTES = 2 * PACK * REC;
  unsigned char* buf;  // [2][PACK][REC]
  char* fp[J];         // this thread's chunk of sweep j: where it goes at the current step ...
  int64_t fs[J];       // ... and how far that moves per step
  unsigned fl[J];      // its place in a record buffer
  // `t_first`: the first step swept; `dir` = +1 / -1: the order of the steps
  __device__ inline void init(unsigned char* lds, const OutArray (&arr)[NA], int tid, int64_t row0,
                              int64_t B, int64_t t_first, int dir) {
    buf = lds;
    const int rows_here = (int)(B - row0 < PACK ? B - row0 : PACK);
    const int n_valid = rows_here * CPR;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      // past the live chunks: repeat an earlier one (same bytes to the same place) — no predicate
      const int c = (tid + j * kThreads) % n_valid;
      const int row = c / CPR, w = (c % CPR) * 16;
      fl[j] = (unsigned)(row * REC + w);
      fp[j] = nullptr;
      fs[j] = 0;
#pragma unroll
      for (int k = 0; k < NA; ++k)
        if (w >= arr[k].rec_off && w < arr[k].rec_off + arr[k].row_bytes) {
          fs[j] = (int64_t)dir * B * arr[k].row_bytes;
          fp[j] = static_cast<char*>(arr[k].base) +
                  (t_first * B + row0 + row) * arr[k].row_bytes + (w - arr[k].rec_off);
        }
    }
  }
  // the record of tile row `li` (< PACK) for the step of parity `par`
  __device__ inline unsigned char* rec(int par, int li) const {
    return buf + (par * PACK + li) * REC;
  }
  // after the barrier behind the step's record writes: out, and on to the next step
  __device__ inline void sweep(int par, int tid) {
    const unsigned char* b = buf + par * PACK * REC;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      if (j * kThreads + (tid & ~63) < NCH)  // wave-uniform: a sweep's last waves may be empty
        *reinterpret_cast<u32x4*>(fp[j]) = *reinterpret_cast<const u32x4*>(b + fl[j]);
      fp[j] += fs[j];
    }
  }
};
constexpr int fwd_rec_bytes(int H, bool TRAIN, bool BF, bool TAIL) {
  return 4 * H + (TRAIN ? 20 * H : 0) + (TRAIN && BF ? 2 * H : 0) + (TAIL ? 2 * H : 0);
}
constexpr int bwd_rec_bytes(int H, bool F32, bool BF, bool PROJ = false) {
  return PROJ ? 12 * H : 12 * H + (F32 ? 12 * H : 0) + (BF ? 6 * H : 0);  // PROJ: two bf16 images
}
constexpr int packed_lds(int rec_bytes, int pack) { return 2 * pack * (rec_bytes + 16); }


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
// PROJ (with TAIL, PACK): the input projection gi_t = y_t W_i + b_i is evaluated per step from
// the bf16 image of the layer in front (the x operand of W_i's dW launch, which exists anyway)
// instead of being read back as fp32 [T, B, 3H]: the Dense chain in front stops one layer
// earlier and 2 x 12 H bytes per row and step never touch memory.  Same MFMA tiles and k order
// as the chain kernel's last layer, bias added to the finished sum: the same bits.
// FRONT (PROJ == 2): the relu Dense(K0 <= 8 -> H) in front is evaluated here too.  The prologue
// brings the workgroup's T x 4 input rows into LDS as bf16 (and leaves that image, the layer's
// dW operand, in memory); y_{t+1} is made while step t waits for its carry — one MFMA per unit
// tile — into an LDS history [T][4][H] that the projection of step t + 1 reads and that goes
// out, whole rows at a time, after the loop (the x operand of W_i's dW, the BPTT's relu' mask).
// Nothing of it touches memory inside the time loop: the first version, with the input rows in
// the register ring and the y rows stored per step, cost the sequence kernel as much as the
// chain launch it replaced (8 loads and 2 stores per step).  Same MFMA tile, bias and relu as
// the chain kernel's layer: the same bits.
struct GruProj {
  const bf16_t* y;   // [T * B][ldy] bf16 image of the GRU's input (in_features == H); FRONT: out
  int64_t ldy;
  const bf16_t* wi;  // forward fragment-major image of W_i [H -> 3H]
  const float* bi;   // [3H]
  const float* x;    // FRONT: [T * B][K0] fp32 input of the front layer
  int K0;
  const bf16_t* w0;  // FRONT: forward fragment-major image of the front kernel [K0 -> H]
  const float* b0;   // FRONT: [H]
  bf16_t* x_bf;      // FRONT: [T * B][8] bf16 image of x (the front layer's dW operand), out
};

struct GruTail {
  GruProj proj;
  const bf16_t* wo;   // forward fragment-major image of the head's kernel [H -> N_out <= 16]
  const float* bo;    // [N_out] or null
  float* ms_out;      // [T * B][N_out] fp32: the head's rows (the sampler backward's input)
  bf16_t* h_bf;       // [T * B][H]: bf16 image of h_out — the x operand of the head's dW
  mippo_sampler::FwdParams samp;  // rows are t * B + env
  int N_out;
};

template <bool TRAIN, int H, bool GUARD, bool BF, bool TAIL = false, int PACK = 0,
          int PROJ = 0>
__global__ void __launch_bounds__(kThreads)
gru_fwd_mfma_kernel(const float* __restrict__ gi, const float* __restrict__ w_h,
                    const float* __restrict__ b_hn, const float* __restrict__ h0,
                    const uint8_t* __restrict__ done, float* __restrict__ h_out,
                    float* __restrict__ h_prev_out, float* __restrict__ gates_out,
                    float* __restrict__ h_final, bf16_t* __restrict__ h_prev_bf, int64_t T,
                    int64_t B, int rpw, GruTail tail) {
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
  DoneWindow dwin = {reinterpret_cast<uint8_t*>(hb1 + GROWS * HROW), 0};
  // PACK: the output records (PackedStores); rpw == PACK
  constexpr int OFF_HP = 4 * H, OFF_G = TRAIN ? 8 * H : 4 * H, OFF_HPB = OFF_G + (TRAIN ? 16 * H : 0);
  constexpr int OFF_HB = OFF_HPB + (TRAIN && BF ? 2 * H : 0);
  constexpr int N_OUT_ARR = 1 + (TRAIN ? 2 : 0) + (TRAIN && BF ? 1 : 0) + (TAIL ? 1 : 0);
  using Packed = PackedStores<N_OUT_ARR, fwd_rec_bytes(H, TRAIN, BF, TAIL), PACK ? PACK : 1>;
  unsigned char* const stg = reinterpret_cast<unsigned char*>(hb1 + GROWS * HROW) + DONE_WIN * GROWS;
  // FRONT: the front layer's input rows [T][4][8] and output rows [T][4][H + 8] (bf16)
  constexpr bool FRONT = PROJ == 2;
  bf16_t* const xs = reinterpret_cast<bf16_t*>(stg + (PACK ? Packed::LDS_BYTES : 0));
  bf16_t* const yhist = xs + (FRONT ? (size_t)T * 4 * 8 : 0);
  // TAIL: [T][16][H + 8], bf16(h_t)
  bf16_t* const hist = yhist + (FRONT ? (size_t)T * 4 * HROW : 0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * rpw;  // rows rpw .. 15 of the tile stay empty
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
  const bool valid = !GUARD || (li < rpw && row < B);
  const unsigned rowc = (unsigned)(valid ? row : B - 1);  // clamped: loads stay inside
  if constexpr (PACK > 0) {
    // ---- a small batch spread thin: rpw == PACK == 4 rows per workgroup ---------------------
    // Of the MFMA result's 16 row slots 4 are live.  With a lane = 4 elements of one row the gate
    // math (~25 VALU instructions and 6 transcendentals per element) ran on 16 lanes and was
    // ~1,200 of a step's ~2,700 cycles, the longest link of its chain.  So the 4 elements of a
    // live lane go out to the dead ones (spread4): lane (li, lq) owns ONE element for the whole
    // sequence — row li & 3, unit 16 ut + 4 lq + (li >> 2) — carry, bias, per-step operands (one
    // dword each) and results alike.  The results leave through per-row records in LDS
    // (PackedStores).  Same expressions per element: the same bits as the other form.
    static_assert(PACK == 4, "the spread form is written for 4 rows per workgroup");
    const int rr = li & 3;
    const int64_t srow = row0 + rr;
    const bool svalid = srow < B;
    const unsigned srowc = (unsigned)(svalid ? srow : B - 1);
    for (int i = tid; i < 2 * GROWS * HROW; i += kThreads) hb0[i] = (bf16_t)0.0f;  // dead rows
    __syncthreads();
    float hc[UTW], bnv[UTW];
    unsigned ucol[UTW];
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui) {
      const int ut = wave + 4 * ui;
      const bool on = ut < UT;
      ucol[ui] = (unsigned)((on ? ut : 0) * 16 + 4 * lq + (li >> 2));
      bnv[ui] = on ? b_hn[ucol[ui]] : 0.0f;
      hc[ui] = (on && svalid) ? h0[srowc * (unsigned)H + ucol[ui]] : 0.0f;
      if (on) hb0[rr * HROW + ucol[ui]] = (bf16_t)hc[ui];
    }
    Packed pk;
    {
      OutArray arr[N_OUT_ARR];
      int k = 0;
      arr[k++] = {h_out, 0, 4 * H};
      if constexpr (TRAIN) {
        arr[k++] = {h_prev_out, OFF_HP, 4 * H};
        arr[k++] = {gates_out, OFF_G, 16 * H};
        if constexpr (BF) arr[k++] = {h_prev_bf, OFF_HPB, 2 * H};
      }
      if constexpr (TAIL) arr[k++] = {tail.h_bf, OFF_HB, 2 * H};
      pk.init(stg, arr, tid, row0, B, 0, +1);
    }
    const int64_t last_t = T - 1;
    // per-step operands, PFW steps ahead: gi of the owned elements — or (PROJ) the B operand of
    // the projection, the 8 inputs of tile row li per k-step
    struct Slot {
      float g[UTW][3];
      bf16x8 y[KS];
    };
    Slot gq[PFW];
    bf16x8 w0f[FRONT ? UTW : 1];
    f32x4 b0v[FRONT ? UTW : 1];
    // FRONT: y_t = relu(x_t W_0 + b_0) of this wave's unit tiles for the 16 tile rows, into the
    // step's LDS tile and (live rows) out; wave 0 also leaves the bf16 image of x_t
    auto front = [&](int64_t t) {
      if constexpr (FRONT) {
        // (tile rows 4 .. 15 repeat rows 0 .. 3: dead rows, never kept)
        const bf16x8 x8 = *reinterpret_cast<const bf16x8*>(xs + ((int)t * 4 + rr) * 8);
        bf16x8 xf;
#pragma unroll
        for (int i = 0; i < 8; ++i) xf[i] = lq == 0 ? x8[i] : (bf16_t)0.0f;
#pragma unroll
        for (int ui = 0; ui < UTW; ++ui) {
          if constexpr (UT % 4 != 0)
            if (wave + 4 * ui >= UT) continue;
          const f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              w0f[ui], xf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          bf16x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (bf16_t)fmaxf(a[e] + b0v[ui][e], 0.0f);
          if (li < 4)
            *reinterpret_cast<bf16x4*>(yhist + ((int)t * 4 + li) * HROW + (wave + 4 * ui) * 16 +
                                       4 * lq) = v;
        }
      }
    };
    bf16x8 wif[PROJ ? UTW : 1][3][KS];
    float biv[PROJ ? UTW : 1][3];
    if constexpr (PROJ) {
      static_assert(TAIL, "PROJ comes with TAIL");
#pragma unroll
      for (int ui = 0; ui < UTW; ++ui) {
        const int ut = wave + 4 * ui < UT ? wave + 4 * ui : 0;
        if constexpr (FRONT) {
          w0f[ui] = *reinterpret_cast<const bf16x8*>(tail.proj.w0 + ((size_t)ut << 9) + lane * 8);
          b0v[ui] = *reinterpret_cast<const f32x4*>(tail.proj.b0 + ut * 16 + 4 * lq);
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          biv[ui][g] = tail.proj.bi[g * H + ucol[ui]];
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)  // block (column tile g UT + ut, ks) of the image
            wif[ui][g][ks] = *reinterpret_cast<const bf16x8*>(
                tail.proj.wi + ((size_t)((g * UT + ut) * KS + ks) << 9) + lane * 8);
        }
      }
    }
    auto load_step = [&](int64_t t, Slot& dst) {
      const int64_t tc = t < last_t ? t : last_t;  // past the end: reload the last step
      if constexpr (FRONT) {
        // (nothing to load: the input rows are in LDS)
      } else if constexpr (PROJ) {
        const bf16_t* yt = tail.proj.y + (tc * B + rowc) * tail.proj.ldy + 8 * lq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) dst.y[ks] = *reinterpret_cast<const bf16x8*>(yt + ks * 32);
      } else {
        const float* gt = gi + tc * B * H3;
#pragma unroll
        for (int ui = 0; ui < UTW; ++ui)
#pragma unroll
          for (int g = 0; g < 3; ++g)
            dst.g[ui][g] = gt[srowc * (unsigned)H3 + (unsigned)(g * H) + ucol[ui]];
      }
    };
#pragma unroll
    for (int d = 0; d < PFW; ++d) load_step(d, gq[d]);
    if constexpr (FRONT) {
      // the workgroup's T x 4 input rows: bf16, zero-padded to 8, into LDS and out
      for (int idx = tid; idx < (int)T * 4; idx += kThreads) {
        const int t = idx >> 2, r = idx & 3;
        const bool on = r < rpw && row0 + r < B;
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          v[i] = (bf16_t)((on && i < tail.proj.K0)
                              ? tail.proj.x[((int64_t)t * B + row0 + r) * tail.proj.K0 + i]
                              : 0.0f);
        *reinterpret_cast<bf16x8*>(xs + idx * 8) = v;
        if (on) *reinterpret_cast<bf16x8*>(tail.proj.x_bf + ((int64_t)t * B + row0 + r) * 8) = v;
      }
      __syncthreads();
      front(0);  // y of step 0
    }
    if (done) dwin.stage(done, 0, T, B, row0, rpw, tid);
    __syncthreads();
    bf16_t* hb = hb0;
    bf16_t* hbn = hb1;
    auto step = [&](int64_t t, Slot& gcur) {
      const bool reset = done != nullptr && dwin.at(t, rr);
      if constexpr (FRONT) {  // this step's y rows (made during the step before)
        const bf16_t* yt = yhist + ((int)t * 4 + rr) * HROW + 8 * lq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) gcur.y[ks] = *reinterpret_cast<const bf16x8*>(yt + ks * 32);
      }
      // the carry's operand reads first; while they are in flight the PREVIOUS step's records go
      // out (their LDS reads and store issues sat behind the barrier, in front of these reads,
      // on every step's chain)
      bf16x8 af[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        af[ks] = *reinterpret_cast<const bf16x8*>(hb + li * HROW + ks * 32 + 8 * lq);
      if (t != 0) pk.sweep((int)((t - 1) & 1), tid);
      if constexpr (PROJ) {  // gi of this step (does not wait for the carry)
#pragma unroll
        for (int ui = 0; ui < UTW; ++ui)
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
              a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wif[ui][g][ks], gcur.y[ks], a, 0, 0, 0);
            gcur.g[ui][g] = spread4(a) + biv[ui][g];
          }
        if constexpr (FRONT)
          if (t + 1 < T) front(t + 1);  // the next step's y, off the carry's chain
      }
      f32x4 acc[UTW][3];
#pragma unroll
      for (int ui = 0; ui < UTW; ++ui)
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[ui][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int ui = 0; ui < UTW; ++ui)
#pragma unroll
          for (int g = 0; g < 3; ++g)  // D[unit = 4*lq + e][row = li]
            acc[ui][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ui][g][ks], af[ks],
                                                                acc[ui][g], 0, 0, 0);
      }
      unsigned char* const rc0 = pk.rec((int)(t & 1), rr);
#pragma unroll
      for (int ui = 0; ui < UTW; ++ui) {
        if constexpr (UT % 4 != 0)
          if (wave + 4 * ui >= UT) continue;  // wave-uniform
        const float hp = hc[ui];
        const float r = fast_sigmoid(gcur.g[ui][0] + spread4(acc[ui][0]));
        const float z = fast_sigmoid(gcur.g[ui][1] + spread4(acc[ui][1]));
        const float qn = spread4(acc[ui][2]) + bnv[ui];
        const float n = fast_tanh(gcur.g[ui][2] + r * qn);
        const float hnew = (1.0f - z) * n + z * hp;
        unsigned char* const rc = rc0 + 4 * ucol[ui];
        *reinterpret_cast<float*>(rc) = hnew;
        if constexpr (TRAIN) {
          *reinterpret_cast<float*>(rc + OFF_HP) = hp;
          *reinterpret_cast<float*>(rc + OFF_G) = r;
          *reinterpret_cast<float*>(rc + OFF_G + 4 * H) = z;
          *reinterpret_cast<float*>(rc + OFF_G + 8 * H) = n;
          *reinterpret_cast<float*>(rc + OFF_G + 12 * H) = qn;
          if constexpr (BF) *reinterpret_cast<bf16_t*>(rc + OFF_HPB - 2 * ucol[ui]) = (bf16_t)hp;
        }
        if constexpr (TAIL) {  // bf16(h_t): the head's operand (before the reset) and its dW's x
          *reinterpret_cast<bf16_t*>(rc + OFF_HB - 2 * ucol[ui]) = (bf16_t)hnew;
          hist[((int)t * GROWS + rr) * HROW + ucol[ui]] = (bf16_t)hnew;
        }
        const float hcn = reset ? 0.0f : hnew;
        hc[ui] = hcn;
        hbn[rr * HROW + ucol[ui]] = (bf16_t)hcn;
      }
      __builtin_amdgcn_sched_barrier(0);  // (the refill behind its slot's use: see the other form)
      load_step(t + PFW, gcur);
      __syncthreads();
      bf16_t* tmp = hb;
      hb = hbn;
      hbn = tmp;
    };
    int64_t t0 = 0;
    if (T >= PFW) {  // first group peeled, the loop inside its branch (see the other form)
#pragma unroll
      for (int d = 0; d < PFW; ++d) step(d, gq[d]);
      for (t0 = PFW; t0 + PFW <= T; t0 += PFW) {
        if (done && !dwin.holds(t0, t0 + PFW - 1)) dwin.stage(done, t0, T, B, row0, rpw, tid);
#pragma unroll
        for (int d = 0; d < PFW; ++d) step(t0 + d, gq[d]);
      }
    }
    if (done && t0 < T && !dwin.holds(t0, T - 1)) dwin.stage(done, t0, T, B, row0, rpw, tid);
#pragma unroll
    for (int d = 0; d < PFW; ++d)
      if (t0 + d < T) step(t0 + d, gq[d]);
    if (T > 0) pk.sweep((int)((T - 1) & 1), tid);  // the last step's records
    if constexpr (FRONT) {  // the y history out: whole rows, 16 bytes per lane
      constexpr int CPR = H / 8;
      for (int idx = tid; idx < (int)T * 4 * CPR; idx += kThreads) {
        const int c = idx % CPR, tr = idx / CPR, t = tr >> 2, r = tr & 3;
        if (r < rpw && row0 + r < B)
          *reinterpret_cast<u32x4*>(const_cast<bf16_t*>(tail.proj.y) +
                                    ((int64_t)t * B + row0 + r) * tail.proj.ldy + c * 8) =
              *reinterpret_cast<const u32x4*>(yhist + tr * HROW + c * 8);
      }
    }
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui) {
      if (wave + 4 * ui >= UT) continue;
      if (svalid) h_final[srowc * (unsigned)H + ucol[ui]] = hc[ui];
    }
  } else {
    f32x4 h[UTW], bn[UTW];
    unsigned ucol[UTW];
    const RowLanes sl(lane, row0, B, rpw, GUARD);  // the arrangement the global stores go out in
    unsigned scol[UTW];
  #pragma unroll
    for (int ui = 0; ui < UTW; ++ui) {
      const int ut = wave + 4 * ui;
      const bool on = ut < UT;
      ucol[ui] = (unsigned)((on ? ut : 0) * 16 + 4 * lq);
      scol[ui] = (unsigned)((on ? ut : 0) * 16) + 4 * sl.chunk;
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
    // gi of steps t .. t+PFW-1 for the owned elements (ring of PFW register slots; the done
    // flags: DoneWindow)
    f32x4 gq[PFW][UTW][3];
    auto load_step = [&](int64_t t, f32x4 (&dst)[UTW][3]) {
      const int64_t tc = t < last_t ? t : last_t;  // past the end: reload the last step
      const float* gt = gi + tc * B * H3;
  #pragma unroll
      for (int ui = 0; ui < UTW; ++ui)
  #pragma unroll
        for (int g = 0; g < 3; ++g)
          dst[ui][g] = *reinterpret_cast<const f32x4*>(gt + rowc * (unsigned)H3 +
                                                       (unsigned)(g * H) + ucol[ui]);
    };
  #pragma unroll
    for (int d = 0; d < PFW; ++d) load_step(d, gq[d]);
    if (done) dwin.stage(done, 0, T, B, row0, rpw, tid);
    __syncthreads();
    bf16_t* hb = hb0;
    bf16_t* hbn = hb1;
    auto step = [&](int64_t t, f32x4 (&gcur)[UTW][3]) {
      const bool reset = done != nullptr && dwin.at(t, li);
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
        if constexpr (UT % 4 != 0)
          if (wave + 4 * ui >= UT) continue;  // wave-uniform (a branch: see RowLanes)
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
        {  // out through the row-major lane arrangement (RowLanes)
          const f32x4 s_hnew = sl(hnew);
          const unsigned o = sl.rowc * (unsigned)H + scol[ui];
          *reinterpret_cast<f32x4*>(ho + o) = s_hnew;
          if constexpr (TRAIN) {
            const f32x4 s_hp = sl(hp), s_r = sl(r), s_z = sl(z), s_n = sl(n), s_qn = sl(qn);
            *reinterpret_cast<f32x4*>(hpo + o) = s_hp;
            if constexpr (BF) *reinterpret_cast<bf16x4*>(hpb + o) = to_bf16x4(s_hp);
            const unsigned og = sl.rowc * (unsigned)(4 * H) + scol[ui];
            *reinterpret_cast<f32x4*>(gto + og) = s_r;
            *reinterpret_cast<f32x4*>(gto + og + (unsigned)H) = s_z;
            *reinterpret_cast<f32x4*>(gto + og + (unsigned)(2 * H)) = s_n;
            *reinterpret_cast<f32x4*>(gto + og + (unsigned)(3 * H)) = s_qn;
          }
          if constexpr (TAIL)  // bf16 image of h_t: the x operand of the head's dW
            *reinterpret_cast<bf16x4*>(tail.h_bf + ((int64_t)t * B + sl.rowc) * H + scol[ui]) =
                to_bf16x4(s_hnew);
        }
        if constexpr (TAIL)  // the head's operand: h_t before the reset select
          *reinterpret_cast<bf16x4*>(hist + ((int)t * GROWS + li) * HROW + ucol[ui]) =
              to_bf16x4(hnew);
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
      // refill this slot (consumed PFW steps from now).  Behind a scheduling barrier: hoisted above
      // the gate math the loads would need registers of their own while the old slot is still
      // live, and the copies back into the slot at the end of a PFW group wait for loads issued
      // moments before (the look-ahead collapses to one round trip per group)
      __builtin_amdgcn_sched_barrier(0);
      load_step(t + PFW, gcur);
    };
    // whole groups of PFW steps: straight-line; then the remainder.  The first group is peeled:
    // the compiler places ONE set of s_waitcnt vmcnt(N) in the loop body, valid for every way
    // into it — entered straight from the prologue (the ring filled back to back, a slot's load
    // only a few memory instructions old when it is consumed) N would be ~12 for every iteration,
    // i.e. each step waiting for the previous step's stores; entered from a group, the loop's
    // counts are the steady state's (a slot's load ~PFW steps of instructions old).
    // (the loop INSIDE the branch of the peeled group: no way into it but through a group)
    int64_t t0 = 0;
    if (T >= PFW) {
  #pragma unroll
      for (int d = 0; d < PFW; ++d) step(d, gq[d]);
      for (t0 = PFW; t0 + PFW <= T; t0 += PFW) {
        if (done && !dwin.holds(t0, t0 + PFW - 1)) dwin.stage(done, t0, T, B, row0, rpw, tid);
  #pragma unroll
        for (int d = 0; d < PFW; ++d) step(t0 + d, gq[d]);
      }
    }
    if (done && t0 < T && !dwin.holds(t0, T - 1)) dwin.stage(done, t0, T, B, row0, rpw, tid);
  #pragma unroll
    for (int d = 0; d < PFW; ++d)
      if (t0 + d < T) step(t0 + d, gq[d]);
  #pragma unroll
    for (int ui = 0; ui < UTW; ++ui) {
      if (wave + 4 * ui >= UT) continue;
      if (valid) *reinterpret_cast<f32x4*>(h_final + rowc * (unsigned)H + ucol[ui]) = h[ui];
    }
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
      if (r < rpw && row0 + r < B)
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
// TAIL (mirror of the forward's GruTail): the sampler's backward and the head's dX are evaluated
// HERE instead of by two launches in front of this one.  Prologue: the workgroup's T x 16 rows
// run the sampler backward row function (d loss / d (mean | pre-softplus std) from
// d loss / d log-likelihood and the regulariser's weight), the rows — rounded to bf16, the head's
// dz image, which is also written out for its dW — wait in LDS; per step the gradient w.r.t. h_t
// is one MFMA k-step of that tile against the head's backward fragments (g_h = dz . W_out^T,
// the product mi_mlp_bwd_dx_bf16 makes), in the lane that consumes it.  Bit-identical to the
// three launches.
// PROJ (mirror of the forward's GruProj): the backward of the input projection rides along.
// dgi leaves as its bf16 image only (what the chain kernel's first act was to make of the fp32
// tensor: the dz operand of W_i's dW), and d loss / d (pre-activation of the relu layer in front)
// = (dgi_bf . W_i^T) . relu'(y) — the product the chain kernel made, 3H / 32 k-steps on the tile
// of dgi this step has in LDS anyway — leaves as ITS bf16 image: the dz operand of that layer's
// dW.  The Dense chain's backward launch, its read of fp32 dgi and the write of it here all go.
struct GruBwdProj {
  const bf16_t* y;     // [T * B][ldy] bf16 image of the GRU's input (post-relu)
  int64_t ldy;
  const bf16_t* wi_b;  // backward fragment-major image of W_i (columns = the H inputs)
  bf16_t* dgi_bf;      // [T * B][3H]
  bf16_t* dz0_bf;      // [T * B][ldy]
};

struct GruBwdTail {
  GruBwdProj proj;
  const bf16_t* wo_b;  // backward fragment-major image of the head's kernel (columns = H inputs)
  bf16_t* dz_out;      // [T * B][ld_dz] bf16 image of the head's output gradient (its dW operand)
  int64_t ld_dz;       // pad8(N_out)
  mippo_sampler::BwdParams sb;  // rows are t * B + env
  int N_out;
};

template <int H, bool GUARD, bool F32, bool BF, bool TAIL = false, int PACK = 0,
          bool PROJ = false>
__global__ void __launch_bounds__(kThreads)
gru_bwd_mfma_kernel(const float* __restrict__ g_h, const float* __restrict__ gates,
                    const float* __restrict__ h_prev, const float* __restrict__ w_h,
                    const uint8_t* __restrict__ done, float* __restrict__ dgi,
                    float* __restrict__ dgh, float* __restrict__ dh0,
                    bf16_t* __restrict__ dgh_bf, int64_t T, int64_t B, int rpw,
                    GruBwdTail tail) {
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
  constexpr int DZROW = 32 + 8;                       // TAIL: one k-step of head columns + pad
  DoneWindow dwin = {reinterpret_cast<uint8_t*>(dg1 + GROWS * GROW), 0};
  // PACK: the output records (PackedStores); rpw == PACK
  constexpr int OFF_GH = 12 * H, OFF_GB = OFF_GH + (F32 ? 12 * H : 0);
  constexpr int N_OUT_ARR = 1 + (F32 ? 1 : 0) + (BF ? 1 : 0);
  using Packed = PackedStores<PROJ ? 2 : N_OUT_ARR, bwd_rec_bytes(H, F32, BF, PROJ), PACK ? PACK : 1>;
  using Packed0 = PackedStores<1, 2 * H, PACK ? PACK : 1>;  // PROJ: the rows of dz0
  unsigned char* const stg = reinterpret_cast<unsigned char*>(dg1 + GROWS * GROW) + DONE_WIN * GROWS;
  // PROJ: the n columns of dgi [2][16][H + 8] (its r and z columns are dgh's), the dz0 records
  constexpr int DNROW = H + 8;
  bf16_t* const dn0 = reinterpret_cast<bf16_t*>(stg + (PACK ? Packed::LDS_BYTES : 0));
  unsigned char* const stg0 = reinterpret_cast<unsigned char*>(dn0 + (PROJ ? 2 * GROWS * DNROW : 0));
  // TAIL: [T][16][DZROW]
  bf16_t* const dzs = reinterpret_cast<bf16_t*>(stg0 + (PROJ ? Packed0::LDS_BYTES : 0));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * rpw;  // rows rpw .. 15 of the tile stay empty

  if constexpr (TAIL) {
    // the sampler backward on the workgroup's T x 16 rows (sampler.hip's row function), the
    // rows rounded to bf16 as the head's dz image: into LDS (k-padded with zeros) and out
    for (int i = tid; i < (int)T * GROWS * DZROW; i += kThreads) dzs[i] = (bf16_t)0.0f;
    __syncthreads();
    for (int idx = tid; idx < (int)T * GROWS; idx += kThreads) {
      const int t = idx / GROWS, r = idx % GROWS;
      if (r < rpw && row0 + r < B) {
        const int64_t b = (int64_t)t * B + row0 + r;
        bf16_t* row_s = dzs + idx * DZROW;
        bf16_t* row_g = tail.dz_out + b * tail.ld_dz;
        mippo_sampler::bwd_row(b, tail.sb, [&](int a, float v) {
          const bf16_t q = (bf16_t)v;
          row_s[a] = q;
          row_g[a] = q;
        });
        for (int a = tail.N_out; a < (int)tail.ld_dz; ++a) row_g[a] = (bf16_t)0.0f;
      }
    }
    __syncthreads();
  }

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
  // TAIL: the head's backward fragments of this wave's unit tiles (columns = units)
  bf16x8 wob[UTW];
  if constexpr (TAIL) {
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui) {
      const int ut = wave + 4 * ui < UT ? wave + 4 * ui : 0;
      wob[ui] = *reinterpret_cast<const bf16x8*>(tail.wo_b + ((size_t)ut << 9) + lane * 8);
    }
  }
  const int64_t row = row0 + li;
  const bool valid = !GUARD || (li < rpw && row < B);
  const unsigned rowc = (unsigned)(valid ? row : B - 1);
  if constexpr (PACK > 0) {
    // ---- a small batch spread thin (rpw == PACK == 4): one element per lane, results through
    // per-row LDS records — the forward's PACK form, same expressions per element as below
    static_assert(PACK == 4, "the spread form is written for 4 rows per workgroup");
    const int rr = li & 3;
    const int64_t srow = row0 + rr;
    const bool svalid = srow < B;
    const unsigned srowc = (unsigned)(svalid ? srow : B - 1);
    for (int i = tid; i < 2 * GROWS * GROW; i += kThreads) dg0[i] = (bf16_t)0.0f;  // dead rows
    if constexpr (PROJ)
      for (int i = tid; i < 2 * GROWS * DNROW; i += kThreads) dn0[i] = (bf16_t)0.0f;
    unsigned ucol[UTW];
    float dh[UTW];
#pragma unroll
    for (int ui = 0; ui < UTW; ++ui) {
      const int ut = wave + 4 * ui < UT ? wave + 4 * ui : 0;
      ucol[ui] = (unsigned)(ut * 16 + 4 * lq + (li >> 2));
      dh[ui] = 0.0f;
    }
    Packed pk;
    Packed0 pk0;
    bf16x8 wib[PROJ ? UTW : 1][KS];
    if constexpr (PROJ) {
      static_assert(TAIL && BF && !F32, "PROJ comes with TAIL (bf16 image of dgh only)");
      OutArray arr[2] = {{tail.proj.dgi_bf, 0, 6 * H}, {dgh_bf, 6 * H, 6 * H}};
      pk.init(stg, arr, tid, row0, B, T - 1, -1);
      OutArray arr0[1] = {{tail.proj.dz0_bf, 0, 2 * H}};
      // (rows of dz0 are ldy wide; ldy == H is required, so a row is its 2H bytes)
      pk0.init(stg0, arr0, tid, row0, B, T - 1, -1);
#pragma unroll
      for (int ui = 0; ui < UTW; ++ui) {
        const int ut = wave + 4 * ui < UT ? wave + 4 * ui : 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          wib[ui][ks] = *reinterpret_cast<const bf16x8*>(
              tail.proj.wi_b + ((size_t)(ut * KS + ks) << 9) + lane * 8);
      }
    } else {
      OutArray arr[N_OUT_ARR];
      int k = 0;
      arr[k++] = {dgi, 0, 12 * H};
      if constexpr (F32) arr[k++] = {dgh, OFF_GH, 12 * H};
      if constexpr (BF) arr[k++] = {dgh_bf, OFF_GB, 6 * H};
      pk.init(stg, arr, tid, row0, B, T - 1, -1);
    }
    struct In {
      float v[UTW][6];  // r, z, n, qn, g_h, h_prev of the owned element
      bf16_t y[UTW];    // PROJ: the projection's input at the owned element (relu' mask)
    };
    In inq[PFW];
    auto load_in = [&](int64_t t, In& dst) {
      const int64_t tc = t > 0 ? t : 0;  // before the start: reload step 0
      const float* gt = gates + tc * B * 4 * H;
      const float* ght = TAIL ? nullptr : g_h + tc * B * H;
      const float* hpt = h_prev + tc * B * H;
#pragma unroll
      for (int ui = 0; ui < UTW; ++ui) {
        const unsigned og = srowc * (unsigned)(4 * H) + ucol[ui];
        const unsigned o = srowc * (unsigned)H + ucol[ui];
        dst.v[ui][0] = gt[og];
        dst.v[ui][1] = gt[og + (unsigned)H];
        dst.v[ui][2] = gt[og + (unsigned)(2 * H)];
        dst.v[ui][3] = gt[og + (unsigned)(3 * H)];
        if constexpr (!TAIL) dst.v[ui][4] = ght[o];
        dst.v[ui][5] = hpt[o];
        if constexpr (PROJ) dst.y[ui] = tail.proj.y[(tc * B + srowc) * tail.proj.ldy + ucol[ui]];
      }
    };
#pragma unroll
    for (int d = 0; d < PFW; ++d) load_in(T - 1 - d, inq[d]);
    if (done) dwin.stage(done, T - DONE_WIN, T, B, row0, rpw, tid);
    __syncthreads();  // the zeroed tiles
    bf16_t* dg = dg0;
    bf16_t* dgn_buf = dg1;
    bf16_t* dnb = dn0;  // PROJ: this step's n columns of dgi (alternates with the other tile)
    auto step = [&](int64_t t, In& in) {
      const bool reset = done != nullptr && dwin.at(t, rr);
      float dhp[UTW];
      bf16_t ycur[UTW];  // (the slot is refilled before its relu' mask is used)
      if constexpr (PROJ) {
#pragma unroll
        for (int ui = 0; ui < UTW; ++ui) ycur[ui] = in.y[ui];
      }
      if constexpr (TAIL) {  // g_h[t] = dz_out[t] . W_out^T: one k-step, D[unit 4 lq + e][row li]
        const bf16x8 dzf = *reinterpret_cast<const bf16x8*>(dzs + ((int)t * GROWS + li) * DZROW +
                                                            8 * lq);
#pragma unroll
        for (int ui = 0; ui < UTW; ++ui)
          in.v[ui][4] = spread4(__builtin_amdgcn_mfma_f32_16x16x32_bf16(
              wob[ui], dzf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0));
      }
      unsigned char* const rc0 = pk.rec((int)(t & 1), rr);
#pragma unroll
      for (int ui = 0; ui < UTW; ++ui) {
        if constexpr (UT % 4 != 0)
          if (wave + 4 * ui >= UT) continue;  // wave-uniform
        const float r = in.v[ui][0], z = in.v[ui][1], n = in.v[ui][2], qn = in.v[ui][3];
        const float dht = in.v[ui][4] + (reset ? 0.0f : dh[ui]);
        const float hp = in.v[ui][5];
        const float dn = dht * (1.0f - z);
        const float dz = dht * (hp - n);
        const float dp = dht * z;
        const float da_n = dn * (1.0f - n * n);
        const float dr = da_n * qn;
        const float da_z = dz * z * (1.0f - z);
        const float da_r = dr * r * (1.0f - r);
        const float dgn = da_n * r;
        // rows past B contribute nothing
        const float a_r = svalid ? da_r : 0.0f, a_z = svalid ? da_z : 0.0f;
        const float a_n = svalid ? da_n : 0.0f, g_n = svalid ? dgn : 0.0f;
        dhp[ui] = svalid ? dp : 0.0f;
        const bf16_t b_r = (bf16_t)a_r, b_z = (bf16_t)a_z, b_n = (bf16_t)g_n;
        if constexpr (PROJ) {  // record = bf16 images of dgi and dgh
          const bf16_t b_an = (bf16_t)a_n;
          unsigned char* const rb = rc0 + 2 * ucol[ui];
          *reinterpret_cast<bf16_t*>(rb) = b_r;
          *reinterpret_cast<bf16_t*>(rb + 2 * H) = b_z;
          *reinterpret_cast<bf16_t*>(rb + 4 * H) = b_an;
          *reinterpret_cast<bf16_t*>(rb + 6 * H) = b_r;
          *reinterpret_cast<bf16_t*>(rb + 8 * H) = b_z;
          *reinterpret_cast<bf16_t*>(rb + 10 * H) = b_n;
          dnb[rr * DNROW + ucol[ui]] = b_an;
        } else {
          unsigned char* const rc = rc0 + 4 * ucol[ui];
          *reinterpret_cast<float*>(rc) = a_r;
          *reinterpret_cast<float*>(rc + 4 * H) = a_z;
          *reinterpret_cast<float*>(rc + 8 * H) = a_n;
          if constexpr (F32) {
            *reinterpret_cast<float*>(rc + OFF_GH) = a_r;
            *reinterpret_cast<float*>(rc + OFF_GH + 4 * H) = a_z;
            *reinterpret_cast<float*>(rc + OFF_GH + 8 * H) = g_n;
          }
          if constexpr (BF) {
            unsigned char* const rb = rc + OFF_GB - 2 * ucol[ui];
            *reinterpret_cast<bf16_t*>(rb) = b_r;
            *reinterpret_cast<bf16_t*>(rb + 2 * H) = b_z;
            *reinterpret_cast<bf16_t*>(rb + 4 * H) = b_n;
          }
        }
        dg[rr * GROW + ucol[ui]] = b_r;
        dg[rr * GROW + H + ucol[ui]] = b_z;
        dg[rr * GROW + 2 * H + ucol[ui]] = b_n;
      }
      __builtin_amdgcn_sched_barrier(0);  // (the refill stays behind its slot's use)
      load_in(t - PFW, in);
      __syncthreads();
      pk.sweep((int)(t & 1), tid);
      if constexpr (PROJ)  // the dz0 rows of the step before (written behind ITS barrier)
        if (t != T - 1) pk0.sweep((int)((t + 1) & 1), tid);
      f32x4 acc[UTW], acc0[PROJ ? UTW : 1];
#pragma unroll
      for (int ui = 0; ui < UTW; ++ui) acc[ui] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (PROJ) {
#pragma unroll
        for (int ui = 0; ui < UTW; ++ui) acc0[ui] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(dg + li * GROW + ks * 32 + 8 * lq);
#pragma unroll
        for (int ui = 0; ui < UTW; ++ui)  // D[unit = 4*lq + e][row = li]
          acc[ui] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ui][ks], af, acc[ui], 0, 0, 0);
        if constexpr (PROJ) {  // dgi . W_i^T: dgi = (dgh's r and z columns | its own n columns)
          bf16x8 ai = af;
          if (ks >= 2 * H / 32)
            ai = *reinterpret_cast<const bf16x8*>(dnb + li * DNROW + (ks - 2 * H / 32) * 32 + 8 * lq);
#pragma unroll
          for (int ui = 0; ui < UTW; ++ui)
            acc0[ui] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wib[ui][ks], ai, acc0[ui], 0, 0, 0);
        }
      }
#pragma unroll
      for (int ui = 0; ui < UTW; ++ui) dh[ui] = dhp[ui] + spread4(acc[ui]);
      if constexpr (PROJ) {
        unsigned char* const r0 = pk0.rec((int)(t & 1), rr);
#pragma unroll
        for (int ui = 0; ui < UTW; ++ui) {
          if constexpr (UT % 4 != 0)
            if (wave + 4 * ui >= UT) continue;
          float v = spread4(acc0[ui]);
          v *= ((float)ycur[ui] > 0.0f ? 1.0f : 0.0f);  // relu' (the chain kernel's expression)
          *reinterpret_cast<bf16_t*>(r0 + 2 * ucol[ui]) = (bf16_t)v;
        }
        dnb = dnb == dn0 ? dn0 + GROWS * DNROW : dn0;
      }
      bf16_t* tmp = dg;  // the next step writes the other tile: one barrier per step
      dg = dgn_buf;
      dgn_buf = tmp;
    };
    int64_t t0 = T - 1;
    if (T >= PFW) {  // the first group peeled, the loop inside its branch (see the forward)
#pragma unroll
      for (int d = 0; d < PFW; ++d) step(t0 - d, inq[d]);
      for (t0 -= PFW; t0 - (PFW - 1) >= 0; t0 -= PFW) {
        if (done && !dwin.holds(t0 - (PFW - 1), t0))
          dwin.stage(done, t0 - (DONE_WIN - 1), T, B, row0, rpw, tid);
#pragma unroll
        for (int d = 0; d < PFW; ++d) step(t0 - d, inq[d]);
      }
    }
    if (done && t0 >= 0 && !dwin.holds(0, t0))
      dwin.stage(done, t0 - (DONE_WIN - 1), T, B, row0, rpw, tid);
#pragma unroll
    for (int d = 0; d < PFW; ++d)
      if (t0 - d >= 0) step(t0 - d, inq[d]);
    if constexpr (PROJ) {  // the dz0 rows of step 0
      __syncthreads();
      pk0.sweep(0, tid);
    }
    if (dh0) {
#pragma unroll
      for (int ui = 0; ui < UTW; ++ui) {
        if (wave + 4 * ui >= UT) continue;
        if (svalid) dh0[srowc * (unsigned)H + ucol[ui]] = dh[ui];
      }
    }
  } else {
    unsigned ucol[UTW];
    f32x4 dh[UTW];
    const RowLanes sl(lane, row0, B, rpw, GUARD);  // the arrangement the global stores go out in
    unsigned scol[UTW];
  #pragma unroll
    for (int ui = 0; ui < UTW; ++ui) {
      const int ut = wave + 4 * ui < UT ? wave + 4 * ui : 0;
      ucol[ui] = (unsigned)(ut * 16 + 4 * lq);
      scol[ui] = (unsigned)(ut * 16) + 4 * sl.chunk;
      dh[ui] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // per-step operands of the owned elements: (r, z, n, qn, g_h, h_prev); ring of PFW (the done
    // flags: DoneWindow)
    struct In {
      f32x4 v[UTW][6];
    };
    In inq[PFW];
    auto load_in = [&](int64_t t, In& dst) {
      const int64_t tc = t > 0 ? t : 0;  // before the start: reload step 0
      const float* gt = gates + tc * B * 4 * H;
      const float* ght = TAIL ? nullptr : g_h + tc * B * H;
      const float* hpt = h_prev + tc * B * H;
  #pragma unroll
      for (int ui = 0; ui < UTW; ++ui) {
        const unsigned og = rowc * (unsigned)(4 * H) + ucol[ui];
        const unsigned o = rowc * (unsigned)H + ucol[ui];
        dst.v[ui][0] = *reinterpret_cast<const f32x4*>(gt + og);
        dst.v[ui][1] = *reinterpret_cast<const f32x4*>(gt + og + (unsigned)H);
        dst.v[ui][2] = *reinterpret_cast<const f32x4*>(gt + og + (unsigned)(2 * H));
        dst.v[ui][3] = *reinterpret_cast<const f32x4*>(gt + og + (unsigned)(3 * H));
        if constexpr (!TAIL) dst.v[ui][4] = *reinterpret_cast<const f32x4*>(ght + o);
        dst.v[ui][5] = *reinterpret_cast<const f32x4*>(hpt + o);
      }
    };
  #pragma unroll
    for (int d = 0; d < PFW; ++d) load_in(T - 1 - d, inq[d]);
    if (done) dwin.stage(done, T - DONE_WIN, T, B, row0, rpw, tid);
    bf16_t* dg = dg0;
    bf16_t* dgn_buf = dg1;
    auto step = [&](int64_t t, In& in) {
      const bool reset = done != nullptr && dwin.at(t, li);
      f32x4 dhp[UTW];
      if constexpr (TAIL) {  // g_h[t] = dz_out[t] . W_out^T: one k-step, D[unit 4 lq + e][row li]
        const bf16x8 dzf = *reinterpret_cast<const bf16x8*>(dzs + ((int)t * GROWS + li) * DZROW +
                                                            8 * lq);
  #pragma unroll
        for (int ui = 0; ui < UTW; ++ui)
          in.v[ui][4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              wob[ui], dzf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      }
      float* gio = dgi + t * B * H3;
      float* gho = F32 ? dgh + t * B * H3 : nullptr;
      // bf16 image of dgh [T*B][3H]: the dz operand of the recurrent kernel's dW launch
      bf16_t* ghb = BF ? dgh_bf + t * B * H3 : nullptr;
  #pragma unroll
      for (int ui = 0; ui < UTW; ++ui) {
        if constexpr (UT % 4 != 0)
          if (wave + 4 * ui >= UT) continue;  // wave-uniform (a branch: see RowLanes)
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
        {  // out through the row-major lane arrangement (RowLanes)
          const f32x4 s_r = sl(a_r), s_z = sl(a_z), s_n = sl(a_n);
          f32x4 s_g = s_n;
          if constexpr (F32 || BF) s_g = sl(g_n);
          const unsigned o3 = sl.rowc * (unsigned)H3 + scol[ui];
          *reinterpret_cast<f32x4*>(gio + o3) = s_r;
          *reinterpret_cast<f32x4*>(gio + o3 + (unsigned)H) = s_z;
          *reinterpret_cast<f32x4*>(gio + o3 + (unsigned)(2 * H)) = s_n;
          if constexpr (F32) {
            *reinterpret_cast<f32x4*>(gho + o3) = s_r;
            *reinterpret_cast<f32x4*>(gho + o3 + (unsigned)H) = s_z;
            *reinterpret_cast<f32x4*>(gho + o3 + (unsigned)(2 * H)) = s_g;
          }
          if constexpr (BF) {
            *reinterpret_cast<bf16x4*>(ghb + o3) = to_bf16x4(s_r);
            *reinterpret_cast<bf16x4*>(ghb + o3 + (unsigned)H) = to_bf16x4(s_z);
            *reinterpret_cast<bf16x4*>(ghb + o3 + (unsigned)(2 * H)) = to_bf16x4(s_g);
          }
        }
        const bf16x4 b_r = to_bf16x4(a_r), b_z = to_bf16x4(a_z), b_n = to_bf16x4(g_n);
        *reinterpret_cast<bf16x4*>(dg + li * GROW + ucol[ui]) = b_r;
        *reinterpret_cast<bf16x4*>(dg + li * GROW + H + ucol[ui]) = b_z;
        *reinterpret_cast<bf16x4*>(dg + li * GROW + 2 * H + ucol[ui]) = b_n;
      }
      __builtin_amdgcn_sched_barrier(0);  // (as in the forward: the refill stays behind its slot's use)
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
    if (T >= PFW) {  // the first group peeled, the loop inside its branch (see the forward)
  #pragma unroll
      for (int d = 0; d < PFW; ++d) step(t0 - d, inq[d]);
      for (t0 -= PFW; t0 - (PFW - 1) >= 0; t0 -= PFW) {
        if (done && !dwin.holds(t0 - (PFW - 1), t0))
          dwin.stage(done, t0 - (DONE_WIN - 1), T, B, row0, rpw, tid);
  #pragma unroll
        for (int d = 0; d < PFW; ++d) step(t0 - d, inq[d]);
      }
    }
    if (done && t0 >= 0 && !dwin.holds(0, t0))
      dwin.stage(done, t0 - (DONE_WIN - 1), T, B, row0, rpw, tid);
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
}

bool mfma_shape_ok(int64_t H) { return H >= 32 && H <= 128 && H % 32 == 0; }
constexpr int kPack = 4;  // rows per workgroup of the packed-store instantiations
// dynamic LDS the tail forms may ask for (one workgroup per CU at these sizes anyway): the
// T x 16-row history of bf16(h_t) is most of it (T = 30, H = 64: 74 KB of ~100)
constexpr int kGruTailLds = 128 * 1024;

// Rows of its 16-row tile a workgroup fills.  A time step is a short dependent chain (LDS
// exchange, 2 KS MFMAs per gate, the gate math) whose length does not depend on the fill, and a
// CU drains the step's stores at ~20 B/clk (T = 30, B = 1024, H = 64 training forward: 15 us with
// the stores compiled out, 33 us with them, on 64 of the 256 CUs).  So a batch that would leave
// CUs idle is spread thinner: fewer rows per workgroup, more workgroups.  MIPPO_GRU_ROWS pins it.
int rows_per_group(int64_t B, int64_t T) {
  const char* e = getenv("MIPPO_GRU_ROWS");  // read per launch: the tests switch it
  const int pinned = e ? atoi(e) : 0;
  if (pinned == 4 || pinned == 8 || pinned == 16) return pinned;
  // a step or two (a rollout step through the generic containers): the launch is its latency,
  // and the 4-row form's prologue (zeroed tiles, record bookkeeping) costs ~2 us more
  if (T < 4) return GROWS;
  // full tiles once they cover the chip; below that 4 rows, whose stores go out packed
  // (PackedStores) — 8 rows with the direct stores measured no better than 16 in the forward
  return mippo::ceil_div(B, (int64_t)GROWS) >= 256 ? GROWS : kPack;
}

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
  const int rpw = rows_per_group(B, T);
  const bool pack = rpw == kPack;
  bf16_t* hpb = static_cast<bf16_t*>(h_prev_bf);
  const size_t lds =
      (size_t)2 * GROWS * (H + 8) * sizeof(bf16_t) + DONE_WIN * GROWS +
      (pack ? packed_lds(fwd_rec_bytes((int)H, h_prev_out != nullptr, hpb != nullptr, false), kPack)
            : 0);
  const dim3 grid((unsigned)mippo::ceil_div(B, (int64_t)rpw));
  hipStream_t st = mippo::as_stream(stream);
  const bool guard = rpw < GROWS || B % GROWS != 0;
#define MI_GRU_FWD(TRAIN, HH, GUARD, BF, PACK)                                                  \
  hipLaunchKernelGGL((gru_fwd_mfma_kernel<TRAIN, HH, GUARD, BF, false, PACK>), grid,            \
                     dim3(kThreads), lds, st, gi, w_h, b_hn, h0, done, h_out, h_prev_out,       \
                     gates_out, h_final, hpb, T, B, rpw, GruTail{})
#define MI_GRU_FWD_G(TRAIN, HH, BF)                    \
  {                                                    \
    if (pack) MI_GRU_FWD(TRAIN, HH, true, BF, kPack);  \
    else if (guard) MI_GRU_FWD(TRAIN, HH, true, BF, 0); \
    else MI_GRU_FWD(TRAIN, HH, false, BF, 0);          \
  }
#define MI_GRU_FWD_H(HH)                              \
  if (H == HH) {                                      \
    if (!h_prev_out) MI_GRU_FWD_G(false, HH, false)   \
    else if (hpb) MI_GRU_FWD_G(true, HH, true)        \
    else MI_GRU_FWD_G(true, HH, false)                \
  }
  MI_GRU_FWD_H(32)
  MI_GRU_FWD_H(64)
  MI_GRU_FWD_H(96)
  MI_GRU_FWD_H(128)
#undef MI_GRU_FWD_H
#undef MI_GRU_FWD_G
#undef MI_GRU_FWD
  return mippo::check_launch("mi_gru_seq_fwd_bf16");
}

// LDS of the TAIL form: the two carry tiles + the T x 16-row history + the head's rows
// (`front`: + the front layer's T x 4 input and output rows)
static size_t gru_tail_lds(int64_t T, int64_t H, int64_t N_out, bool front = false) {
  return (size_t)(2 + T) * GROWS * (H + 8) * sizeof(bf16_t) + DONE_WIN * GROWS +
         packed_lds(fwd_rec_bytes((int)H, true, true, true), kPack) +  // (whether packed or not)
         (front ? (size_t)T * 4 * (8 + H + 8) * sizeof(bf16_t) : 0) +
         (size_t)T * GROWS * N_out * 4;
}

extern "C" int mi_gru_seq_fwd_tail_supported(int64_t T, int64_t H, int64_t N_out) {
  return T >= 1 && mfma_shape_ok(H) && N_out >= 2 && N_out <= 16 && N_out % 2 == 0 &&
         gru_tail_lds(T, H, N_out) <= kGruTailLds;
}

extern "C" int mi_gru_seq_bwd_tail_supported(int64_t T, int64_t H, int64_t N_out);

namespace {

// The TAIL launches: `proj` null = gi read from memory, else the projection inside (PROJ).
int gru_fwd_tail_launch(const char* who, const GruProj* proj, const float* gi, const float* w_h,
                        const float* b_hn, const float* h0, const uint8_t* done, float* h_out,
                        float* h_prev_out, float* gates_out, float* h_final, void* h_prev_bf,
                        const void* w_out, const float* b_out, int64_t N_out, float* ms_out,
                        void* h_bf_out, const float* extras, const uint64_t* rng_state,
                        uint64_t offset_add, const float* eps2, float min_std, float std_scale,
                        float entropy_weight, float* loglik, float* reg, int64_t T, int64_t B,
                        int64_t H, mi_stream_t stream) {
  MI_REQUIRE(B >= 1 && B * 4 * H < (1LL << 31) && mi_gru_seq_fwd_tail_supported(T, H, N_out),
             "%s: T=%lld H=%lld N_out=%lld outside the supported class", who, (long long)T,
             (long long)H, (long long)N_out);
  MI_REQUIRE((gi || proj) && w_h && b_hn && h0 && h_out && h_prev_out && gates_out && h_final &&
                 h_prev_bf && w_out && ms_out && h_bf_out && extras && loglik && reg,
             "%s: null pointer", who);
  MI_REQUIRE(rng_state || eps2, "%s: need rng_state or injected entropy noise", who);
  MI_REQUIRE(al16(w_out) && al16(h_bf_out), "%s: buffers must be 16-byte aligned", who);
  GruTail tail = {proj ? *proj : GruProj{},
                  static_cast<const bf16_t*>(w_out),
                  b_out,
                  ms_out,
                  static_cast<bf16_t*>(h_bf_out),
                  {extras, {rng_state, offset_add, eps2, eps2}, nullptr, nullptr, nullptr, nullptr,
                   loglik, reg, (int)(N_out / 2), min_std, std_scale, entropy_weight, 0},
                  (int)N_out};
  const size_t lds = gru_tail_lds(T, H, N_out, proj && proj->x);
  MI_REQUIRE(lds <= (size_t)kGruTailLds, "%s: %zu bytes of LDS needed, %d available", who, lds,
             kGruTailLds);
  const int rpw = rows_per_group(B, T);
  MI_REQUIRE(!proj || rpw == kPack, "%s: the projection rides in the small-batch form only", who);
  const dim3 grid((unsigned)mippo::ceil_div(B, (int64_t)rpw));
  hipStream_t st = mippo::as_stream(stream);
  const bool guard = rpw < GROWS || B % GROWS != 0;
  bf16_t* hpb = static_cast<bf16_t*>(h_prev_bf);
#define MI_GRU_TAIL(HH, GUARD, PACK, PROJ)                                                         \
  {                                                                                                \
    static const hipError_t attr = hipFuncSetAttribute(                                            \
        reinterpret_cast<const void*>(                                                             \
            &gru_fwd_mfma_kernel<true, HH, GUARD, true, true, PACK, PROJ>),                        \
        hipFuncAttributeMaxDynamicSharedMemorySize, kGruTailLds);                                    \
    MI_REQUIRE(attr == hipSuccess, "%s: cannot raise the LDS limit", who);                         \
    hipLaunchKernelGGL((gru_fwd_mfma_kernel<true, HH, GUARD, true, true, PACK, PROJ>), grid,       \
                       dim3(kThreads), lds, st, gi, w_h, b_hn, h0, done, h_out, h_prev_out,        \
                       gates_out, h_final, hpb, T, B, rpw, tail);                                  \
  }
#define MI_GRU_TAIL_H(HH)                                  \
  if (H == HH) {                                           \
    if (proj && proj->x) MI_GRU_TAIL(HH, true, kPack, 2)   \
    else if (proj) MI_GRU_TAIL(HH, true, kPack, 1)         \
    else if (rpw == kPack) MI_GRU_TAIL(HH, true, kPack, 0) \
    else if (guard) MI_GRU_TAIL(HH, true, 0, 0)            \
    else MI_GRU_TAIL(HH, false, 0, 0)                      \
  }
  MI_GRU_TAIL_H(32)
  MI_GRU_TAIL_H(64)
  MI_GRU_TAIL_H(96)
  MI_GRU_TAIL_H(128)
#undef MI_GRU_TAIL_H
#undef MI_GRU_TAIL
  return mippo::check_launch(who);
}

}  // namespace

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
  MI_REQUIRE(gi, "mi_gru_seq_fwd_tail_bf16: null gi");
  return gru_fwd_tail_launch("mi_gru_seq_fwd_tail_bf16", nullptr, gi, w_h, b_hn, h0, done, h_out,
                             h_prev_out, gates_out, h_final, h_prev_bf, w_out, b_out, N_out,
                             ms_out, h_bf_out, extras, rng_state, offset_add, eps2, min_std,
                             std_scale, entropy_weight, loglik, reg, T, B, H, stream);
}

// 1 when the input projection can ride in the sequence launches too (mi_gru_seq_fwd_proj_tail_bf16
// / mi_gru_seq_bwd_proj_tail_bf16): the tail forms' class, a GRU whose input is as wide as its
// state, and a batch small enough for the 4-rows-per-workgroup form (`rows_per_group`).
extern "C" int mi_gru_seq_proj_supported(int64_t T, int64_t B, int64_t H, int64_t K_in,
                                         int64_t N_out) {
  return B >= 1 && K_in == H && mi_gru_seq_fwd_tail_supported(T, H, N_out) &&
         mi_gru_seq_bwd_tail_supported(T, H, N_out) && rows_per_group(B, T) == kPack;
}

// mi_gru_seq_fwd_tail_bf16 with gi = y W_i + b_i evaluated inside the launch — see GruProj.
// y_bf [T*B, ldy]: bf16 image of the GRU's input (K_in == H); w_i: forward fragment-major image
// of W_i; b_i [3H].  Bit-identical to the Dense chain's last layer + mi_gru_seq_fwd_tail_bf16.
extern "C" int mi_gru_seq_fwd_proj_tail_bf16(
    const void* y_bf, int64_t ldy, const void* w_i, const float* b_i, const float* w_h,
    const float* b_hn, const float* h0, const uint8_t* done, float* h_out, float* h_prev_out,
    float* gates_out, float* h_final, void* h_prev_bf, const void* w_out, const float* b_out,
    int64_t N_out, float* ms_out, void* h_bf_out, const float* extras, const uint64_t* rng_state,
    uint64_t offset_add, const float* eps2, float min_std, float std_scale, float entropy_weight,
    float* loglik, float* reg, int64_t T, int64_t B, int64_t H, mi_stream_t stream) {
  const char* who = "mi_gru_seq_fwd_proj_tail_bf16";
  MI_REQUIRE(y_bf && w_i && b_i && ldy >= H && ldy % 8 == 0 && al16(y_bf) && al16(w_i),
             "%s: projection operands missing or misaligned", who);
  MI_REQUIRE(mi_gru_seq_proj_supported(T, B, H, H, N_out), "%s: outside the supported class", who);
  const GruProj proj = {static_cast<const bf16_t*>(y_bf), ldy, static_cast<const bf16_t*>(w_i),
                        b_i,     nullptr, 0, nullptr, nullptr, nullptr};
  return gru_fwd_tail_launch(who, &proj, nullptr, w_h, b_hn, h0, done, h_out, h_prev_out,
                             gates_out, h_final, h_prev_bf, w_out, b_out, N_out, ms_out, h_bf_out,
                             extras, rng_state, offset_add, eps2, min_std, std_scale,
                             entropy_weight, loglik, reg, T, B, H, stream);
}

// 1 when the layer in front can ride in the forward launch too (its T x 4 input and output rows
// need room in LDS beside the history of h).
extern "C" int mi_gru_seq_front_supported(int64_t T, int64_t B, int64_t H, int64_t K0,
                                          int64_t N_out) {
  return K0 >= 1 && K0 <= 8 && mi_gru_seq_proj_supported(T, B, H, H, N_out) &&
         gru_tail_lds(T, H, N_out, true) <= (size_t)kGruTailLds;
}

// mi_gru_seq_fwd_proj_tail_bf16 with the relu Dense(K0 <= 8 -> H) in front of the GRU inside the
// launch as well — see GruProj (FRONT).  x [T*B, K0] fp32: the front layer's input; w_0: forward
// fragment-major image of its kernel; b_0 [H].  Out, besides mi_gru_seq_fwd_tail_bf16's: x_bf_out
// [T*B, 8] and y_bf_out [T*B, H], the bf16 images of the layer's input and output (the x
// operands of its own and of W_i's dW; y_bf_out is also what mi_gru_seq_bwd_proj_tail_bf16
// reads).  Bit-identical to mi_mlp_fwd_bf16 on that layer + mi_gru_seq_fwd_proj_tail_bf16.
extern "C" int mi_gru_seq_fwd_front_proj_tail_bf16(
    const float* x, int64_t K0, const void* w_0, const float* b_0, void* x_bf_out, void* y_bf_out,
    const void* w_i, const float* b_i, const float* w_h, const float* b_hn, const float* h0,
    const uint8_t* done, float* h_out, float* h_prev_out, float* gates_out, float* h_final,
    void* h_prev_bf, const void* w_out, const float* b_out, int64_t N_out, float* ms_out,
    void* h_bf_out, const float* extras, const uint64_t* rng_state, uint64_t offset_add,
    const float* eps2, float min_std, float std_scale, float entropy_weight, float* loglik,
    float* reg, int64_t T, int64_t B, int64_t H, mi_stream_t stream) {
  const char* who = "mi_gru_seq_fwd_front_proj_tail_bf16";
  MI_REQUIRE(x && w_0 && b_0 && x_bf_out && y_bf_out && w_i && b_i && K0 >= 1 && K0 <= 8 &&
                 al16(w_0) && al16(b_0) && al16(x_bf_out) && al16(y_bf_out) && al16(w_i),
             "%s: front / projection operands missing, misaligned or K0 outside 1..8", who);
  MI_REQUIRE(mi_gru_seq_front_supported(T, B, H, K0, N_out), "%s: outside the supported class",
             who);
  const GruProj proj = {static_cast<const bf16_t*>(y_bf_out),
                        H,
                        static_cast<const bf16_t*>(w_i),
                        b_i,
                        x,
                        (int)K0,
                        static_cast<const bf16_t*>(w_0),
                        b_0,
                        static_cast<bf16_t*>(x_bf_out)};
  return gru_fwd_tail_launch(who, &proj, nullptr, w_h, b_hn, h0, done, h_out, h_prev_out,
                             gates_out, h_final, h_prev_bf, w_out, b_out, N_out, ms_out, h_bf_out,
                             extras, rng_state, offset_add, eps2, min_std, std_scale,
                             entropy_weight, loglik, reg, T, B, H, stream);
}

extern "C" int mi_gru_seq_bwd_bf16(const float* g_h, const float* gates, const float* h_prev,
                                   const float* w_h, const uint8_t* done, float* dgi, float* dgh,
                                   float* dh0, void* dgh_bf, int64_t T, int64_t B, int64_t H,
                                   mi_stream_t stream) {
  MI_REQUIRE(T >= 1 && B >= 1 && mfma_shape_ok(H) && B * 4 * H < (1LL << 31),
             "mi_gru_seq_bwd_bf16: bad shape");
  MI_REQUIRE(g_h && gates && h_prev && w_h && dgi && (dgh || dgh_bf),
             "mi_gru_seq_bwd_bf16: null pointer (dgh or its bf16 image is required)");
  const int rpw = rows_per_group(B, T);
  const bool pack = rpw == kPack;
  bf16_t* gb = static_cast<bf16_t*>(dgh_bf);
  const size_t lds =
      (size_t)2 * GROWS * (3 * H + 8) * sizeof(bf16_t) + DONE_WIN * GROWS +
      (pack ? packed_lds(bwd_rec_bytes((int)H, dgh != nullptr, gb != nullptr), kPack) : 0);
  const dim3 grid((unsigned)mippo::ceil_div(B, (int64_t)rpw));
  hipStream_t st = mippo::as_stream(stream);
  const bool guard = rpw < GROWS || B % GROWS != 0;
#define MI_GRU_BWD(HH, GUARD, F32, BF, PACK)                                                    \
  hipLaunchKernelGGL((gru_bwd_mfma_kernel<HH, GUARD, F32, BF, false, PACK>), grid,              \
                     dim3(kThreads), lds, st, g_h, gates, h_prev, w_h, done, dgi, dgh, dh0, gb, \
                     T, B, rpw, GruBwdTail{})
#define MI_GRU_BWD_G(HH, F32, BF)                     \
  {                                                   \
    if (pack) MI_GRU_BWD(HH, true, F32, BF, kPack);   \
    else if (guard) MI_GRU_BWD(HH, true, F32, BF, 0); \
    else MI_GRU_BWD(HH, false, F32, BF, 0);           \
  }
#define MI_GRU_BWD_H(HH)                        \
  if (H == HH) {                                \
    if (dgh && gb) MI_GRU_BWD_G(HH, true, true) \
    else if (gb) MI_GRU_BWD_G(HH, false, true)  \
    else MI_GRU_BWD_G(HH, true, false)          \
  }
  MI_GRU_BWD_H(32)
  MI_GRU_BWD_H(64)
  MI_GRU_BWD_H(96)
  MI_GRU_BWD_H(128)
#undef MI_GRU_BWD_H
#undef MI_GRU_BWD_G
#undef MI_GRU_BWD
  return mippo::check_launch("mi_gru_seq_bwd_bf16");
}

static size_t gru_bwd_tail_lds(int64_t T, int64_t H) {
  return (size_t)2 * GROWS * (3 * H + 8) * sizeof(bf16_t) + DONE_WIN * GROWS +
         packed_lds(bwd_rec_bytes((int)H, false, true), kPack) +  // (whether packed or not)
         (size_t)2 * GROWS * (H + 8) * sizeof(bf16_t) + packed_lds(2 * (int)H, kPack) +  // (PROJ)
         (size_t)T * GROWS * (32 + 8) * 2;
}

extern "C" int mi_gru_seq_bwd_tail_supported(int64_t T, int64_t H, int64_t N_out) {
  return T >= 1 && mfma_shape_ok(H) && N_out >= 2 && N_out <= 16 && N_out % 2 == 0 &&
         gru_bwd_tail_lds(T, H) <= kGruTailLds;
}

namespace {

// The BPTT TAIL launches: `proj` null = fp32 dgi out, else the projection's backward inside.
int gru_bwd_tail_launch(const char* who, const GruBwdProj* proj, const float* gates,
                        const float* h_prev, const float* w_h, const uint8_t* done, float* dgi,
                        float* dh0, void* dgh_bf, const void* w_out_bwd, int64_t N_out,
                        const float* mean_and_std, const float* extras, const uint64_t* rng_state,
                        uint64_t offset_add, const float* eps2, const float* g_loglik, float g_reg,
                        float min_std, float std_scale, float entropy_weight, void* dz_out_bf,
                        int64_t T, int64_t B, int64_t H, mi_stream_t stream) {
  MI_REQUIRE(B >= 1 && B * 4 * H < (1LL << 31) && mi_gru_seq_bwd_tail_supported(T, H, N_out),
             "%s: T=%lld H=%lld N_out=%lld outside the supported class", who, (long long)T,
             (long long)H, (long long)N_out);
  MI_REQUIRE(gates && h_prev && w_h && (dgi || proj) && dgh_bf && w_out_bwd && mean_and_std &&
                 extras && dz_out_bf,
             "%s: null pointer", who);
  MI_REQUIRE(rng_state || eps2, "%s: need rng_state or injected entropy noise", who);
  MI_REQUIRE(al16(w_out_bwd) && al16(dz_out_bf), "%s: buffers must be 16-byte aligned", who);
  GruBwdTail tail = {proj ? *proj : GruBwdProj{},
                     static_cast<const bf16_t*>(w_out_bwd),
                     static_cast<bf16_t*>(dz_out_bf),
                     mippo::ceil_div(N_out, 8) * 8,
                     {mean_and_std, extras, {rng_state, offset_add, eps2, eps2}, g_loglik, g_reg,
                      (int)(N_out / 2), min_std, std_scale, entropy_weight},
                     (int)N_out};
  const size_t lds = gru_bwd_tail_lds(T, H);
  const int rpw = rows_per_group(B, T);
  MI_REQUIRE(!proj || rpw == kPack, "%s: the projection rides in the small-batch form only", who);
  const dim3 grid((unsigned)mippo::ceil_div(B, (int64_t)rpw));
  hipStream_t st = mippo::as_stream(stream);
  const bool guard = rpw < GROWS || B % GROWS != 0;
  bf16_t* gb = static_cast<bf16_t*>(dgh_bf);
#define MI_GRU_BT(HH, GUARD, PACK, PROJ)                                                         \
  {                                                                                              \
    static const hipError_t attr = hipFuncSetAttribute(                                          \
        reinterpret_cast<const void*>(                                                           \
            &gru_bwd_mfma_kernel<HH, GUARD, false, true, true, PACK, PROJ>),                     \
        hipFuncAttributeMaxDynamicSharedMemorySize, kGruTailLds);                                  \
    MI_REQUIRE(attr == hipSuccess, "%s: cannot raise the LDS limit", who);                       \
    hipLaunchKernelGGL((gru_bwd_mfma_kernel<HH, GUARD, false, true, true, PACK, PROJ>), grid,    \
                       dim3(kThreads), lds, st, nullptr, gates, h_prev, w_h, done, dgi, nullptr, \
                       dh0, gb, T, B, rpw, tail);                                                \
  }
#define MI_GRU_BT_H(HH)                                      \
  if (H == HH) {                                             \
    if (proj) MI_GRU_BT(HH, true, kPack, true)               \
    else if (rpw == kPack) MI_GRU_BT(HH, true, kPack, false) \
    else if (guard) MI_GRU_BT(HH, true, 0, false)            \
    else MI_GRU_BT(HH, false, 0, false)                      \
  }
  MI_GRU_BT_H(32)
  MI_GRU_BT_H(64)
  MI_GRU_BT_H(96)
  MI_GRU_BT_H(128)
#undef MI_GRU_BT_H
#undef MI_GRU_BT
  return mippo::check_launch(who);
}

}  // namespace

// mi_gru_seq_bwd_bf16 (bf16 image of dgh out) with the sampler's backward and the head's dX in
// front of it INSIDE the launch — see GruBwdTail.  w_out_bwd: backward fragment-major image of
// the head's kernel; ms / extras / rng / g_ll / g_reg: mi_tanh_gauss_bwd_f32's operands with
// rows t * B + env; dz_out_bf [T*B, pad8(N_out)]: the head's output-gradient image (its dW
// operand), written here.
extern "C" int mi_gru_seq_bwd_tail_bf16(
    const float* gates, const float* h_prev, const float* w_h, const uint8_t* done, float* dgi,
    float* dh0, void* dgh_bf, const void* w_out_bwd, int64_t N_out, const float* mean_and_std,
    const float* extras, const uint64_t* rng_state, uint64_t offset_add, const float* eps2,
    const float* g_loglik, float g_reg, float min_std, float std_scale, float entropy_weight,
    void* dz_out_bf, int64_t T, int64_t B, int64_t H, mi_stream_t stream) {
  MI_REQUIRE(dgi, "mi_gru_seq_bwd_tail_bf16: null dgi");
  return gru_bwd_tail_launch("mi_gru_seq_bwd_tail_bf16", nullptr, gates, h_prev, w_h, done, dgi,
                             dh0, dgh_bf, w_out_bwd, N_out, mean_and_std, extras, rng_state,
                             offset_add, eps2, g_loglik, g_reg, min_std, std_scale,
                             entropy_weight, dz_out_bf, T, B, H, stream);
}

// mi_gru_seq_bwd_tail_bf16 with the backward of the input projection inside — see GruBwdProj.
// y_bf [T*B, ldy == H]: the post-relu bf16 image the forward read; w_i_bwd: backward
// fragment-major image of W_i.  Out: dgi_bf [T*B, 3H] (instead of fp32 dgi) and dz0_bf [T*B, H],
// the dz operands of W_i's and of the relu layer's dW.  Bit-identical to
// mi_gru_seq_bwd_tail_bf16 + the Dense chain's backward (mi_mlp_bwd_dx_bf16) on its dgi.
extern "C" int mi_gru_seq_bwd_proj_tail_bf16(
    const void* y_bf, int64_t ldy, const void* w_i_bwd, void* dgi_bf, void* dz0_bf,
    const float* gates, const float* h_prev, const float* w_h, const uint8_t* done, float* dh0,
    void* dgh_bf, const void* w_out_bwd, int64_t N_out, const float* mean_and_std,
    const float* extras, const uint64_t* rng_state, uint64_t offset_add, const float* eps2,
    const float* g_loglik, float g_reg, float min_std, float std_scale, float entropy_weight,
    void* dz_out_bf, int64_t T, int64_t B, int64_t H, mi_stream_t stream) {
  const char* who = "mi_gru_seq_bwd_proj_tail_bf16";
  MI_REQUIRE(y_bf && w_i_bwd && dgi_bf && dz0_bf && ldy == H && al16(w_i_bwd) && al16(dgi_bf) &&
                 al16(dz0_bf),
             "%s: projection operands missing, misaligned or ldy != H", who);
  MI_REQUIRE(mi_gru_seq_proj_supported(T, B, H, H, N_out), "%s: outside the supported class", who);
  const GruBwdProj proj = {static_cast<const bf16_t*>(y_bf), ldy,
                           static_cast<const bf16_t*>(w_i_bwd), static_cast<bf16_t*>(dgi_bf),
                           static_cast<bf16_t*>(dz0_bf)};
  return gru_bwd_tail_launch(who, &proj, gates, h_prev, w_h, done, nullptr, dh0, dgh_bf,
                             w_out_bwd, N_out, mean_and_std, extras, rng_state, offset_add, eps2,
                             g_loglik, g_reg, min_std, std_scale, entropy_weight, dz_out_bf, T, B,
                             H, stream);
}
