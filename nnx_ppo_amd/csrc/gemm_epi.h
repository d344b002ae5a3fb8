// Epilogue descriptor of the per-layer NT GEMMs (gemm_bf16.hip: 128-row tiles, register
// staged; gemm256_bf16.hip: 256 x 256 tiles, direct-to-LDS loads), shared so that both
// translation units evaluate the same fused epilogues.
#pragma once
#include "bf16_common.h"

namespace mippo_gemm {

using mippo_bf16::bf16_t;

enum { EPI_FWD = 0, EPI_DX = 1 };

struct Epi {
  const float* bias;   // FWD: [J] or null
  int act;             // FWD
  float* out_f32;      // [I][ld_f32] or null
  int64_t ld_f32;
  bf16_t* out_bf;      // [I][ld_bf] or null (padding columns are written as zero)
  int64_t ld_bf;
  bf16_t* aux_bf;      // FWD: pre-activation [I][ld_bf] or null
  const bf16_t* prev;  // DX: previous layer's output (pre-activation for swish) or null
  int64_t ld_prev;
  int prev_act;        // DX
};

// gemm256_bf16.hip: 1 if the 256 x 256 kernel took the product (launched), 0 if the shape is
// outside its class (the caller runs the 128-row kernel), negative on error.
int nt256_launch(int epi, const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int64_t I,
                 int64_t J, int64_t R, const Epi& ep, hipStream_t st);

// dW on 256 x 256 tiles (gemm256_bf16.hip): which problems it takes, its split plan, its
// launch (slabs[l]: [S][K * N + N] fp32, the layout the 128-row path leaves).
bool dw256_takes(int64_t K, int64_t N, int64_t M);
bool dw256_candidate(int64_t K, int64_t N, int64_t M);
void dw256_plan(int64_t M, int64_t tiles, int64_t* rows, int64_t* S);
int64_t dw256_tiles(int64_t K, int64_t N);
int dw256_launch(int n, const bf16_t* const* x_bf, const bf16_t* const* dz_bf, const int64_t* K,
                 const int64_t* N, int64_t M, float* const* slabs, int64_t rows, int64_t S,
                 hipStream_t st);
bool dw128_takes(int64_t M, int64_t rows_per_split, int n, int64_t wide_tiles, int64_t tiles);
int dw128_launch(int n, const bf16_t* const* x_bf, const bf16_t* const* dz_bf, const int64_t* K,
                 const int64_t* N, int64_t M, float* const* slabs, int64_t rows, int64_t S,
                 hipStream_t st);

}  // namespace mippo_gemm
