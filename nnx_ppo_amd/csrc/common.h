// Shared host-side helpers for libmippo (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <errno.h>
#include <stdint.h>

#include "mippo.h"

namespace mippo {

// printf-style; stores into a thread_local buffer returned by mi_last_error().
void set_error(const char* fmt, ...);

// Checks hipGetLastError() after a launch; returns 0 or -EIO (and records it).
int check_launch(const char* what);

inline hipStream_t as_stream(mi_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Memory-bound kernels: cap the grid at 8 blocks per CU and grid-stride the rest.
constexpr int kNumCU = 256;
constexpr int kMaxStreamBlocks = kNumCU * 8;

}  // namespace mippo

#define MI_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      ::mippo::set_error(__VA_ARGS__);   \
      return -EINVAL;                    \
    }                                    \
  } while (0)
