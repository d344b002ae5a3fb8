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

#ifdef __HIPCC__
// "Last block done" ticket: every block calls this after THREAD 0 wrote the
// block's partial results; it returns true (to all threads) in the block that
// arrives last.  A thread of that block that reads the other blocks' partials
// executes __threadfence() first (acquire).  `counter` must be zero before the
// launch and is left zero by the caller (`*counter = 0` in the last block), so
// launches ordered on one stream can share it.  Saves the separate one-block
// "finalize" launch of a two-pass reduction (~5 us + a dependency edge at this
// workload's sizes) while keeping a fixed summation order.  Only thread 0 fences:
// a device-scope release writes back the XCD's L2, and 16 waves doing so per block
// cost more than the launch it saves.
__device__ inline bool last_block_ticket(unsigned int* counter) {
  __shared__ bool is_last;
  if (threadIdx.x == 0) {
    __threadfence();  // release: this block's partials are visible device-wide
    const unsigned int blocks = gridDim.x * gridDim.y * gridDim.z;
    is_last = atomicAdd(counter, 1u) == blocks - 1;
  }
  __syncthreads();
  return is_last;
}
#endif

}  // namespace mippo

#define MI_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      ::mippo::set_error(__VA_ARGS__);   \
      return -EINVAL;                    \
    }                                    \
  } while (0)
