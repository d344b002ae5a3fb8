// Device-side view of the one-shot peer exchange's regions (comm.hip has the protocol): the
// region layout, the per-chunk flags and the header, shared with kernels of other translation
// units that exchange through the same regions (trunk_ws.hip: the advantage statistics of a
// sharded minibatch inside mi_policy_ws_bwd_gae_bf16).
#pragma once
#include "bf16_common.h"

namespace mippo_comm {

constexpr int kCommThreads = 256;
constexpr int kMaxWorld = 16;
constexpr int64_t kChunkBytes = 4096;     // one block-pass: 256 threads x 16 B
constexpr int64_t kHeaderBytes = 4096;
constexpr int kMaxBlocks = 256;

struct CommHeader {
  unsigned long long seq;      // collectives completed on this rank
  unsigned int ticket;         // blocks finished in the running collective
  unsigned int errors;         // spins that timed out (sticky)
  unsigned long long timeout;  // wall-clock ticks (100 MHz) a spin may last
};

struct CommDev {               // by value in the kernel arguments
  int rank, world;
  int64_t slot_bytes, chunks;
  unsigned int* error_word;    // nullable: bumped with hdr->errors (mi_comm_set_error_word)
  char* peer[kMaxWorld];
};

__host__ __device__ inline int64_t flags_off(const int64_t chunks, int world, int parity, int r) {
  return kHeaderBytes + ((int64_t)(parity * world + r) * chunks) * 4;
}
__host__ __device__ inline int64_t slots_base(const int64_t chunks, int world) {
  const int64_t f = kHeaderBytes + (int64_t)2 * world * chunks * 4;
  return (f + 4095) / 4096 * 4096;
}
__host__ __device__ inline int64_t slot_off(const int64_t chunks, int world, int64_t slot_bytes,
                                            int parity, int r) {
  return slots_base(chunks, world) + (int64_t)(parity * world + r) * slot_bytes;
}


// comm.hip: the device view of a communicator handle (false: null / not connected)
bool dev_view_of(const void* comm, CommDev* out);

// The collective counts itself once every block has finished (so a block that starts late
// still reads the old `seq`).
__device__ inline void comm_finish(CommHeader* hdr, unsigned long long seq_new) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int blocks = gridDim.x;
    if (__hip_atomic_fetch_add(&hdr->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
        blocks - 1) {
      hdr->seq = seq_new;
      __hip_atomic_store(&hdr->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace mippo_comm
