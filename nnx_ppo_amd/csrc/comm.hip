// e — one-shot peer exchange over IPC-mapped device buffers (SURVEY §5 last row, §8b
// `mi_allreduce_oneshot`, §8e "Transport").  The reference is single-device: it has no
// collective at all, so everything here is new work that BASELINE.json's north_star
// asks for (env-sharded data parallelism, gradient all-reduce over xGMI).
//
// Why not a ring: the messages are tiny (the whole gradient arena is 322 KB at C2/C5, the
// advantage statistics are 24 bytes) and an 8-GPU MI355X node is fully connected
// point-to-point (7 xGMI links per GPU), so a ring all-reduce pays 14 dependent hops of
// latency for nothing.  One shot: every rank WRITES its contribution into a slot it owns
// in every peer's buffer (7 concurrent link transfers, one hop), raises a per-chunk flag
// behind a system-scope release, waits for the same flags from its peers, and reduces the
// `world` slots locally in rank order — so every rank computes bit-identical sums.  The
// exchange is a plain kernel on the launch stream: a sharded iteration is ONE HIP graph.
//
// Memory (one hipExtMallocWithFlags(..., hipDeviceMallocUncached) region per rank, exported
// with hipIpcGetMemHandle and mapped by every peer):
//   [0, 4096)                 header (local use): seq, finish ticket, error count
//   flags [2][world][chunks]  uint32 sequence numbers, written by the owning peer
//   slots [2][world][slot]    payload, slot (p, r) written by rank r
// Two parities: a rank can run at most one collective ahead of a peer (finishing
// collective k needs every peer's flags for k, which a peer raises only after it has
// finished READING collective k-1), so slot set k&1 is never overwritten while in use.
//
// Every spin is bounded (wall clock): a peer that never arrives makes the kernel count an
// error and finish with garbage instead of hanging the GPU; the host reads the count
// (mi_comm_status) and raises.
#include "bf16_common.h"
#include "comm_common.h"
#include "optim_common.h"

namespace {

using namespace mippo_comm;
constexpr int kThreads = kCommThreads;

struct Comm {                  // host object behind the opaque handle
  int rank, world;
  int64_t slot_bytes, chunks, region_bytes;
  char* local;                 // this rank's region
  char* peer[kMaxWorld];       // every rank's region as mapped here (peer[rank] == local)
  bool opened[kMaxWorld];
  unsigned int* error_word;    // caller-owned device word that mirrors hdr->errors (nullable)
};

CommDev dev_view(const Comm* c) {
  CommDev d;
  d.rank = c->rank;
  d.world = c->world;
  d.slot_bytes = c->slot_bytes;
  d.chunks = c->chunks;
  d.error_word = c->error_word;
  for (int r = 0; r < kMaxWorld; ++r) d.peer[r] = r < c->world ? c->peer[r] : nullptr;
  return d;
}

// ---- device side ---------------------------------------------------------------------

// Push one chunk of this rank's payload into slot (parity, me) of every peer, then raise
// the chunk's flag there.  `src` points at the chunk (nbytes <= kChunkBytes of it valid).
__device__ inline void push_chunk(const CommDev& c, int parity, unsigned int seq, int64_t chunk,
                                  const char* __restrict__ src, int64_t nbytes) {
  const int64_t off = slot_off(c.chunks, c.world, c.slot_bytes, parity, c.rank) +
                      chunk * kChunkBytes;
  const int tid = threadIdx.x;
  const bool vec = (nbytes % 16 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
  for (int r = 0; r < c.world; ++r) {
    if (r == c.rank) continue;  // own contribution is read from `src` directly
    char* dst = c.peer[r] + off;
    if (vec) {
      if ((int64_t)tid * 16 < nbytes)
        reinterpret_cast<uint4*>(dst)[tid] = reinterpret_cast<const uint4*>(src)[tid];
    } else {
      for (int64_t i = tid; i < nbytes; i += kThreads) dst[i] = src[i];
    }
  }
  // system-scope release by every storing thread (the stores of one wave to a peer are
  // not ordered with another wave's), then one lane raises the flags
  __threadfence_system();
  __syncthreads();
  if (tid < c.world && tid != c.rank) {
    unsigned int* flag = reinterpret_cast<unsigned int*>(
        c.peer[tid] + flags_off(c.chunks, c.world, parity, c.rank)) + chunk;
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// push_chunk with the payload in LDS (a chunk = kChunkBytes / 16 = 256 16-byte pieces, one
// per thread; the tail of a short chunk is zero-filled by the caller)
__device__ inline void push_chunk_lds(const CommDev& c, int parity, unsigned int seq,
                                      int64_t chunk, const float* __restrict__ src_lds,
                                      int64_t nbytes) {
  const int64_t off = slot_off(c.chunks, c.world, c.slot_bytes, parity, c.rank) +
                      chunk * kChunkBytes;
  const int tid = threadIdx.x;
  const uint4 mine = reinterpret_cast<const uint4*>(src_lds)[tid];
  for (int r = 0; r < c.world; ++r) {
    if (r == c.rank) continue;
    if ((int64_t)tid * 16 < nbytes) reinterpret_cast<uint4*>(c.peer[r] + off)[tid] = mine;
  }
  __threadfence_system();
  __syncthreads();
  if (tid < c.world && tid != c.rank) {
    unsigned int* flag = reinterpret_cast<unsigned int*>(
        c.peer[tid] + flags_off(c.chunks, c.world, parity, c.rank)) + chunk;
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Wait until every peer has raised this chunk's flag for `seq` (bounded), then make the
// payload they wrote visible to every wave of the workgroup.  Returns false (to every
// thread of the workgroup) if a wait ran out — the chunk's slots then hold garbage and the
// caller must not use them: the all-reduce / all-gather write NaN, the fused optimiser
// skips the update.  The error is sticky (hdr->errors, mirrored into the caller's error
// word that travels to the host with every iteration's metrics).
__device__ inline bool wait_chunk(const CommDev& c, CommHeader* hdr, int parity,
                                  unsigned int seq, int64_t chunk) {
  const int tid = threadIdx.x;
  int bad = 0;
  if (tid < c.world && tid != c.rank) {
    const unsigned int* flag = reinterpret_cast<const unsigned int*>(
        c.peer[c.rank] + flags_off(c.chunks, c.world, parity, tid)) + chunk;
    const unsigned long long t0 = wall_clock64();
    const unsigned long long limit = hdr->timeout;
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > limit) {
        atomicAdd(&hdr->errors, 1u);
        if (c.error_word) atomicAdd(c.error_word, 1u);
        bad = 1;
        break;
      }
    }
  }
  bad = __syncthreads_or(bad);
  // every wave invalidates for itself (an acquire only covers the issuing wave's later
  // loads); system scope: the payload was written by another device
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  return !bad;
}

template <typename T>
__device__ inline T poison();
template <>
__device__ inline float poison<float>() { return __builtin_nanf(""); }
template <>
__device__ inline double poison<double>() { return __builtin_nan(""); }

template <typename T>
__global__ void __launch_bounds__(kThreads)
allreduce_kernel(CommDev c, T* __restrict__ buf, int64_t n, T scale) {
  CommHeader* hdr = reinterpret_cast<CommHeader*>(c.peer[c.rank]);
  const unsigned long long seq64 = hdr->seq + 1;
  const unsigned int seq = (unsigned int)seq64;
  const int parity = (int)(seq64 & 1);
  constexpr int kPer = (int)(kChunkBytes / sizeof(T));  // elements per chunk
  const int64_t nchunks = mippo::ceil_div(n, (int64_t)kPer);
  // Sticky: after one lost peer this rank neither pushes nor waits any more (it would run
  // ahead of the two-parity protocol and overwrite slots a late peer is still reading) and
  // hands back NaN; its peers then time out on IT, so the failure reaches every rank
  const bool failed = hdr->errors != 0;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t e0 = ch * kPer;
    const int64_t cnt = n - e0 < kPer ? n - e0 : kPer;
    if (!failed)
      push_chunk(c, parity, seq, ch, reinterpret_cast<const char*>(buf + e0),
                 cnt * (int64_t)sizeof(T));
    const bool ok = !failed && wait_chunk(c, hdr, parity, seq, ch);  // lost peer: no more waits
    for (int64_t i = threadIdx.x; i < cnt; i += kThreads) {
      T s = ok ? T(0) : poison<T>();  // a peer never arrived: NaN, never a partial sum
      for (int r = 0; r < c.world; ++r) {  // fixed rank order: identical on every rank
        const T v = r == c.rank
                        ? buf[e0 + i]
                        : reinterpret_cast<const T*>(
                              c.peer[c.rank] + slot_off(c.chunks, c.world, c.slot_bytes, parity, r) +
                              ch * kChunkBytes)[i];
        s += v;
      }
      buf[e0 + i] = s * scale;
    }
    __syncthreads();  // the flags of the next chunk are raised by other lanes
  }
  comm_finish(hdr, seq64);
}

__global__ void __launch_bounds__(kThreads)
allgather_kernel(CommDev c, const char* __restrict__ src, int64_t nbytes, char* __restrict__ dst) {
  CommHeader* hdr = reinterpret_cast<CommHeader*>(c.peer[c.rank]);
  const unsigned long long seq64 = hdr->seq + 1;
  const unsigned int seq = (unsigned int)seq64;
  const int parity = (int)(seq64 & 1);
  const int64_t nchunks = mippo::ceil_div(nbytes, kChunkBytes);
  const bool failed = hdr->errors != 0;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t b0 = ch * kChunkBytes;
    const int64_t cnt = nbytes - b0 < kChunkBytes ? nbytes - b0 : kChunkBytes;
    if (!failed) push_chunk(c, parity, seq, ch, src + b0, cnt);
    const bool ok = !failed && wait_chunk(c, hdr, parity, seq, ch);
    for (int r = 0; r < c.world; ++r) {
      const char* from = r == c.rank
                             ? src + b0
                             : c.peer[c.rank] +
                                   slot_off(c.chunks, c.world, c.slot_bytes, parity, r) + b0;
      // a peer never arrived: all-ones bytes (NaN as fp32 / fp64) instead of its stale slot
      for (int64_t i = threadIdx.x; i < cnt; i += kThreads)
        dst[(int64_t)r * nbytes + b0 + i] = (ok || r == c.rank) ? from[i] : (char)0xFF;
    }
    __syncthreads();
  }
  comm_finish(hdr, seq64);
}

// The optimiser step with the gradient exchange inside it: chunk by chunk, push this
// rank's gradients, wait for the peers', reduce in rank order, scale by 1/world (the
// all-reduce-MEAN of equal shards) and run the Adam body on the result.
__global__ void __launch_bounds__(kThreads)
adam_allreduce_kernel(CommDev c, mippo_optim::AdamArgs a) {
  CommHeader* hdr = reinterpret_cast<CommHeader*>(c.peer[c.rank]);
  const unsigned long long seq64 = hdr->seq + 1;
  const unsigned int seq = (unsigned int)seq64;
  const int parity = (int)(seq64 & 1);
  const mippo_optim::AdamStep st = mippo_optim::adam_begin(a);
  constexpr int kPer = (int)(kChunkBytes / sizeof(float));  // 1024 elements = 4 passes of 256
  const int64_t nchunks = mippo::ceil_div(a.n, (int64_t)kPer);
  const float inv_world = 1.0f / (float)c.world;
  // this rank's gradient chunk WITH the pending dW slabs summed in (mi_adam_step_slabs_f32's
  // sums, the order of reduce_slabs_grouped_kernel): staged in LDS, pushed from there — the
  // slab reduction needs no launch of its own in front of the exchange
  __shared__ __attribute__((aligned(16))) float s_g[kChunkBytes / sizeof(float)];
  // Sticky: once a peer has failed to arrive (this launch or an earlier one) no update is
  // applied any more — parameters, moments and bf16 images keep their last good values and
  // the error word stops the run at this iteration's host sync (loop.IterationRunner.collect)
  const bool failed = hdr->errors != 0;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
    const int64_t e0 = ch * kPer;
    const int64_t cnt = a.n - e0 < kPer ? a.n - e0 : kPer;
#pragma unroll
    for (int pass = 0; pass < kPer; pass += kThreads) {
      const int64_t i = e0 + pass + threadIdx.x;
      float gv = 0.0f;
      if (pass + threadIdx.x < cnt) {
        gv = a.g[i];
        if (a.slabs.n) gv = gv + mippo_optim::slab_sum(a, i, e0 + pass);
      }
      s_g[pass + threadIdx.x] = gv;
    }
    __syncthreads();
    if (!failed) push_chunk_lds(c, parity, seq, ch, s_g, cnt * 4);
    const bool ok = !failed && wait_chunk(c, hdr, parity, seq, ch);  // lost peer: no more waits
    for (int64_t pass = 0; pass < cnt && ok; pass += kThreads) {
      const int64_t i = e0 + pass + threadIdx.x;
      if (i < a.n && pass + threadIdx.x < cnt) {
        float s = 0.0f;
        for (int r = 0; r < c.world; ++r) {
          const float v = r == c.rank
                              ? s_g[pass + threadIdx.x]
                              : reinterpret_cast<const float*>(
                                    c.peer[c.rank] +
                                    slot_off(c.chunks, c.world, c.slot_bytes, parity, r) +
                                    ch * kChunkBytes)[pass + threadIdx.x];
          s += v;
        }
        mippo_optim::adam_element(a, st, i, e0 + pass, s * inv_world);
      }
    }
    __syncthreads();
  }
  mippo_optim::adam_end(a, st);
  comm_finish(hdr, seq64);
}

int grid_for(int64_t nchunks) {
  return (int)(nchunks < 1 ? 1 : (nchunks > kMaxBlocks ? kMaxBlocks : nchunks));
}

}  // namespace

// ---- host side --------------------------------------------------------------------------

namespace mippo_comm {
bool dev_view_of(const void* comm, CommDev* out) {
  if (!comm || !out) return false;
  const Comm* c = static_cast<const Comm*>(comm);
  for (int r = 0; r < c->world; ++r)
    if (!c->peer[r]) return false;
  *out = dev_view(c);
  return true;
}
}  // namespace mippo_comm

extern "C" int64_t mi_comm_handle_bytes(void) { return (int64_t)sizeof(hipIpcMemHandle_t); }

extern "C" int mi_comm_create(int rank, int world, int64_t slot_bytes, double timeout_seconds,
                              void** comm_out, void* handle_out) {
  MI_REQUIRE(comm_out && handle_out, "mi_comm_create: null pointer");
  MI_REQUIRE(world >= 1 && world <= kMaxWorld && rank >= 0 && rank < world,
             "mi_comm_create: bad rank %d / world %d (world <= %d)", rank, world, kMaxWorld);
  MI_REQUIRE(slot_bytes >= kChunkBytes && slot_bytes % kChunkBytes == 0,
             "mi_comm_create: slot_bytes must be a positive multiple of %lld",
             (long long)kChunkBytes);
  MI_REQUIRE(timeout_seconds > 0.0, "mi_comm_create: timeout must be positive");
  Comm* c = new Comm();
  c->rank = rank;
  c->world = world;
  c->slot_bytes = slot_bytes;
  c->chunks = slot_bytes / kChunkBytes;
  c->region_bytes = slot_off(c->chunks, world, slot_bytes, 2, 0);  // end of the last slot
  for (int r = 0; r < kMaxWorld; ++r) {
    c->peer[r] = nullptr;
    c->opened[r] = false;
  }
  c->error_word = nullptr;
  void* p = nullptr;
  // uncached (fine-grained) device memory: peers' stores and this device's loads meet in
  // memory, not in an L2 that the other side cannot see
  hipError_t e = hipExtMallocWithFlags(&p, (size_t)c->region_bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags(&p, (size_t)c->region_bytes, hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    mippo::set_error("mi_comm_create: cannot allocate %lld bytes of fine-grained device memory: %s",
                     (long long)c->region_bytes, hipGetErrorString(e));
    delete c;
    return -ENOMEM;
  }
  c->local = static_cast<char*>(p);
  c->peer[rank] = c->local;
  e = hipMemset(p, 0, (size_t)c->region_bytes);
  if (e == hipSuccess) {
    CommHeader h = {};
    h.timeout = (unsigned long long)(timeout_seconds * 1e8);  // wall_clock64: 100 MHz
    e = hipMemcpy(p, &h, sizeof(h), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess)
    e = hipIpcGetMemHandle(static_cast<hipIpcMemHandle_t*>(handle_out), p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(p);
    delete c;
    mippo::set_error("mi_comm_create: %s", hipGetErrorString(e));
    return -EIO;
  }
  *comm_out = c;
  return 0;
}

extern "C" int mi_comm_connect(void* comm, const void* all_handles) {
  MI_REQUIRE(comm && all_handles, "mi_comm_connect: null pointer");
  Comm* c = static_cast<Comm*>(comm);
  const hipIpcMemHandle_t* hs = static_cast<const hipIpcMemHandle_t*>(all_handles);
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank || c->opened[r]) continue;
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, hs[r], hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      mippo::set_error("mi_comm_connect: hipIpcOpenMemHandle(rank %d) failed: %s", r,
                       hipGetErrorString(e));
      return -EIO;
    }
    c->peer[r] = static_cast<char*>(p);
    c->opened[r] = true;
  }
  return 0;
}

extern "C" int mi_comm_destroy(void* comm) {
  if (!comm) return 0;
  Comm* c = static_cast<Comm*>(comm);
  (void)hipDeviceSynchronize();
  for (int r = 0; r < c->world; ++r)
    if (c->opened[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
  if (c->local) (void)hipFree(c->local);
  (void)hipGetLastError();
  delete c;
  return 0;
}

extern "C" int mi_comm_status(void* comm, int64_t* seq_out, int64_t* errors_out) {
  MI_REQUIRE(comm, "mi_comm_status: null comm");
  Comm* c = static_cast<Comm*>(comm);
  CommHeader h = {};
  hipError_t e = hipMemcpy(&h, c->local, sizeof(h), hipMemcpyDeviceToHost);  // synchronises
  if (e != hipSuccess) {
    mippo::set_error("mi_comm_status: %s", hipGetErrorString(e));
    return -EIO;
  }
  if (seq_out) *seq_out = (int64_t)h.seq;
  if (errors_out) *errors_out = (int64_t)h.errors;
  return 0;
}

extern "C" int mi_comm_set_error_word(void* comm, void* device_word) {
  MI_REQUIRE(comm, "mi_comm_set_error_word: null comm");
  MI_REQUIRE((reinterpret_cast<uintptr_t>(device_word) & 3) == 0,
             "mi_comm_set_error_word: the word must be 4-byte aligned");
  static_cast<Comm*>(comm)->error_word = static_cast<unsigned int*>(device_word);
  return 0;
}

extern "C" int64_t mi_comm_slot_bytes(void* comm) {
  return comm ? static_cast<Comm*>(comm)->slot_bytes : -EINVAL;
}

#define COMM_READY(c, who)                                                          \
  for (int r_ = 0; r_ < (c)->world; ++r_)                                            \
    MI_REQUIRE((c)->peer[r_], who ": rank %d is not connected (mi_comm_connect)", r_)

extern "C" int mi_allreduce_oneshot_f32(void* comm, float* buf, int64_t n, float scale,
                                        mi_stream_t stream) {
  MI_REQUIRE(comm && (buf || n == 0) && n >= 0, "mi_allreduce_oneshot_f32: bad arguments");
  Comm* c = static_cast<Comm*>(comm);
  COMM_READY(c, "mi_allreduce_oneshot_f32");
  MI_REQUIRE(n * 4 <= c->slot_bytes, "mi_allreduce_oneshot_f32: %lld bytes exceed the slot (%lld)",
             (long long)(n * 4), (long long)c->slot_bytes);
  if (n == 0) return 0;
  hipLaunchKernelGGL((allreduce_kernel<float>), dim3(grid_for(mippo::ceil_div(n * 4, kChunkBytes))),
                     dim3(kThreads), 0, mippo::as_stream(stream), dev_view(c), buf, n, scale);
  return mippo::check_launch("mi_allreduce_oneshot_f32");
}

extern "C" int mi_allreduce_oneshot_f64(void* comm, double* buf, int64_t n, double scale,
                                        mi_stream_t stream) {
  MI_REQUIRE(comm && (buf || n == 0) && n >= 0, "mi_allreduce_oneshot_f64: bad arguments");
  Comm* c = static_cast<Comm*>(comm);
  COMM_READY(c, "mi_allreduce_oneshot_f64");
  MI_REQUIRE(n * 8 <= c->slot_bytes, "mi_allreduce_oneshot_f64: %lld bytes exceed the slot (%lld)",
             (long long)(n * 8), (long long)c->slot_bytes);
  if (n == 0) return 0;
  hipLaunchKernelGGL((allreduce_kernel<double>), dim3(grid_for(mippo::ceil_div(n * 8, kChunkBytes))),
                     dim3(kThreads), 0, mippo::as_stream(stream), dev_view(c), buf, n, scale);
  return mippo::check_launch("mi_allreduce_oneshot_f64");
}

extern "C" int mi_allgather_oneshot(void* comm, const void* src, int64_t nbytes, void* dst,
                                    mi_stream_t stream) {
  MI_REQUIRE(comm && nbytes >= 0 && ((src && dst) || nbytes == 0),
             "mi_allgather_oneshot: bad arguments");
  Comm* c = static_cast<Comm*>(comm);
  COMM_READY(c, "mi_allgather_oneshot");
  MI_REQUIRE(nbytes <= c->slot_bytes, "mi_allgather_oneshot: %lld bytes exceed the slot (%lld)",
             (long long)nbytes, (long long)c->slot_bytes);
  if (nbytes == 0) return 0;
  hipLaunchKernelGGL(allgather_kernel, dim3(grid_for(mippo::ceil_div(nbytes, kChunkBytes))),
                     dim3(kThreads), 0, mippo::as_stream(stream), dev_view(c),
                     static_cast<const char*>(src), nbytes, static_cast<char*>(dst));
  return mippo::check_launch("mi_allgather_oneshot");
}

extern "C" int mi_adam_step_allreduce_f32(
    void* comm, float* params, float* grads, float* m, float* v, int64_t n, float lr, float b1,
    float b2, float eps, float weight_decay, int64_t* step, void* begin_next_ticket,
    int64_t n_shadows, const int64_t* shadow_begin, const int64_t* shadow_K,
    const int64_t* shadow_N, void* const* w_bf, void* const* wt_bf, void* const* frag_fwd,
    void* const* frag_bwd, int64_t n_slab_leaves, const void* const* slab_ptr,
    const int64_t* n_slabs, const int64_t* slab_K, const int64_t* slab_N,
    const int64_t* gw_offset, const int64_t* gb_offset, const int64_t* gb_first,
    mi_stream_t stream) {
  MI_REQUIRE(comm, "mi_adam_step_allreduce_f32: null comm");
  Comm* c = static_cast<Comm*>(comm);
  COMM_READY(c, "mi_adam_step_allreduce_f32");
  MI_REQUIRE(n >= 1 && n * 4 <= c->slot_bytes,
             "mi_adam_step_allreduce_f32: %lld bytes exceed the slot (%lld)", (long long)(n * 4),
             (long long)c->slot_bytes);
  mippo_optim::AdamArgs a;
  int rc = mippo_optim::fill_adam_args(a, "mi_adam_step_allreduce_f32", params, grads, m, v, n, lr,
                                       b1, b2, eps, weight_decay, step, nullptr, 0.0f,
                                       begin_next_ticket, n_shadows, shadow_begin, shadow_K,
                                       shadow_N, w_bf, wt_bf, frag_fwd, frag_bwd);
  if (rc) return rc;
  MI_REQUIRE(n_slab_leaves >= 0 && n_slab_leaves <= mippo_optim::kMaxSlabLeaves,
             "mi_adam_step_allreduce_f32: 0 <= n_slab_leaves <= %d", mippo_optim::kMaxSlabLeaves);
  a.slabs.n = (int)n_slab_leaves;
  for (int64_t l = 0; l < n_slab_leaves; ++l) {
    MI_REQUIRE(slab_ptr && n_slabs && slab_K && slab_N && gw_offset && gb_offset && slab_ptr[l] &&
                   n_slabs[l] >= 1 && slab_K[l] >= 1 && slab_N[l] >= 1 && gw_offset[l] >= 0 &&
                   gw_offset[l] + slab_K[l] * slab_N[l] <= n &&
                   (!gb_first || (gb_first[l] >= 0 && gb_first[l] < slab_N[l])) &&
                   (gb_offset[l] < 0 ||
                    gb_offset[l] + slab_N[l] - (gb_first ? gb_first[l] : 0) <= n),
               "mi_adam_step_allreduce_f32: bad slab leaf %lld", (long long)l);
    mippo_optim::SlabLeaf& lf = a.slabs.leaf[l];
    lf.slabs = static_cast<const float*>(slab_ptr[l]);
    lf.S = (int)n_slabs[l];
    lf.KN = (int)(slab_K[l] * slab_N[l]);
    lf.N = (int)slab_N[l];
    lf.gw_off = gw_offset[l];
    lf.gb_off = gb_offset[l];
    lf.b_lo = gb_first ? (int)gb_first[l] : 0;
  }
  hipLaunchKernelGGL(adam_allreduce_kernel, dim3(grid_for(mippo::ceil_div(n * 4, kChunkBytes))),
                     dim3(kThreads), 0, mippo::as_stream(stream), dev_view(c), a);
  return mippo::check_launch("mi_adam_step_allreduce_f32");
}
