/*
 * mippo.h — C ABI of libmippo.so: the MI355X (gfx950) kernels behind the
 * nnx-ppo `ppo_step` hot path.
 *
 * The reference (emiwar/nnx-ppo) is pure Python on JAX and has NO FFI / plugin
 * registry; every entry point below replaces an *implicit* XLA op sequence of
 * the reference, cited per function as `file:line` relative to the reference
 * root.  The Python host side (`nnx_ppo_amd/`) binds these with ctypes; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - `extern "C"`, plain device pointers + explicit int64 sizes, scalars by
 *     value, the HIP stream as an opaque `mi_stream_t` (a `hipStream_t`).
 *   - The caller owns every buffer (inputs, outputs, workspaces).  Nothing here
 *     allocates, frees, synchronises or throws; kernels are only enqueued on
 *     `stream`, so every call is HIP-graph-capturable.
 *   - Return 0 on success or a negative errno value (-EINVAL bad shape or null
 *     pointer, -EIO launch failure).  `mi_last_error()` returns a description
 *     of the last failure on the calling thread.
 *   - All tensors are dense row-major; time-major rollouts are `[T, N, *feat]`
 *     as in the reference (`nnx_ppo/algorithms/rollout.py:61-66`).
 *   - Flags (`done`, `truncated`) are uint8 (0/1) — torch.bool storage.
 */
#ifndef MIPPO_H
#define MIPPO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mi_stream_t; /* hipStream_t */

/* Activation codes shared by the dense / MLP entry points
 * (`nnx_ppo/networks/factories.py:107-110`: relu | swish | tanh; 0 = none). */
enum { MI_ACT_NONE = 0, MI_ACT_RELU = 1, MI_ACT_TANH = 2, MI_ACT_SWISH = 3 };

/* ---- library ---------------------------------------------------------- */

/* ABI version of this header (bumped when a signature changes). */
int mi_abi_version(void);

/* Description of the last error on this thread ("" if none). */
const char* mi_last_error(void);

/* ---- a13: GAE reverse scan -------------------------------------------- */

/* Generalised advantage estimation, `nnx_ppo/algorithms/ppo.py:351-394`:
 *   nv    = done[t] ? 0 : V[t+1]            (V[T] = last_value)
 *   delta = r[t] + gamma*nv - V[t];  delta = truncated[t] ? 0 : delta
 *   A[t]  = delta + (1-done[t]) * gamma * lambda * A[t+1],   A[T] = 0
 * rewards, values, done, truncated, advantages: [T, N]; last_value: [N].
 * `targets` (nullable) receives V + A (`ppo.py:456-458`).
 * One thread per env, time loop in registers, rows coalesced. */
int mi_gae_f32(const float* rewards, const float* values,
               const float* last_value, const uint8_t* done,
               const uint8_t* truncated, float* advantages, float* targets,
               int64_t T, int64_t N, float gamma, float lambda,
               mi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MIPPO_H */
