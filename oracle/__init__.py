"""CPU oracle for the nnx-ppo `ppo_step` hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (numpy / torch-CPU,
fp64 by default) of the reference algorithm in emiwar/nnx-ppo, written from the
reference sources cited per function (`file:line` relative to the reference
root).  It exists to check the HIP path; it is never part of the product:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
    leg may import it;
  * nothing under `nnx_ppo_amd/` imports it, and the product path has no CPU
    fallback (it raises when libmippo.so is missing).

Pinning status:
  * GAE (`oracle.gae`) is PINNED by the reference's own known-answer test
    (`nnx_ppo/algorithms/ppo_test.py:229-264`, seed 23, tol 1e-6) via
    `tests/golden/gae_seed23.npz`.
  * Normaliser / sampler replay / rollout lock-step / step counters are pinned
    by the data-independent identities the reference tests assert
    (`normalizer_test.py:33-65`, `adapter_test.py:61-75`,
    `rollout_test.py:121-222`, `ppo_test.py:38-62,340-349`), re-run here on
    numpy-generated data because the reference's data come from JAX's RNG.
  * The key scheme (`oracle.keys`: reset keys, minibatch permutations, observation
    noise) is a numpy uint64 restatement independent of the product, PINNED by the
    published SplitMix64 outputs (`tests/test_oracle_keys.py`); the product's CPU and
    HIP statements are checked against it bit for bit.
  * Adam / AdamW / global-norm clip (optax), Linear init (flax), LSTM / GRU cell
    arithmetic (flax) and all RNG streams (jax.random) live in third-party
    dependencies that are absent from the reference tree and from this image:
    their published formulas are restated and are PARITY UNPINNED.
  * The reference itself cannot be imported here (jax/flax/optax are not
    installed: ordinary ModuleNotFoundError, no permission denial).
"""
