"""GAE oracle (numpy).  TEST INFRASTRUCTURE — see oracle/__init__.py.

Two independent statements of the same quantity:

* `gae_known_answer` — the mask-and-assign reverse loop the reference's own test
  uses as ground truth (`nnx_ppo/algorithms/ppo_test.py:244-253`).
* `gae` — the scan the reference ships (`nnx_ppo/algorithms/ppo.py:351-394`),
  restated step by step with the reference's operation order so that the fp32
  variant is what a plain fp32 evaluation of that code produces.

`tests/test_oracle_gae.py` checks one against the other and both against the
committed fixture.
"""
from __future__ import annotations

import numpy as np


def gae_known_answer(rewards, values_incl_last, done, truncation, gamma, lambda_):
    """ppo_test.py:244-253 — values_incl_last is [T+1, N]; float64 in, float64 out."""
    T = rewards.shape[0]
    adv = np.full(rewards.shape, np.nan, dtype=np.float64)
    for t in range(T - 1, -1, -1):
        bootstrap = np.where(done[t], 0.0, values_incl_last[t + 1])
        row = rewards[t] + gamma * bootstrap - values_incl_last[t]
        row = np.where(truncation[t], 0.0, row)
        if t + 1 < T:
            row = row + gamma * lambda_ * adv[t + 1] * (1 - done[t])
        adv[t] = row
    return adv


def gae(rewards, values_excl_last, last_value, done, truncation, lambda_, gamma,
        dtype=np.float64):
    """ppo.py:351-394 (signature and argument order as the reference).

    inner_step (ppo.py:364-377), evaluated in `dtype`:
        next_value = where(done, 0, next_value)
        new_value  = reward + gamma * next_value
        advantage  = new_value - old_value
        advantage  = where(truncated, 0, advantage)
        gae_adv    = advantage + (1 - done) * gamma * lambda_ * next_advantage
    """
    dt = np.dtype(dtype).type
    r = np.asarray(rewards, dtype=dtype)
    v = np.concatenate(
        [np.asarray(values_excl_last, dtype=dtype),
         np.asarray(last_value, dtype=dtype).reshape(1, -1)], axis=0)
    assert v.shape == (r.shape[0] + 1, r.shape[1])  # ppo.py:362
    d = np.asarray(done, dtype=bool)
    tr = np.asarray(truncation, dtype=bool)
    g, lam = dt(gamma), dt(lambda_)
    T = r.shape[0]
    out = np.empty_like(r)
    nxt = np.zeros(r.shape[1], dtype=dtype)
    for t in range(T - 1, -1, -1):
        next_value = np.where(d[t], dt(0), v[t + 1])
        new_value = r[t] + g * next_value
        advantage = new_value - v[t]
        advantage = np.where(tr[t], dt(0), advantage)
        keep = (dt(1) - d[t].astype(dtype))
        nxt = advantage + ((keep * g) * lam) * nxt
        out[t] = nxt
    return out


def make_seed23_case():
    """Inputs of the reference's known-answer test (ppo_test.py:230-242):
    np.random.seed(23), T=100, N=512, gamma=0.8, lambda=0.95, P(done)=0.01,
    truncation a random subset of done."""
    T, N = 100, 512
    np.random.seed(23)
    rewards = np.random.normal(size=(T, N))
    values = np.random.normal(size=(T + 1, N))
    done = np.random.choice([True, False], size=(T, N), p=[0.01, 0.99])
    truncation = np.random.choice([True, False], size=(T, N))
    truncation = np.logical_and(done, truncation)
    return dict(rewards=rewards, values=values, done=done, truncation=truncation,
                gamma=0.8, lambda_=0.95)
