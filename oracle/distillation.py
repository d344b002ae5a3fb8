"""CPU restatement of the reference's distillation loss
(`nnx_ppo/algorithms/distillation.py:160-230`) — TEST INFRASTRUCTURE ONLY (imported by
`tests/`; the product never imports `oracle/`).

The student is scanned over time step by step, as the reference does, with the
TEACHER's rollout_extras as the replay channel; gradients come from autograd.
Parity status: pinned by the reference's own identities for this path
(`distillation_test.py:46-199`: step counting, finite metrics, teacher parameters
untouched) — the reference holds no numeric golden vector for the loss value, so the
loss/gradient comparison is against this restatement alone ("parity unpinned" for the
numbers themselves)."""
from __future__ import annotations

import torch

from .networks import Module, _map
from .ppo import _leaves, _samplers, tree_where


def distillation_loss(student: Module, student_state, obs, done, teacher_rollout_extras):
    """distillation.py:188-222.  Returns (total, dict(distillation_nll, regularization))."""
    T = done.shape[0]
    for s in _samplers(student):
        s.begin_replay(T)
    state = student_state
    lls, regs = [], []
    for t in range(T):  # step_network, distillation.py:194-197
        obs_t = _map(lambda x: x[t], obs)
        ex_t = _map(lambda x: x[t], teacher_rollout_extras)
        out = student(state, obs_t, ex_t)
        state = tree_where(done[t], student.reset_state(out.next_state), out.next_state)
        lls.append(out.output.loglikelihoods)
        regs.append(out.regularization_loss)
    ll = _map(lambda *v: torch.stack(v, 0), lls[0], *lls[1:])
    n_env = done.shape[1]
    reg = torch.stack([r.expand(n_env) if r.dim() == 0 else r for r in regs], 0)
    nll = sum(-x.mean() for x in _leaves(ll))  # per head, summed (distillation.py:213-216)
    regl = reg.mean()
    return nll + regl, dict(distillation_nll=nll.detach(), regularization=regl.detach())
