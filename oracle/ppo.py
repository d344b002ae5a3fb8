"""PPO iteration of the oracle (torch-CPU fp64 + autograd).
TEST INFRASTRUCTURE — see oracle/__init__.py.

Restated from the reference:
  tree_where / single_transition / unroll_env   nnx_ppo/algorithms/rollout.py:11-73,270-279
  gae                                            nnx_ppo/algorithms/ppo.py:351-394
  ppo_loss                                       nnx_ppo/algorithms/ppo.py:397-531
  ppo_step                                       nnx_ppo/algorithms/ppo.py:254-348
  new_training_state                             nnx_ppo/algorithms/ppo.py:534-572
  Adam / AdamW / clip_by_global_norm             optax (third-party, unpinned): published formulas

Envs are whatever batched playground-API env the test passes in (the product's
synthetic envs are device-agnostic torch code and run here on CPU).  Key
derivation (`split` / `fold_in` / `permutation`) is passed in as `keys` so both
sides consume the same integer streams; minibatch indices can also be injected.
"""
from __future__ import annotations

from typing import Any, Optional

import torch

from .networks import DTYPE, Module, _leaves, _map


# --------------------------------------------------------------------- pytrees
def _tmap(fn, tree, *rest):
    """tree map that also descends into env State-like dataclasses."""
    import dataclasses

    if tree is None:
        return None
    if dataclasses.is_dataclass(tree) and not isinstance(tree, type):
        kw = {f.name: _tmap(fn, getattr(tree, f.name), *[getattr(r, f.name) for r in rest])
              for f in dataclasses.fields(tree)}
        return dataclasses.replace(tree, **kw)
    if isinstance(tree, dict):
        return {k: _tmap(fn, tree[k], *[r[k] for r in rest]) for k in tree}
    if isinstance(tree, (list, tuple)):
        return type(tree)(_tmap(fn, v, *[r[i] for r in rest]) for i, v in enumerate(tree))
    return fn(tree, *rest)


def tree_where(cond, on_true, on_false):
    """rollout.py:270-279."""

    def leaf(x, y):
        if not isinstance(x, torch.Tensor) or x.dim() == 0 or x.shape[0] != cond.shape[0]:
            return x
        c = cond.reshape(cond.shape + (1,) * (x.dim() - cond.dim()))
        return torch.where(c, x, y.to(x.dtype))

    return _tmap(leaf, on_true, on_false)


def _as_bool(x):
    return x if x.dtype == torch.bool else x != 0


# --------------------------------------------------------------------- rollout
class Transition:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def single_transition(env, networks: Module, carry, reset_keys):
    """rollout.py:11-45."""
    net_state, env_state = carry
    with torch.no_grad():
        out = networks(net_state, env_state.obs)
    nxt = env.step(env_state, _tmap(lambda a: a.to(torch.float32), out.output.actions))
    done = _as_bool(nxt.done)
    trunc = nxt.info.get("truncated", None)
    trunc = torch.zeros_like(done) if trunc is None else _as_bool(trunc)
    tr = dict(obs=env_state.obs, actions=out.output.actions, loglikelihoods=out.output.loglikelihoods,
              value_estimates=out.output.value_estimates, rewards=nxt.reward, done=done,
              truncated=trunc, next_obs=nxt.obs, rollout_extras=out.rollout_extras)
    reset_states = env.reset(reset_keys)
    nxt = tree_where(done, reset_states, nxt)
    reset_net = networks.reset_state(out.next_state)
    next_net = tree_where(done, reset_net, out.next_state)
    return (next_net, nxt), tr


def unroll_env(env, env_state, networks: Module, network_state, unroll_length: int, keys_tb):
    """rollout.py:48-73.  `keys_tb`: int64 [T, B] reset keys."""
    carry = (network_state, env_state)
    steps = []
    for t in range(unroll_length):
        carry, tr = single_transition(env, networks, carry, keys_tb[t])
        steps.append(tr)
    stacked = {k: _tmap(lambda *xs: torch.stack(xs, 0), steps[0][k], *[s[k] for s in steps[1:]])
               for k in steps[0]}
    # rollout.py:67-72: value_estimates leaves must match rewards leaves
    for v, r in zip(_leaves(stacked["value_estimates"]), _leaves(stacked["rewards"])):
        assert v.shape == r.shape
    # the carry that leaves the rollout is DATA (the reference stores arrays in
    # TrainingState; `nnx.grad` differentiates the loss w.r.t. the parameters only,
    # ppo.py:298-312): a learnable initial state reaches the loss through the resets
    # inside the replay, not through the stored carry
    final_net = _tmap(lambda x: x.detach() if isinstance(x, torch.Tensor) else x, carry[0])
    return final_net, carry[1], Transition(**stacked)


# ------------------------------------------------------------------------ gae
def gae(rewards, values_excl_last, last_value, done, truncation, lambda_, gamma):
    """ppo.py:351-394 in torch (dtype of `rewards`)."""
    values = torch.cat([values_excl_last, last_value.reshape(1, -1)], dim=0)
    assert values.shape == (rewards.shape[0] + 1, rewards.shape[1])
    nxt = torch.zeros(rewards.shape[1], dtype=rewards.dtype)
    out = []
    for t in range(rewards.shape[0] - 1, -1, -1):
        next_value = torch.where(done[t], torch.zeros_like(values[t + 1]), values[t + 1])
        advantage = rewards[t] + gamma * next_value - values[t]
        advantage = torch.where(truncation[t], torch.zeros_like(advantage), advantage)
        nxt = advantage + (1 - done[t].to(rewards.dtype)) * gamma * lambda_ * nxt
        out.append(nxt)
    return torch.stack(out[::-1], 0).detach()


# ----------------------------------------------------------------------- loss
def _samplers(networks: Module):
    return [m for m in networks.modules() if type(m).__name__ == "NormalTanhSampler"]


def _leaves(t):
    if isinstance(t, dict):
        return [x for k in t for x in _leaves(t[k])]
    if isinstance(t, (list, tuple)):
        return [x for v in t for x in _leaves(v)]
    return [] if t is None else [t]


def ppo_loss(networks: Module, network_state, mb: Transition, clip_range, normalize_advantages,
             discounting_factor, gae_lambda, critic_loss_weight, combine_advantages=False):
    """ppo.py:397-531 (rewards / value heads / log-likelihoods: tensors or PyTrees).
    Returns (total, dict(actor, critic, regularization, clipping_fraction, advantages))."""
    T = mb.done.shape[0]
    for s in _samplers(networks):
        s.begin_replay(T)
    state = network_state
    vals, lls, regs = [], [], []
    for t in range(T):  # ppo.py:415-431
        obs_t = _map(lambda x: x[t], mb.obs)
        ex_t = _map(lambda x: x[t], mb.rollout_extras)
        out = networks(state, obs_t, ex_t)
        reset = networks.reset_state(out.next_state)
        state = tree_where(mb.done[t], reset, out.next_state)  # ppo.py:411-413
        vals.append(out.output.value_estimates)
        lls.append(out.output.loglikelihoods)
        regs.append(out.regularization_loss)
    stack = lambda xs: _map(lambda *v: torch.stack(v, 0), xs[0], *xs[1:])
    values = stack(vals)
    ll_new = stack(lls)
    n_env = mb.done.shape[1]
    reg = torch.stack([r.expand(n_env) if r.dim() == 0 else r for r in regs], 0)

    last_obs = _map(lambda x: x[-1], mb.next_obs)
    last_values = networks.forward_value(state, last_obs)  # ppo.py:433-437, value branch only

    dt = reg.dtype
    det = lambda t: _map(lambda x: x.detach(), t)
    # one GAE per reward key; done / truncated are shared (ppo.py:440-456)
    adv = _map(lambda r, v, lv: gae(r.to(dt), v, lv, mb.done, mb.truncated, gae_lambda,
                                    discounting_factor),
               mb.rewards, det(values), det(last_values))
    target = _map(lambda v, a: (v + a).detach(), values, adv)  # ppo.py:456-458
    a = adv
    if combine_advantages:  # ppo.py:460-474
        summed = sum(_leaves(adv))
        a = summed if isinstance(ll_new, torch.Tensor) else _map(lambda _: summed, ll_new)
    if normalize_advantages:  # ppo.py:477-480 (population std, per leaf)
        a = _map(lambda x: (x - x.mean()) / (x.std(unbiased=False) + 1e-8), a)

    def clipped(ln, lo, ad):
        ratio = torch.exp(ln - lo.to(dt))
        c1 = ratio * ad
        c2 = torch.clamp(ratio, 1 - clip_range, 1 + clip_range) * ad
        return -torch.mean(torch.minimum(c1, c2))

    actor_tree = _map(clipped, ll_new, mb.loglikelihoods, a)        # ppo.py:494-499
    critic_tree = _map(lambda v, t: 0.5 * torch.mean((v - t) ** 2), values, target)
    actor = sum(_leaves(actor_tree))                                # ppo.py:505-507
    critic = sum(_leaves(critic_tree))
    regl = reg.mean()
    total = actor + critic_loss_weight * critic + regl
    clip_leaves = _leaves(_map(
        lambda ln, lo: (torch.abs(torch.exp(ln.detach() - lo.to(dt)) - 1.0) > clip_range)
        .to(dt).mean(), ll_new, mb.loglikelihoods))
    clipfrac = sum(clip_leaves) / len(clip_leaves)
    return total, dict(actor=actor.detach(), critic=critic.detach(),
                       regularization=regl.detach(), clipping_fraction=clipfrac,
                       advantages=adv, values=det(values), ll_new=det(ll_new),
                       # the trees the reference logs under losses/actor, losses/critic
                       # (ppo.py:509-513): one entry per policy term / reward key
                       actor_tree=det(actor_tree), critic_tree=det(critic_tree))


# ------------------------------------------------------------------ optimiser
class Adam:
    """optax.chain([clip_by_global_norm(c)]?, adam | adamw) — ppo.py:555-569.
        clip:  g <- g if ||g|| < c else g / ||g|| * c
        adam:  m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; t += 1
               u = (m / (1 - b1^t)) / (sqrt(v / (1 - b2^t)) + eps)
        adamw: u += weight_decay * p ;  p <- p - lr * u"""

    def __init__(self, params, lr=1e-4, gradient_clipping=None, weight_decay=None,
                 b1=0.9, b2=0.999, eps=1e-8):
        self.params = list(params)
        self.lr, self.clip = lr, gradient_clipping
        if weight_decay is None or weight_decay is False:
            self.wd = 0.0
        elif weight_decay is True:
            self.wd = 1e-4
        else:
            self.wd = float(weight_decay)
        self.b1, self.b2, self.eps = b1, b2, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def update(self, grads):
        if self.clip is not None:
            gn = torch.sqrt(sum((g ** 2).sum() for g in grads))
            if not bool(gn < self.clip):
                grads = [g / gn * self.clip for g in grads]
        self.t += 1
        with torch.no_grad():
            for i, (p, g) in enumerate(zip(self.params, grads)):
                self.m[i] = self.b1 * self.m[i] + (1 - self.b1) * g
                self.v[i] = self.b2 * self.v[i] + (1 - self.b2) * g * g
                mh = self.m[i] / (1 - self.b1 ** self.t)
                vh = self.v[i] / (1 - self.b2 ** self.t)
                u = mh / (torch.sqrt(vh) + self.eps)
                if self.wd:
                    u = u + self.wd * p
                p -= self.lr * u


# ------------------------------------------------------------------- ppo_step
class TrainingState:
    def __init__(self, networks, network_states, env_states, optimizer, rng_key, steps_taken):
        self.networks, self.network_states, self.env_states = networks, network_states, env_states
        self.optimizer, self.rng_key, self.steps_taken = optimizer, rng_key, steps_taken


def new_training_state(env, networks: Module, n_envs, seed, keys, learning_rate=1e-4,
                       gradient_clipping=None, weight_decay=None):
    """ppo.py:534-572.  `keys`: module providing key/split (integer key plumbing)."""
    key = keys.key(seed)
    ks = keys.split(key)
    key, training_key = ks[0], ks[1]
    env_states = env.reset(keys.split(key, n_envs))
    network_states = _tmap(lambda x: x.detach() if isinstance(x, torch.Tensor) else x,
                           networks.initialize_state(n_envs))
    opt = Adam(networks.parameters(), learning_rate, gradient_clipping, weight_decay)
    return TrainingState(networks, network_states, env_states, opt, training_key, 0)


def ppo_step(env, ts: TrainingState, n_envs, rollout_length, gae_lambda, discounting_factor,
             clip_range, normalize_advantages, n_epochs, n_minibatches, keys,
             critic_loss_weight=1.0, minibatch_inds: Optional[torch.Tensor] = None,
             combine_advantages: bool = False):
    """ppo.py:254-348.  Returns (state, dict(loss rows [n_grad_steps] per term, rollout))."""
    networks = ts.networks
    ks = keys.split(ts.rng_key)
    reset_key, new_key = ks[0], ks[1]
    next_net, next_env, ro = unroll_env(env, ts.env_states, networks, ts.network_states,
                                        rollout_length,
                                        keys.split(reset_key, (rollout_length, n_envs)))
    mb_size = n_envs // n_minibatches
    if minibatch_inds is None:
        rows = []
        for e in range(n_epochs):  # ppo.py:284-294
            perm = keys.permutation(keys.fold_in(new_key, e), n_envs)
            rows.append(perm[: n_minibatches * mb_size].reshape(n_minibatches, mb_size))
        minibatch_inds = torch.cat(rows, 0)
    rows_out = {k: [] for k in ("actor", "critic", "regularization", "clipping_fraction")}
    diag: dict = {"grad_norm": [], "advantages": [], "critic_R^2": []}
    grads_first = None
    trees: dict = {}
    for i in range(n_epochs * n_minibatches):
        inds = minibatch_inds[i]
        mb = Transition(**{k: _tmap(lambda x: x[:, inds], v) for k, v in ro.__dict__.items()})
        st = _tmap(lambda x: x[inds], ts.network_states)  # PRE-rollout carry, ppo.py:298-300
        params = networks.parameters()
        total, lm = ppo_loss(networks, st, mb, clip_range, normalize_advantages,
                             discounting_factor, gae_lambda, critic_loss_weight,
                             combine_advantages)
        grads = torch.autograd.grad(total, params, allow_unused=True)
        grads = [torch.zeros_like(p) if g is None else g for p, g in zip(params, grads)]
        if grads_first is None:
            grads_first = [g.clone() for g in grads]
        if isinstance(lm["advantages"], torch.Tensor):  # single reward key
            from .metrics import step_diagnostics

            for k, v in step_diagnostics(grads, lm, normalize_advantages).items():
                diag[k].append(v)
        ts.optimizer.update(grads)
        for k in rows_out:
            rows_out[k].append(lm[k])
        for k in ("actor_tree", "critic_tree"):
            trees.setdefault(k, []).append(lm[k])
    networks.update_statistics(ro.rollout_extras)  # ppo.py:336 — after all updates
    new = TrainingState(networks, next_net, next_env, ts.optimizer, new_key,
                        ts.steps_taken + rollout_length * n_envs)
    info = {k: torch.stack(v) for k, v in rows_out.items()}
    info["rollout"] = ro
    for k, v in trees.items():  # stacked over the gradient steps, per key
        info[k] = _map(lambda *xs: torch.stack(xs), v[0], *v[1:])
    for k, v in diag.items():  # rows behind GRAD_NORM / CRITIC_EXTRA (ppo.py:313-315,520-528)
        if v:
            info[k] = torch.stack(v)
    info["grads_first"] = grads_first
    info["minibatch_inds"] = minibatch_inds
    return new, info
