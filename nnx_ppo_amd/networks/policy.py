"""The network `make_mlp_actor_critic` builds (`nnx_ppo/networks/factories.py:88-146`:
`Sequential([Normalizer?, PPOAdapter(action=Sequential([Dense.., sampler]),
value=Sequential([Dense..]))])`) with its evaluation as ONE launch on the bf16 path.

Module tree, parameters, state / extras / metrics structure and every method are
those of the plain `Sequential` — `MLPActorCritic` only recognises its own shape and
routes `__call__` (rollout / inference) and `replay` / `replay_backward` (loss) to
`mi_policy_fwd_bf16` / `mi_policy_ws_fwd_bf16`: normaliser in the input stage, both trunks
side by side (they read the same input, adapter.py:75-117), sampler on the action trunk's
output.  At this workload's sizes a policy step is bound by launch and dependency latency,
not arithmetic: four launches on two streams become one.  Trunks in the shape class of
csrc/trunk_ws.hip take the weights-stationary kernels (a trunk's fragments requested once,
up front, and kept in registers); the tile kernels of csrc/mlp_bf16.hip take the rest —
same results bit for bit, so the choice is invisible.  Anything outside the
pattern (PyTree observations, fp32 compute, CPU tensors, injected rollout extras,
trunks the kernel cannot take) goes through the generic container path."""
from __future__ import annotations

import os
from typing import Any

import torch

from .. import config, ops, parallel
from . import dense_chain
from .adapter import PPOAdapter, _Fork, _can_fork
from .containers import Sequential
from ..tree import canonicalize, tree_leaves
from .feedforward import Dense
from .normalizer import Normalizer
from .sampling_layers import NormalTanhSampler
from .types import PPONetworkOutput, StatefulModuleOutput
from .utils import Flattener


# MIPPO_FUSED_POLICY=0 sends everything through the generic containers (A/B timing)
FUSED = os.environ.get("MIPPO_FUSED_POLICY", "1") != "0"
# MIPPO_WS_POLICY=0 keeps the loss replay on the tile kernels (A/B timing)
WS_POLICY = os.environ.get("MIPPO_WS_POLICY", "1") != "0"
# the weights-stationary backward (both trunks in one launch) measures ~1 % of an iteration
# better than policy_bwd_kernel at C2 (2.105 vs 2.125 ms); MIPPO_WS_POLICY_BWD=0 for the latter
WS_POLICY_BWD = os.environ.get("MIPPO_WS_POLICY_BWD", "1") != "0"
# MIPPO_WS_ROLLOUT=0 keeps rollout / evaluation steps (<= 8192 rows) on the tile kernel
WS_POLICY_ROLLOUT = os.environ.get("MIPPO_WS_ROLLOUT", "1") != "0"
WS_MIN_ROWS = 8192
# MIPPO_WS_MASKS=0: the weights-stationary backward reads relu' from the bf16 images (A/B)
USE_MASKS = os.environ.get("MIPPO_WS_MASKS", "1") != "0"
# MIPPO_FUSED_ROLLOUT=0 keeps the rollout stepwise (two launches per step) where the one-launch
# rollout (mi_rollout_mock_ws_bf16) would apply (A/B timing, bit-identity tests)
FUSED_ROLLOUT = os.environ.get("MIPPO_FUSED_ROLLOUT", "1") != "0"
# the GAE scan, the advantage statistics and the loss gradients inside the backward launch
# (mi_policy_ws_bwd_gae_bf16) instead of a launch of their own between forward and backward
GAE_IN_BWD = os.environ.get("MIPPO_GAE_IN_BWD", "1") != "0"


class MLPActorCritic(Sequential):
    """`[Normalizer, PPOAdapter]` (flat observations) or `[Normalizer(pytree), Flattener,
    PPOAdapter]` (PyTree observations, BASELINE config 3: per-leaf statistics, leaves
    concatenated in sorted-key order in front of the trunks)."""

    def __init__(self, layers):
        super().__init__(layers)
        adapter = self.layers[-1]
        assert isinstance(adapter, PPOAdapter) and len(self.layers) in (2, 3)
        self._norm = self.layers[0]
        self._adapter = adapter
        # a PyTree-observation network has the Flattener between normaliser and adapter
        self._flattener = self.layers[1] if len(self.layers) == 3 else None
        assert self._flattener is None or (isinstance(self._flattener, Flattener)
                                           and self._flattener.preserve_levels == 0)

    def _ws_dual(self, a_dims, a_acts, c_dims, c_acts) -> bool:
        key = (tuple(a_dims), tuple(a_acts), tuple(c_dims), tuple(c_acts))
        cache = self.__dict__.setdefault("_ws_dual_cache", {})
        if key not in cache:
            cache[key] = ops.policy_ws_dual_supported(a_dims, a_acts, c_dims, c_acts)
        return cache[key]

    def _flat_stats(self):
        """Per-leaf normaliser statistics as ONE vector each, leaves in the Flattener's
        order — what the kernel's input stage takes.  The leaves are re-homed as views of the
        flat vectors (the way the optimiser re-homes parameters), so the in-place Welford
        merge of `update_statistics` and the generic path keep working on them; anything
        that REPLACED a leaf since (a `.to(device)`, a checkpoint load) is noticed by its
        address and re-homed again."""
        n = self._norm
        flats = self.__dict__.get("_stat_flats")
        out = []
        for vi, var in enumerate((n.mean, n.M2)):
            leaves = tree_leaves(var.value)
            flat = None if flats is None else flats[vi]
            off, ok = 0, flat is not None and flat.device == leaves[0].device
            for t in leaves:
                ok = ok and t.data_ptr() == flat.data_ptr() + 4 * off and t.is_contiguous()
                off += t.numel()
            if not ok or off != flat.numel():
                from ..tree import tree_map

                flat = torch.cat([t.reshape(-1) for t in leaves])
                views, off = {}, 0
                for t in leaves:
                    views[id(t)] = flat[off:off + t.numel()].view(t.shape)
                    off += t.numel()
                var.value = tree_map(lambda t: views[id(t)], var.value)
            out.append(flat)
        self.__dict__["_stat_flats"] = out
        return out[0], out[1]

    def _flat_seq(self, x_seq):
        """`_flat_obs` for a `[T, B, w]` sequence tree -> `[T, B, K0]` (else the input)."""
        if not isinstance(x_seq, dict):
            return x_seq
        leaves = tree_leaves(canonicalize(x_seq))
        shapes = tree_leaves(self._norm.mean.value)
        if len(leaves) != len(shapes) or not leaves:
            return x_seq
        for t, m in zip(leaves, shapes):
            if not (isinstance(t, torch.Tensor) and t.dim() == 3 and t.is_cuda
                    and t.dtype == torch.float32 and t.shape[:2] == leaves[0].shape[:2]
                    and m.dim() == 1 and t.shape[2] == m.shape[0]):
                return x_seq
        return torch.cat(leaves, dim=2)

    def _flat_obs(self, obs):
        """A PyTree observation as the `[B, K0]` tensor the Flattener would hand the trunks
        (one concatenation), or None if it is not a tree of 2-D fp32 GPU leaves that matches
        the normaliser's."""
        if self._flattener is None or not isinstance(obs, dict) or self._norm is None:
            return None
        leaves = tree_leaves(canonicalize(obs))
        shapes = tree_leaves(self._norm.mean.value)
        if len(leaves) != len(shapes) or not leaves:
            return None
        B = leaves[0].shape[0] if isinstance(leaves[0], torch.Tensor) and leaves[0].dim() == 2 else -1
        for t, m in zip(leaves, shapes):
            if not (isinstance(t, torch.Tensor) and t.dim() == 2 and t.is_cuda
                    and t.dtype == torch.float32 and t.shape[0] == B and m.dim() == 1
                    and t.shape[1] == m.shape[0]):
                return None
        return torch.cat(leaves, dim=1)

    # ---- pattern ---------------------------------------------------------------------
    def _parts(self):
        a_layers = self._adapter.action.layers[:-1]
        sampler = self._adapter.action.layers[-1]
        c_layers = self._adapter.value.layers
        return a_layers, sampler, c_layers

    def _fusable(self, x, M: int) -> bool:
        if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dtype != torch.float32:
            return False
        return self._trunks_fusable(M)

    def _trunks_fusable(self, M: int) -> bool:
        """The part of `_fusable` that does not look at the input: settings, module pattern
        and trunk shapes for M rows (asked BEFORE a PyTree input is concatenated)."""
        if not FUSED or config.compute_dtype() != "bf16":
            return False
        norm = self._norm
        if norm is not None and not isinstance(norm, Normalizer):
            return False
        if norm is not None and self._flattener is None and not (
                isinstance(norm.mean.value, torch.Tensor) and norm.mean.value.dim() == 1):
            return False
        if not (isinstance(self._adapter.action, Sequential)
                and isinstance(self._adapter.value, Sequential)):
            return False
        a_layers, sampler, c_layers = self._parts()
        if type(sampler) is not NormalTanhSampler or not a_layers or not c_layers:
            return False
        if not all(type(l) is Dense for l in list(a_layers) + list(c_layers)):
            return False
        if a_layers[-1].out_features % 2 or a_layers[-1].out_features > 128:
            return False
        for layers in (a_layers, c_layers):
            width = max(max(l.in_features, l.out_features) for l in layers)
            if len(layers) > dense_chain.FUSED_MAX_LAYERS:
                return False
            if width > dense_chain.FUSED_MAX_WIDTH:
                return False
        return True

    def _launch(self, x2: torch.Tensor, extras2, train: bool, value_tail=None):
        a_layers, sampler, c_layers = self._parts()
        dense_chain.refresh(list(a_layers) + list(c_layers))  # one launch for both trunks
        chain = lambda ls: ([l._ff for l in ls], [dense_chain._bias(l) for l in ls],
                            [ls[0].in_features] + [l.out_features for l in ls],
                            [l.act_code for l in ls])
        norm = None
        if self._norm is not None:
            n = self._norm
            mean, m2 = (n.mean.value, n.M2.value) if self._flattener is None \
                else self._flat_stats()
            norm = (mean, m2, n.counter.value, n.epsilon)
        M = x2.shape[0]
        A = a_layers[-1].out_features // 2
        eps, eps2 = sampler._noise(M, A, x2.device)
        off = sampler._next_offset()
        ca, cc = chain(a_layers), chain(c_layers)
        rows = M + (0 if value_tail is None else value_tail.shape[0])
        if rows > WS_MIN_ROWS:  # training sizes: any pair in the class (one launch where the
            #                     pair is instantiated, else one per trunk)
            ws = (WS_POLICY and train and 2 * A <= 64
                  and ops.policy_ws_supported(ca[2], ca[3], cc[2], cc[3]))
        else:                   # rollout / evaluation sizes: only as one launch for both
            ws = (WS_POLICY_ROLLOUT and 2 * A <= 64
                  and self._ws_dual(ca[2], ca[3], cc[2], cc[3]))
        r = ops.policy_fwd_bf16(
            x2, norm, ca, cc, sampler._state(x2.device), off,
            deterministic=sampler.deterministic, extras=extras2, eps=eps, eps2=eps2,
            train=train, want_stats=not train, value_tail=value_tail, ws=ws, **sampler._kw())
        return r, off, eps2

    # ---- rollout / inference (adapter.py:75-117 over the whole stack) -----------------
    def __call__(self, network_state, obs: Any, rollout_extras: Any = None):
        if rollout_extras is not None or not FUSED or config.compute_dtype() != "bf16":
            return super().__call__(network_state, obs, rollout_extras)
        if self._flattener is None:
            x2 = obs if isinstance(obs, torch.Tensor) and obs.dim() == 2 else None
        else:  # PyTree observation: one concatenation, then the same launch
            lead = tree_leaves(obs)
            ok = (lead and isinstance(lead[0], torch.Tensor) and lead[0].dim() == 2
                  and self._trunks_fusable(lead[0].shape[0]))
            x2 = self._flat_obs(obs) if ok else None
        if x2 is None or not self._fusable(x2, x2.shape[0]):
            return super().__call__(network_state, obs, rollout_extras)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        r, _, _ = self._launch(x2, None, train=False)
        a_layers, _, c_layers = self._parts()
        value = r["value"]
        if value.shape[-1] == 1:
            value = value.squeeze(-1)
        a_state = network_state[-1]["action"]
        v_state = network_state[-1]["value"]
        a_metrics = {i: {} for i in range(len(a_layers))}
        a_metrics[len(a_layers)] = {"mu": r["mu"], "sigma": r["sigma"]}
        adapter_out = dict(
            next_state={"action": list(a_state), "value": list(v_state)},
            metrics={"action": a_metrics, "value": {i: {} for i in range(len(c_layers))}},
            rollout_extras={"action": [None] * len(a_layers) + [r["raw"]],
                            "value": [None] * len(c_layers)})
        # the layers in front of the adapter: normaliser (its extras are the raw observation,
        # normalizer.py:63-96) and, for PyTree observations, the Flattener (no state, no extras)
        n_pre = len(self.layers) - 1
        pre_extras = [canonicalize(obs)] + [None] * (n_pre - 1)
        return StatefulModuleOutput(
            next_state=[()] * n_pre + [adapter_out["next_state"]],
            output=PPONetworkOutput(actions=r["action"], loglikelihoods=r["log_likelihood"],
                                    value_estimates=value),
            regularization_loss=r["reg"],
            metrics=dict(enumerate([{}] * n_pre + [adapter_out["metrics"]])),
            rollout_extras=pre_extras + [adapter_out["rollout_extras"]])

    # ---- the whole rollout in one launch (rollout.py:48-73) ---------------------------
    def unroll_fused(self, env, env_state, network_state, unroll_length: int, reset_key):
        """`unroll_env` for EpisodeWrapper(MockEnv) — the env with a device-side step — as ONE
        launch (`mi_rollout_mock_ws_bf16`: a workgroup owns a tile of envs for all T steps).
        Returns what `unroll_env` returns, every leaf bit-identical to the stepwise rollout,
        or None when the fused form does not apply (any other env, PyTree observations,
        trunks outside the class, injected noise): the caller then steps."""
        if not (FUSED and FUSED_ROLLOUT) or config.compute_dtype() != "bf16":
            return None
        from .. import random as rnd
        from ..algorithms.types import State, Transition
        from ..envs.constants import constant
        from ..envs.synthetic import MockEnv
        from ..wrappers.episode_wrapper import EpisodeWrapper

        if type(env) is not EpisodeWrapper or type(env.env) is not MockEnv \
                or not isinstance(env.env.obs_size, int) or rnd._TORCH_ONLY[0] \
                or env.env.obs_law != "uniform":
            return None
        if self._flattener is not None or not isinstance(env_state, State):
            return None
        obs = env_state.obs
        if not (isinstance(obs, torch.Tensor) and obs.dim() == 2 and self._fusable(obs, obs.shape[0])):
            return None
        N, K0 = obs.shape
        data, info = env_state.data, env_state.info
        try:
            key, count, counter = data["key"], data["step_count"], info["step_counter"]
        except (KeyError, TypeError):
            return None
        if set(data) != {"key", "step_count"} or set(info) != {"step_counter", "truncated"} \
                or env_state.metrics:
            return None
        for t in (key, count, counter):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int64
                    and t.shape == (N,)):
                return None
        if not (isinstance(reset_key, torch.Tensor) and reset_key.is_cuda
                and reset_key.dtype == torch.int64 and reset_key.dim() == 0):
            return None
        a_layers, sampler, c_layers = self._parts()
        if sampler.noise_override is not None or unroll_length < 1 or K0 != env.env.obs_size:
            return None
        chain = lambda ls: ([l._ff for l in ls], [dense_chain._bias(l) for l in ls],
                            [ls[0].in_features] + [l.out_features for l in ls],
                            [l.act_code for l in ls])
        dense_chain.refresh(list(a_layers) + list(c_layers))
        ca, cc = chain(a_layers), chain(c_layers)
        cache = self.__dict__.setdefault("_rollout_fused_cache", {})
        ck = (tuple(ca[2]), tuple(ca[3]), tuple(cc[2]), tuple(cc[3]))
        if ck not in cache:
            cache[ck] = ops.rollout_mock_ws_supported(ca[2], ca[3], cc[2], cc[3])
        if not cache[ck]:
            return None
        norm = None
        if self._norm is not None:
            n = self._norm
            norm = (n.mean.value, n.M2.value, n.counter.value, n.epsilon)
        T = int(unroll_length)
        off = sampler._pending          # step t takes offset off + t: T calls of _next_offset()
        sampler._pending += T
        c = lambda t: t if t.is_contiguous() else t.contiguous()
        r = ops.rollout_mock_ws(
            c(key), c(count), c(counter), c(obs), reset_key, env.env.max_steps, env.max_len, T,
            norm, ca, cc, sampler._state(obs.device), off,
            deterministic=sampler.deterministic, **sampler._kw())
        value = r["value"]
        if value.shape[-1] == 1:
            value = value.squeeze(-1)
        La, Lc = len(a_layers), len(c_layers)
        a_metrics = {i: {} for i in range(La)}
        a_metrics[La] = {"mu": r["mu"], "sigma": r["sigma"]}
        n_pre = len(self.layers) - 1
        net_metrics = dict(enumerate(
            [{}] * n_pre + [{"action": a_metrics, "value": {i: {} for i in range(Lc)}}]))
        extras = [r["obs"]] + [None] * (n_pre - 1) + [
            {"action": [None] * La + [r["raw"]], "value": [None] * Lc}]
        rollout = Transition(
            obs=r["obs"],
            network_output=PPONetworkOutput(actions=r["action"],
                                            loglikelihoods=r["log_likelihood"],
                                            value_estimates=value),
            rewards=r["reward"], done=r["done"], truncated=r["truncated"],
            next_obs=r["next_obs"], metrics={"env": {}, "net": net_metrics},
            rollout_extras=extras)
        dev = obs.device
        final_env = State(
            data={"key": r["key_out"], "step_count": r["count_out"]}, obs=r["obs_out"],
            reward=r["reward_out"],
            # after the reset select both flags are down: a finished episode was replaced by
            # its reset state, an unfinished one has neither (rollout.py:41-44)
            done=constant((N,), torch.float32, 0.0, dev), metrics={},
            info={"step_counter": r["counter_out"],
                  "truncated": constant((N,), torch.bool, 0, dev)})
        # the carry: stateless layers only (the fusable pattern), passed through as the
        # stepwise call passes it
        final_net = [()] * n_pre + [{"action": list(network_state[-1]["action"]),
                                    "value": list(network_state[-1]["value"])}]
        return final_net, final_env, rollout

    # ---- loss replay (ppo.py:411-431) ------------------------------------------------
    def replay_with_bootstrap(self, state0, x_seq, done_seq, extras_seq, last_obs):
        """`replay` plus the value estimate of `last_obs` [B, K0] (the bootstrap of
        ppo.py:433-437): the extra rows ride along in the value trunk of the same launch.
        Returns (ctx, out, reg, final_state, last_values) or None when the fused path
        does not apply."""
        if self._flattener is not None:
            lead = tree_leaves(x_seq)
            if not (lead and isinstance(lead[0], torch.Tensor) and lead[0].dim() == 3
                    and self._trunks_fusable(lead[0].shape[0] * lead[0].shape[1])):
                return None
            last_obs = self._flat_obs(last_obs)
        if not (isinstance(last_obs, torch.Tensor) and last_obs.dim() == 2):
            return None
        return self.replay(state0, x_seq, done_seq, extras_seq, need_input_grad=False,
                           _bootstrap=last_obs)

    def replay(self, state0, x_seq, done_seq, extras_seq, need_input_grad=True,
               _bootstrap=None):
        x_tree = x_seq
        if self._flattener is not None and not need_input_grad and extras_seq is not None:
            lead = tree_leaves(x_seq)
            if (lead and isinstance(lead[0], torch.Tensor) and lead[0].dim() == 3
                    and self._trunks_fusable(lead[0].shape[0] * lead[0].shape[1])):
                x_seq = self._flat_seq(x_seq)  # PyTree observation: one concatenation
        fus = (not need_input_grad and extras_seq is not None
               and isinstance(x_seq, torch.Tensor) and x_seq.dim() == 3
               and self._fusable(x_seq, x_seq.shape[0] * x_seq.shape[1]))
        if not fus:
            if _bootstrap is not None:
                return None
            ctx, out, reg, fs = super().replay(state0, x_tree, done_seq, extras_seq,
                                               need_input_grad)
            return ("generic", ctx), out, reg, fs
        T, B, K0 = x_seq.shape
        M = T * B
        a_layers, sampler, c_layers = self._parts()
        A = a_layers[-1].out_features // 2
        raw = extras_seq[-1]["action"][len(a_layers)]
        ex2 = raw.reshape(M, A)
        if not ex2.is_contiguous():
            ex2 = ex2.contiguous()
        x2 = x_seq.reshape(M, K0)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        tail = None
        if _bootstrap is not None:
            tail = _bootstrap if _bootstrap.is_contiguous() else _bootstrap.contiguous()
        r, off, eps2 = self._launch(x2, ex2, train=True, value_tail=tail)
        value = r["value"].view(T, B, -1)
        squeezed = value.shape[-1] == 1
        if squeezed:
            value = value.squeeze(-1)
        out = PPONetworkOutput(actions=None, loglikelihoods=r["log_likelihood"].view(T, B),
                               value_estimates=value)
        s_ctx = (r["mean_and_std"], ex2, off, eps2, (T, B, 2 * A))
        shadows = lambda ls, saved: [(xb, aux, dense_chain._shadows(l)[0])
                                     for (xb, aux), l in zip(saved, ls)]
        masks = None if r.get("actor_masks") is None else (r["actor_masks"], r["critic_masks"])
        ctx = ("fused", s_ctx, (shadows(a_layers, r["actor_saved"]), M, False),
               (shadows(c_layers, r["critic_saved"]), M, False), squeezed, (T, B), masks)
        final_state = [()] * (len(self.layers) - 1) + [{"action": list(state0[-1]["action"]),
                                                       "value": list(state0[-1]["value"])}]
        if _bootstrap is not None:
            lv = r["value_tail_out"]
            return ctx, out, r["reg"].view(T, B), final_state, (lv.squeeze(-1) if squeezed
                                                               else lv)
        return ctx, out, r["reg"].view(T, B), final_state

    def replay_backward(self, ctx, g_out, g_reg):
        if ctx[0] == "generic":
            return super().replay_backward(ctx[1], g_out, g_reg)
        _, s_ctx, a_ctx, v_ctx, squeezed, (T, B), masks = ctx
        a_layers, sampler, c_layers = self._parts()
        M = T * B
        g_v = g_out.value_estimates.reshape(M, -1)
        if not g_v.is_contiguous():
            g_v = g_v.contiguous()
        g_ll = g_out.loglikelihoods
        g_ll = None if g_ll is None else g_ll.reshape(M)
        linear_heads = (a_layers[-1].act_code == ops.ACT_NONE
                        and c_layers[-1].act_code == ops.ACT_NONE
                        and len(a_layers) >= 2 and len(c_layers) >= 2)
        if not linear_heads:
            return self._backward_per_port(s_ctx, a_ctx, v_ctx, g_out, g_reg, g_v, M)
        ms2, ex2, off, eps2, _ = s_ctx
        dense_chain.refresh(list(a_layers) + list(c_layers))
        desc = lambda ls, c: ([l._fb for l in ls],
                              [ls[0].in_features] + [l.out_features for l in ls],
                              [l.act_code for l in ls], [sv[1] for sv in c[0]])
        da, dc = desc(a_layers, a_ctx), desc(c_layers, v_ctx)
        ws = (WS_POLICY_BWD and M > WS_MIN_ROWS and ms2.shape[1] <= 64
              and ops.policy_ws_supported(da[1], da[2], dc[1], dc[2]))
        a_dz, c_dz = ops.policy_bwd_bf16(
            ms2, ex2, sampler._state(ms2.device), off, g_ll, g_reg, g_v, da, dc, eps2=eps2,
            ws=ws, masks=(masks if ws and USE_MASKS else None), **sampler._kw())
        # dW / db of every layer of both trunks: one grouped launch per tile class
        problems = []
        for ls, c, dz in ((c_layers, v_ctx, c_dz), (a_layers, a_ctx, a_dz)):
            for i in range(len(ls) - 1, -1, -1):
                l = ls[i]
                problems.append((c[0][i][0], dz[i], l.kernel.grad,
                                 l.bias.grad if l.bias is not None else None))
        ops.dense_bwd_dw_grouped_bf16(problems, accumulate=True)
        return None

    def _bwd_desc(self, a_ctx, v_ctx):
        a_layers, _, c_layers = self._parts()
        desc = lambda ls, c: ([l._fb for l in ls],
                              [ls[0].in_features] + [l.out_features for l in ls],
                              [l.act_code for l in ls], [sv[1] for sv in c[0]])
        return desc(a_layers, a_ctx), desc(c_layers, v_ctx)

    def gae_backward_supported(self, ctx, T: int, B: int) -> bool:
        """True when `replay_backward_gae` can take this replay context: the fused class on
        the weights-stationary backward with masks, and a [T, B] the launch supports."""
        if not GAE_IN_BWD or ctx[0] != "fused":
            return False
        _, s_ctx, a_ctx, v_ctx, squeezed, tb, masks = ctx
        a_layers, _, c_layers = self._parts()
        M = T * B
        if (masks is None or not USE_MASKS or not squeezed or tb != (T, B)
                or not (WS_POLICY_BWD and M > WS_MIN_ROWS and s_ctx[0].shape[1] <= 64)):
            return False
        if not (a_layers[-1].act_code == ops.ACT_NONE and c_layers[-1].act_code == ops.ACT_NONE
                and len(a_layers) >= 2 and len(c_layers) >= 2):
            return False
        da, dc = self._bwd_desc(a_ctx, v_ctx)
        return ops.policy_bwd_gae_supported(T, B, da, dc)

    def replay_backward_gae(self, ctx, g_reg, rewards, values, last_values, done, truncated,
                            ll_new, ll_old, reg, gamma, lambda_, normalize, clip_range,
                            critic_weight, loss_out=None, defer=None):
        """`ops.gae_ppo_loss` followed by `replay_backward`, as ONE launch + the dW launch
        (`gae_backward_supported` must hold).  Returns loss_out[4] — with `defer` (a list)
        filled only by `ops.policy_loss_finalize(defer)`."""
        _, s_ctx, a_ctx, v_ctx, _, (T, B), masks = ctx
        a_layers, sampler, c_layers = self._parts()
        ms2, ex2, off, eps2, _ = s_ctx
        dense_chain.refresh(list(a_layers) + list(c_layers))
        da, dc = self._bwd_desc(a_ctx, v_ctx)
        a_dz, c_dz, loss_out = ops.policy_bwd_gae_bf16(
            ms2, ex2, sampler._state(ms2.device), off, g_reg, da, dc, masks, rewards, values,
            last_values, done, truncated, ll_new, ll_old, reg, gamma, lambda_, normalize,
            clip_range, critic_weight, eps2=eps2, loss_out=loss_out, defer=defer,
            comm=parallel.peer_comm(), **sampler._kw())
        problems = []
        for ls, c, dz in ((c_layers, v_ctx, c_dz), (a_layers, a_ctx, a_dz)):
            for i in range(len(ls) - 1, -1, -1):
                l = ls[i]
                problems.append((c[0][i][0], dz[i], l.kernel.grad,
                                 l.bias.grad if l.bias is not None else None))
        ops.dense_bwd_dw_grouped_bf16(problems, accumulate=True)
        return loss_out

    def _backward_per_port(self, s_ctx, a_ctx, v_ctx, g_out, g_reg, g_v, M):
        a_layers, sampler, c_layers = self._parts()
        g_ms = sampler.replay_backward(s_ctx, {"action": None,
                                               "log_likelihood": g_out.loglikelihoods}, g_reg)
        g_ms = g_ms.reshape(M, g_ms.shape[-1])
        if _can_fork(g_v, getattr(self.layers[-1], "_wide", True)):
            fork = _Fork(g_v)
            with fork:
                dense_chain.backward(c_layers, v_ctx, g_v)
            dense_chain.backward(a_layers, a_ctx, g_ms)
            fork.join()
        else:
            dense_chain.backward(a_layers, a_ctx, g_ms)
            dense_chain.backward(c_layers, v_ctx, g_v)
        return None


class GRUActorCritic(Sequential):
    """The network `make_gru_actor_critic` builds — `Sequential([Normalizer, PPOAdapter(
    action=Sequential([Dense, GRU, Dense, NormalTanhSampler]), value=Sequential([Dense..]))])`
    — with its ROLLOUT / evaluation step as one launch on the bf16 path
    (`mi_gru_policy_step_bf16`).  A single step of a recurrent actor is row-local like an
    MLP's, so normaliser, both Dense layers, the GRU's input projection, its recurrent product
    and gate arithmetic and the sampler run on one row tile with every weight in registers,
    beside the value trunk: one launch for the generic containers' seven.  Module tree,
    parameters, carry / extras / metrics structure are the plain `Sequential`'s; the loss
    replay (whole sequences, BPTT) stays on the sequence kernels of networks/recurrent.py.
    Anything outside the pattern goes through the generic container path."""

    def __init__(self, layers):
        super().__init__(layers)
        assert len(self.layers) == 2 and isinstance(self.layers[-1], PPOAdapter)
        self._norm = self.layers[0]
        self._adapter = self.layers[-1]

    def _parts(self):
        a = self._adapter.action.layers
        return a[0], a[1], a[2], a[3], self._adapter.value.layers

    def _fusable(self, obs) -> bool:
        from .recurrent import GRU

        if not FUSED or config.compute_dtype() != "bf16":
            return False
        if not (isinstance(obs, torch.Tensor) and obs.dim() == 2 and obs.is_cuda
                and obs.dtype == torch.float32):
            return False
        n = self._norm
        if not (isinstance(n, Normalizer) and isinstance(n.mean.value, torch.Tensor)
                and n.mean.value.dim() == 1):
            return False
        act, val = self._adapter.action, self._adapter.value
        if not (isinstance(act, Sequential) and isinstance(val, Sequential)
                and len(act.layers) == 4):
            return False
        d_in, gru, d_out, sampler, c_layers = self._parts()
        if not (type(d_in) is Dense and type(gru) is GRU and type(d_out) is Dense
                and type(sampler) is NormalTanhSampler and c_layers
                and all(type(l) is Dense for l in c_layers)):
            return False
        if d_in.act_code != ops.ACT_RELU or d_out.act_code != ops.ACT_NONE:
            return False
        if d_in.bias is None or d_out.bias is None or not gru._mfma():
            return False
        H = gru.hidden_features
        if d_in.out_features != H or gru.in_features != H or d_out.in_features != H:
            return False
        key = (obs.shape[1], H, d_out.out_features,
               tuple(l.out_features for l in c_layers), tuple(l.act_code for l in c_layers))
        cache = self.__dict__.setdefault("_step_cache", {})
        if key not in cache:
            cache[key] = ops.gru_policy_step_supported(
                obs.shape[1], H, d_out.out_features,
                [c_layers[0].in_features] + [l.out_features for l in c_layers],
                [l.act_code for l in c_layers])
        return cache[key]

    def __call__(self, network_state, obs: Any, rollout_extras: Any = None):
        if rollout_extras is not None or not self._fusable(obs):
            return super().__call__(network_state, obs, rollout_extras)
        d_in, gru, d_out, sampler, c_layers = self._parts()
        proj = gru._proj()
        dense_chain.refresh([d_in, proj, d_out, *c_layers])  # one launch for all of them
        n = self._norm
        M = obs.shape[0]
        A = d_out.out_features // 2
        a_state = network_state[-1]["action"]
        v_state = network_state[-1]["value"]
        h = a_state[1]
        eps, eps2 = sampler._noise(M, A, obs.device)
        r = ops.gru_policy_step(
            obs if obs.is_contiguous() else obs.contiguous(),
            (n.mean.value, n.M2.value, n.counter.value, n.epsilon),
            (d_in._ff, d_in.bias.data), (proj._ff, proj.bias.data if proj.bias is not None
                                         else None),
            gru.w_h.data, gru.b_hn.data, (d_out._ff, d_out.bias.data),
            h if h.is_contiguous() else h.contiguous(),
            ([l._ff for l in c_layers], [dense_chain._bias(l) for l in c_layers],
             [c_layers[0].in_features] + [l.out_features for l in c_layers],
             [l.act_code for l in c_layers]),
            sampler._state(obs.device), sampler._next_offset(),
            deterministic=sampler.deterministic, eps=eps, eps2=eps2, **sampler._kw())
        value = r["value"]
        if value.shape[-1] == 1:
            value = value.squeeze(-1)
        next_action = list(a_state)
        next_action[1] = r["h"]
        a_metrics = {0: {}, 1: {}, 2: {}, 3: {"mu": r["mu"], "sigma": r["sigma"]}}
        return StatefulModuleOutput(
            next_state=[(), {"action": next_action, "value": list(v_state)}],
            output=PPONetworkOutput(actions=r["action"], loglikelihoods=r["log_likelihood"],
                                    value_estimates=value),
            regularization_loss=r["reg"],
            metrics={0: {}, 1: {"action": a_metrics,
                                "value": {i: {} for i in range(len(c_layers))}}},
            rollout_extras=[obs, {"action": [None, None, None, r["raw"]],
                                  "value": [None] * len(c_layers)}])

    # ---- the whole rollout in one launch (rollout.py:48-73) ---------------------------
    def unroll_fused(self, env, env_state, network_state, unroll_length: int, reset_key):
        """`MLPActorCritic.unroll_fused` for the recurrent actor-critic
        (`mi_rollout_mock_gru_ws_bf16`): the GRU carry stays on chip for the T steps and is
        reset (zeros) where a step ends an episode.  Every leaf bit-identical to the stepwise
        rollout; None when the fused form does not apply."""
        if not (FUSED and FUSED_ROLLOUT):
            return None
        from .. import random as rnd
        from ..algorithms.types import State, Transition
        from ..envs.constants import constant
        from ..envs.synthetic import MockEnv
        from ..wrappers.episode_wrapper import EpisodeWrapper

        if type(env) is not EpisodeWrapper or type(env.env) is not MockEnv \
                or not isinstance(env.env.obs_size, int) or rnd._TORCH_ONLY[0] \
                or env.env.obs_law != "uniform" or not isinstance(env_state, State):
            return None
        obs = env_state.obs
        if not self._fusable(obs) or unroll_length < 1 or obs.shape[1] != env.env.obs_size:
            return None
        N, K0 = obs.shape
        data, info = env_state.data, env_state.info
        if not (isinstance(data, dict) and isinstance(info, dict)
                and set(data) == {"key", "step_count"}
                and set(info) == {"step_counter", "truncated"}) or env_state.metrics:
            return None
        key, count, counter = data["key"], data["step_count"], info["step_counter"]
        for t in (key, count, counter):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int64
                    and t.shape == (N,)):
                return None
        if not (isinstance(reset_key, torch.Tensor) and reset_key.is_cuda
                and reset_key.dtype == torch.int64 and reset_key.dim() == 0):
            return None
        d_in, gru, d_out, sampler, c_layers = self._parts()
        if sampler.noise_override is not None or getattr(gru, "trainable_initial_state", False):
            return None
        a_state = network_state[-1]["action"]
        v_state = network_state[-1]["value"]
        h = a_state[1]
        if not (isinstance(h, torch.Tensor) and h.is_cuda and h.dtype == torch.float32
                and h.shape == (N, gru.hidden_features)):
            return None
        proj = gru._proj()
        dense_chain.refresh([d_in, proj, d_out, *c_layers])
        n = self._norm
        T = int(unroll_length)
        off = sampler._pending
        sampler._pending += T
        c = lambda t: t if t.is_contiguous() else t.contiguous()
        r = ops.rollout_mock_gru_ws(
            c(key), c(count), c(counter), c(obs), reset_key, env.env.max_steps, env.max_len, T,
            (n.mean.value, n.M2.value, n.counter.value, n.epsilon),
            (d_in._ff, d_in.bias.data),
            (proj._ff, proj.bias.data if proj.bias is not None else None),
            gru.w_h.data, gru.b_hn.data, (d_out._ff, d_out.bias.data), c(h),
            ([l._ff for l in c_layers], [dense_chain._bias(l) for l in c_layers],
             [c_layers[0].in_features] + [l.out_features for l in c_layers],
             [l.act_code for l in c_layers]),
            sampler._state(obs.device), off, deterministic=sampler.deterministic,
            **sampler._kw())
        value = r["value"]
        if value.shape[-1] == 1:
            value = value.squeeze(-1)
        a_metrics = {0: {}, 1: {}, 2: {}, 3: {"mu": r["mu"], "sigma": r["sigma"]}}
        rollout = Transition(
            obs=r["obs"],
            network_output=PPONetworkOutput(actions=r["action"],
                                            loglikelihoods=r["log_likelihood"],
                                            value_estimates=value),
            rewards=r["reward"], done=r["done"], truncated=r["truncated"],
            next_obs=r["next_obs"],
            metrics={"env": {}, "net": {0: {}, 1: {"action": a_metrics,
                                                   "value": {i: {} for i in
                                                             range(len(c_layers))}}}},
            rollout_extras=[r["obs"], {"action": [None, None, None, r["raw"]],
                                       "value": [None] * len(c_layers)}])
        dev = obs.device
        final_env = State(
            data={"key": r["key_out"], "step_count": r["count_out"]}, obs=r["obs_out"],
            reward=r["reward_out"], done=constant((N,), torch.float32, 0.0, dev), metrics={},
            info={"step_counter": r["counter_out"],
                  "truncated": constant((N,), torch.bool, 0, dev)})
        next_action = list(a_state)
        next_action[1] = r["h_out"]
        final_net = [(), {"action": next_action, "value": list(v_state)}]
        return final_net, final_env, rollout
