"""`ppo_step` against the oracle on the network SHAPES of the BASELINE configs the other
tests only cover at kernel level ("configs untested"):

  C3  CheetahRun-shaped: dict obs {position 8, velocity 9} -> Normalizer -> Flattener ->
      actor 4x256 / critic 2x512, bf16, at M = T * mb > 8192 rows per gradient step, which
      is where the 512-wide trunk leaves the 16-row rollout shape of the kernels
      (`csrc/mlp_bf16.hip: launch_chain`, RT = 4) — the dispatch the C3 bench runs.
  C4  CartpoleBalance-shaped GRU actor (Dense 5->64, GRU 64, Dense 64->2) / critic 2x256 on
      the bf16 matrix-core recurrence (`csrc/gru_mfma.hip`), with `max_steps=5` resets so
      about a fifth of the steps reset the carry.

bf16 operands against an fp64 oracle: events (obs stream, flags, keys) stay bit-exact;
losses within the bf16 end-to-end bounds of DESIGN §5 (5e-2 actor / regulariser, 2e-3
critic); parameters after the Adam steps within 2e-2."""
import numpy as np
import pytest
import torch

from nnx_ppo_amd import random as keys
from oracle import keys as okeys  # the oracle's own key scheme (numpy; tests/test_oracle_keys.py)
from oracle import networks as on
from oracle import ppo as op

pytestmark = pytest.mark.gpu


def _c3_net(seed=17):
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.adapter import PPOAdapter
    from nnx_ppo_amd.networks.containers import Sequential
    from nnx_ppo_amd.networks.normalizer import Normalizer
    from nnx_ppo_amd.networks.sampling_layers import NormalTanhSampler
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.networks.utils import Flattener

    rngs = Rngs(seed)
    actor = factories.make_mlp_layers([17] + [256] * 4 + [12], rngs, activation_last_layer=False)
    critic = factories.make_mlp([17] + [512] * 2 + [1], rngs, activation_last_layer=False)
    sampler = NormalTanhSampler(rngs, entropy_weight=1e-2, min_std=1e-1)
    return Sequential([Normalizer({"position": 8, "velocity": 9}), Flattener(),
                       PPOAdapter(action=Sequential([*actor, sampler]), value=critic)])


def _c3_net_factory(seed=17):
    """The same network from `make_mlp_actor_critic` with a dict `obs_size`: the rollout step
    is then one concatenation + one launch (networks/policy.py)."""
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs

    return factories.make_mlp_actor_critic({"position": 8, "velocity": 9}, 6, [256] * 4,
                                           [512] * 2, Rngs(seed))


def _called(prof) -> set:
    return {name for name, *_ in prof.records}


@pytest.mark.parametrize("build", ["by_hand", "factory"])
def test_c3_shape_ppo_step_vs_oracle(dev, build):
    from nnx_ppo_amd import _lib, config
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import cheetah_shaped
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    N, T = 512, 30                       # one minibatch: M = 15 360 rows per gradient step
    mk_env = lambda: EpisodeWrapper(cheetah_shaped(max_steps=11), 40)
    with config.use_compute_dtype("bf16"):
        from oracle import envs as oe

        # the oracle side on the oracle's own env / wrapper / keys (oracle/envs.py, keys.py)
        env = mk_env()
        oenv = oe.EpisodeWrapper(oe.MockEnv({"position": 8, "velocity": 9}, 6, max_steps=11), 40)
        net = _c3_net() if build == "by_hand" else _c3_net_factory()
        ts = ppo.new_training_state(env, net, N, 18, 3e-4, device=dev)
        onet = on.from_product(net)
        ots = op.new_training_state(oenv, onet, N, 18, okeys, 3e-4)
        for k in range(2):
            with _lib.profiler as prof:
                ts, m = ppo.ppo_step(env, ts, N, T, 0.95, 0.99, 0.2, True, False, 2, 1)
            ots, info = op.ppo_step(oenv, ots, N, T, 0.95, 0.99, 0.2, True, 2, 1, okeys)
            # the kernels this shape is meant to exercise: whole-trunk forward / dX chain
            # at M > 8192 and the grouped dW — per trunk for the hand-composed network, both
            # trunks + normaliser + sampler (+ bootstrap rows) in ONE launch each way for the
            # factory's (since round 3 also with a trunk wider than 256 at training sizes)
            used = _called(prof)
            if build == "by_hand":
                assert {"mi_mlp_fwd_bf16", "mi_mlp_bwd_dx_bf16"} <= used, used
            else:
                assert {"mi_policy_fwd_bf16", "mi_policy_bwd_bf16"} <= used, used
                assert not {"mi_tanh_gauss_fwd_f32", "mi_tanh_gauss_bwd_f32",
                            "mi_normalize_fwd_f32"} & used, used
            # (its slab reduction rides on the Adam launch when nothing reads the
            # gradient in between)
            assert used & {"mi_dense_bwd_dw_grouped_bf16",
                           "mi_dense_bwd_dw_grouped_slabs_bf16"}, used
            assert int(ts.steps_taken) == (k + 1) * N * T
            for name in ("position", "velocity"):   # events stay exact under bf16
                assert torch.equal(ts.env_states.obs[name].cpu(), ots.env_states.obs[name])
            a, c, r = (info[n].numpy().mean() for n in ("actor", "critic", "regularization"))
            assert np.allclose(m["losses/actor/mean"].item(), a, rtol=5e-2, atol=5e-4), \
                (k, m["losses/actor/mean"].item(), a)
            assert np.allclose(m["losses/critic/mean"].item(), c, rtol=3e-3), \
                (k, m["losses/critic/mean"].item(), c)
            assert np.allclose(m["losses/regularization/mean"].item(), r, rtol=5e-2, atol=1e-4), \
                (k, m["losses/regularization/mean"].item(), r)
        norm, onorm = net.layers[0], onet.layers[0]
        assert float(norm.counter.value) == 2 * N * T == float(onorm.counter)
        for key in ("position", "velocity"):
            assert np.allclose(norm.mean.value[key].cpu().numpy(), onorm.mean[key].numpy(),
                               atol=1e-5)
        worst = 0.0
        for p, q in zip(net.parameters(), onet.parameters()):
            assert torch.isfinite(p.data).all()
            worst = max(worst, float((p.data.cpu() - q.detach()).abs().max()))
        assert worst < 2e-2, worst


def test_c4_gru_bf16_mfma_ppo_step_vs_oracle(dev):
    from nnx_ppo_amd import _lib, config
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import cartpole_shaped
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs

    N, T = 256, 30
    with config.use_compute_dtype("bf16"):
        net = factories.make_gru_actor_critic(5, 1, 64, [256, 256], Rngs(42))
        from oracle import envs as oe

        env, oenv = cartpole_shaped(max_steps=5), oe.MockEnv(5, 1, max_steps=5)
        ts = ppo.new_training_state(env, net, N, 42, 3e-4, 1.0, device=dev)
        onet = on.from_product(net)
        ots = op.new_training_state(oenv, onet, N, 42, okeys, 3e-4, 1.0)
        for k in range(2):
            with _lib.profiler as prof:
                ts, m = ppo.ppo_step(env, ts, N, T, 0.95, 0.99, 0.2, True, False, 2, 2)
            ots, info = op.ppo_step(oenv, ots, N, T, 0.95, 0.99, 0.2, True, 2, 2, okeys)
            used = _called(prof)
            # the matrix-core recurrence — since round 3 with the head Dense and the sampler's
            # replay riding in the forward sequence launch
            assert {"mi_gru_seq_fwd_front_proj_tail_bf16", "mi_gru_seq_bwd_proj_tail_bf16"} <= used, used
            assert "mi_gru_seq_fwd_f32" not in used   # the matrix-core recurrence, not VALU
            done = info["rollout"].done
            assert int(done.sum()) >= int(0.15 * N * T)  # reset-heavy: ~20 % of the steps
            assert torch.equal(ts.env_states.obs.cpu(), ots.env_states.obs)
            a, c, r = (info[n].numpy().mean() for n in ("actor", "critic", "regularization"))
            assert np.allclose(m["losses/actor/mean"].item(), a, rtol=5e-2, atol=5e-4), \
                (k, m["losses/actor/mean"].item(), a)
            assert np.allclose(m["losses/critic/mean"].item(), c, rtol=3e-3), \
                (k, m["losses/critic/mean"].item(), c)
            assert np.allclose(m["losses/regularization/mean"].item(), r, rtol=5e-2, atol=1e-4), \
                (k, m["losses/regularization/mean"].item(), r)
        # the carry the next iteration starts from (reset rows are exact zeros on both sides)
        carry = [t for t in _leaves(ts.network_states) if t.dim() == 2 and t.shape[1] == 64][0]
        ocarry = [t for t in _leaves(ots.network_states) if t.dim() == 2 and t.shape[1] == 64][0]
        assert np.allclose(carry.cpu().numpy(), ocarry.numpy(), atol=3e-2)
        worst = 0.0
        for p, q in zip(net.parameters(), onet.parameters()):
            assert torch.isfinite(p.data).all()
            worst = max(worst, float((p.data.cpu() - q.detach()).abs().max()))
        assert worst < 2e-2, worst


def _leaves(tree):
    from nnx_ppo_amd.tree import tree_leaves

    return [t for t in tree_leaves(tree) if isinstance(t, torch.Tensor)]
