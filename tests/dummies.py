"""Fake networks for tests, restating the reference's test doubles
(`nnx_ppo/test_dummies/dummy_counter.py:46-75`, `stateful_nets.py:12-40`) on the
product's StatefulModule protocol.  Test fakes: plain torch ops."""
import torch

from nnx_ppo_amd.networks.types import (PPONetworkOutput, StatefulModule, StatefulModuleOutput,
                                        zero_scalar)


class DummyCounterNet(StatefulModule):
    """Outputs the number of steps since its last reset, whatever the input."""

    def __call__(self, state, obs, rollout_extras=None):
        old = state["counter_state"]["counter"]
        new = old + 1
        f = new.to(torch.float32)
        return StatefulModuleOutput(
            next_state={"counter_state": {"counter": new}},
            output=PPONetworkOutput(actions=f[:, None], loglikelihoods=torch.ones_like(f),
                                    value_estimates=torch.ones_like(f)),
            regularization_loss=zero_scalar(obs.device), metrics={}, rollout_extras=None)

    def initialize_state(self, batch_size):
        return {"counter_state": {"counter": torch.zeros(batch_size, dtype=torch.int64,
                                                         device=self.device)}}

    def reset_state(self, prev_state):
        return {"counter_state": {"counter": torch.zeros_like(prev_state["counter_state"]["counter"])}}


class RepeatAndCountNet(StatefulModule):
    """Outputs its input as action and counts how many samples it has seen."""

    def __init__(self):
        self.n_calls = 0

    def __call__(self, state, obs, rollout_extras=None):
        b = obs.shape[0]
        self.n_calls += b
        one = torch.ones(b, device=obs.device)
        return StatefulModuleOutput((), PPONetworkOutput(obs, one, one), zero_scalar(obs.device),
                                    {}, None)


# ---- single-env (un-batched) envs, written the way the reference writes them -----------
# (`nnx_ppo/test_dummies/dummy_counter.py:10-43`, `move_to_center_env.py:10-50`: scalar
# counters, `[2]` positions, a scalar key) — what `envs.VmapEnv` lifts to the batched
# convention.
class SingleDummyCounterEnv:
    observation_size = 1
    action_size = 1

    def reset(self, rng):
        from nnx_ppo_amd import random as rnd
        from nnx_ppo_amd.algorithms.types import State

        zero = torch.zeros((), dtype=torch.int64, device=rng.device)
        return State(data={"current_step": zero, "reset_step": rnd.randint(rng, (), 3, 10)},
                     obs=torch.zeros(1, dtype=torch.float32, device=rng.device),
                     info={"current_step": zero}, reward=1.0, done=0.0, metrics={})

    def step(self, state, action):
        from nnx_ppo_amd.algorithms.types import State

        cur = state.data["current_step"] + 1
        data = {"current_step": cur, "reset_step": state.data["reset_step"]}
        done = (cur >= data["reset_step"]).to(torch.float32)
        a = action.reshape(())
        return State(data=data, obs=torch.zeros(1, dtype=torch.float32, device=cur.device),
                     info={"current_step": cur},
                     reward=torch.where(a == cur.to(a.dtype), 1.0, 0.0).to(torch.float32),
                     done=done, metrics=state.metrics)


class SingleMoveToCenterEnv:
    observation_size = 2
    action_size = 2

    def __init__(self, reward_falloff: float = 0.5, border_radius: float = 2.0):
        self.reward_falloff = reward_falloff
        self.border_radius = border_radius

    def reset(self, rng):
        from nnx_ppo_amd import random as rnd

        u = rnd.uniform(rng, (2,))
        phi, rad = u[0], u[1] * (self.border_radius * 0.9)
        ang = 2 * torch.pi * phi
        return self._get_state({"pos": torch.stack([torch.cos(ang) * rad, torch.sin(ang) * rad])})

    def step(self, state, action):
        return self._get_state({"pos": state.data["pos"] + torch.clamp(action, -1, 1)})

    def _get_state(self, data):
        from nnx_ppo_amd.algorithms.types import State

        d_sqr = torch.square(data["pos"]).sum(-1)
        reward = torch.exp(-(d_sqr / (self.reward_falloff ** 2) / 2))
        done = torch.where(d_sqr > self.border_radius ** 2, 1.0, 0.0).to(torch.float32)
        return State(data=data, obs=data["pos"] / 10.0, info={}, reward=reward, done=done,
                     metrics={})
