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
