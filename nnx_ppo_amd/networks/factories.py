"""MLP builders (counterpart of `nnx_ppo/networks/factories.py`):
`make_mlp_layers` 14-45, `make_mlp` 48-69, `make_mlp_actor_critic` 72-146 —
same signatures and defaults."""
from __future__ import annotations

from typing import Any, Union

from . import activations, initializers
from .adapter import PPOAdapter
from .containers import Sequential
from .feedforward import Dense
from .normalizer import Normalizer
from .recurrent import GRU
from .sampling_layers import NormalTanhSampler
from .types import Rngs, StatefulModule


def make_mlp_layers(sizes: list[int], rngs: Rngs, activation: Any = activations.relu,
                    activation_last_layer: bool = True, **linear_kwargs) -> list[Dense]:
    layers = []
    for i, (din, dout) in enumerate(zip(sizes[:-1], sizes[1:])):
        is_last = i == len(sizes) - 2
        act = activation if (not is_last or activation_last_layer) else None
        layers.append(Dense(din, dout, rngs, activation=act, **linear_kwargs))
    return layers


def make_mlp(sizes: list[int], rngs: Rngs, activation: Any = activations.relu,
             activation_last_layer: bool = True, **linear_kwargs) -> Sequential:
    return Sequential(
        make_mlp_layers(sizes, rngs, activation, activation_last_layer, **linear_kwargs))


def make_mlp_actor_critic(
    obs_size: Union[int, dict],
    action_size: int,
    actor_hidden_sizes: list[int],
    critic_hidden_sizes: list[int],
    rngs: Rngs,
    activation: Union[Any, str] = activations.relu,
    normalize_obs: bool = True,
    initializer_scale: float = 1.0,
    entropy_weight: float = 1e-2,
    min_std: float = 1e-1,
    std_scale: float = 1.0,
) -> StatefulModule:
    """Sequential([Normalizer(obs_size)?, PPOAdapter(action=Sequential([actor...,
    NormalTanhSampler]), value=critic)]) — factories.py:88-146.

    `obs_size` may also be a dict `{name: width}` (a PyTree observation of flat leaves,
    BASELINE config 3): the network is then what a user of the reference composes by hand,
    `Sequential([Normalizer(obs_size)?, Flattener(), PPOAdapter(...)])` — per-leaf running
    statistics, the leaves concatenated in sorted-key order in front of the trunks."""
    tree_obs = isinstance(obs_size, dict)
    if tree_obs:
        from .utils import Flattener

        obs_tree = {k: int(v) for k, v in obs_size.items()}
        obs_size = sum(obs_tree.values())
    if isinstance(activation, str):
        activation = {"swish": activations.swish, "tanh": activations.tanh,
                      "relu": activations.relu}[activation]
    kernel_init = initializers.variance_scaling(initializer_scale, "fan_in", "uniform")
    actor_layers = make_mlp_layers([obs_size] + list(actor_hidden_sizes) + [action_size * 2],
                                   rngs, activation, activation_last_layer=False,
                                   kernel_init=kernel_init)
    critic = make_mlp([obs_size] + list(critic_hidden_sizes) + [1], rngs, activation,
                      activation_last_layer=False, kernel_init=kernel_init)
    sampler = NormalTanhSampler(rngs, entropy_weight=entropy_weight, min_std=min_std,
                                std_scale=std_scale)
    adapter = PPOAdapter(action=Sequential([*actor_layers, sampler]), value=critic)
    # the same module tree as the reference's; `MLPActorCritic` is a `Sequential` that
    # evaluates this particular tree in one launch on the bf16 path (networks/policy.py)
    from .policy import MLPActorCritic

    if tree_obs:
        if normalize_obs:
            return MLPActorCritic([Normalizer(obs_tree), Flattener(), adapter])
        return Sequential([Flattener(), adapter])
    if normalize_obs:
        return MLPActorCritic([Normalizer(obs_size), adapter])
    return adapter


def make_gru_actor_critic(
    obs_size: int,
    action_size: int,
    hidden_size: int,
    critic_hidden_sizes: list[int],
    rngs: Rngs,
    activation: Union[Any, str] = activations.relu,
    normalize_obs: bool = True,
    initializer_scale: float = 1.0,
    entropy_weight: float = 1e-2,
    min_std: float = 1e-1,
    std_scale: float = 1.0,
) -> StatefulModule:
    """Recurrent actor / feed-forward critic (BASELINE.json config 4).  The
    reference has no recurrent factory; this composes the network the way its
    tests hand-compose an LSTM actor (`recurrent_test.py:245-261`):
    actor = Dense(obs -> H, act) -> GRU(H -> H) -> Dense(H -> 2A) -> sampler."""
    if isinstance(activation, str):
        activation = {"swish": activations.swish, "tanh": activations.tanh,
                      "relu": activations.relu}[activation]
    kernel_init = initializers.variance_scaling(initializer_scale, "fan_in", "uniform")
    actor_layers = [
        Dense(obs_size, hidden_size, rngs, activation=activation, kernel_init=kernel_init),
        GRU(hidden_size, hidden_size, rngs, kernel_init=kernel_init,
            recurrent_kernel_init=kernel_init),
        Dense(hidden_size, action_size * 2, rngs, activation=None, kernel_init=kernel_init),
    ]
    critic = make_mlp([obs_size] + list(critic_hidden_sizes) + [1], rngs, activation,
                      activation_last_layer=False, kernel_init=kernel_init)
    sampler = NormalTanhSampler(rngs, entropy_weight=entropy_weight, min_std=min_std,
                                std_scale=std_scale)
    adapter = PPOAdapter(action=Sequential([*actor_layers, sampler]), value=critic)
    if normalize_obs:
        # the same module tree; `GRUActorCritic` is a `Sequential` that evaluates a rollout
        # step of this particular tree in one launch on the bf16 path (networks/policy.py)
        from .policy import GRUActorCritic

        return GRUActorCritic([Normalizer(obs_size), adapter])
    return adapter
