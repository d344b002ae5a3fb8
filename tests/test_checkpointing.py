"""Checkpoint / resume — the cases of `nnx_ppo/algorithms/checkpointing_test.py`
(step directory naming 103-112, base directory creation 114-124, contents 126-137,
round trips with / without config 139-179, `steps_taken` 181-198, weights 200-225,
accumulation 227-243, normaliser statistics 264-...), on the torch-native format.
CPU: building and saving a training state launches no kernel."""
import os

import pytest
import torch

from nnx_ppo_amd.algorithms import ppo
from nnx_ppo_amd.algorithms.checkpointing import load_checkpoint, make_checkpoint_fn
from nnx_ppo_amd.algorithms.config import EvalConfig, PPOConfig, TrainConfig
from nnx_ppo_amd.algorithms.types import TrainingState
from nnx_ppo_amd.envs import cartpole_shaped
from nnx_ppo_amd.networks import factories
from nnx_ppo_amd.networks.types import Rngs
from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper


def _state(seed=17, n_envs=4):
    env = EpisodeWrapper(cartpole_shaped(max_steps=50), 50)
    net = factories.make_mlp_actor_critic(5, 1, [16, 16], [32], Rngs(seed))
    return ppo.new_training_state(env, net, n_envs, 42, device="cpu")


def test_default_and_custom_checkpoint_every_steps():
    assert TrainConfig().checkpoint_every_steps == 500_000
    assert TrainConfig(checkpoint_every_steps=100_000).checkpoint_every_steps == 100_000


def test_directories_and_contents(tmp_path):
    base = tmp_path / "new_subdir"
    assert not base.exists()
    fn = make_checkpoint_fn(str(base))
    st = _state()
    fn(st, step=5000)
    step_dir = base / "step_0000005000"
    assert step_dir.is_dir()
    for name in ("networks.pt", "optimizer.pt", "metadata.pt"):
        assert (step_dir / name).is_file()
    fn(st, step=1000)
    fn(st, step=2000)
    assert sorted(os.listdir(base)) == ["step_0000001000", "step_0000002000", "step_0000005000"]


def test_files_load_without_executing_anything(tmp_path):
    make_checkpoint_fn(str(tmp_path))(_state(), step=1)
    for name in ("networks.pt", "optimizer.pt", "metadata.pt"):
        obj = torch.load(tmp_path / "step_0000000001" / name, weights_only=True)
        assert isinstance(obj, dict)


@pytest.mark.parametrize("with_config", [False, True])
def test_round_trip(tmp_path, with_config):
    config = TrainConfig(ppo=PPOConfig(n_envs=4, total_steps=1000),
                         eval=EvalConfig(enabled=False)) if with_config else None
    st = _state(seed=17)
    # make the saved state distinguishable
    norm = st.networks.layers[0]
    norm.mean.value.copy_(torch.arange(5.0))
    norm.counter.value.fill_(123.0)
    st.optimizer.m.normal_()
    st.optimizer.step.fill_(7)
    st = st.replace(steps_taken=torch.tensor(9999, dtype=torch.int64))
    make_checkpoint_fn(str(tmp_path), config=config)(st, step=1234)

    tmpl = _state(seed=99)  # same architecture, different weights
    assert not torch.equal(tmpl.optimizer.params, st.optimizer.params)
    ckpt = load_checkpoint(str(tmp_path / "step_0000001234"), tmpl.networks, tmpl.optimizer)
    assert set(ckpt) == {"training_state", "step", "config"} and ckpt["step"] == 1234
    loaded = ckpt["training_state"]
    assert isinstance(loaded, TrainingState)
    assert int(loaded.steps_taken) == 9999
    assert loaded.networks is tmpl.networks and loaded.optimizer is tmpl.optimizer
    for (n1, p1), (n2, p2) in zip(st.networks.named_parameters(),
                                  loaded.networks.named_parameters()):
        assert n1 == n2 and torch.equal(p1.data, p2.data)
    assert torch.equal(loaded.optimizer.m, st.optimizer.m)
    assert int(loaded.optimizer.step) == 7
    ln = loaded.networks.layers[0]
    assert torch.equal(ln.mean.value, torch.arange(5.0)) and float(ln.counter.value) == 123.0
    assert torch.equal(loaded.rng_key, st.rng_key)
    from nnx_ppo_amd.tree import tree_leaves

    for a, b in zip(tree_leaves(loaded.env_states), tree_leaves(st.env_states)):
        assert torch.equal(a, b)
    assert type(loaded.env_states) is type(st.env_states)
    if with_config:
        assert ckpt["config"] == config and isinstance(ckpt["config"].ppo, PPOConfig)
    else:
        assert ckpt["config"] is None


def test_mismatched_architecture_is_refused(tmp_path):
    make_checkpoint_fn(str(tmp_path))(_state(), step=1)
    env = EpisodeWrapper(cartpole_shaped(max_steps=50), 50)
    other = factories.make_mlp_actor_critic(5, 1, [16, 16, 16], [32], Rngs(0))
    tmpl = ppo.new_training_state(env, other, 4, 1, device="cpu")
    with pytest.raises(ValueError):
        load_checkpoint(str(tmp_path / "step_0000000001"), tmpl.networks, tmpl.optimizer)


def test_crafted_class_references_are_refused(tmp_path):
    """A checkpoint names classes; a name is honoured only for allow-listed modules and
    only if it is the kind of class its tag claims (`__enum__: os:system`)."""
    from nnx_ppo_amd.algorithms import checkpointing as ck

    for bad in ({"__enum__": "os:getenv", "value": "HOME"},
                {"__enum__": "builtins:print", "value": "x"},
                {"__enum__": "nnx_ppo_amd.algorithms.checkpointing:make_checkpoint_fn",
                 "value": "/tmp/x"},
                {"__enum__": "nnx_ppo_amd.algorithms.config:TrainConfig", "value": 1},
                {"__dataclass__": "os:system", "fields": {}},
                {"__dataclass__": "nnx_ppo_amd.algorithms.types:LoggingLevel", "fields": {}},
                {"__dataclass__": "nnx_ppo_amd.algorithms.config:TrainConfig",
                 "fields": {"not_a_field": 1}},
                {"__dataclass__": "nnx_ppo_amd.algorithms.checkpointing:_lookup", "fields": {}}):
        with pytest.raises(RuntimeError):
            ck._decode(bad, "cpu")
    # the legitimate forms still decode
    from nnx_ppo_amd.algorithms.types import LoggingLevel

    enc = ck._encode(TrainConfig(ppo=PPOConfig(logging_level=LoggingLevel.LOSSES)))
    assert ck._decode(enc, "cpu") == TrainConfig(ppo=PPOConfig(logging_level=LoggingLevel.LOSSES))
    # and a crafted metadata.pt is refused by load_checkpoint as a whole
    make_checkpoint_fn(str(tmp_path))(_state(), step=1)
    meta_path = tmp_path / "step_0000000001" / "metadata.pt"
    meta = torch.load(meta_path, weights_only=True)
    meta["config"] = {"__enum__": "os:getenv", "value": "HOME"}
    torch.save(meta, meta_path)
    tmpl = _state(seed=3)
    with pytest.raises(RuntimeError):
        load_checkpoint(str(tmp_path / "step_0000000001"), tmpl.networks, tmpl.optimizer)


def test_user_module_needs_allow_listing():
    import dataclasses as dc
    import sys
    import types

    from nnx_ppo_amd.algorithms import checkpointing as ck

    mod = types.ModuleType("user_env_mod")

    @dc.dataclass
    class S:
        a: int = 0

    S.__module__, S.__qualname__ = "user_env_mod", "S"
    mod.S = S
    sys.modules["user_env_mod"] = mod
    try:
        enc = ck._encode(S(3))
        with pytest.raises(RuntimeError):
            ck._decode(enc, "cpu")
        ck.allow_checkpoint_module("user_env_mod")
        assert ck._decode(enc, "cpu") == S(3)
    finally:
        ck._ALLOWED_MODULE_PREFIXES.remove("user_env_mod")
        del sys.modules["user_env_mod"]
