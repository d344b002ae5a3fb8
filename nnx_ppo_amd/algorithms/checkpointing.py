"""Checkpoint / resume (counterpart of `nnx_ppo/algorithms/checkpointing.py:42-204`:
`make_checkpoint_fn`, `load_checkpoint`; `train_ppo(..., checkpoint_fn=, initial_state=)`
is the consumer, `ppo.py:52,91-102`).

Same contract, torch-native files: each checkpoint is `{directory}/step_{step:010d}/`
with

  networks.pt    parameters by name, the modules' `Variable`s (normaliser statistics)
                 and sampler RNG state `{seed, offset}`            (reference: orbax dir)
  optimizer.pt   the flat arenas: params, Adam m / v, step, layout  (reference: orbax dir)
  metadata.pt    `network_states`, `env_states`, `rng_key`, `steps_taken`, `step` and
                 the optional `TrainConfig`                         (reference: pickle)

Every file holds only tensors, numbers, strings, lists and dicts, so it loads with
`torch.load(..., weights_only=True)` — nothing in a checkpoint is executed on load.
Dataclass nodes (env `State`, configs) are stored as tagged dicts and rebuilt from
classes that are already imported."""
from __future__ import annotations

import dataclasses
import enum
import os
import sys
from typing import Any, Optional

import torch

from ..networks.types import StatefulModule, Variable, bump_param_epoch
from ..optim import Optimizer
from .config import TrainConfig
from .types import TrainingState

_TAG = "__dataclass__"
_ENUM = "__enum__"
_TUPLE = "__tuple__"


def _encode(x: Any) -> Any:
    if isinstance(x, torch.Tensor):
        return x.detach().cpu()
    if dataclasses.is_dataclass(x) and not isinstance(x, type):
        cls = type(x)
        return {_TAG: f"{cls.__module__}:{cls.__qualname__}",
                "fields": {f.name: _encode(getattr(x, f.name)) for f in dataclasses.fields(x)}}
    if isinstance(x, enum.Enum):
        cls = type(x)
        return {_ENUM: f"{cls.__module__}:{cls.__qualname__}", "value": x.value}
    if isinstance(x, dict):
        return {k: _encode(v) for k, v in x.items()}
    if isinstance(x, tuple):
        return {_TUPLE: [_encode(v) for v in x]}
    if isinstance(x, list):
        return [_encode(v) for v in x]
    if x is None or isinstance(x, (bool, int, float, str)):
        return x
    raise TypeError(f"checkpoint: cannot store a {type(x).__name__} (tensors, dataclasses, "
                    "enums, dicts, lists, tuples, numbers and strings only)")


# Classes a checkpoint may name.  A checkpoint only ever NAMES classes (it holds no code),
# and a name is honoured only if (a) its module is on this allow-list, (b) the module is
# already imported, and (c) the object found there is the kind of class the tag claims — a
# dataclass type for `__dataclass__`, an `enum.Enum` subclass for `__enum__`.  Without
# these checks a crafted metadata.pt could name any callable of any imported module
# (`os:system`) and have it called with a value from the file.
_ALLOWED_MODULE_PREFIXES: list[str] = ["nnx_ppo_amd."]


def allow_checkpoint_module(module: str) -> None:
    """Let checkpoints restore dataclasses / enums defined in `module` (e.g. the module of
    a user env's `State`).  Exact module name, or a package prefix ending in '.'."""
    if module not in _ALLOWED_MODULE_PREFIXES:
        _ALLOWED_MODULE_PREFIXES.append(module)


def _module_allowed(module: str) -> bool:
    for a in _ALLOWED_MODULE_PREFIXES:
        if module == a or module == a.rstrip(".") or (a.endswith(".") and module.startswith(a)):
            return True
    return False


def _lookup(spec: str, kind: str):
    if not isinstance(spec, str):
        raise RuntimeError(f"checkpoint: malformed class reference {spec!r}")
    module, _, qual = spec.partition(":")
    if not _module_allowed(module):
        raise RuntimeError(
            f"checkpoint refers to {spec}, whose module is not on the allow-list; call "
            f"checkpointing.allow_checkpoint_module({module!r}) before loading if you trust it")
    mod = sys.modules.get(module)
    if mod is None:
        raise RuntimeError(f"checkpoint refers to {spec}; import {module} before loading "
                           "(classes are looked up, never imported, while loading)")
    obj = mod
    for part in qual.split("."):
        if not part or part.startswith("_"):
            raise RuntimeError(f"checkpoint: refusing private attribute in {spec}")
        obj = getattr(obj, part)
    if not isinstance(obj, type):
        raise RuntimeError(f"checkpoint: {spec} is not a class")
    if kind == "dataclass":
        if not dataclasses.is_dataclass(obj):
            raise RuntimeError(f"checkpoint: {spec} is not a dataclass")
    elif kind == "enum":
        if not issubclass(obj, enum.Enum):
            raise RuntimeError(f"checkpoint: {spec} is not an Enum")
    else:  # pragma: no cover
        raise AssertionError(kind)
    return obj


def _decode(x: Any, device) -> Any:
    if isinstance(x, torch.Tensor):
        return x.to(device)
    if isinstance(x, dict):
        if _TAG in x:
            cls = _lookup(x[_TAG], "dataclass")
            names = {f.name for f in dataclasses.fields(cls)}
            fields = x.get("fields")
            if not isinstance(fields, dict) or not set(fields) <= names:
                raise RuntimeError(f"checkpoint: fields of {x[_TAG]} do not match the class")
            return cls(**{k: _decode(v, device) for k, v in fields.items()})
        if _ENUM in x:
            return _lookup(x[_ENUM], "enum")(x["value"])
        if _TUPLE in x:
            return tuple(_decode(v, device) for v in x[_TUPLE])
        return {k: _decode(v, device) for k, v in x.items()}
    if isinstance(x, list):
        return [_decode(v, device) for v in x]
    return x


def _named_variables(networks: StatefulModule) -> list[tuple[str, Variable]]:
    out: list[tuple[str, Variable]] = []
    seen: set[int] = set()

    def rec(m, prefix):
        for name, v in vars(m).items():
            if isinstance(v, Variable) and id(v) not in seen:
                seen.add(id(v))
                out.append((prefix + name, v))
        for cname, c in m._children():
            rec(c, f"{prefix}{cname}.")

    rec(networks, "")
    return out


def _named_samplers(networks: StatefulModule) -> list[tuple[str, Any]]:
    out = []

    def rec(m, prefix):
        if hasattr(m, "advance_rng") and hasattr(m, "seed"):
            out.append((prefix.rstrip("."), m))
        for cname, c in m._children():
            rec(c, f"{prefix}{cname}.")

    rec(networks, "")
    return out


def make_checkpoint_fn(directory: str, config: Optional[TrainConfig] = None):
    """checkpointing.py:42-114 — returns `checkpoint_fn(training_state, step)` for
    `train_ppo(..., checkpoint_fn=...)`."""
    abs_directory = os.path.abspath(directory)

    def checkpoint_fn(training_state: TrainingState, step: int) -> None:
        step_dir = os.path.join(abs_directory, f"step_{int(step):010d}")
        os.makedirs(step_dir, exist_ok=True)
        nets: StatefulModule = training_state.networks
        samplers = {}
        for name, m in _named_samplers(nets):
            off = 0 if m.rng_state is None else int(m.rng_state[1].item())
            samplers[name] = {"seed": int(m.seed), "offset": off + int(m._pending)}
        torch.save({
            "parameters": {n: p.data.detach().cpu().clone() for n, p in nets.named_parameters()},
            "variables": {n: _encode(v.value) for n, v in _named_variables(nets)},
            "samplers": samplers,
        }, os.path.join(step_dir, "networks.pt"))
        torch.save(_encode(training_state.optimizer.state_dict()),
                   os.path.join(step_dir, "optimizer.pt"))
        torch.save({
            "network_states": _encode(training_state.network_states),
            "env_states": _encode(training_state.env_states),
            "rng_key": _encode(training_state.rng_key),
            "steps_taken": _encode(training_state.steps_taken),
            "step": int(step),
            "config": _encode(config),
        }, os.path.join(step_dir, "metadata.pt"))

    return checkpoint_fn


def load_checkpoint(path: str, networks: StatefulModule, optimizer: Optimizer) -> dict:
    """checkpointing.py:117-204 — `networks` / `optimizer` are structural templates
    (same architecture; their values are overwritten in place).  Returns
    `{"training_state", "step", "config"}`."""
    path = os.path.abspath(path)
    device = optimizer.device
    load = lambda name: torch.load(os.path.join(path, name), map_location="cpu",
                                   weights_only=True)
    net = load("networks.pt")
    params = dict(networks.named_parameters())
    if set(params) != set(net["parameters"]):
        raise ValueError("load_checkpoint: the network's parameters do not match the "
                         f"checkpoint ({sorted(set(params) ^ set(net['parameters']))})")
    opt_sd = _decode(load("optimizer.pt"), device)
    if list(opt_sd["names"]) != list(optimizer.names) or \
            [tuple(s) for s in opt_sd["shapes"]] != [tuple(s) for s in optimizer.shapes]:
        raise ValueError("load_checkpoint: the optimizer layout does not match the checkpoint")
    optimizer.load_state_dict(opt_sd)  # parameters live in the optimizer's arena
    for n, p in params.items():
        if tuple(p.shape) != tuple(net["parameters"][n].shape):
            raise ValueError(f"load_checkpoint: shape of {n} differs")
        p.data.copy_(net["parameters"][n].to(device))
    variables = dict(_named_variables(networks))
    for n, val in net["variables"].items():
        from ..tree import tree_map

        tree_map(lambda dst, src: dst.copy_(src.to(dst.device)), variables[n].value,
                 _decode(val, device))
    samplers = dict(_named_samplers(networks))
    for n, st in net["samplers"].items():
        m = samplers[n]
        m.seed = int(st["seed"])
        m.rng_state = None
        m._pending = 0
        from .. import ops

        m.rng_state = ops.make_rng_state(m.seed, device, int(st["offset"]))
    bump_param_epoch()
    meta = load("metadata.pt")
    training_state = TrainingState(
        networks=networks,
        network_states=_decode(meta["network_states"], device),
        env_states=_decode(meta["env_states"], device),
        optimizer=optimizer,
        rng_key=_decode(meta["rng_key"], device),
        steps_taken=_decode(meta["steps_taken"], device),
    )
    return {"training_state": training_state, "step": int(meta["step"]),
            "config": _decode(meta["config"], device)}
