#!/usr/bin/env python
"""Benchmark of the PPO hot path: env-steps/sec on a synthetic 4096-env x 30-step
rollout + update (BASELINE.json configs[1]; per GPU when sharded).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--compute f32|bf16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full `ppo_step` iteration (rollout -> n_epochs x n_minibatches
replay/GAE/loss/Adam steps -> normaliser update) run exactly as `train_ppo` runs it
(`nnx_ppo_amd/algorithms/loop.py`: one HIP-graph launch per iteration and ONE host sync
per iteration — the read of the iteration's metrics, the counterpart of the reference's
`int(steps_taken)`, nnx_ppo/algorithms/ppo.py:209 — inside the timed region; iteration
i+1 is enqueued before the host waits for iteration i, as `train_ppo` does when no
callback is due).

Timing: W untimed warm-up steps, then windows of EXACTLY K steps, each
bracketed by barrier + device synchronise on both sides and reduced with MAX over ranks;
the window is repeated until >= 1 s has been timed.  `value` = env-steps of one window /
the MEDIAN window time; `iteration_ms` = p10 / p50 / p90 of the per-iteration wall times
(host sync to host sync) over all windows.  `back_to_back` is the same graph replayed K
times with no host read in between (round 1's definition), and `train_ppo` is
`throughput/train_sps` as the drop-in entry point itself reports it on this box.

Prints ONE JSON line on rank 0 (contract in the task statement), with
`roofline` (dominant kernel, timed live with HIP events on the launch stream)
and `cpu_baseline` (the CPU oracle timed on this box's host cores, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

# workload = BASELINE.json configs[1] (C2 in SURVEY §8): CartpoleBalance-shaped
N_ENVS = 4096
T = 30
OBS, ACT = 5, 1
ACTOR_H, CRITIC_H = [64, 64, 64, 64], [256, 256]
N_EPOCHS, N_MB = 4, 4
SEED = 17
FWD_FLOP_PER_SAMPLE = 2 * (OBS * 64 + 3 * 64 * 64 + 64 * 2 * ACT + OBS * 256 + 256 * 256 + 256)
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16
PEAK_HBM_GBPS = 8000.0          # HBM3E, MI355X_MICROARCH.md
PMC_TRAFFIC_FILE = "r03_pmc_traffic.json"  # HBM bytes per launch from the PMC passes


# symbol -> (M, K, N) from the integer arguments of a recorded call (in call order;
# pointers and the stream handle are filtered out by the profiler)
GEMM_SYMBOLS = {
    "mi_dense_fwd_f32": lambda a: (a[0], a[1], a[2]),
    "mi_dense_bwd_dx_f32": lambda a: (a[0], a[1], a[2]),
    "mi_dense_bwd_dw_f32": lambda a: (a[0], a[1], a[2]),
    "mi_dense_fwd_bf16": lambda a: (a[3], a[4], a[5]),      # ldx ldwt ldy M K N act
    "mi_dense_bwd_dx_bf16": lambda a: (a[5], a[6], a[7]),   # lddz ldw ldprev act ldgx M K N
    "mi_dense_bwd_dw_bf16": lambda a: (a[2], a[3], a[4]),   # ldx lddz M K N acc
    "mi_mlp_fwd_bf16": None,                                # flops annotated by the wrapper
    "mi_mlp_bwd_dx_bf16": None,
    "mi_policy_fwd_bf16": None,
    "mi_policy_bwd_bf16": None,
    "mi_policy_ws_fwd_bf16": None,
    "mi_policy_ws_bwd_bf16": None,
    "mi_policy_ws_bwd_gae_bf16": None,
    "mi_dense_bwd_dw_grouped_bf16": None,
    "mi_dense_bwd_dw_grouped_slabs_bf16": None,
}


def build(device, state: bool = True):
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import cartpole_shaped
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    env = EpisodeWrapper(cartpole_shaped(max_steps=1000), 1000)
    net = factories.make_mlp_actor_critic(OBS, ACT, ACTOR_H, CRITIC_H, Rngs(SEED),
                                          normalize_obs=True)
    ts = ppo.new_training_state(env, net, N_ENVS, SEED, 1e-4, device=device) if state else None
    return env, net, ts


def one_iter(env, ts):
    from nnx_ppo_amd.algorithms import ppo

    ts, metrics = ppo.ppo_step(env, ts, N_ENVS, T, 0.95, 0.99, 0.2, True, False, N_EPOCHS, N_MB)
    return ts, metrics  # no host read: train_ppo counts the steps on the host


def roofline_of_dominant_kernel(env, ts):
    """One instrumented iteration: every C-ABI call bracketed by HIP events on
    the launch stream.  Dominant kernel = the dense GEMM family (MFMA-bound);
    achieved = algorithmic FLOPs (2*M*K*N per GEMM) / device time."""
    from nnx_ppo_amd import _lib

    # Eager launches arrive slower (~7 us each) than these kernels run (~3-20 us),
    # so events around them would include host gaps.  A spin kernel first holds the
    # stream while the host enqueues the whole iteration; the kernels then run back
    # to back and each event pair brackets device time only (agrees with rocprofv3).
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda._sleep(10_000_000)
    e1.record()
    torch.cuda.synchronize()
    cycles_per_ms = 10_000_000 / e0.elapsed_time(e1)
    torch.cuda._sleep(int(cycles_per_ms * 60))  # ~60 ms head start for the host
    # an event pair with nothing in between still measures ~5 us (two marker packets):
    # calibrate it behind the same blocker and subtract it from every bracket
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
             for _ in range(64)]
    for a, b in pairs:
        a.record()
        b.record()
    with _lib.profiler as prof:
        ts, _ = one_iter(env, ts)
    summ = prof.summary()
    gaps = sorted(a.elapsed_time(b) for a, b in pairs)
    pair_ms = gaps[len(gaps) // 2]
    for d in summ.values():
        d["args"] = [(ints, max(ms - pair_ms, 5e-4)) for ints, ms in d["args"]]
        d["ms"] = sum(ms for _, ms in d["args"])
        d["avg_ms"] = d["ms"] / max(d["calls"], 1)
    flops = 0.0
    ms = 0.0
    per_kernel = {}
    # kernel classes of the whole-trunk launches, named as rocprofv3 shows them
    # (csrc/mlp_bf16.hip: RT = 1 up to 8192 rows, RT = 4 above)
    classes: dict = {}

    def add(cls, t_ms, work):
        c = classes.setdefault(cls, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        c["launches"] += 1
        c["ms"] += t_ms
        c["flops"] += work[0] or 0.0
        c["bytes"] += work[1] or 0.0

    for name, d in summ.items():
        per_kernel[name] = {"calls": d["calls"], "ms": round(d["ms"], 4)}
        if name in ("mi_mlp_fwd_bf16", "mi_mlp_bwd_dx_bf16"):
            bwd = "true" if name == "mi_mlp_bwd_dx_bf16" else "false"
            for (ints, t_ms), work in zip(d["args"], d["work"]):
                M = ints[1] if name == "mi_mlp_bwd_dx_bf16" else ints[0]
                add(f"mlp_chain_kernel<{1 if M <= 8192 else 4}, {bwd}>", t_ms, work)
        if name in ("mi_policy_fwd_bf16", "mi_policy_bwd_bf16"):
            kern = "policy_kernel" if name == "mi_policy_fwd_bf16" else "policy_bwd_kernel"
            # the instantiation csrc/mlp_bf16.hip picks: <RT, NB> of the action trunk, then of
            # the value trunk — 16 rows x 4 column tiles per wave at rollout sizes; at training
            # sizes 64 x 4, and 256 x 1 for a trunk no wider than 64
            narrow = max(ACTOR_H) <= 64 and os.environ.get("MIPPO_NARROW_TRUNK", "1")[:1] != "0"
            for (ints, t_ms), work in zip(d["args"], d["work"]):
                M = ints[0] if name == "mi_policy_fwd_bf16" else ints[1]  # (offset_add, M, ..)
                shape = "1, 4, 1, 4" if M <= 8192 else ("16, 1, 4, 4" if narrow else "4, 4, 4, 4")
                add(f"{kern}<{shape}>", t_ms, work)
        if name == "mi_policy_ws_fwd_bf16":
            # csrc/trunk_ws.hip: both trunks in ONE launch (policy_ws_dual_kernel<HV, NHV, HA,
            # NHA, RT>): 32-row tiles, one per workgroup, at rollout sizes; 64-row tiles with
            # the CUs split between the trunks at replay sizes
            for (ints, t_ms), work in zip(d["args"], d["work"]):
                rt = 2 if ints[0] <= 8192 else 4
                add(f"policy_ws_dual_kernel<256, 1, 64, 3, {rt}>", t_ms, work)
        if name == "mi_policy_ws_bwd_bf16":
            for (ints, t_ms), work in zip(d["args"], d["work"]):
                add("policy_ws_bwd_dual_kernel<256, 1, 64, 3, 4>", t_ms, work)
        if name == "mi_policy_ws_bwd_gae_bf16":
            # the same launch with the GAE scan, the advantage statistics and the loss
            # gradients inside (no mi_gae_ppo_loss_f32 launch in front of it)
            for (ints, t_ms), work in zip(d["args"], d["work"]):
                add("policy_ws_bwd_gae_kernel<256, 1, 64, 3>", t_ms, work)
        if name == "mi_dense_bwd_dw_grouped_bf16":
            for (ints, t_ms), work in zip(d["args"], d["work"]):
                add("dW group (tn_gemm_dw_all_kernel + reduce_slabs_grouped)",
                    t_ms, work)
        if name == "mi_dense_bwd_dw_grouped_slabs_bf16":
            # csrc/gemm_bf16.hip dw_grouped_launch: at training sizes (M a multiple of 32) the
            # group's tiles go through the DMA-staged 128 x 128 kernel of gemm256_bf16.hip
            # (MIPPO_DW128_DMA=0: the register-staged tile kernel) — same tiles, same slabs
            # (MIPPO_DW128_DMA=2 forces that kernel; by default it takes short splits of groups
            # made mostly of full-width tiles — not the headline's two trunks, 7 of whose 13
            # tiles are at most 64 columns wide)
            forced = os.environ.get("MIPPO_DW128_DMA", "1")[:1] == "2"
            for (ints, t_ms), work in zip(d["args"], d["work"]):
                kern = "tn128_kernel" if forced else "tn_gemm_dw_all_kernel"
                add(f"{kern} (slabs reduced by adam_kernel)", t_ms, work)
        if name in GEMM_SYMBOLS:
            if GEMM_SYMBOLS[name] is None:
                flops += d["flops"]
                ms += d["ms"]
                continue
            for ints, t_ms in d["args"]:
                M, K, N = GEMM_SYMBOLS[name](ints)
                flops += 2.0 * M * K * N
                ms += t_ms
    achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    per_kernel["_note"] = ("HIP events around each C-ABI call of one eager iteration queued "
                           "behind a spin kernel (device time, no host gaps)")
    from nnx_ppo_amd import config as mi_config

    bf16 = mi_config.compute_dtype() == "bf16"
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
    gemm_family = {
        "achieved_tflops": round(achieved, 3), "peak_tflops": peak,
        "frac": round(achieved / peak, 5), "ms_per_iter": round(ms, 3),
        "flop_per_iter": flops,
        "note": "every dense launch (trunk fwd, dX chain, dW): SURVEY 8(d)'s 2*M*K*N count",
    }
    traffic_db, traffic_meta = {}, {}
    prof_dir = Path(__file__).resolve().parent / "profiles"
    # the newest traffic file (rNN_pmc_traffic.json), unless the named one exists
    pmc = prof_dir / PMC_TRAFFIC_FILE
    if not pmc.exists():
        older = sorted(prof_dir.glob("r*_pmc_traffic.json"))
        pmc = older[-1] if older else pmc
    if pmc.exists():
        traffic_meta = json.loads(pmc.read_text())
        traffic_db = traffic_meta.get("kernels", {})
    from nnx_ppo_amd.csrc.build import source_signature

    sig_now = source_signature()
    sig_file = traffic_meta.get("kernel_signature")
    traffic_stale = sig_file != sig_now  # a file without a signature is stale by definition
    trunk = {k: v for k, v in classes.items()
             if k.startswith(("mlp_chain_kernel", "policy_kernel", "policy_bwd_kernel",
                              "trunk_ws_", "policy_ws_", "tn_gemm_dw", "tn128", "dW group"))}
    if trunk:
        # Dominant kernel = the dense class (trunk forward / backward, dW) with the most
        # device time.  Their arithmetic intensity (50-150 flop/B with the activations kept
        # for the backward) is below the bf16 ridge (2500 TF/s / 8 TB/s = 312 flop/B), so HBM
        # bounds them.
        dom = max(trunk, key=lambda k: trunk[k]["ms"])
        c = trunk[dom]
        gbps = c["bytes"] / (c["ms"] * 1e-3) / 1e9
        tf = c["flops"] / (c["ms"] * 1e-3) / 1e12
        # rocprofv3 names carry every template argument; the class names above only the
        # tile shape: match on the common prefix
        stem = dom.split(" pair")[0].split(" (")[0].rstrip(">")
        t = traffic_db.get(dom) or next(
            (v for k, v in traffic_db.items() if k.startswith(stem)), None)
        roof = {
            "bound": "hbm", "kernel": dom, "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS,
            "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 5),
            "traffic": None if t is None else t["hbm_bytes_per_launch"],
            "traffic_source": None if t is None else
            f"profiles/{pmc.name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
            "this command; FETCH_SIZE x2 on gfx950)",
            # the PMC passes are a separate run: this says whether they were taken on the
            # kernels that are being timed now (sha256 over csrc/ + include/)
            "traffic_stale": None if t is None else traffic_stale,
            "traffic_kernel_signature": sig_file, "kernel_signature": sig_now,
            "traffic_git_head": traffic_meta.get("git_head"),
            "launches_per_iter": c["launches"],
            "avg_launch_us": round(c["ms"] / c["launches"] * 1e3, 2),
            "algorithmic_bytes_per_launch": round(c["bytes"] / c["launches"]),
            "arithmetic_intensity_flop_per_byte": round(c["flops"] / max(c["bytes"], 1.0), 1),
            "mfma_tflops": round(tf, 2), "mfma_frac": round(tf / peak, 5),
            "event_pair_overhead_us": round(pair_ms * 1e3, 2),
            "classes": {k: {"launches": v["launches"],
                            "avg_launch_us": round(v["ms"] / v["launches"] * 1e3, 2),
                            "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                            "TFLOPs": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)}
                        for k, v in classes.items()},
            "gemm_family": gemm_family,
        }
    else:  # fp32 path: per-layer MFMA GEMMs only
        dom = max(GEMM_SYMBOLS, key=lambda k: summ.get(k, {"ms": 0})["ms"])
        roof = {
            "bound": "mfma", "kernel": "dense GEMM family (fwd, dX, dW); largest: " + dom,
            "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 5), "traffic": None,
            "event_pair_overhead_us": round(pair_ms * 1e3, 2),
            "gemm_ms_per_iter": round(ms, 3), "gemm_flop_per_iter": flops,
        }
    return ts, roof, per_kernel


def cpu_baseline(min_seconds: float = 12.0, max_iters: int = 12):
    """The CPU oracle (torch-CPU fp32 restatement of the reference's ppo_step,
    autograd through the T-step scan) on the same workload, on ALL of this box's host
    cores.  Bounded sample: 1 warm-up iteration, then full-size iterations until
    `min_seconds` of CPU work have been timed (at most `max_iters`)."""
    from nnx_ppo_amd import random as keys
    from nnx_ppo_amd.envs import cartpole_shaped
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper
    from oracle import networks as on
    from oracle import ppo as op

    cores, how = _usable_cores()
    torch.set_num_threads(cores)
    net = factories.make_mlp_actor_critic(OBS, ACT, ACTOR_H, CRITIC_H, Rngs(SEED))
    onet = on.from_product(net, torch.float32)
    env = EpisodeWrapper(cartpole_shaped(max_steps=1000), 1000)
    ts = op.new_training_state(env, onet, N_ENVS, SEED, keys)
    ts, _ = op.ppo_step(env, ts, N_ENVS, T, 0.95, 0.99, 0.2, True, N_EPOCHS, N_MB, keys)
    times = []
    while sum(times) < min_seconds and len(times) < max_iters:
        t0 = time.perf_counter()
        ts, _ = op.ppo_step(env, ts, N_ENVS, T, 0.95, 0.99, 0.2, True, N_EPOCHS, N_MB, keys)
        times.append(time.perf_counter() - t0)
        _log(f"  cpu iteration {len(times)}: {times[-1]:.2f} s on {cores} threads")
    times.sort()
    med = times[len(times) // 2]
    return {
        "value": round(N_ENVS * T / med, 1), "unit": "env-steps/s", "cores": cores,
        "kind": "port",
        "sample": f"median of {len(times)} timed ppo_step iterations (+1 warm-up, "
                  f"{sum(times):.1f} s of CPU work) of the CPU oracle at the full workload "
                  f"({N_ENVS} envs x {T} steps, {N_EPOCHS}x{N_MB} grad steps), torch-CPU fp32, "
                  f"{cores} threads = every core this process may use ({how})",
    }


_T0 = time.perf_counter()


def _usable_cores():
    """All the host cores this process may actually run on: the scheduler affinity,
    capped by the container's CPU quota (cgroup cpu.max) — asking torch for more threads
    than the quota allows only makes them take turns."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = int(txt[0]) / int(txt[1])
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                    quota = q / period
            break
        except (OSError, ValueError, IndexError):
            continue
    cores = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return cores, f"affinity {aff}, cgroup quota {'none' if quota is None else round(quota, 1)}"


def _log(msg: str) -> None:
    """Progress on stderr (the JSON line on stdout stays alone)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def _pct(xs, q):
    xs = sorted(xs)
    if not xs:
        return float("nan")
    pos = q * (len(xs) - 1)
    lo = int(pos)
    hi = min(lo + 1, len(xs) - 1)
    return xs[lo] + (xs[hi] - xs[lo]) * (pos - lo)


def train_ppo_throughput(device, compute: str, iterations: int = 45):
    """`throughput/train_sps` as the drop-in `train_ppo` reports it (fresh state, same
    workload): median over the iterations after the eager first one and the capture."""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.config import EvalConfig, PPOConfig, TrainConfig
    from nnx_ppo_amd.algorithms.types import LoggingLevel

    env, net, _ = build(device, state=False)
    cfg = TrainConfig(
        ppo=PPOConfig(n_envs=N_ENVS, rollout_length=T, total_steps=N_ENVS * T * iterations,
                      n_epochs=N_EPOCHS, n_minibatches=N_MB, learning_rate=1e-4,
                      logging_level=LoggingLevel.LOSSES | LoggingLevel.THROUGHPUT),
        eval=EvalConfig(enabled=False), seed=SEED, checkpoint_every_steps=0)
    sps = []
    ppo.train_ppo(env, net, cfg, compute_dtype=compute,
                  log_fn=lambda m, s: sps.append(float(m["throughput/train_sps"])))
    steady = sps[5:]
    return {"train_sps_median": round(_pct(steady, 0.5), 1),
            "train_sps_p10": round(_pct(steady, 0.1), 1),
            "train_sps_p90": round(_pct(steady, 0.9), 1), "iterations": len(steady),
            "note": "train_ppo(..., log_fn=...) with LoggingLevel.THROUGHPUT, eval disabled; "
                    "first 5 iterations (eager + capture) dropped"}


# The other single-GPU configs of BASELINE.json (SURVEY 8's C3 and C4), timed for a short
# window each AFTER the headline measurement and reported under "configs" — the headline
# `value` / `config` stay C2's.  FLOPs per iteration: SURVEY 8(d) (forward FLOPs per sample x
# [rollout + n_epochs x (3 x replay + bootstrap)]).
OTHER_CONFIGS = {
    "C3": {"n_envs": 8192, "flop_per_iter": 3.069e12,
           "workload": "BASELINE configs[2]: CheetahRun-shaped dict obs {position 8, velocity "
                       "9}, act 6, MLP actor 4x256 / critic 2x512, normalize_obs, n_envs=8192"},
    "C4": {"n_envs": 4096, "flop_per_iter": 2.973e11,
           "workload": "BASELINE configs[3]: CartpoleBalance-shaped, actor Dense-GRU(64)-Dense "
                       "/ critic 2x256, carry through rollout + replay, max_steps=5 "
                       "(reset-on-done on ~20 % of the steps), n_envs=4096"},
}


def build_config(name: str, n_envs: int, device):
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import cartpole_shaped, cheetah_shaped
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    rngs = Rngs(SEED)
    if name == "C3":
        env = EpisodeWrapper(cheetah_shaped(max_steps=1000), 1000)
        net = factories.make_mlp_actor_critic({"position": 8, "velocity": 9}, 6, [256] * 4,
                                              [512] * 2, rngs)
    elif name == "C4":
        env = EpisodeWrapper(cartpole_shaped(max_steps=5), 1000)
        net = factories.make_gru_actor_critic(5, 1, 64, [256, 256], rngs)
    else:
        raise ValueError(name)
    ts = ppo.new_training_state(env, net, n_envs, SEED, 1e-4, device=device)
    return env, net, ts


def other_configs(device, compute: str, iters: int = 12, windows: int = 3) -> dict:
    """C3 and C4 by the headline's own definition (train_ppo's loop: one graph launch + one
    host sync per iteration), `windows` windows of `iters` iterations each, median window."""
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.algorithms.loop import IterationRunner

    peak = (PEAK_BF16_MFMA_TFLOPS if compute == "bf16" else PEAK_F32_MFMA_TFLOPS) * 1e12
    out = {}
    for name, c in OTHER_CONFIGS.items():
        n = c["n_envs"]
        env, net, ts = build_config(name, n, device)
        runner = IterationRunner(
            lambda st, env=env, n=n: ppo.ppo_step(env, st, n, T, 0.95, 0.99, 0.2, True, False,
                                                  N_EPOCHS, N_MB), ts)

        def window(k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ticket = runner.launch()
            for i in range(k):
                nxt = runner.launch() if i + 1 < k else None
                m = runner.collect(ticket)
                ticket = nxt
            torch.cuda.synchronize()
            return time.perf_counter() - t0, m

        window(3)  # eager, recorded, replayed
        times = []
        for _ in range(windows):
            dt, m = window(iters)
            times.append(dt)
        times.sort()
        dt = times[len(times) // 2]
        tf = c["flop_per_iter"] * iters / dt
        out[name] = {
            "value": round(n * T * iters / dt, 1), "unit": "env-steps/s",
            "ms_per_step": round(dt / iters * 1e3, 3), "n_envs": n, "steps": iters,
            "windows": windows, "mfma_tflops": round(tf / 1e12, 1),
            "mfma_frac": round(tf / peak, 5), "launch_mode": runner.launch_mode,
            "workload": c["workload"],
            "final_losses": {k: float(v) for k, v in m.items() if k.startswith("losses/")},
        }
        _log(f"  {name}: {out[name]['value'] / 1e6:.2f} M env-steps/s, "
             f"{out[name]['ms_per_step']} ms/iteration")
        del runner, ts, env, net
        torch.cuda.empty_cache()
    return out


def _self_launch(args) -> int:
    """`python3 bench.py --gpus N` (N > 1) with no launcher around it: start the N ranks
    here, BEFORE anything in this process touches the GPU (a process that has initialised
    HIP must never be replaced or forked into GPU work).  One child per GPU with the
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment `torch.distributed.run` would
    give it; rank 0's stdout (the ONE JSON line) is relayed, every child's stderr is
    inherited, and the exit code is non-zero if any rank's is.  A rank that fails takes
    the others down (their exact PIDs) instead of leaving them in a collective."""
    import socket
    import subprocess
    import tempfile

    n = args.gpus
    single = os.environ.get("MIPPO_SINGLE_DEVICE") == "1"
    n_dev = torch.cuda.device_count()  # counts devices without initialising the runtime
    if n_dev < n and not single:
        raise SystemExit(
            f"[bench] --gpus {n} but only {n_dev} GPU(s) are visible.  For a rehearsal of "
            "the N > 1 path on fewer GPUs (every rank on cuda:0, rendezvous over gloo) set "
            "MIPPO_SINGLE_DEVICE=1")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL, peer regions)
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    if single:
        base.setdefault("MIPPO_DIST_BACKEND", "gloo")  # RCCL cannot put two ranks on one GPU
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=str(port), MIPPO_SELF_LAUNCHED="1")
    cmd = [sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]]
    out0 = tempfile.TemporaryFile(mode="w+")
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, stdout=out0 if r == 0 else sys.stderr))
    _log(f"self-launched {n} ranks (pids {[p.pid for p in procs]}), rendezvous 127.0.0.1:{port}")
    deadline = time.monotonic() + float(os.environ.get("MIPPO_BENCH_TIMEOUT_S", "1500"))
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
        if bad or time.monotonic() > deadline:
            failed = bad or [("timeout", None)]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.monotonic() + 15
            while any(p.poll() is None for p in procs) and time.monotonic() < t_kill:
                time.sleep(0.2)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    out0.seek(0)
    text = out0.read()
    if failed or any(codes):
        sys.stderr.write(f"[bench] ranks exited with {codes} (first failure: {failed})\n")
        sys.stderr.write(text)
        return next((c for c in codes if c), 1) or 1
    sys.stdout.write(text)
    sys.stdout.flush()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short C3 / C4 windows reported under `configs`")
    ap.add_argument("--no-train-ppo", action="store_true",
                    help="skip the cross-check run of train_ppo itself")
    ap.add_argument("--min-timed-seconds", type=float, default=1.0)
    ap.add_argument("--compute", choices=["f32", "bf16"], default="bf16",
                    help="MFMA path of the Dense layers (BASELINE configs[1] is bf16)")
    ap.add_argument("--eager", action="store_true",
                    help="launch every kernel from Python instead of replaying the captured "
                         "HIP graph of the iteration")
    ap.add_argument("--transport", choices=["auto", "oneshot", "rccl"], default="auto",
                    help="N > 1: how gradients / statistics travel.  oneshot = the one-shot "
                         "peer kernels over IPC-mapped buffers (one HIP graph per iteration); "
                         "rccl = torch.distributed collectives between HIP-graph segments; "
                         "auto = oneshot if its self-check against RCCL passes, else rccl")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: this process only starts the ranks and relays rank 0's line
        raise SystemExit(_self_launch(args))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank "
                         "per GPU (or run `python3 bench.py --gpus N`, which starts them itself)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: MIPPO_DIST_BACKEND=gloo MIPPO_SINGLE_DEVICE=1 runs the
    # N > 1 code path (collectives through gloo, every rank on cuda:0)
    backend = os.environ.get("MIPPO_DIST_BACKEND", "nccl")
    if os.environ.get("MIPPO_SINGLE_DEVICE") == "1":
        local_rank = 0
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    transport = "none"
    oneshot_check = "n/a (one rank)"
    if world > 1:
        import datetime

        import torch.distributed as dist

        # a short timeout: a wedged collective should fail this run, not hang the node
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=180), **kw)
        from nnx_ppo_amd import parallel

        oneshot_check = "not attempted (--transport rccl)"
        if args.transport in ("auto", "oneshot"):
            ok, why = parallel.enable_oneshot(device)
            oneshot_check = "passed (bit-identical to torch.distributed on every rank)" if ok \
                else f"failed: {why}"
            if not ok:
                if args.transport == "oneshot":
                    raise SystemExit(f"[bench] one-shot transport unavailable: {why}")
                if rank == 0:
                    print(f"[bench] one-shot transport not used ({why}); RCCL collectives "
                          "between graph segments", file=sys.stderr)
        transport = parallel.transport()

    from nnx_ppo_amd import config as mi_config
    from nnx_ppo_amd.algorithms.loop import IterationRunner

    mi_config.set_compute_dtype(args.compute)
    env, net, ts = build(device)
    # a capture failure raises (GraphCaptureError): nothing here falls back to eager
    # launches by itself — rerun with --eager for that number
    runner = IterationRunner(lambda st: one_iter(env, st), ts, hip_graph=not args.eager)

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    def run_window(k: int, stamps=None) -> float:
        """EXACTLY k iterations as the training loop issues them; returns wall seconds."""
        barrier()
        t0 = time.perf_counter()
        ticket = runner.launch()
        for i in range(k):
            nxt = runner.launch() if i + 1 < k else None
            m = runner.collect(ticket)  # the iteration's host sync
            if stamps is not None:
                stamps.append(time.perf_counter())
            ticket = nxt
        barrier()
        run_window.metrics = m
        return time.perf_counter() - t0

    # warm-up: iteration 1 is eager, iteration 2 records the graph — at least 3 so the
    # timed region only sees replays
    warm = max(args.warmup, 1 if args.eager else 3)
    _log(f"built; warm-up {warm} iterations (1 eager, 1 recorded, rest replayed)")
    run_window(warm)
    _log("warm-up done")
    # size the repetition count from one untimed window
    est = run_window(args.steps)
    if world > 1:
        # every rank must run the SAME number of windows (each window holds two barriers): the
        # count comes from the slowest rank's estimate, not from each rank's own clock
        import torch.distributed as dist

        et = torch.tensor([est], dtype=torch.float64, device=device)
        if backend != "nccl":
            et = et.cpu()
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        est = float(et.cpu()[0])
    reps = max(1, min(200, int(args.min_timed_seconds / max(est, 1e-6)) + 1))
    _log(f"one window of {args.steps} iterations = {est * 1e3:.1f} ms; timing {reps} windows")
    windows, iter_ms = [], []
    for _ in range(reps):
        stamps = [time.perf_counter()]
        windows.append(run_window(args.steps, stamps))
        # the first interval holds the window's launch of its first iteration; the
        # steady-state intervals are host sync to host sync
        iter_ms += [(b - a) * 1e3 for a, b in zip(stamps[1:-1], stamps[2:])]
    metrics = run_window.metrics
    own_windows = list(windows)  # this rank's own clock, before the MAX over ranks
    wt = torch.tensor(windows, dtype=torch.float64, device=device)
    if world > 1:
        import torch.distributed as dist

        if backend != "nccl":
            wt = wt.cpu()
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
    windows = [float(x) for x in wt.cpu()]
    elapsed = _pct(windows, 0.5)
    # one line per rank: what it ran on, how its exchanges travel, its own clock
    props = torch.cuda.get_device_properties(device)
    comm = None
    if world > 1:
        from nnx_ppo_amd import parallel

        comm = parallel.peer_comm()
    me = {
        "rank": rank, "device": str(device), "gpu": props.name,
        "gcn_arch": getattr(props, "gcnArchName", None),
        "pid": os.getpid(), "world_seen": world if world == 1 else dist.get_world_size(),
        "dist_backend": None if world == 1 else dist.get_backend(),
        "transport": transport, "oneshot_self_check": oneshot_check,
        "oneshot_timeouts": None if comm is None else comm.status()[1],
        "ms_per_step_own_clock": round(_pct(own_windows, 0.5) / args.steps * 1e3, 4),
        "iteration_ms_p50_own_clock": round(_pct(iter_ms, 0.5), 4),
    }
    ranks = [me]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)

    _log(f"timed: median window {elapsed * 1e3:.2f} ms")
    # the same graph replayed back to back, no host read in between (round 1's number)
    b2b = None
    if runner._graph is not None:
        barrier()
        t0 = time.perf_counter()
        with torch.cuda.stream(runner.stream):
            for _ in range(args.steps):
                runner._graph()
        barrier()
        b2b = time.perf_counter() - t0

    # instrumented iteration on every rank (collectives must match), reported by rank 0
    _log("instrumented iteration (HIP events around every C-ABI call)")
    ts, roof, per_kernel = roofline_of_dominant_kernel(env, runner.state)
    _log("roofline done")

    if rank == 0:
        total_env_steps = world * N_ENVS * T * args.steps
        line = {
            "metric": "env-steps/sec (whole node), 4096-env x 30-step PPO rollout+update",
            "value": round(total_env_steps / elapsed, 1),
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": warm,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "per_gpu": {"value": round(total_env_steps / elapsed / world, 1),
                        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                        "note": "weak scaling: every GPU runs the N = 1 workload, so this "
                                "ms_per_step compares directly with the N = 1 line's"},
            "ranks": ranks,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.compute,
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: CartpoleBalance-shaped synthetic env "
                            "(obs 5, act 1), MLP actor 4x64 / critic 2x256, "
                            f"n_envs={N_ENVS}/GPU, rollout_length={T}, {N_EPOCHS} epochs x "
                            f"{N_MB} minibatches, normalize_obs, Adam",
                "n_envs_per_gpu": N_ENVS, "rollout_length": T,
                "global_n_envs": world * N_ENVS, "parallelism": f"env-sharded dp{world}",
                "transport": transport,
            },
            "timing": {
                "definition": "train_ppo's loop: one graph launch + one host sync (metric "
                              "read) per iteration; window = exactly `steps` iterations "
                              "between barrier+synchronize pairs, MAX over ranks; value from "
                              "the median window",
                "windows": len(windows), "timed_s": round(sum(windows), 3),
                "window_ms": {"p10": round(_pct(windows, 0.1) * 1e3, 3),
                              "p50": round(_pct(windows, 0.5) * 1e3, 3),
                              "p90": round(_pct(windows, 0.9) * 1e3, 3)},
                "iteration_ms": {"p10": round(_pct(iter_ms, 0.1), 4),
                                 "p50": round(_pct(iter_ms, 0.5), 4),
                                 "p90": round(_pct(iter_ms, 0.9), 4), "n": len(iter_ms)},
            },
            "back_to_back": None if b2b is None else {
                "value": round(total_env_steps / b2b, 1),
                "ms_per_step": round(b2b / args.steps * 1e3, 3),
                "note": "K graph replays enqueued with no host read, one final synchronise "
                        "(rank 0's clock)"},
            "launch_mode": runner.launch_mode,
            "roofline": roof,
            "kernels_ms_per_iter": per_kernel,
            "final_losses": {k: float(v) for k, v in metrics.items() if k.startswith("losses/")},
        }
        if world == 1 and not args.no_train_ppo and not args.eager:
            _log("train_ppo cross-check")
            line["train_ppo"] = train_ppo_throughput(device, args.compute)
        if world == 1 and not args.no_other_configs and not args.eager:
            _log("other configs (C3, C4): short windows, same timing definition")
            line["configs"] = other_configs(device, args.compute)
        if world == 1 and not args.no_cpu_baseline:
            _log("cpu baseline (oracle on the host cores)")
            line["cpu_baseline"] = cpu_baseline()
        _log("done")
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
