#!/usr/bin/env python
"""Benchmark of the PPO hot path: env-steps/sec on a synthetic 4096-env x 30-step
rollout + update (BASELINE.json configs[1]; per GPU when sharded).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--compute f32|bf16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full `ppo_step` iteration (rollout -> n_epochs x n_minibatches
replay/GAE/loss/Adam steps -> normaliser update) including the reference's one
host sync per iteration (`int(steps_taken)`, nnx_ppo/algorithms/ppo.py:209).
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

CAPTURE_FAILED_RC = 17


def _supervise() -> None:
    """N > 1: every rank's bench runs as a CHILD of this (GPU-free, torch-free)
    process.  The iteration's RCCL all-reduces are captured into the HIP graph with it;
    if that capture is refused, the ranks agree on it and exit with CAPTURE_FAILED_RC,
    and the benchmark is started again with eager launches in a FRESH process — an
    aborted capture leaves HIP streams unusable, so falling back inside the same
    process is not reliable (it segfaulted in rehearsal)."""
    import subprocess

    env = dict(os.environ, MIPPO_BENCH_CHILD="1", MIPPO_BENCH_ATTEMPT="0")
    cmd = [sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]]
    rc = subprocess.call(cmd, env=env)
    if rc == CAPTURE_FAILED_RC and "--eager" not in sys.argv:
        env["MIPPO_BENCH_ATTEMPT"] = "1"
        rc = subprocess.call(cmd + ["--eager"], env=env)
    sys.exit(rc)


if (__name__ == "__main__" and int(os.environ.get("WORLD_SIZE", "1")) > 1
        and os.environ.get("MIPPO_BENCH_CHILD") != "1"):
    _supervise()

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
    "mi_dense_bwd_dw_grouped_bf16": None,
}


def build(device):
    from nnx_ppo_amd.algorithms import ppo
    from nnx_ppo_amd.envs import cartpole_shaped
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper

    env = EpisodeWrapper(cartpole_shaped(max_steps=1000), 1000)
    net = factories.make_mlp_actor_critic(OBS, ACT, ACTOR_H, CRITIC_H, Rngs(SEED),
                                          normalize_obs=True)
    ts = ppo.new_training_state(env, net, N_ENVS, SEED, 1e-4, device=device)
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
        if name == "mi_dense_bwd_dw_grouped_bf16":
            for (ints, t_ms), work in zip(d["args"], d["work"]):
                add("dW group (tn_gemm_dw_all_kernel + reduce_slabs_grouped)",
                    t_ms, work)
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
    traffic_db = {}
    pmc = Path(__file__).resolve().parent / "profiles" / "r01_pmc_traffic.json"
    if pmc.exists():
        traffic_db = json.loads(pmc.read_text()).get("kernels", {})
    trunk = {k: v for k, v in classes.items()
             if k.startswith(("mlp_chain_kernel", "policy_kernel", "policy_bwd_kernel"))}
    if trunk:
        # Dominant kernel = the trunk class with the most device time.  Its arithmetic
        # intensity (~100-150 flop/B with the activations kept for the backward) is below
        # the bf16 ridge (2500 TF/s / 8 TB/s = 312 flop/B), so HBM bounds it.
        dom = max(trunk, key=lambda k: trunk[k]["ms"])
        c = trunk[dom]
        gbps = c["bytes"] / (c["ms"] * 1e-3) / 1e9
        tf = c["flops"] / (c["ms"] * 1e-3) / 1e12
        t = traffic_db.get(dom)
        roof = {
            "bound": "hbm", "kernel": dom, "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS,
            "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 5),
            "traffic": None if t is None else t["hbm_bytes_per_launch"],
            "traffic_source": None if t is None else
            "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
            "this command; FETCH_SIZE x2 on gfx950)",
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


def cpu_baseline(iters: int = 2):
    """The CPU oracle (torch-CPU fp32 restatement of the reference's ppo_step,
    autograd through the T-step scan) on the same workload, on this box's host
    cores.  Bounded sample: 1 warm-up + `iters` timed iterations at full C2 size."""
    from nnx_ppo_amd import random as keys
    from nnx_ppo_amd.envs import cartpole_shaped
    from nnx_ppo_amd.networks import factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.wrappers.episode_wrapper import EpisodeWrapper
    from oracle import networks as on
    from oracle import ppo as op

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    threads = min(cores, 16)
    torch.set_num_threads(threads)
    net = factories.make_mlp_actor_critic(OBS, ACT, ACTOR_H, CRITIC_H, Rngs(SEED))
    onet = on.from_product(net, torch.float32)
    env = EpisodeWrapper(cartpole_shaped(max_steps=1000), 1000)
    ts = op.new_training_state(env, onet, N_ENVS, SEED, keys)
    ts, _ = op.ppo_step(env, ts, N_ENVS, T, 0.95, 0.99, 0.2, True, N_EPOCHS, N_MB, keys)
    t0 = time.perf_counter()
    for _ in range(iters):
        ts, _ = op.ppo_step(env, ts, N_ENVS, T, 0.95, 0.99, 0.2, True, N_EPOCHS, N_MB, keys)
    dt = time.perf_counter() - t0
    return {
        "value": round(N_ENVS * T * iters / dt, 1), "unit": "env-steps/s", "cores": threads,
        "kind": "port",
        "sample": f"{iters} timed ppo_step iterations (+1 warm-up) of the CPU oracle at the "
                  f"full workload ({N_ENVS} envs x {T} steps, {N_EPOCHS}x{N_MB} grad steps), "
                  "torch-CPU fp32",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--compute", choices=["f32", "bf16"], default="bf16",
                    help="MFMA path of the Dense layers (BASELINE configs[1] is bf16)")
    ap.add_argument("--eager", action="store_true",
                    help="launch every kernel from Python instead of replaying the captured "
                         "HIP graph of the iteration")
    ap.add_argument("--capture-collectives", action="store_true",
                    help="N > 1: capture the RCCL all-reduces INTO the iteration's HIP graph "
                         "(one graph launch per iteration) instead of the default sequence of "
                         "graphs with eager collectives between them")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: MIPPO_DIST_BACKEND=gloo MIPPO_SINGLE_DEVICE=1 runs the
    # N > 1 code path (collectives through gloo, every rank on cuda:0, eager launches)
    backend = os.environ.get("MIPPO_DIST_BACKEND", "nccl")
    if os.environ.get("MIPPO_SINGLE_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        import datetime

        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        # a short timeout: a wedged collective should fail this run, not hang the node
        kw = {"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}
        # own key prefix per attempt: a second attempt (see _supervise) must not read the
        # first one's rendezvous keys from the launcher's store
        agent_store = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true"
        store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world,
                              is_master=(rank == 0 and not agent_store),
                              timeout=datetime.timedelta(seconds=180))
        store = dist.PrefixStore("mippo_bench_" + os.environ.get("MIPPO_BENCH_ATTEMPT", "0"),
                                 store)
        dist.init_process_group(backend, store=store, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=180), **kw)
    elif args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from nnx_ppo_amd import config as mi_config

    mi_config.set_compute_dtype(args.compute)
    env, net, ts = build(device)
    graphed = None
    if not args.eager:
        from nnx_ppo_amd.algorithms.graph import GraphedPPOStep, SegmentedPPOStep

        # N > 1: by default a sequence of graphs with the collectives between them (needs
        # nothing from RCCL); --capture-collectives records them into ONE graph
        Recorder = GraphedPPOStep if (world == 1 or args.capture_collectives) else SegmentedPPOStep
        try:
            graphed = Recorder(env, ts, N_ENVS, T, 0.95, 0.99, 0.2, True, False, N_EPOCHS,
                               N_MB, warmup=2)
        except Exception as exc:  # capture unsupported for something in the iteration
            print(f"[bench] rank {rank}: HIP-graph capture failed ({exc!r}); running eager",
                  file=sys.stderr)
            graphed = None
        if world > 1:
            # every rank must take the same path (the collectives of a replayed graph and
            # of eager launches pair up only if all ranks issue them the same way)
            import torch.distributed as dist

            ok = torch.tensor([1 if graphed is not None else 0], dtype=torch.int32)
            ok = ok.to(device) if backend == "nccl" else ok
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                # start over with eager launches in a fresh process (see _supervise)
                sys.stderr.flush()
                os._exit(CAPTURE_FAILED_RC)
    if graphed is None:
        args.eager = True
        ts_box = [ts]

        def run_one():
            ts_box[0], m = one_iter(env, ts_box[0])
            return m
    else:
        ts_box = [graphed.ts]

        def run_one():
            ts_, m = graphed()  # no host read: train_ppo counts the steps on the host
            return m
    for _ in range(args.warmup):
        run_one()

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        metrics = run_one()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist

        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    roof = per_kernel = None
    if rank == 0:
        pass
    # instrumented iteration on every rank (collectives must match), reported by rank 0
    ts, roof, per_kernel = roofline_of_dominant_kernel(env, ts_box[0])

    if rank == 0:
        total_env_steps = world * N_ENVS * T * args.steps
        line = {
            "metric": "env-steps/sec (whole node), 4096-env x 30-step PPO rollout+update",
            "value": round(total_env_steps / elapsed, 1),
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
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
            },
            "launch_mode": "eager" if args.eager else (
                "hip-graph (one hipGraphLaunch per iteration)"
                if (world == 1 or args.capture_collectives) else
                f"{sum(1 for x in graphed.program if isinstance(x, torch.cuda.CUDAGraph))} HIP "
                f"graphs per iteration with "
                f"{sum(1 for x in graphed.program if not isinstance(x, torch.cuda.CUDAGraph))} "
                "eager collectives between them"),
            "roofline": roof,
            "kernels_ms_per_iter": per_kernel,
            "final_losses": {k: float(v) for k, v in metrics.items() if k.startswith("losses/")},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
