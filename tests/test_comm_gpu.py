"""The one-shot peer exchange (`csrc/comm.hip`, `comm.py`) with two
processes sharing the one GPU of the box: the regions are exported and mapped through HIP
IPC exactly as between two GPUs; the kernels, flags, fences and the double buffering are
the ones a node runs.  (xGMI itself needs a multi-GPU node: not covered here.)

Also the equivalence VERDICT r1 asked for: 2 ranks x N/2 envs reproduce the
single-process iteration on the union of the envs — parameters, Adam moments, normaliser
statistics and logged losses."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = Path(__file__).resolve().parent


def _run_pair(tmp_path, scenario, timeout=240, env=None):
    store = tmp_path / f"store_{scenario}"
    procs = [subprocess.Popen([sys.executable, str(HERE / "_comm_worker.py"), str(r), "2",
                               str(store), scenario, str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"RANK {r} OK" in out, (scenario, r, p.returncode, out[-3000:])
    return outs


@pytest.mark.parametrize("scenario", ["collectives", "stress", "graph", "fused_adam", "timeout"])
def test_oneshot_exchange_two_processes(dev, tmp_path, scenario):
    _run_pair(tmp_path, scenario)


def test_sharded_ppo_step_equals_single_process(dev, tmp_path):
    """2 ranks x 64 envs (one-shot exchanges: fused all-reduce + Adam, advantage
    statistics, normaliser Chan merge, loss rows) == 1 process x 128 envs with the union
    minibatches.  Equal up to summation order: 2e-5 on parameters after 8 Adam steps of
    lr 1e-3, 1e-5 relative on statistics and losses; the two ranks bit-identical."""
    sys.path.insert(0, str(HERE))
    from _sharded_case import N, T, build_state, run_iterations

    _run_pair(tmp_path, "sharded_ppo")
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    env, ts, inds = build_state(dev, 0, 1)
    ref = run_iterations(env, ts, inds)
    # replicas: bit-identical parameters, moments and normaliser statistics
    for k in ("params", "adam_m", "norm_mean", "norm_m2"):
        assert torch.equal(r0[k], r1[k]), k
    assert r0["step"] == r1["step"] == ref["step"] == 8
    assert r0["norm_count"] == r1["norm_count"] == ref["norm_count"] == 2.0 * N * T
    assert r0["steps_taken"] == ref["steps_taken"] == 2 * N * T  # whole-job steps
    # the shards' event streams are the halves of the single process's
    assert torch.equal(torch.cat([r0["obs"], r1["obs"]]), ref["obs"].cpu())
    cpu = lambda t: t.cpu().double().numpy()
    assert np.allclose(cpu(r0["params"]), cpu(ref["params"]), rtol=0, atol=2e-5)
    assert np.allclose(cpu(r0["adam_m"]), cpu(ref["adam_m"]), rtol=1e-3, atol=1e-6)
    assert np.allclose(cpu(r0["norm_mean"]), cpu(ref["norm_mean"]), rtol=1e-5, atol=1e-6)
    assert np.allclose(cpu(r0["norm_m2"]), cpu(ref["norm_m2"]), rtol=1e-5, atol=1e-4)
    for it in range(2):
        for k, want in ref["metrics"][it].items():
            if not k.startswith("losses/"):
                continue  # rollout statistics (loglikelihood/...) are per shard by nature
            got0, got1 = r0["metrics"][it][k], r1["metrics"][it][k]
            assert got0 == got1, k                       # logged values agree across ranks
            assert np.isclose(got0, want, rtol=2e-4, atol=2e-6), (it, k, got0, want)
    # and it was not a trivial run
    assert not torch.equal(ref["params"].cpu(), build_state(dev, 0, 1)[1].optimizer.params.cpu())


def test_sharded_gradient_step_is_the_four_launches_of_a_single_gpu(dev, tmp_path):
    """on the one-shot transport a sharded rank's gradient step is forward,
    backward WITH the GAE scan / global advantage statistics / loss inside
    (`mi_policy_ws_bwd_gae_bf16` exchanging the statistics partials through the peers' regions),
    dW, and `mi_adam_step_allreduce_f32` summing the dW slabs — no GAE / statistics-exchange /
    loss / slab-reduction launches of their own.  2 ranks x 1024 envs (C2's network, bf16)
    against 1 process x 2048 envs with the union minibatches: parameters within the bf16
    path's order bound, the ranks bit-identical.  (Two ranks share the one GPU of the box: the
    kernels and the protocol are a node's; xGMI itself is not covered.)"""
    import os

    sys.path.insert(0, str(HERE))
    env_vars = dict(os.environ, MIPPO_SHARDED_CASE="big")
    _run_pair(tmp_path, "sharded_ppo", timeout=400, env=env_vars)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=True)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=True)
    # the single-process reference in a child too (the case module reads its size at import)
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "from _sharded_case import build_state, run_iterations;"
            "dev = torch.device('cuda', 0); env, ts, inds = build_state(dev, 0, 1);"
            "r = run_iterations(env, ts, inds); torch.cuda.synchronize();"
            "torch.save({k: (v.cpu() if isinstance(v, torch.Tensor) else v) "
            "for k, v in r.items()}, %r)"
            % (str(HERE.parent), str(HERE), str(tmp_path / "ref.pt")))
    p = subprocess.run([sys.executable, "-c", code], env=env_vars, capture_output=True,
                       text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-3000:]
    ref = torch.load(tmp_path / "ref.pt", weights_only=True)
    per_step = {"mi_policy_ws_fwd_bf16", "mi_policy_ws_bwd_gae_bf16",
                "mi_dense_bwd_dw_grouped_slabs_bf16"}
    for r in (r0, r1):
        used = set(r["used"])
        assert per_step | {"mi_adam_step_allreduce_f32"} <= used, sorted(used)
        # none of the launches the sharded step used to need
        assert not used & {"mi_gae_stats_f32", "mi_gae_f32", "mi_ppo_loss_f32",
                           "mi_allreduce_oneshot_f64", "mi_reduce_slabs_grouped_f32",
                           "mi_policy_ws_bwd_bf16", "mi_adam_step_f32",
                           "mi_adam_step_slabs_f32"}, sorted(used)
    assert per_step | {"mi_adam_step_slabs_f32"} <= set(ref["used"])
    for k in ("params", "adam_m", "norm_mean", "norm_m2"):
        assert torch.equal(r0[k], r1[k]), k              # replicas bit-identical
    assert r0["step"] == r1["step"] == ref["step"]
    assert torch.equal(torch.cat([r0["obs"], r1["obs"]]), ref["obs"])
    cpu = lambda t: t.double().numpy()
    # bf16 images of gradients that differ in fp64 summation order of the statistics: a few
    # roundings flip; Adam moves a parameter by <= lr = 1e-3 per step whatever the gradient
    worst = float(np.abs(cpu(r0["params"]) - cpu(ref["params"])).max())
    assert worst < 5e-4, worst
    assert np.allclose(cpu(r0["norm_mean"]), cpu(ref["norm_mean"]), rtol=1e-5, atol=1e-6)
    for it in range(2):
        for k, want in ref["metrics"][it].items():
            if not k.startswith("losses/"):
                continue
            got0, got1 = r0["metrics"][it][k], r1["metrics"][it][k]
            assert got0 == got1, k
            assert np.isclose(got0, want, rtol=5e-3, atol=2e-5), (it, k, got0, want)
