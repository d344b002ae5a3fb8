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


def _run_pair(tmp_path, scenario, timeout=240):
    store = tmp_path / f"store_{scenario}"
    procs = [subprocess.Popen([sys.executable, str(HERE / "_comm_worker.py"), str(r), "2",
                               str(store), scenario, str(tmp_path)],
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
