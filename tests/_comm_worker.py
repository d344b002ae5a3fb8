"""One rank of the two-process tests of the one-shot peer exchange (tests/test_comm_gpu.py).
Both ranks share cuda:0 (a one-GPU box): the regions travel through HIP IPC exactly as
they do between two GPUs of a node; what a one-GPU box cannot show is xGMI itself.

    python tests/_comm_worker.py RANK WORLD STORE_FILE SCENARIO OUT_DIR
"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import datetime  # noqa: E402

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, store_file, scenario, out_dir = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3],
                                                  sys.argv[4], Path(sys.argv[5]))
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    store = dist.FileStore(store_file, world)
    dist.init_process_group("gloo", store=store, rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=120))
    from nnx_ppo_amd import parallel

    timeout = 2.0 if scenario == "timeout" else 20.0
    ok, why = parallel.enable_oneshot(dev, slot_bytes=1 << 20, timeout_s=timeout)
    if not ok:
        print(f"ENABLE FAILED: {why}", flush=True)
        sys.exit(5)
    assert parallel.transport() == "oneshot"
    comm = parallel.peer_comm()
    globals()["scenario_" + scenario](rank, world, dev, comm, out_dir)
    torch.cuda.synchronize()
    if scenario != "timeout":
        comm.check()
    dist.barrier()
    print(f"RANK {rank} OK", flush=True)
    dist.destroy_process_group()


def _rank_order_sum(parts):
    s = parts[0].clone()
    for p in parts[1:]:
        s = s + p
    return s


def scenario_collectives(rank, world, dev, comm, out_dir):
    g = torch.Generator().manual_seed(100 + rank)
    for dtype in (torch.float32, torch.float64):
        for n in (1, 3, 255, 1024, 1025, 80_640, 262_144 + 7, 700_001):  # > slot: pieces
            x = torch.randn(n, generator=g, dtype=dtype).to(dev)
            # a tensor larger than one slot goes through in pieces (all-reduce only)
            parts = comm.allgather(x) if n * x.element_size() <= comm.slot_bytes else None
            y = comm.allreduce_(x.clone(), 0.5)
            if parts is not None:
                want = _rank_order_sum(list(parts.unbind(0))) * 0.5
                assert torch.equal(y, want), (dtype, n)      # rank-order sum, bit for bit
                assert torch.equal(parts[rank], x)
            s = y.double().sum().reshape(1)
            both = comm.allgather(s)
            torch.cuda.synchronize()
            assert bool((both == both[0]).all()), (dtype, n)  # identical on every rank
    # other dtypes / shapes through allgather (bytes)
    for t in (torch.arange(7, dtype=torch.int64, device=dev) * (rank + 1),
              torch.full((3, 5), rank + 1, dtype=torch.uint8, device=dev),
              torch.randn(3, 4, 5, device=dev)):
        parts = comm.allgather(t)
        assert parts.shape == (world, *t.shape) and torch.equal(parts[rank], t)
    p = comm.allgather(torch.tensor([rank], device=dev))
    assert p.flatten().tolist() == list(range(world))


def scenario_stress(rank, world, dev, comm, out_dir):
    """Hundreds of collectives back to back with the ranks deliberately out of step: the
    double-buffered slots and per-chunk flags must never hand over stale or torn data."""
    g = torch.Generator().manual_seed(7 + rank)
    n = 50_000
    x = torch.empty(n, device=dev)
    tri = world * (world + 1) / 2
    for k in range(300):
        if int(torch.randint(0, 4, (1,), generator=g)) == 0:
            torch.cuda._sleep(int(torch.randint(1, 2_000_000, (1,), generator=g)))
        x.fill_(float((rank + 1) * (k % 97 + 1)))
        comm.allreduce_(x, 1.0)
        if k % 10 == 0:
            assert float(x.min()) == float(x.max()) == tri * (k % 97 + 1), k
    st = torch.tensor([float(rank + 1), 2.0, 3.0], dtype=torch.float64, device=dev)
    for k in range(200):  # tiny messages, as the advantage statistics are
        y = comm.allreduce_(st.clone(), 1.0)
    assert y.tolist() == [tri, 2.0 * world, 3.0 * world]


def scenario_graph(rank, world, dev, comm, out_dir):
    """The collectives are plain launches: capture three of them in a HIP graph and replay
    it with changing inputs."""
    a = torch.zeros(80_640, device=dev)
    b = torch.zeros(3, dtype=torch.float64, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        comm.allreduce_(a, 1.0)  # warm-up outside the capture
        torch.cuda.synchronize()
        gph = torch.cuda.CUDAGraph()
        gph.capture_begin(capture_error_mode="thread_local")
        comm.allreduce_(a, 1.0 / world)
        comm.allreduce_(b, 1.0)
        parts = comm.allgather(b)
        gph.capture_end()
        for k in range(25):
            a.fill_(float(rank + k))
            b.fill_(float(rank * 10 + k))
            gph.replay()
            torch.cuda.synchronize()
            mean = sum(r + k for r in range(world)) / world
            tot = float(sum(r * 10 + k for r in range(world)))
            assert float(a[0]) == float(a[-1]) == mean, (k, float(a[0]), mean)
            assert b.tolist() == [tot] * 3
            assert parts.tolist() == [[tot] * 3] * world  # gathered AFTER the reduce


def scenario_fused_adam(rank, world, dev, comm, out_dir):
    """mi_adam_step_allreduce_f32 == all-reduce-mean, then mi_adam_step_f32, bit for bit —
    on a bare arena and through Optimizer.update on a network with bf16 images."""
    from nnx_ppo_amd import config, ops, parallel
    from nnx_ppo_amd.networks import dense_chain, factories
    from nnx_ppo_amd.networks.types import Rngs
    from nnx_ppo_amd.optim import Optimizer

    g = torch.Generator().manual_seed(3)
    n = 80_640
    base = {k: torch.randn(n, generator=g).to(dev) for k in ("p", "m")}
    base["v"] = torch.rand(n, generator=g).to(dev)
    grads = torch.randn(n, generator=torch.Generator().manual_seed(50 + rank)).to(dev)
    out = []
    for fused in (True, False):
        p, m, v = base["p"].clone(), base["m"].clone(), base["v"].clone()
        gr = grads.clone()
        step = torch.full((1,), 4, dtype=torch.int64, device=dev)
        kw = dict(lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, weight_decay=1e-4)
        if fused:
            assert comm.adam_step_allreduce(p, gr, m, v, step, shadows=[], **kw)
        else:
            comm.allreduce_(gr, 1.0 / world)
            ops.adam_step(p, gr, m, v, step, begin_next=True, **kw)
        torch.cuda.synchronize()
        assert int(step) == 5 and float(gr.abs().sum()) == 0.0
        out.append((p, m, v))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    # through the optimiser, with the bf16 images of a Dense trunk refreshed by the launch
    with config.use_compute_dtype("bf16"):
        res = []
        for fused in (True, False):
            net = factories.make_mlp([5, 64, 33, 7], Rngs(4), activation_last_layer=False)
            net.to(dev)
            opt = Optimizer(net, 1e-2, device=dev)
            dense_chain.refresh(net.layers)
            opt.begin()
            opt.grads.copy_(torch.randn(opt.n, generator=torch.Generator().manual_seed(9 + rank)))
            if fused:
                opt.update()
            else:
                norm = torch.zeros(1, device=dev)
                opt.update(norm_out=norm)          # logging the norm takes the unfused path
            torch.cuda.synchronize()
            res.append([opt.params.clone()] + [t.clone() for l in net.layers
                                               for t in (l._w_bf, l._wt_bf, l._ff, l._fb)])
        for a, b in zip(*res):
            assert torch.equal(a, b)
    both = comm.allgather(res[0][0])
    assert torch.equal(both[0], both[1])              # replicas stay bit-identical


def scenario_timeout(rank, world, dev, comm, out_dir):
    """A peer that never arrives: the kernel's waits are bounded, it finishes and the
    host is told — the GPU is not left spinning."""
    import time

    x = torch.ones(1000, device=dev)
    comm.allreduce_(x, 1.0)           # one good collective
    torch.cuda.synchronize()
    seq0, err0 = comm.status()
    assert err0 == 0
    if rank == 0:
        t0 = time.perf_counter()
        comm.allreduce_(x, 1.0)       # rank 1 never issues this one
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        seq, err = comm.status()
        assert err >= 1 and seq == seq0 + 1, (seq0, seq, err)
        assert 1.5 < dt < 15.0, dt
        try:
            comm.check()
        except Exception as exc:  # noqa: BLE001
            assert "timed out" in str(exc)
        else:
            raise AssertionError("check() did not raise")
        # never a partial sum: the chunk whose peer did not arrive comes back as NaN, and
        # the count is mirrored into the word the training loop reads every iteration
        assert bool(torch.isnan(x).all()), x[:4]
        assert int(comm.error_word.item()) == err
        from nnx_ppo_amd import ops
        assert any(w.data_ptr() == comm.error_word.data_ptr() for w in ops.health_words(dev))
        # sticky: the fused optimiser applies no update once a peer has been lost
        n = 5000
        p, m, v = (torch.full((n,), c, device=dev) for c in (1.0, 0.5, 0.25))
        g = torch.ones(n, device=dev)
        step = torch.full((1,), 4, dtype=torch.int64, device=dev)
        assert comm.adam_step_allreduce(p, g, m, v, step, lr=1e-2, b1=0.9, b2=0.999, eps=1e-8,
                                        weight_decay=0.0, shadows=[])
        torch.cuda.synchronize()
        assert float(p.min()) == float(p.max()) == 1.0
        assert float(m.min()) == float(m.max()) == 0.5
        assert float(v.min()) == float(v.max()) == 0.25
    else:
        time.sleep(9.0)


def scenario_sharded_ppo(rank, world, dev, comm, out_dir):
    """One sharded ppo_step (N / world envs per rank, gradients + advantage statistics +
    normaliser statistics + loss rows exchanged by the one-shot kernels) — the results go
    to OUT_DIR for the parent to compare with its single-process run on all N envs."""
    from _sharded_case import build_state, run_iterations

    env, ts, inds = build_state(dev, rank, world)
    from nnx_ppo_amd import parallel

    assert parallel.transport() == "oneshot"
    result = run_iterations(env, ts, inds)
    torch.cuda.synchronize()
    torch.save({k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in result.items()},
               out_dir / f"rank{rank}.pt")


if __name__ == "__main__":
    main()
