"""Host side of the one-shot peer exchange (`csrc/comm.hip`, `include/mippo.h` section e).

`PeerComm` owns this rank's IPC-exported device region, swaps the 64-byte handles with
the peers over whatever `torch.distributed` backend is up (RCCL on a GPU node, gloo in a
rehearsal) and maps their regions.  After that a collective is one kernel launch on
torch's current stream — no library call, nothing the host waits for — so it can live
inside a captured HIP graph.  The reference has no collective (it is single-device,
SURVEY §2); what is exchanged and why is SURVEY §8e / DESIGN §7.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch
import torch.distributed as dist

from . import ops
from ._lib import MippoError, check, lib, ptr, stream

DEFAULT_SLOT_BYTES = 8 << 20   # per (parity, rank) slot; a larger tensor goes in pieces
DEFAULT_TIMEOUT_S = float(os.environ.get("MIPPO_COMM_TIMEOUT_S", "20"))


class PeerComm:
    def __init__(self, device, slot_bytes: int = DEFAULT_SLOT_BYTES,
                 timeout_s: float = DEFAULT_TIMEOUT_S, group=None):
        if not (dist.is_available() and dist.is_initialized()):
            raise MippoError("PeerComm needs an initialised torch.distributed process group "
                             "(it carries the IPC handles; the data path does not use it)")
        self.device = torch.device(device)
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.slot_bytes = int(slot_bytes) // 4096 * 4096
        self._h = ctypes.c_void_p()
        nb = int(lib().mi_comm_handle_bytes())
        mine = ctypes.create_string_buffer(nb)
        with torch.cuda.device(self.device):
            check(lib().mi_comm_create(self.rank, self.world, self.slot_bytes, float(timeout_s),
                                       ctypes.byref(self._h), mine), "mi_comm_create")
            handles: list = [None] * self.world
            dist.all_gather_object(handles, bytes(mine.raw), group=group)
            blob = b"".join(handles)
            if len(blob) != nb * self.world:
                raise MippoError("PeerComm: handle exchange returned the wrong size")
            check(lib().mi_comm_connect(self._h, ctypes.create_string_buffer(blob, len(blob))),
                  "mi_comm_connect")
            # the sticky timeout count, mirrored into a word the training loop copies to the
            # host with every iteration's metrics (ops.health_words)
            self.error_word = torch.zeros(1, dtype=torch.int32, device=self.device)
            check(lib().mi_comm_set_error_word(self._h, ptr(self.error_word)),
                  "mi_comm_set_error_word")
        dist.barrier(group=group)  # every rank has mapped every peer before anyone launches

    # ---- collectives (one launch each; capturable) -------------------------------------
    def allreduce_(self, t: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
        """In place: t <- scale * sum over ranks (rank order; bit-identical everywhere)."""
        if not (t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.float64)):
            raise MippoError("PeerComm.allreduce_: contiguous fp32 / fp64 GPU tensor expected")
        flat = t.view(-1)
        per = self.slot_bytes // t.element_size()
        fn = (lib().mi_allreduce_oneshot_f32 if t.dtype == torch.float32
              else lib().mi_allreduce_oneshot_f64)
        for o in range(0, flat.numel(), per):
            piece = flat[o:o + per]
            check(fn(self._h, ptr(piece), piece.numel(), float(scale), stream()),
                  "mi_allreduce_oneshot")
        return t

    def allgather(self, t: torch.Tensor) -> torch.Tensor:
        """[world, *t.shape]: every rank's `t`, in rank order."""
        if not (t.is_cuda and t.is_contiguous()):
            raise MippoError("PeerComm.allgather: contiguous GPU tensor expected")
        nbytes = t.numel() * t.element_size()
        if nbytes > self.slot_bytes:
            raise MippoError(f"PeerComm.allgather: {nbytes} bytes exceed the slot")
        out = torch.empty((self.world, *t.shape), dtype=t.dtype, device=t.device)
        check(lib().mi_allgather_oneshot(self._h, ptr(t), nbytes, ptr(out), stream()),
              "mi_allgather_oneshot")
        return out

    def adam_step_allreduce(self, params, grads, m, v, step, *, lr, b1, b2, eps, weight_decay,
                            shadows, slabs=None) -> bool:
        """`ops.adam_step(..., begin_next=True)` with the gradient all-reduce-mean inside
        the launch.  `slabs`: pending dW slabs in `ops.adam_step`'s form — summed into this
        rank's gradient before it is pushed (no reduction launch in front of the exchange).
        False (nothing launched) if the arena does not fit one slot."""
        n = params.numel()
        if n * 4 > self.slot_bytes:
            return False
        ticket = ops.workspace(params.device, "adam_ticket", 16, zeroed=True)
        sh = list(shadows or [])[:16]
        ns = len(sh)
        I = ctypes.c_int64 * max(ns, 1)
        P = ctypes.c_void_p * max(ns, 1)
        col = lambda j: [t[j] for t in sh]
        bf = torch.bfloat16
        check(lib().mi_adam_step_allreduce_f32(
            self._h, ptr(params, torch.float32), ptr(grads, torch.float32),
            ptr(m, torch.float32), ptr(v, torch.float32), n, float(lr), float(b1), float(b2),
            float(eps), float(weight_decay), ptr(step, torch.int64), ptr(ticket), ns,
            I(*col(0)) if ns else None, I(*col(1)) if ns else None, I(*col(2)) if ns else None,
            P(*[ptr(t, bf) for t in col(3)]) if ns else None,
            P(*[ptr(t, bf) for t in col(4)]) if ns else None,
            P(*[ptr(t, bf) for t in col(5)]) if ns else None,
            P(*[ptr(t, bf) for t in col(6)]) if ns else None, *self._slab_args(slabs), stream()),
            "mi_adam_step_allreduce_f32")
        return True

    @staticmethod
    def _slab_args(slabs):
        if slabs is None:
            return (0, None, None, None, None, None, None, None)
        sp, S, Ks, Ns, gw_off, gb_off = slabs[:6]
        b_lo = slabs[6] if len(slabs) > 6 else None
        nl = len(Ks)
        Pl = ctypes.c_void_p * nl
        Il = ctypes.c_int64 * nl
        return (nl, Pl(*sp), Il(*S), Il(*Ks), Il(*Ns), Il(*gw_off), Il(*gb_off),
                Il(*b_lo) if b_lo is not None and any(b_lo) else None)

    # ---- host-side checks (synchronise) --------------------------------------------------
    def status(self) -> tuple[int, int]:
        """(collectives completed on this rank, waits that timed out) — synchronises."""
        seq, err = ctypes.c_int64(), ctypes.c_int64()
        check(lib().mi_comm_status(self._h, ctypes.byref(seq), ctypes.byref(err)),
              "mi_comm_status")
        return int(seq.value), int(err.value)

    def check(self) -> None:
        seq, err = self.status()
        if err:
            raise MippoError(f"one-shot peer exchange: {err} wait(s) timed out after "
                             f"{seq} collectives on rank {self.rank} (a peer did not arrive)")

    def close(self) -> None:
        if self._h:
            lib().mi_comm_set_error_word(self._h, None)
            lib().mi_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):  # best effort
        try:
            self.close()
        except Exception:
            pass


def self_check(comm: PeerComm, device, backend_moves_gpu_tensors: bool) -> Optional[str]:
    """Compare the peer path with torch.distributed on seeded data.  Returns None if every
    rank agrees that it is correct, else the reason.  Collective: all ranks call it."""
    why = None
    try:
        g = torch.Generator(device="cpu").manual_seed(1234 + comm.rank)
        for n in (1, 1000, 80_640, 300_001):
            x = torch.randn(n, generator=g, dtype=torch.float32).to(device)
            ref = x.clone()
            if backend_moves_gpu_tensors:
                dist.all_reduce(ref, group=comm.group)
            else:
                r = ref.cpu()
                dist.all_reduce(r, group=comm.group)
                ref = r.to(device)
            y = comm.allreduce_(x.clone(), 1.0)
            torch.cuda.synchronize(device)
            if not torch.allclose(y, ref, rtol=1e-5, atol=1e-5):
                why = f"all-reduce of {n} floats differs from torch.distributed"
                break
            # bit-identical on every rank: compare a checksum
            s = y.view(torch.int32).to(torch.int64).sum().reshape(1)
            parts = comm.allgather(s)
            torch.cuda.synchronize(device)
            if not bool((parts == parts[0]).all()):
                why = f"all-reduce of {n} floats is not bit-identical across ranks"
                break
        _, err = comm.status()
        if err and why is None:
            why = f"{err} wait(s) timed out"
    except Exception as exc:  # noqa: BLE001 - reported as the reason
        why = f"{type(exc).__name__}: {exc}"
    flag = torch.tensor([0 if why is None else 1], dtype=torch.int32)
    if backend_moves_gpu_tensors:
        flag = flag.to(device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=comm.group)
    if int(flag.item()) and why is None:
        why = "a peer rank reported a failure"
    return why
