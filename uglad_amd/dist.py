"""Collectives of the sharded GLAD path (one process per GPU, RCCL over xGMI through torch.distributed's "nccl" backend).

The reference is single-process; sharding the batch of matrices over ranks creates exactly three exchange points
(SURVEY.md section 8e):
  (i)   per unroll step: SUM of one fp32 -- the batch-wide ||Z - theta_half||_F^2 behind get_frobenius_norm (glad.py:60-71,147);
  (ii)  per epoch: SUM of the 42 parameter gradients (+ the loss), after loss.backward() (main.py:408/504/624/772);
  (iii) once per missing-data fit: MIN of |Theta| and SUM of sign(Theta) (main.py:703,710).
All three are tiny and latency-bound, so they ride on plain all_reduce calls.  The object is injectable so that the
sharded == unsharded property can be asserted without a cluster (tests use gloo with world_size 2, and an in-process fake).
"""
from __future__ import annotations

from typing import Optional

import torch


class Collective:
    """World of size 1: every exchange is the identity."""

    world_size = 1
    rank = 0

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        return t

    def all_reduce_min(self, t: torch.Tensor) -> torch.Tensor:
        return t

    def all_gather_cat(self, t: torch.Tensor) -> torch.Tensor:
        return t

    def shard(self, n: int):
        """Contiguous slice [lo, hi) of n items owned by this rank (every rank gets n // world_size, remainder to the first)."""
        w, r = self.world_size, self.rank
        base, rem = divmod(n, w)
        lo = r * base + min(r, rem)
        return lo, lo + base + (1 if r < rem else 0)


class TorchCollective(Collective):
    def __init__(self, group=None):
        import torch.distributed as dist

        self._dist = dist
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._rccl = None  # (function pointer, communicator) of the native per-step exchange, made on first use

    def native_exchange(self):
        """Exchange (i) without Python in the loop: an RCCL communicator of the library's own (uglad_rccl_comm_init: the unique id is made
        on rank 0 and broadcast through this process group), whose ncclAllReduce the C pass issues itself on the compute stream
        (uglad_glad_forward_sharded).  None when the group is not on GPUs (gloo on the CPU: the tests' rehearsals) or RCCL is missing;
        UGLAD_NATIVE_EXCHANGE=0 in the environment keeps the per-step torch.distributed calls (A/B)."""
        import os

        if self._rccl is not None:
            return self._rccl or None
        self._rccl = False
        if os.environ.get("UGLAD_NATIVE_EXCHANGE", "1") == "0" or self._dist.get_backend(self.group) != "nccl":
            return None
        from . import _lib

        lib = _lib.get_lib()
        try:
            uid = [lib.rccl_unique_id() if self.rank == 0 else None]
            self._dist.broadcast_object_list(uid, src=self._dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                             group=self.group)
            comm = lib.rccl_comm_init(uid[0], self.world_size, self.rank)
        except _lib.UgladError:
            return None
        self._comm = comm
        self._rccl = lib.rccl_exchange(comm)
        return self._rccl

    def all_reduce_sum(self, t):
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t

    def all_reduce_min(self, t):
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN, group=self.group)
        return t

    def all_gather_cat(self, t):
        sizes = [torch.zeros(1, dtype=torch.int64, device=t.device) for _ in range(self.world_size)]
        self._dist.all_gather(sizes, torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device), group=self.group)
        sizes = [int(s.item()) for s in sizes]
        mx = max(sizes)
        pad = t
        if t.shape[0] < mx:
            pad = torch.cat([t, t.new_zeros((mx - t.shape[0],) + tuple(t.shape[1:]))])
        outs = [torch.empty_like(pad) for _ in range(self.world_size)]
        self._dist.all_gather(outs, pad.contiguous(), group=self.group)
        return torch.cat([o[:n] for o, n in zip(outs, sizes)])


_override: Optional[Collective] = None


def set_collective(c: Optional[Collective]) -> None:
    """Install (or clear, with None) a process-wide collective; used by tests to inject a fake."""
    global _override
    _override = c


def get_collective() -> Collective:
    if _override is not None:
        return _override
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return TorchCollective()
    except Exception:
        pass
    return Collective()
