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

import atexit
import logging
import os
from typing import Optional

import torch

_log = logging.getLogger("uglad_amd.dist")


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


# One RCCL communicator of the library's own per process group, made on first use and destroyed at interpreter exit (or by
# release_native_exchanges(), e.g. before destroy_process_group): get_collective() builds a fresh TorchCollective per fit(), and a
# communicator per fit() would leak its device buffers and proxy threads.
_native_cache: dict = {}


def release_native_exchanges() -> None:
    """ncclCommDestroy on every cached communicator (idempotent)."""
    from . import _lib

    for key, rec in list(_native_cache.items()):
        comm = rec.get("comm")
        if comm:
            try:
                _lib.get_lib().rccl_comm_destroy(comm)
            except Exception as exc:  # noqa: BLE001 -- at exit there is nobody to raise to
                _log.warning("uglad_amd.dist: ncclCommDestroy failed: %s", exc)
        _native_cache.pop(key, None)


atexit.register(release_native_exchanges)


class TorchCollective(Collective):
    def __init__(self, group=None):
        import torch.distributed as dist

        self._dist = dist
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def native_status(self) -> dict:
        """What native_exchange() decided for this group: {"native": bool, "nranks": ncclCommCount or None, "fallback_reason": str or None}
        (bench.py puts it into its JSON line, so that a multi-GPU record shows whether RCCL was driven from C and how many ranks it saw)."""
        rec = _native_cache.get(self._key())
        if rec is None:
            return {"native": False, "nranks": None, "fallback_reason": "native_exchange() not called"}
        return {k: rec.get(k) for k in ("native", "nranks", "fallback_reason")}

    def _key(self):
        return id(self.group) if self.group is not None else "WORLD"

    def native_exchange(self):
        """Exchange (i) without Python in the loop: an RCCL communicator of the library's own (uglad_rccl_comm_init: the unique id is made
        on rank 0 and broadcast through this process group), whose ncclAllReduce the C pass issues itself on the compute stream
        (uglad_glad_forward_sharded).  None -- and the per-step torch.distributed.all_reduce, which is RCCL too -- unless
          * UGLAD_NATIVE_EXCHANGE=1 is set: OFF by default, because RCCL between GPUs has never executed this path (no multi-GPU node was
            available to any round; one rank on one GPU and two gloo ranks are what ran), and a rank that fails inside a pass would leave
            the others waiting in ncclAllReduce;
          * the group's backend is nccl (gloo on the CPU: the tests' rehearsals);
          * EVERY rank of the group got its communicator: the broadcast always runs (rank 0 sends a failure marker when it has no id), then
            an all_reduce(MIN) of an ok flag decides for the whole group -- no rank decides alone, so no rank waits in a collective the
            others have skipped.  A rank that did get a communicator while another failed destroys it again.
        The decision is cached per process group, logged (never swallowed) and reported by native_status()."""
        key = self._key()
        rec = _native_cache.get(key)
        if rec is not None:
            return rec["exchange"]
        rec = {"native": False, "nranks": None, "fallback_reason": None, "exchange": None, "comm": None}
        _native_cache[key] = rec
        if os.environ.get("UGLAD_NATIVE_EXCHANGE", "0") != "1":
            rec["fallback_reason"] = "UGLAD_NATIVE_EXCHANGE != 1 (default: per-step torch.distributed.all_reduce; see uglad_amd/dist.py)"
            return None
        if self._dist.get_backend(self.group) != "nccl":
            rec["fallback_reason"] = f"process group backend is {self._dist.get_backend(self.group)}, not nccl"
            return None
        from . import _lib

        lib = _lib.get_lib()
        reason = None
        uid = [None]
        if self.rank == 0:
            try:
                uid = [lib.rccl_unique_id()]
            except _lib.UgladError as exc:
                uid, reason = [b""], f"rank 0: {exc}"  # the marker: everybody still takes part in the broadcast
        src = self._dist.get_global_rank(self.group, 0) if self.group is not None else 0
        self._dist.broadcast_object_list(uid, src=src, group=self.group)
        comm = None
        if uid[0]:
            try:
                comm = lib.rccl_comm_init(uid[0], self.world_size, self.rank)
            except _lib.UgladError as exc:
                reason = f"rank {self.rank}: {exc}"
        elif reason is None:
            reason = "rank 0 could not make an RCCL unique id"
        ok = torch.tensor([1 if comm else 0], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
        self._dist.all_reduce(ok, op=self._dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) != 1:
            if comm:
                lib.rccl_comm_destroy(comm)
            rec["fallback_reason"] = reason or "another rank could not initialise its RCCL communicator"
            _log.warning("uglad_amd.dist: native RCCL exchange unavailable (%s); using torch.distributed.all_reduce per step", rec["fallback_reason"])
            return None
        try:
            rec["nranks"] = lib.rccl_comm_count(comm)
        except _lib.UgladError as exc:
            _log.warning("uglad_amd.dist: ncclCommCount failed: %s", exc)
        rec.update(native=True, comm=comm, exchange=lib.rccl_exchange(comm))
        return rec["exchange"]

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
