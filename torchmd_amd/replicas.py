"""Replica fan-out over the GPUs of one node.

Replicas are independent rows of the `[R,N,3]` state tensors (the reference loops over them serially,
`torchmd/forces.py:116`), so they shard with NO per-step data-path collective: one process per GPU,
each owning `R_local` replicas.  The only traffic is control-plane: a one-time consistency check of the
topology and, per output period, an all-gather of three scalars per replica (Epot, Ekin, T) so that
rank 0 can write the monitor rows the reference's `run.py:276-285` writes.  With `torch.distributed`
backend "nccl" this is RCCL over xGMI; the CPU tests run the same code over "gloo".
"""

from __future__ import annotations

import hashlib
import os

import numpy as np
import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def replica_slice(total_replicas: int, rank: int, world: int):
    """Contiguous block of replica indices owned by `rank` (sizes differ by at most one)."""
    if total_replicas < world:
        raise ValueError(f"{total_replicas} replicas cannot be spread over {world} ranks")
    base, extra = divmod(total_replicas, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


class ReplicaFanout:
    def __init__(self, total_replicas: int, device=None, group=None):
        self.group = group
        self.active = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.active else 0
        self.world = dist.get_world_size(group) if self.active else 1
        self.total = total_replicas
        self.local = replica_slice(total_replicas, self.rank, self.world)
        self.counts = [len(replica_slice(total_replicas, r, self.world)) for r in range(self.world)]
        backend = dist.get_backend(group) if self.active else None
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        if backend == "nccl" and self.device.type != "cuda":
            raise RuntimeError("the nccl (RCCL) backend needs device tensors")

    def check_same_topology(self, *arrays) -> None:
        """All ranks must simulate the same system: compare a digest of the topology arrays."""
        h = hashlib.sha256()
        for a in arrays:
            h.update(np.ascontiguousarray(a).tobytes())
        digest = torch.tensor(list(h.digest()[:8]), dtype=torch.int64, device=self.device)
        if not self.active:
            return
        ref = digest.clone()
        dist.broadcast(ref, src=0, group=self.group)
        ok = torch.tensor([int(torch.equal(ref, digest))], dtype=torch.int64, device=self.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) != 1:
            raise RuntimeError("replica ranks were set up with different topologies")

    def gather_observables(self, ekin, epot, temp):
        """Each argument: length-R_local sequence.  Returns [total_replicas, 3] float64 (Ekin, Epot, T)
        ordered by global replica index, on every rank."""
        mine = torch.tensor(
            np.stack([np.asarray(ekin, dtype=np.float64), np.asarray(epot, dtype=np.float64),
                      np.asarray(temp, dtype=np.float64)], axis=1).reshape(-1, 3),
            dtype=torch.float64, device=self.device,
        )
        if mine.shape[0] != len(self.local):
            raise ValueError("observable count does not match the number of local replicas")
        if not self.active:
            return mine.cpu().numpy()
        width = max(self.counts)
        pad = torch.zeros(width, 3, dtype=torch.float64, device=self.device)
        pad[: mine.shape[0]] = mine
        out = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(out, pad, group=self.group)
        rows = [o[:c] for o, c in zip(out, self.counts)]
        return torch.cat(rows, dim=0).cpu().numpy()

    def max_over_ranks(self, value: float) -> float:
        if not self.active:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def barrier(self):
        if self.active:
            if self.device.type == "cuda":
                dist.barrier(group=self.group, device_ids=[self.device.index])
            else:
                dist.barrier(group=self.group)
