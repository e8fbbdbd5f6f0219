"""Cross-rank exchange for the query-axis shard (DESIGN.md section e).

One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in
the CPU tests).  The only data that ever crosses ranks on the hot path are the per-direction
exchange vectors of ``pccm_reduce`` (<= n/128 + 8191 doubles) and two extrema; full per-point
columns are all-gathered only when a getter materialises them.
"""
from __future__ import annotations

import typing

import numpy as np


class Collective:
    def __init__(self, group=None, enabled: bool = False, device: typing.Optional[int] = None):
        self.group = group
        self.device = device        # GPU whose memory stages the nccl exchange (None: torch's current device)
        self.rank, self.world = 0, 1
        self._dist = None
        self._backend = None
        self._buffers = {}
        if enabled or group is not None:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed is not initialised")
            self._dist = dist
            self.rank = dist.get_rank(group)
            self.world = dist.get_world_size(group)
            self._backend = dist.get_backend(group)

    @property
    def sharded(self) -> bool:
        return self.world > 1

    def _tensor(self, arr: np.ndarray):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        return t.to(self._cuda()) if self._backend == "nccl" else t

    def _cuda(self):
        import torch
        return torch.device("cuda", self.device if self.device is not None else torch.cuda.current_device())

    def _staging(self, n: int, dtype):
        """Pinned host + device buffers for the small per-report exchange (nccl): allocated once per size, so
        a report costs two async copies and one collective instead of fresh allocations and pageable copies."""
        import torch
        key = (n, np.dtype(dtype).str)
        buf = self._buffers.get(key)
        if buf is None:
            tdt = torch.from_numpy(np.empty(0, dtype=dtype)).dtype
            buf = (torch.empty(n, dtype=tdt).pin_memory(), torch.empty(n, dtype=tdt, device=self._cuda()))
            self._buffers[key] = buf
        return buf

    def allreduce(self, arr: np.ndarray, op: str = "sum") -> np.ndarray:
        if not self.sharded:
            return arr
        dist = self._dist
        rop = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op]
        arr = np.ascontiguousarray(arr)
        if self._backend == "nccl" and arr.ndim == 1:
            import torch
            host, dev = self._staging(arr.shape[0], arr.dtype)
            host.numpy()[:] = arr
            dev.copy_(host, non_blocking=True)
            dist.all_reduce(dev, op=rop, group=self.group)
            host.copy_(dev, non_blocking=True)
            torch.cuda.current_stream(self._cuda()).synchronize()
            return host.numpy().copy()
        t = self._tensor(arr)
        dist.all_reduce(t, op=rop, group=self.group)
        return t.cpu().numpy()

    def allgather_rows(self, local: np.ndarray, counts) -> np.ndarray:
        """Concatenate the ranks' row blocks (rank r contributes counts[r] rows)."""
        if not self.sharded:
            return local
        import torch
        width = max(counts)
        pad = np.zeros((width,) + local.shape[1:], dtype=local.dtype)
        pad[:local.shape[0]] = local
        t = self._tensor(pad)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(outs, t, group=self.group)
        return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(outs, counts)], axis=0)
