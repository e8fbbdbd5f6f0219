"""Many pairs, many GPUs: the frames of a sequence (or the decoded versions of a frame) are independent units.

The reference evaluates one pair per process run (handler.py:44-71).  A codec study evaluates hundreds; with one
process per GPU (``torch.distributed``) those shard by *pair*, not by point: rank r takes chains r, r + W, r + 2W, ... (a chain =
consecutive pairs that share their origin cloud: one reference against several decoded versions keeps the reference resident)
and runs each whole report on its own GPU -- no collective in the data path (the per-pair context pool of
``_native.acquire_engine`` makes consecutive pairs cheap), one small ``all_gather_object`` of the finished rows at the
end.  That is the weak-scaling way to use a node; sharding the query axis of ONE pair (``CloudPair(group=...)``) is the
strong-scaling way and only pays for very large clouds.
"""
from __future__ import annotations

import typing

from .calculator import MetricCalculator
from .cloud_pair import CloudPair
from .options import CalculateOptions, transform_options


def evaluate_pairs(pairs: typing.Iterable[typing.Tuple[typing.Any, typing.Any]], options: CalculateOptions, *, group=None,
                   device: typing.Optional[int] = None, workers: int = 2,
                   **pair_kwargs) -> typing.List[typing.Dict[typing.Tuple, typing.Any]]:
    """Reports (``CalculateResult.as_dict()``) of every ``(origin_cloud, reconst_cloud)`` in ``pairs``, in input order.

    ``group``: a ``torch.distributed`` process group (or ``True`` for the default group); every rank must pass the same
    sequence (only its own share is loaded if the items are callables returning the two clouds).  ``pair_kwargs`` go to
    ``CloudPair`` (``extent=``, ``normal_index=``, ``nn_engine=`` ...).  Without a group: one GPU, all pairs.
    ``workers``: host threads per rank, each with its own pooled context -- one pair's upload overlaps another's
    kernels (ctypes releases the GIL inside the library): 1.81 -> 1.57 ms per fresh 1M-point pair with two."""
    rank, world, dist = 0, 1, None
    if group is not None:
        import torch.distributed as dist
        if group is True:
            group = dist.group.WORLD
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    if device is None and world > 1:
        import os
        device = int(os.environ.get("LOCAL_RANK", "0"))
    def run_chain(chain):
        """The items of one chain, one after the other on one context; consecutive items with the SAME origin cloud object keep
        it resident (CloudPair.with_reconst: upload, normals, extent and self search of the origin happen once)."""
        out, pair, prev_origin = {}, None, None
        try:
            for i, item in chain:
                origin, reconst = item() if callable(item) else item
                if pair is not None and origin is prev_origin:
                    pair = pair.with_reconst(reconst)
                else:
                    if pair is not None:
                        pair.close()
                    # (items that are LOADED here are freed here, while the context works on: their bytes go through its own buffers)
                    pair = CloudPair(origin, reconst, device=device, **{"staged_io": callable(item), **pair_kwargs})
                out[i] = MetricCalculator(pair).calculate(transform_options(options)).as_dict()
                prev_origin = origin
        finally:
            if pair is not None:
                pair.close()
        return out

    # chains: runs of consecutive items that share their origin cloud (by identity; items given as callables are loaded by
    # the rank that owns them, so each stands alone).  Ranks and worker threads take whole chains.
    chains, total, last_origin = [], 0, None
    for i, item in enumerate(pairs):
        total = i + 1
        origin = None if callable(item) else item[0]
        if chains and origin is not None and origin is last_origin:
            chains[-1].append((i, item))
        else:
            chains.append([(i, item)])
        last_origin = origin
    todo = [c for k, c in enumerate(chains) if k % world == rank]
    mine: typing.Dict[int, typing.Dict] = {}
    if workers > 1 and len(todo) > 1:
        import sys
        from concurrent.futures import ThreadPoolExecutor
        # The threads hand the interpreter to each other around every call into the library; a thread that comes back from an
        # upload while the other runs Python waits for the interpreter's switch interval -- 5 ms by default, four pairs' worth.
        # 20 us for the duration of the sequence: 1.01-1.08 ms per fresh 1M-point pair instead of 1.05-1.46.
        interval = sys.getswitchinterval()
        sys.setswitchinterval(2e-5)
        try:
            with ThreadPoolExecutor(max_workers=int(workers)) as pool:
                for rows in pool.map(run_chain, todo):
                    mine.update(rows)
        finally:
            sys.setswitchinterval(interval)
    else:
        for chain in todo:
            mine.update(run_chain(chain))
    if world == 1:
        return [mine[i] for i in range(total)]
    shares: typing.List[typing.Optional[dict]] = [None] * world
    dist.all_gather_object(shares, mine, group=group)
    merged: typing.Dict[int, typing.Dict] = {}
    for share in shares:
        merged.update(share)
    return [merged[i] for i in range(total)]
