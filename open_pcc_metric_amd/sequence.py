"""Many pairs, many GPUs: the frames of a sequence (or the decoded versions of a frame) are independent units.

The reference evaluates one pair per process run (handler.py:44-71).  A codec study evaluates hundreds; with one
process per GPU (``torch.distributed``) those shard by *pair*, not by point: rank r takes pairs r, r + W, r + 2W, ...
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
    def report(item):
        origin, reconst = item() if callable(item) else item
        with CloudPair(origin, reconst, device=device, **pair_kwargs) as pair:
            return MetricCalculator(pair).calculate(transform_options(options)).as_dict()

    todo, total = [], 0
    for i, item in enumerate(pairs):
        total = i + 1
        if i % world == rank:
            todo.append((i, item))
    mine: typing.Dict[int, typing.Dict] = {}
    if workers > 1 and len(todo) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(workers)) as pool:
            for (i, _), rows in zip(todo, pool.map(report, [item for _, item in todo])):
                mine[i] = rows
    else:
        for i, item in todo:
            mine[i] = report(item)
    if world == 1:
        return [mine[i] for i in range(total)]
    shares: typing.List[typing.Optional[dict]] = [None] * world
    dist.all_gather_object(shares, mine, group=group)
    merged: typing.Dict[int, typing.Dict] = {}
    for share in shares:
        merged.update(share)
    return [merged[i] for i in range(total)]
