#!/usr/bin/env python3
"""What ONE rank's GPU does per step when a pair is split over W ranks: HIP-event sums per kernel class
from pccm_profile_get, for world = 1/2/4/8, the direction-first split of round 2 and the rows-only split of round 1.

One process plays rank r of W on one MI355X (pccm_set_shard_dir with the plan CloudPair would use), so the numbers are
a rank's own kernel times, free of the time slicing that several processes on one GPU would add.  No collective is
timed here: this is the GPU side of a step ("unmeasured on multi-GPU hardware" otherwise).

    python scripts/rank_profile.py [points ...]      # default: 1000000 8000000
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from open_pcc_metric_amd import _native as nat  # noqa: E402
from open_pcc_metric_amd.cloud_pair import shard_plan  # noqa: E402


def one_rank(eng, plan, rank, steps=10):
    for d in (0, 1, 2):
        eng.set_shard_dir(d, *plan[d][rank])
    sharded = any(plan[d][rank] != (0, 1) for d in (0, 1, 2))
    reqs = [(d, m) for d in (0, 1) for m in (nat.METRIC_D1, nat.METRIC_D2) if eng.shard_range(d)[1] > eng.shard_range(d)[0]]

    def aligned(d):
        b, e = eng.shard_range(d)
        return b % 8192 == 0 and (e % 8192 == 0 or e == eng.n_iter(d))
    chunked = all(aligned(d) for d, _ in reqs)

    def step():
        eng.drop_caches()
        eng.nn_pair("grid")
        eng.reduce_prefetch_many(reqs, "row")
        if not sharded:
            eng.reduce_total_many(reqs, "row")
        elif chunked:
            eng.reduce_chunks_many(reqs, "row")        # one number per 8192-row chunk for the exchange (what CloudPair does)
        else:
            for d, m in reqs:
                eng.reduce(d, m, "row")                # per-leaf sums for the exchange vector
    for _ in range(3):
        step()
    eng.profile(True)
    eng.profile_reset()
    for _ in range(steps):
        step()
    eng.sync()
    prof = {k: eng.profile_get(k) for k in nat.KERNEL_CLASSES}
    eng.profile(False)
    out = {k: round(v[0] / steps * 1e3, 1) for k, v in prof.items() if v[1]}
    out["gpu_total_us"] = round(sum(out.values()), 1)
    out["rows"] = [list(eng.shard_range(d)) for d in (0, 1)]
    return out


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [1_000_000, 8_000_000]
    table = {}
    for n in sizes:
        a, b, na, nb = bench.synth(n)
        eng = nat.Engine(0)
        eng.set_cloud(0, a); eng.set_cloud(1, b)
        eng.set_normals(0, na); eng.set_normals(1, nb)
        for d in (0, 1):
            eng.nn_fuse(d, "row")
        eng.nn_want_idx(False)          # what CloudPair sets for clouds without colours (the bench's configuration): W = 1 below is
                                        # then the single-GPU step of bench.py, kernel for kernel (VERDICT r2, weak #7c)
        for world in (1, 2, 4, 8):
            for mode in ("direction", "rows"):
                if world == 1 and mode == "rows":
                    continue
                plan = shard_plan(world, mode)
                ranks = sorted({0, world - 1})
                res = {f"rank{r}": one_rank(eng, plan, r) for r in ranks}
                table[f"{n}/{mode}/world{world}"] = res
                worst = max(v["gpu_total_us"] for v in res.values())
                print(f"n={n:>8} {mode:9s} world={world}: slowest rank {worst:8.1f} us of GPU time per step  " +
                      "  ".join(f"{k}: build {v.get('grid_build', 0):.0f} query {v.get('grid_query', 0):.0f} finish {v.get('grid_finish', 0):.0f} "
                                f"reduce {v.get('reduce', 0):.0f}" for k, v in res.items()), flush=True)
        eng.close()
    print(json.dumps(table))
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "rank_profile.json"), "w") as fh:
        json.dump(table, fh, indent=1)


if __name__ == "__main__":
    main()
