"""Round-4 developer scratch: one reference cloud against decoded clouds of falling density, chained (with_reconst) and alone:
search time, tail length and grid decisions of every pair."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from open_pcc_metric_amd import _native as nat  # noqa: E402
from open_pcc_metric_amd.cloud_pair import CloudPair  # noqa: E402
from open_pcc_metric_amd.point_cloud import PointCloud  # noqa: E402

ca, cb = bench.synth_content()
rng = np.random.default_rng(78)
decoded = [cb]
for step in (2, 4):
    q = np.unique((np.round(ca / step) * step).astype(np.float32), axis=0)
    decoded.append(np.ascontiguousarray(q[rng.random(len(q)) >= 0.03]))


def unit(n, seed):
    g = np.random.default_rng(seed).standard_normal((n, 3)).astype(np.float32)
    return g / np.linalg.norm(g, axis=1, keepdims=True)


def colours(p):
    c = np.stack([128 + 100 * np.sin(p[:, 0] / 37.0), 128 + 100 * np.cos(p[:, 1] / 23.0), 128 + 90 * np.sin(p[:, 2] / 51.0)], 1)
    return np.clip(np.rint(c + rng.normal(0, 6, c.shape)), 0, 255).astype(np.uint8) / 255.0


WITH_COLOURS = os.environ.get("COLOURS", "1") == "1"
col_a = colours(ca) if WITH_COLOURS else None
col_d = [colours(d) if WITH_COLOURS else None for d in decoded]
na = unit(len(ca), 1)
nd = [unit(len(d), 2 + k) for k, d in enumerate(decoded)]


def look(pair, tag):
    eng = pair._engine
    eng.sync()
    eng.profile(True)
    eng.profile_reset()
    t0 = time.perf_counter()
    pair.recompute()
    eng.sync()
    dt = time.perf_counter() - t0
    prof = {k: eng.profile_get(k) for k in ("grid_build", "grid_query", "grid_finish")}
    eng.profile(False)
    st = [eng.nn_stats(d) for d in (0, 1)]
    print(tag, "points", [len(pair.clouds[0].points), len(pair.clouds[1].points)], "recompute ms %.3f" % (dt * 1e3),
          {k: round(v[0] * 1e3, 1) for k, v in prof.items()}, "tails", [s["tail_queries"] for s in st], "cells", st[0]["splits"], "fallback", [s["fallback_queries"] for s in st])


eng = nat.acquire_engine(0)
for rnd in range(2):
    pair = None
    for k, d in enumerate(decoded):
        eng.sync()
        if os.environ.get("SLEEP"):
            time.sleep(float(os.environ["SLEEP"]))
        if os.environ.get("FRESH"):
            d = np.array(d)                            # a fresh host array, as a reader hands out
        eng.profile(True)
        eng.profile_reset()
        if pair is None:
            eng.reset()
            eng.profile(True)
            eng.profile_reset()
            pair = CloudPair(PointCloud(ca, None, col_a), PointCloud(d, None, col_d[k]), normal_index="neighbour", _engine=eng, **({} if os.environ.get("EXTENT") else {"extent": [1, 1, 1]}))
        else:
            pair = pair.with_reconst(PointCloud(d, None, col_d[k]))
        eng.sync()
        print(f"chained round {rnd} rate {k}: FIRST search (inside the constructor)", {kk: round(eng.profile_get(kk)[0] * 1e3, 1) for kk in ("grid_build", "grid_query", "grid_finish")},
              "tails", [eng.nn_stats(dd)["tail_queries"] for dd in (0, 1)])
        eng.profile(False)
        look(pair, f"chained round {rnd} rate {k} (no normals yet)")
        pair._require_normals(0)
        pair._require_normals(1)                       # estimated on the GPU, as the command line does for files without normals
        look(pair, f"   after the normals' estimation")
        from open_pcc_metric_amd.calculator import MetricCalculator
        from open_pcc_metric_amd.options import CalculateOptions, transform_options
        t0 = time.perf_counter()
        MetricCalculator(pair).calculate(transform_options(CalculateOptions("ycc" if WITH_COLOURS else None, True, True)))
        print("   report ms %.2f" % ((time.perf_counter() - t0) * 1e3))
        if os.environ.get("EXTENT"):
            pair.get_extent()
        look(pair, f"   after the report")
for k, d in enumerate(decoded):
    eng.reset()
    p = CloudPair(PointCloud(ca, na), PointCloud(d, nd[k]), normal_index="neighbour", _engine=eng, extent=[1, 1, 1])
    look(p, f"alone rate {k}")
nat.release_engine(eng)
