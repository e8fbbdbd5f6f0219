"""Round-4 developer scratch: the headline step's kernels through the C ABI, without torch or the metric DAG.

    N=1000000 python scripts/dev_r4_brick.py            # prints per-class HIP-event averages and a checksum of the results

The step is what CloudPair.recompute() + the report's reductions enqueue: drop_caches, nn_pair (both directions, matched-record
results), four reductions.  The checksum (sums and maxima of the D1 / D2 columns of both directions) must not move between kernel
variants; the parity tests are the judge of correctness, this is the quick look between them.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat  # noqa: E402

n = int(os.environ.get("N", 1000000))
steps = int(os.environ.get("STEPS", 30))
a = np.random.default_rng(1234).random((n, 3), dtype=np.float32)
b = np.random.default_rng(5678).random((n, 3), dtype=np.float32)


def unit(seed):
    g = np.random.default_rng(seed).standard_normal((n, 3), dtype=np.float32)
    return (g / np.linalg.norm(g, axis=1, keepdims=True)).astype(np.float32)


e = nat.Engine(0)
e.set_cloud(0, a)
e.set_normals(0, unit(4321))
e.set_cloud(1, b)
e.set_normals(1, unit(8765))
e.nn_fuse(nat.DIR_LEFT, "row")
e.nn_fuse(nat.DIR_RIGHT, "row")
e.nn_want_idx(False)
req = [(nat.DIR_LEFT, nat.METRIC_D1), (nat.DIR_LEFT, nat.METRIC_D2), (nat.DIR_RIGHT, nat.METRIC_D1), (nat.DIR_RIGHT, nat.METRIC_D2)]


def step():
    e.drop_caches()
    e.nn_pair("grid")
    e.reduce_prefetch_many(req, "row")
    return e.reduce_total_many(req, "row")


for _ in range(4):
    tot = step()
e.sync()
e.profile(True)
e.profile_reset()
t0 = time.perf_counter()
for _ in range(steps):
    tot = step()
e.sync()
dt = (time.perf_counter() - t0) / steps
out = {k: e.profile_get(k) for k in nat.KERNEL_CLASSES}
e.profile(False)
chk = " ".join(f"{float(v):.17g}" for t in tot for v in t)
import hashlib  # noqa: E402
print("RESULT n", n, "eager_ms/step %.4f" % (dt * 1e3), " ".join(f"{k} {v[0] / v[1] * 1e3:.1f}" for k, v in out.items() if v[1]),
      "| chk", hashlib.sha1(chk.encode()).hexdigest()[:12], "| tails", [e.nn_stats(d)["tail_queries"] for d in (0, 1)], "| tag", os.environ.get("TAG", ""))
