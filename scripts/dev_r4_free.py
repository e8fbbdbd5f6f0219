"""Round-4 developer scratch: searches that follow an upload from a FRESH host array (a new address every time, as every real
pair has) against uploads from one reused array; what the runtime does with a user pointer it has not seen (pin, map, unmap)
stalls the kernels that come next."""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from open_pcc_metric_amd import _native as nat  # noqa: E402

ca, cb = bench.synth_content()
e = nat.Engine(0)
e.set_cloud(0, ca)
e.set_cloud(1, cb)
e.nn_want_idx(True)
other = nat.Engine(0)
persistent = np.array(cb[:500_000], dtype=np.float64)
keep = []


def trial(name, make, pause=0.0, same_ctx=False, rows=500_000):
    times, ups = [], []
    for it in range(10):
        buf = make(rows)
        t0 = time.perf_counter()
        (e if same_ctx else other).set_cloud(1, buf)
        ups.append((time.perf_counter() - t0) * 1e3)
        if same_ctx:
            e.set_cloud(1, cb)
        keep.append(buf)
        if pause:
            time.sleep(pause)
        e.drop_caches()
        t0 = time.perf_counter()
        e.nn_pair("grid")
        e.sync()
        times.append((time.perf_counter() - t0) * 1e3)
    print(f"{name:44s} upload ms median {sorted(ups)[5]:.2f} | search ms:", " ".join(f"{t:.2f}" for t in times))


trial("reused host array (12 MB)", lambda r: persistent)
trial("fresh host array (12 MB) every time", lambda r: np.array(cb[:r], dtype=np.float64))
trial("fresh array, 50 ms pause before the search", lambda r: np.array(cb[:r], dtype=np.float64), pause=0.05)
trial("fresh small array (0.7 MB)", lambda r: np.array(cb[:r], dtype=np.float64), rows=30_000)
trial("reused host array again", lambda r: persistent)


def trial_free(name, drop):
    times, gpu = [], []
    for it in range(12):
        buf = np.array(cb[:500_000], dtype=np.float64)
        other.set_cloud(1, buf)
        e.drop_caches()
        e.profile(True)
        e.profile_reset()
        t0 = time.perf_counter()
        e.nn_pair("grid")                  # launched; the kernels run while ...
        if drop:
            del buf                        # ... the host gives the upload's source back to the system (munmap: 12 MB)
        else:
            keep.append(buf)
        e.sync()
        times.append((time.perf_counter() - t0) * 1e3)
        gpu.append(sum(e.profile_get(k)[0] for k in ("grid_build", "grid_query", "grid_finish")) * 1e3)
        e.profile(False)
    print(f"{name:44s} search wall ms:", " ".join(f"{t:.2f}" for t in times), "| GPU ms by events:", " ".join(f"{t:.2f}" for t in gpu))


trial_free("source kept alive", False)
trial_free("source freed while the kernels run", True)
trial_free("source kept alive", False)
trial_free("source freed while the kernels run", True)
