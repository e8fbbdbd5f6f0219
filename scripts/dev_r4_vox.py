"""Round-4 dev: the content pair's searches through the C ABI: tail lengths and kernel times, distances only vs with rows."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_pcc_metric_amd import _native as nat
a, b = bench.synth_content()
for rows in (False, True):
    e = nat.Engine(0)
    e.set_cloud(0, a); e.set_cloud(1, b)
    e.nn_want_idx(rows)
    for _ in range(3):
        e.drop_caches(); e.nn_pair("grid"); e.nn(2, "grid")
    e.sync()
    print("rows", rows, "fallback", [e.nn_stats(d)["fallback_queries"] for d in (0, 1, 2)], "cells", e.nn_stats(0)["splits"])
    e.profile(True); e.profile_reset()
    for _ in range(20):
        e.drop_caches(); e.nn_pair("grid")
    e.sync()
    print("   pair only:", {k: round(e.profile_get(k)[0] / 20 * 1e3, 1) for k in nat.KERNEL_CLASSES if e.profile_get(k)[1]})
    e.profile_reset()
    for _ in range(20):
        e.nn(2, "grid")
    e.sync()
    print("   self only:", {k: round(e.profile_get(k)[0] / 20 * 1e3, 1) for k in nat.KERNEL_CLASSES if e.profile_get(k)[1]})
    e.close()
