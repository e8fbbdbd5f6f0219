"""round-3 dev: where does a fresh pair's time go?  (pageable fp32 inputs, 1M points + normals each)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_pcc_metric_amd import _native as nat
a, b, na, nb = bench.synth(1000000)
e = nat.Engine(0)
for rep in range(5):
    t = [time.perf_counter()]
    e.set_cloud(0, a); t.append(time.perf_counter())
    e.set_normals(0, na); t.append(time.perf_counter())
    e.set_cloud(1, b); t.append(time.perf_counter())
    e.set_normals(1, nb); t.append(time.perf_counter())
    e.sync(); t.append(time.perf_counter())
    e.nn_fuse(0, "row"); e.nn_fuse(1, "row"); e.nn_want_idx(False)
    e.nn_pair("auto"); e.sync(); t.append(time.perf_counter())
    r = e.reduce_total_many([(0, 0), (1, 0), (0, 1), (1, 1)]); t.append(time.perf_counter())
    d = [1e3 * (y - x) for x, y in zip(t, t[1:])]
    print("rep %d: cloudA %.3f | nrmA %.3f | cloudB %.3f | nrmB %.3f | drain %.3f | search %.3f | reduce %.3f | total %.3f ms" % (rep, *d, 1e3 * (t[-1] - t[0])))
