import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
import bench
n = 1000000
a, b, na, nb = bench.synth(n)
e = nat.Engine(0); e.profile(True)
for _ in range(3):
    t = time.perf_counter(); e.set_cloud(0, a); e.set_cloud(1, b); dt = time.perf_counter() - t
print("set_cloud x2 ms", dt * 1e3, "ingest kernel", e.profile_get("ingest"))
for _ in range(3):
    t = time.perf_counter()
    pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1, 1, 1])
    t1 = time.perf_counter()
    pair.recompute() if hasattr(pair, "recompute") else None
    pair._engine.sync()
    t2 = time.perf_counter()
    print("  ctor ms", (t1 - t) * 1e3, "recompute ms", (t2 - t1) * 1e3)
for _ in range(3):
    t = time.perf_counter()
    pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1, 1, 1])
    res = MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, True, True))).as_dict()
    dt = time.perf_counter() - t
    print("end-to-end single shot (H2D + ingest + report incl. self search) ms", dt * 1e3)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1, 1, 1])
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
