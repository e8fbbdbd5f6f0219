"""Developer scratch: where does the host time of one bench step go?"""
import cProfile, pstats, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
import open_pcc_metric_amd.metric as m
from open_pcc_metric_amd import _native as nat

n = int(os.environ.get("N", 1000000))
a, b, na, nb = bench.synth(n)
pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0])
eng = pair._engine
opts = CalculateOptions(None, False, True)
def step():
    pair.recompute()
    metrics = transform_options(opts)[2:] + [m.GeoHausdorffDistance(True, False), m.GeoHausdorffDistance(False, False)]
    return MetricCalculator(pair).calculate(metrics).as_dict()
for _ in range(5): step()
eng.sync()
K = 50
t = time.perf_counter()
for _ in range(K): step()
eng.sync(); print("ms/step", (time.perf_counter() - t) / K * 1e3)
# split: recompute only (async) then sync
t = time.perf_counter()
for _ in range(K): pair.recompute()
t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter()
print("recompute enqueue ms", (t1 - t) / K * 1e3, "drain ms total", (t2 - t1) * 1e3)
t = time.perf_counter()
for _ in range(K): pair.recompute(); eng.sync()
print("recompute+sync ms", (time.perf_counter() - t) / K * 1e3)
for metric in (nat.METRIC_D1, nat.METRIC_D2):
    t = time.perf_counter()
    for _ in range(K): eng.reduce(0, metric)
    print("reduce metric", metric, "ms", (time.perf_counter() - t) / K * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(K): step()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
