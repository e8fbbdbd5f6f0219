import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
import open_pcc_metric_amd.metric as m
n = 1000000
a, b, na, nb = bench.synth(n)
pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], use_graph=True)
eng = pair._engine
opts = CalculateOptions(None, False, True)
T = {"recompute": 0.0, "build_metrics": 0.0, "plan": 0.0, "eval": 0.0, "asdict": 0.0}
def step():
    t0 = time.perf_counter(); pair.recompute(); t1 = time.perf_counter()
    metrics = transform_options(opts)[2:] + [m.GeoHausdorffDistance(True, False), m.GeoHausdorffDistance(False, False)]
    t2 = time.perf_counter()
    calc = MetricCalculator(pair); calc._plan(metrics); t3 = time.perf_counter()
    res = [calc._metric_recursive_calculate(x) for x in metrics]; t4 = time.perf_counter()
    d = {x._key(): x.value for x in res}; t5 = time.perf_counter()
    T["recompute"] += t1 - t0; T["build_metrics"] += t2 - t1; T["plan"] += t3 - t2; T["eval"] += t4 - t3; T["asdict"] += t5 - t4
    return d
for _ in range(6): step()
for k in T: T[k] = 0.0
K = 100
t = time.perf_counter()
for _ in range(K): step()
tot = time.perf_counter() - t
print("ms/step", tot / K * 1e3, {k: round(v / K * 1e6, 1) for k, v in T.items()}, "(us)")
# eval split: first reduce_total waits for the GPU
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(K): step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
