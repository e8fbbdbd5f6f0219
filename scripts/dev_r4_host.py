"""Round-4 dev: what the host does behind the last kernel of a headline step (GPU box).

Stamps the return of the one blocking call of a report (Engine.reduce_total_many) and the end of calculate() + as_dict():
the difference is Python that no GPU work hides."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_pcc_metric_amd import _native as nat, metric as m
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud

n = int(os.environ.get("N", 1000000))
a, b, na, nb = bench.synth(n)
pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], use_graph=True)
eng = pair._engine
options = CalculateOptions(color=None, hausdorff=False, point_to_plane=True)
stamps = {}
orig = type(eng).reduce_total_many


def wrapped(self, *a, **k):
    stamps["call"] = time.perf_counter()
    r = orig(self, *a, **k)
    stamps["ret"] = time.perf_counter()
    return r


type(eng).reduce_total_many = wrapped


def metrics():
    return transform_options(options)[2:] + [m.GeoHausdorffDistance(True, False), m.GeoHausdorffDistance(False, False)]


def step():
    pair.recompute()
    return MetricCalculator(pair).calculate(metrics()).as_dict()


for _ in range(6):
    step()
import gc
gc.collect(); gc.freeze()
eng.sync()
K = 500
pre = wait = post = 0.0
T0 = time.perf_counter()
for _ in range(K):
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    pre += stamps["call"] - t0
    wait += stamps["ret"] - stamps["call"]
    post += t1 - stamps["ret"]
tot = (time.perf_counter() - T0) / K
print(f"RESULT n {n}: step {tot * 1e6:.1f} us = before the blocking call {pre / K * 1e6:.1f} + inside it {wait / K * 1e6:.1f} + after it {post / K * 1e6:.1f}")
