"""Developer scratch: a bench step (hipGraph replay) split into enqueue, GPU wait and the Python tail."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import open_pcc_metric_amd.metric as m
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud

n = 1000000
a, b, na, nb = bench.synth(n)
pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], use_graph=True)
eng = pair._engine
options = CalculateOptions(color=None, hausdorff=False, point_to_plane=True)
def metrics_list():
    ms = transform_options(options)[2:]
    return ms + [m.GeoHausdorffDistance(True, False), m.GeoHausdorffDistance(False, False)]
def step():
    pair.recompute()
    return MetricCalculator(pair).calculate(metrics_list()).as_dict()
for _ in range(5): step()
eng.sync()
K = 200
t = time.perf_counter()
for _ in range(K): step()
eng.sync(); print("ms/step", (time.perf_counter() - t) / K * 1e3)
# pieces
acc = np.zeros(5)
for _ in range(K):
    t0 = time.perf_counter(); pair.recompute()
    t1 = time.perf_counter(); ms = metrics_list()
    t2 = time.perf_counter(); eng.sync()
    t3 = time.perf_counter(); res = MetricCalculator(pair).calculate(ms)
    t4 = time.perf_counter(); d = res.as_dict()
    t5 = time.perf_counter()
    acc += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]
print("us: recompute(enqueue) %.1f | build metric list %.1f | wait for GPU %.1f | calculate after GPU is done %.1f | as_dict %.1f" % tuple(acc / K * 1e6))
import cProfile, pstats
pr = cProfile.Profile()
for _ in range(K):
    pair.recompute(); ms = metrics_list(); eng.sync()
    pr.enable(); MetricCalculator(pair).calculate(ms).as_dict(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
