"""Developer scratch: the evaluation phase of a report, metric by metric (who blocks, who costs)."""
import sys, os, time, collections
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
for _ in range(5):
    pair.recompute(); MetricCalculator(pair).calculate(metrics_list())
K = 300
rec = 0.0
import gc
if os.environ.get('NOGC'): gc.disable()
lists = []
per = collections.OrderedDict()
tot = np.zeros(4)
for _ in range(K):
    t0 = time.perf_counter()
    pair.recompute()
    ta = time.perf_counter()
    calc = MetricCalculator(pair)
    ml = metrics_list()
    tb = time.perf_counter()
    program, requested = calc._plan(ml)
    t1 = time.perf_counter()
    tot[3] += tb - ta
    lists.append(tb - ta)
    rec = rec + (ta - t0) if _ else ta - t0
    for metric, resolved in program:
        s = time.perf_counter()
        metric.calculate(pair) if resolved is None else metric.calculate(**resolved)
        e = time.perf_counter()
        k = type(metric).__name__ + (":sym" if isinstance(metric, m.SymmetricMetric) else "")
        per[k] = per.get(k, 0.0) + (e - s)
    t2 = time.perf_counter()
    tot += [t1 - t0, t2 - t1, t2 - t0, 0]
print("list us median %.1f p90 %.1f max %.1f" % tuple(np.percentile(np.array(lists) * 1e6, [50, 90, 100])))
print("recompute us", rec / K * 1e6, "list us", tot[3] / K * 1e6)
print("us/step: enqueue+plan %.1f | evaluation (incl. the wait for the GPU) %.1f | total %.1f" % tuple(tot[:3] / K * 1e6))
for k, v in per.items():
    print("  %-34s %7.1f us/step" % (k, v / K * 1e6))
