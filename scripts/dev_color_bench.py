import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rng = np.random.default_rng(0)
a = rng.random((n, 3), dtype=np.float32)
b = rng.random((n, 3), dtype=np.float32)
ca = (rng.integers(0, 256, (n, 3)) / 255.0)
cb = (rng.integers(0, 256, (n, 3)) / 255.0)
e = nat.Engine(0)
e.profile(True)
x = ((rng.integers(0, 256, (n, 3)) - rng.integers(0, 256, (n, 3))) / 255.0) ** 2
for _ in range(3):
    e.profile_reset()
    t = time.perf_counter(); s = e.seq_colsum(x); dt = time.perf_counter() - t
    print("seq_colsum wall ms", dt * 1e3, "kernel", e.profile_get("reduce"))
t = time.perf_counter(); w = np.add.reduce(x, axis=0); print("numpy add.reduce ms", (time.perf_counter() - t) * 1e3, (s == w).all())
pair = CloudPair(PointCloud(a, None, ca), PointCloud(b, None, cb), extent=[1, 1, 1])
pair._engine.profile(True)
for scheme in ("rgb", "ycc"):
    for _ in range(3):
        pair.recompute()
        pair._engine.sync()
        pair._engine.profile_reset()
        t = time.perf_counter()
        res = MetricCalculator(pair).calculate(transform_options(CalculateOptions(color=scheme))).as_dict()
        dt = time.perf_counter() - t
    print(scheme, "report with colour rows ms", dt * 1e3, "point", pair._engine.profile_get("point"), "reduce", pair._engine.profile_get("reduce"))
