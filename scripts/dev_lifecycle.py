"""Developer scratch: what a fresh CloudPair costs beyond the work itself (context create / destroy, allocations)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
a, b, na, nb = bench.synth(1000000)
for rep in range(4):
    t0 = time.perf_counter(); e = nat.Engine(0)
    t1 = time.perf_counter(); e.set_cloud(0, a); e.set_cloud(1, b); e.set_normals(0, na); e.set_normals(1, nb)
    t2 = time.perf_counter(); e.nn_pair("auto"); e.sync()
    t3 = time.perf_counter(); e.close()
    t4 = time.perf_counter()
    print(f"engine create {1e3*(t1-t0):6.2f} | uploads {1e3*(t2-t1):6.2f} | search {1e3*(t3-t2):6.2f} | close {1e3*(t4-t3):6.2f} ms")
opts = transform_options(CalculateOptions(None, True, True))
for rep in range(4):
    t0 = time.perf_counter()
    pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1, 1, 1])
    res = MetricCalculator(pair).calculate(opts).as_dict()
    t1 = time.perf_counter()
    del pair, res
    t2 = time.perf_counter()
    print(f"fresh CloudPair + full report {1e3*(t1-t0):6.2f} ms | teardown {1e3*(t2-t1):6.2f} ms")
