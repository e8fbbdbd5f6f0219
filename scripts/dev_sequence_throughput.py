"""Developer scratch: fresh 1M-point pairs back to back -- one host thread vs two (uploads overlap the other pair's work)."""
import sys, os, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
a, b, na, nb = bench.synth(1000000)
opts = CalculateOptions(None, True, True)
def one(_):
    with CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1, 1, 1]) as pair:
        return MetricCalculator(pair).calculate(transform_options(opts)).as_dict()
for workers in (1, 2, 3):
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(one, range(2 * workers)))          # warm the pooled contexts
        t = time.perf_counter(); list(ex.map(one, range(60))); dt = time.perf_counter() - t
    print(f"{workers} thread(s): {dt / 60 * 1e3:.2f} ms per fresh pair  ({2e6 / (dt / 60) / 1e6:.0f} Mpoints/s end to end)")
