"""Developer scratch: a report on PCC-like content -- voxelised surface, colours, NO normals (estimated on the GPU)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud

rng = np.random.default_rng(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200000
v = rng.standard_normal((n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
r = 380 + 40 * np.sin(3 * np.arctan2(v[:, 1], v[:, 0])) * np.sin(5 * np.arccos(v[:, 2]))
a = np.unique(np.round(512 + r[:, None] * v), axis=0).astype(np.float32)
b = np.unique(np.round(a + rng.normal(0, 0.7, a.shape)), axis=0).astype(np.float32)
ca = rng.integers(0, 256, a.shape) / 255.0
cb = rng.integers(0, 256, b.shape) / 255.0
print("A", len(a), "B", len(b))
opts = CalculateOptions(color="ycc", hausdorff=True, point_to_plane=True)
for it in range(4):
    t0 = time.perf_counter()
    pair = CloudPair(PointCloud(a, None, ca), PointCloud(b, None, cb), normal_index="neighbour")
    t1 = time.perf_counter()
    res = MetricCalculator(pair).calculate(transform_options(opts)).as_dict()
    t2 = time.perf_counter()
    print(f"run {it}: CloudPair (H2D, ingest, both sweeps) {1e3 * (t1 - t0):7.2f} ms | report (normals x2, self search, min-OBB, D1/D2/Hausdorff/colour) {1e3 * (t2 - t1):7.2f} ms")
    pair.close()
import cProfile, pstats
pair = CloudPair(PointCloud(a, None, ca), PointCloud(b, None, cb), normal_index="neighbour")
pr = cProfile.Profile(); pr.enable()
res = MetricCalculator(pair).calculate(transform_options(opts)).as_dict()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
