"""Round-4 dev: per-kernel-class time of the three rates of the content_cli record (coarser decoded lattices)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
ca, cb = bench.synth_content()
rng = np.random.default_rng(78)
decoded = [cb]
for step in (2, 4):
    q = np.unique((np.round(ca / step) * step).astype(np.float32), axis=0)
    decoded.append(np.ascontiguousarray(q[rng.random(len(q)) >= 0.03]))
copts = CalculateOptions(color=None, hausdorff=True, point_to_plane=True)
for d in decoded:
    for rep in range(2):
        pair = CloudPair(PointCloud(ca), PointCloud(d), normal_index="neighbour", extent=[511.0, 322.0, 505.0])
        e = pair._engine
        e.sync()
        e.profile(True); e.profile_reset()
        t0 = time.perf_counter()
        pair.recompute(); e.sync()
        t1 = time.perf_counter()
        pair._require_normals(0); pair._require_normals(1); e.sync()
        t2 = time.perf_counter()
        with np.errstate(divide="ignore"):
            MetricCalculator(pair).calculate(transform_options(copts)).as_dict()
        e.sync()
        t3 = time.perf_counter()
        prof = {k: round(e.profile_get(k)[0] * 1e3, 1) for k in nat.KERNEL_CLASSES if e.profile_get(k)[1]}
        e.profile(False)
        if rep:
            print(f"n_dec {len(d)}: searches {1e3 * (t1 - t0):.2f} ms normals {1e3 * (t2 - t1):.2f} ms report {1e3 * (t3 - t2):.2f} ms | kernel us {prof} | cells {e.nn_stats(0)['splits']} fallback {[e.nn_stats(k)['fallback_queries'] for k in (0,1)]} tails {[e.nn_stats(k)['tail_queries'] for k in (0,1)]} pairs {e.nn_stats(0)['pairs']}")
        pair.close()
