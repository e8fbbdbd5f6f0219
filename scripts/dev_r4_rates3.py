"""Round-4 dev: bench.py's content_cli three-rate flow with the kernel-class time of every stage."""
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
cols = lambda p: np.clip(np.rint(128 + 90 * np.sin(p / 31.0)), 0, 255).astype(np.uint8) / 255.0
copts = CalculateOptions(color="ycc", hausdorff=True, point_to_plane=True)


def snap(e):
    return {k: e.profile_get(k)[0] * 1e3 for k in nat.KERNEL_CLASSES}


for rep in range(3):
    pair = None
    origin = PointCloud(ca, None, cols(ca))
    for d in decoded:
        dec = PointCloud(d, None, cols(d))
        t0 = time.perf_counter()
        if pair is None:
            pair = CloudPair(origin, dec, normal_index="neighbour")
            e = pair._engine
            e.profile(True); e.profile_reset()
        else:
            pair = pair.with_reconst(dec)
        e.sync(); t1 = time.perf_counter(); s1 = snap(e)
        pair._require_normals(0); pair._require_normals(1); e.sync(); t2 = time.perf_counter(); s2 = snap(e)
        pair.get_extent(); t3 = time.perf_counter()
        with np.errstate(divide="ignore"):
            MetricCalculator(pair).calculate(transform_options(copts)).as_df().to_string()
        e.sync(); t4 = time.perf_counter(); s4 = snap(e)
        if rep == 2:
            print(f"n_dec {len(d)}: pair {1e3*(t1-t0):.2f} normals {1e3*(t2-t1):.2f} extent {1e3*(t3-t2):.2f} report {1e3*(t4-t3):.2f} ms | build us: pair-stage {s1['grid_build']:.0f} normals-stage {s2['grid_build']-s1['grid_build']:.0f} report-stage {s4['grid_build']-s2['grid_build']:.0f}")
        e.profile_reset()
    e.profile(False)
    pair.close()
