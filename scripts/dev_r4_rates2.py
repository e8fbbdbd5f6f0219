"""Round-4 dev: the with_reconst chain of the content_cli record, per stage and kernel class."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.point_cloud import PointCloud
ca, cb = bench.synth_content()
rng = np.random.default_rng(78)
decoded = [cb]
for step in (2, 4):
    q = np.unique((np.round(ca / step) * step).astype(np.float32), axis=0)
    decoded.append(np.ascontiguousarray(q[rng.random(len(q)) >= 0.03]))
cols = lambda p: np.clip(np.rint(128 + 90 * np.sin(p / 31.0)), 0, 255).astype(np.uint8) / 255.0
for rep in range(2):
    pair = None
    for d in decoded:
        t0 = time.perf_counter()
        if pair is None:
            pair = CloudPair(PointCloud(ca, None, cols(ca)), PointCloud(d, None, cols(d)), normal_index="neighbour", extent=[511.0, 322.0, 505.0])
            e = pair._engine
            e.profile(True)
        else:
            e.profile_reset()
            pair = pair.with_reconst(PointCloud(d, None, cols(d)))
        e.sync()
        t1 = time.perf_counter()
        prof = {k: round(e.profile_get(k)[0] * 1e3, 1) for k in nat.KERNEL_CLASSES if e.profile_get(k)[1]}
        if rep:
            print(f"n_dec {len(d)}: construct/with_reconst {1e3 * (t1 - t0):.2f} ms | kernel us {prof} | cells {e.nn_stats(0)['splits']} pairs {e.nn_stats(0)['pairs']}")
    e.profile(False)
    pair.close()
