"""Round-4 developer scratch: a loop over FRESH pairs -- new host arrays every iteration, freed behind it, as a loop over files has
them -- in direct and in staged I/O mode, against the bench's end_to_end loop (the same arrays every time)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from open_pcc_metric_amd.calculator import MetricCalculator  # noqa: E402
from open_pcc_metric_amd.cloud_pair import CloudPair  # noqa: E402
from open_pcc_metric_amd.options import CalculateOptions, transform_options  # noqa: E402
from open_pcc_metric_amd.point_cloud import PointCloud  # noqa: E402

a, b, na, nb = bench.synth(1_000_000)
opts = CalculateOptions(color=None, hausdorff=False, point_to_plane=True)


def loop(fresh, staged, reps=12):
    out = []
    for it in range(reps + 2):
        if fresh:
            clouds = (PointCloud(np.array(a), np.array(na)), PointCloud(np.array(b), np.array(nb)))      # (the copies are not timed)
        else:
            clouds = (PointCloud(a, na), PointCloud(b, nb))
        t0 = time.perf_counter()
        with CloudPair(*clouds, extent=[1.0, 1.0, 1.0], staged_io=staged) as p:
            MetricCalculator(p).calculate(transform_options(opts)[2:]).as_dict()
        dt = time.perf_counter() - t0
        del clouds
        if it >= 2:
            out.append(dt * 1e3)
    return out


for fresh in (False, True):
    for staged in (False, True):
        t = loop(fresh, staged)
        print(f"{'fresh arrays' if fresh else 'same arrays '} {'staged' if staged else 'direct'}: median {sorted(t)[len(t) // 2]:.2f} ms  all", " ".join(f"{x:.2f}" for x in t))
