"""Round-4 developer scratch: fresh 1M + 1M pairs through evaluate_pairs, one and two host threads, several repetitions."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from open_pcc_metric_amd.options import CalculateOptions  # noqa: E402
from open_pcc_metric_amd.point_cloud import PointCloud  # noqa: E402
from open_pcc_metric_amd.sequence import evaluate_pairs  # noqa: E402

a, b, na, nb = bench.synth(1_000_000)
opts = CalculateOptions(color=None, hausdorff=False, point_to_plane=True)
for workers, interval in ((1, 0), (2, 0), (2, 0), (2, 0)):        # (evaluate_pairs sets the interpreter's switch interval itself)
    items = [(PointCloud(a, na), PointCloud(b, nb)) for _ in range(16)]
    evaluate_pairs(items[:workers * 2], opts, workers=workers, extent=[1.0, 1.0, 1.0])
    out = []
    for _ in range(4):
        t0 = time.perf_counter()
        evaluate_pairs(items, opts, workers=workers, extent=[1.0, 1.0, 1.0])
        out.append((time.perf_counter() - t0) / len(items) * 1e3)
    print("switch interval", interval, "workers", workers, "ms per pair", " ".join(f"{x:.3f}" for x in out), "| host threads", len(os.sched_getaffinity(0)))
