"""Run by tests/test_gpu_ab_paths.py in a child process whose environment selects one of the library's A/B code paths
(the switches are read once per process): a 150k-point report through the product against the oracle, bit for bit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd.calculator import MetricCalculator  # noqa: E402
from open_pcc_metric_amd.cloud_pair import CloudPair  # noqa: E402
from open_pcc_metric_amd.options import CalculateOptions, transform_options  # noqa: E402
from open_pcc_metric_amd.point_cloud import PointCloud  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    n = 150_000
    a = np.random.default_rng(11).random((n, 3), dtype=np.float32)
    b = np.random.default_rng(12).random((n, 3), dtype=np.float32)
    na = np.random.default_rng(13).standard_normal((n, 3), dtype=np.float32)
    nb = np.random.default_rng(14).standard_normal((n, 3), dtype=np.float32)
    want = orc.OraclePair(a, b, na, nb, method="kdtree").report(hausdorff=True, point_to_plane_=True, peak=1.0)
    for use_graph in (False, True):
        pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], device=0, use_graph=use_graph)
        for _ in range(3 if use_graph else 1):          # eager, capture, replay
            pair.recompute()
            got = MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, True, True))).as_dict()
            for key, val in want.items():
                if not (got[key] == val):
                    raise AssertionError(f"{key}: HIP {got[key]!r} != oracle {val!r} (use_graph={use_graph})")
    # a voxelised pair (integer coordinates, exact ties, unequal sizes): the lattice kernel or whatever PCCM_LATTICE=0 puts in its place
    rng = np.random.default_rng(21)
    v = rng.standard_normal((120_000, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    va = np.unique(np.round(200 + 150 * v), axis=0).astype(np.float32)
    vb = np.unique(np.round(va + rng.normal(0, 0.6, va.shape)), axis=0).astype(np.float32)[: len(va) - 77]
    vwant = orc.OraclePair(va, vb, None, None, method="kdtree").report(hausdorff=True, point_to_plane_=False, peak=300.0)
    pair = CloudPair(PointCloud(va), PointCloud(vb), extent=[300.0, 300.0, 300.0], device=0)
    for _ in range(2):                                  # a first search, then a rebuild (spatial order after pccm_drop_caches)
        pair.recompute()
        vgot = MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, True, False))).as_dict()
        for key, val in vwant.items():
            if not (vgot[key] == val):
                raise AssertionError(f"voxelised pair, {key}: HIP {vgot[key]!r} != oracle {val!r}")
    print("ab path ok:", " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("PCCM_")))


if __name__ == "__main__":
    main()
