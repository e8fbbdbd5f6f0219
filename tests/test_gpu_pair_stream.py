"""Two pairs in flight on two contexts of one GPU (bench.py's `pair_stream`): the launch of one pair is issued before the report of
the other is read.  Every report must be the one the pair gives alone -- the contexts share nothing but the device."""
import numpy as np
import pytest

from conftest import same_bits
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from oracle import oracle as orc
from test_gpu_parity import clouds, unit_normals

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_graph", [True, False])
def test_two_pairs_taking_turns(use_graph):
    opts = CalculateOptions(None, True, True)
    n = 120_000
    a, b = clouds("uniform32", n, n, seed=3)              # (equal sizes: the row-indexed normals of quirk Q1 need them)
    c, d = clouds("surface", 50_000, 40_000, seed=4)
    na, nb, nc, nd = unit_normals(len(a), 1), unit_normals(len(b), 2), unit_normals(len(c), 3), unit_normals(len(d), 4)
    want0 = orc.OraclePair(a, b, na, nb, method="kdtree").report(hausdorff=True, point_to_plane_=True, peak=1.0)
    want1 = orc.OraclePair(c, d, nc, nd, method="kdtree", normal_index="neighbour").report(hausdorff=True, point_to_plane_=True, peak=1.0)
    with CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], use_graph=use_graph) as p0, \
            CloudPair(PointCloud(c, nc), PointCloud(d, nd), extent=[1.0, 1.0, 1.0], normal_index="neighbour", use_graph=use_graph) as p1:
        assert p0._engine is not p1._engine

        def finish(p):
            return MetricCalculator(p).calculate(transform_options(opts)).as_dict()

        for p in (p0, p1):                                  # eager, capture, first replay
            for _ in range(3):
                p.recompute()
                finish(p)
        p0.recompute()
        for _ in range(12):
            p1.recompute()                                  # the other pair's launch ...
            r0 = finish(p0)                                 # ... before this pair's report
            p0.recompute()
            r1 = finish(p1)
            for got, want in ((r0, want0), (r1, want1)):
                assert list(got.keys()) == list(want.keys())
                for k in want:
                    assert same_bits(got[k], want[k]), k
        finish(p0)
