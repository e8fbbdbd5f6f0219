"""Round-3 pieces of the brick kernel (open_pcc_metric_amd/csrc/pccm_brick.hip) against the oracle, bit for bit:

* the ring-1 stop rule evaluated in fp32 (face distances with a slack that covers the fp32 evaluation) on clouds far from
  the origin, where that slack is a visible fraction of a cell -- the rule may only ever send MORE queries to the tail
  kernels, never settle one that the fp64 rule (face_bound, pccm_grid.h) would not;
* nearer-of-pair tracking: exact ties and near ties inside one pair of LDS records (duplicated points in the searched cloud
  are always neighbours in the cell-sorted array)."""
import numpy as np
import pytest

from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def report_equal(a, b, na, nb, extent):
    want = orc.OraclePair(a, b, na, nb, method="kdtree").report(hausdorff=True, point_to_plane_=True, peak=1.0)
    with CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=extent, device=0) as pair:
        for _ in range(2):
            pair.recompute()
            got = MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, True, True))).as_dict()
            for key, val in want.items():
                assert got[key] == val, key
        return pair._engine.nn_stats(0), pair._engine.nn_stats(1)


@pytest.mark.parametrize("offset", [0.0, 1.0e3, 1.0e5, -3.0e6])
def test_fp32_stop_rule_far_from_the_origin(offset):
    n = 200_000
    rng = np.random.default_rng(5)
    a = (rng.random((n, 3)) * 64.0 + offset).astype(np.float32)       # fp32-exact by construction; ulp(3e6) = 0.25 of a 1.2-unit cell
    b = (rng.random((n, 3)) * 64.0 + offset).astype(np.float32)
    na = rng.standard_normal((n, 3)).astype(np.float32)
    nb = rng.standard_normal((n, 3)).astype(np.float32)
    report_equal(a, b, na, nb, [1.0, 1.0, 1.0])


def test_pairs_with_ties_and_duplicates():
    n = 150_000
    rng = np.random.default_rng(6)
    a = rng.random((n, 3), dtype=np.float32)
    b = rng.random((n, 3), dtype=np.float32)
    b[1::2] = b[0::2]                                                   # every point of B twice: the winner's partner is its twin
    b[2::4] = np.nextafter(b[2::4], np.float32(2.0))                    # ... or one ulp away from the next pair's
    na = rng.standard_normal((n, 3)).astype(np.float32)
    nb = rng.standard_normal((n, 3)).astype(np.float32)
    report_equal(a, b, na, nb, [1.0, 1.0, 1.0])
