"""Point-to-plane on the matched point's normal (--normal-index neighbour) formed by the reductions themselves from matched
records (UnitJob::defer 4 / 5, round 4): no separate point pass, the same bits.  Against the oracle, against the unfused path
(pccm_point_metric + NumPy's own sum) and with both normal modes in one batch."""
import numpy as np
import pytest

from conftest import same_bits
from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from oracle import oracle as orc
from test_gpu_parity import clouds, unit_normals

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,n,m", [("uniform32", 100_000, 90_001), ("surface", 60_000, 50_000), ("dup", 30_000, 20_000)])
@pytest.mark.parametrize("exact32", [True, False])
def test_neighbour_normals_full_report_bit_exact(kind, n, m, exact32):
    a, b = clouds(kind, n, m, seed=77)
    na, nb = unit_normals(len(a), 1), unit_normals(len(b), 2)
    if not exact32:                                  # normals that need all of their fp64 bits: the fp64 gather (defer 5)
        na = na.astype(np.float64) * (1.0 + 2.0 ** -40)
        nb = nb.astype(np.float64) * (1.0 - 2.0 ** -41)
    pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], nn_engine="grid", normal_index="neighbour")
    res = MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, True, True))).as_dict()
    want = orc.OraclePair(a, b, na, nb, method="kdtree", normal_index="neighbour").report(hausdorff=True, point_to_plane_=True, peak=1.0)
    assert list(res.keys()) == list(want.keys())
    for k in want:
        assert same_bits(res[k], want[k]), (k, res[k], want[k])
    pair.close()


def test_row_and_neighbour_columns_in_one_batch():
    """One reduction job reads one normal per record: a batch that asks for both modes must not fold them into one job."""
    n = 40_000
    a, b = clouds("uniform32", n, n, seed=5)
    na, nb = unit_normals(n, 3), unit_normals(n, 4)
    e = nat.Engine(0)
    e.set_cloud(0, a)
    e.set_cloud(1, b)
    e.set_normals(0, na)
    e.set_normals(1, nb)
    e.nn_want_idx(True)
    e.nn_pair("grid")
    L, R = nat.DIR_LEFT, nat.DIR_RIGHT
    req = [(L, nat.METRIC_D1), (L, nat.METRIC_D2), (L, nat.METRIC_D2), (R, nat.METRIC_D2), (R, nat.METRIC_D2)]
    modes = ["row", "row", "neighbour", "neighbour", "row"]
    got = e.reduce_total_many(req, modes)
    for (d, metric), mode, g in zip(req, modes, got):
        col = e.point_metric(d, metric, mode)        # the unfused column, reduced by NumPy itself
        assert same_bits(g[0], np.sum(col)) and same_bits(g[1], col.min()) and same_bits(g[2], col.max()), (d, metric, mode)
    e.close()
