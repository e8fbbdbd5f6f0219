"""pccm_tie_exposure / CloudPair.tie_exposure: the reference keeps whichever of several equidistant nearest
neighbours nanoflann meets (cloud_pair.py:22-23), this package the smallest row; the point-to-plane projection
(metric.py:146-153) depends on the pick.  The diagnostic reports the tie rate and the interval of D2 MSE values any tie rule can
produce.  Checked here against a dense NumPy enumeration of all nearest neighbours (an independent restatement of the
definition, not the oracle's search), on tie-laden lattices and on tie-free data."""
import numpy as np
import pytest

from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.point_cloud import PointCloud

pytestmark = pytest.mark.gpu


def dense_exposure(a, b, nb_normals, mode):
    """All nearest neighbours of every a_i in b by a dense distance matrix (fp64, the reference's summation order);
    -> (tied, sum_min, sum_max, sum_pick, max_mult) with the smallest row as the pick."""
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    tied = mult = 0
    smin = smax = spick = 0.0
    for i in range(len(a64)):
        d = a64[i] - b64
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        best = d2.min()
        cand = np.flatnonzero(d2 == best)
        nrm = nb_normals[i][None, :] if mode == "row" else nb_normals[cand]
        p = np.einsum("ij,ij->i", d[cand], np.broadcast_to(nrm, (len(cand), 3)))
        v = p * p
        smin += v.min(); smax += v.max(); spick += v[0]
        tied += len(cand) > 1
        mult = max(mult, len(cand))
    return tied, smin, smax, spick, mult


@pytest.mark.parametrize("mode", ["row", "neighbour"])
def test_lattice_pair_interval_contains_the_pick_and_matches_the_enumeration(mode):
    rng = np.random.default_rng(3)
    a = np.unique(rng.integers(0, 24, (2600, 3)), axis=0).astype(np.float32)
    b = np.unique(rng.integers(0, 24, (2600, 3)), axis=0).astype(np.float32)
    n = min(len(a), len(b))
    a, b = a[rng.permutation(len(a))[:n]], b[rng.permutation(len(b))[:n]]          # equal sizes: row-indexed normals are legal
    na = rng.standard_normal((n, 3))
    nb = rng.standard_normal((n, 3))
    with CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[24.0, 24.0, 24.0], normal_index=mode) as pair:
        for is_left, (q, r, nr) in ((True, (a, b, nb)), (False, (b, a, na))):
            t = pair.tie_exposure(is_left, point_to_plane=True)
            tied, smin, smax, spick, mult = dense_exposure(q, r, nr, mode)
            assert t["queries"] == n and t["not_enumerated"] == 0
            assert t["tied_queries"] == tied and tied > n // 10          # a lattice: ties are the rule
            assert t["max_multiplicity"] == mult
            assert np.isclose(t["d2_mse_min"], smin / n, rtol=1e-12, atol=0)
            assert np.isclose(t["d2_mse_max"], smax / n, rtol=1e-12, atol=0)
            assert np.isclose(t["d2_mse_pick"], spick / n, rtol=1e-12, atol=0)
            assert t["d2_mse_min"] <= t["d2_mse_pick"] <= t["d2_mse_max"] and t["d2_mse_min"] < t["d2_mse_max"]
            # the report's own GeoMSE(point_to_plane=True) is the pick's value
            import open_pcc_metric_amd.metric as m
            from open_pcc_metric_amd.calculator import MetricCalculator
            mse = MetricCalculator(pair).calculate([m.GeoMSE(is_left, True)]).as_dict()[("GeoMSE", is_left, True)]
            assert np.isclose(float(mse), t["d2_mse_pick"], rtol=1e-12, atol=0)
            assert t["d2_mse_min"] <= float(mse) * (1 + 1e-12) and float(mse) <= t["d2_mse_max"] * (1 + 1e-12)


def test_tie_free_data_collapses_the_interval():
    rng = np.random.default_rng(5)
    n = 20000
    a, b = rng.random((n, 3), dtype=np.float32), rng.random((n, 3), dtype=np.float32)
    nb = rng.standard_normal((n, 3)).astype(np.float32)
    na = rng.standard_normal((n, 3)).astype(np.float32)
    with CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0]) as pair:
        for is_left in (True, False):
            t = pair.tie_exposure(is_left, point_to_plane=True)
            assert t["tied_queries"] == 0 and t["tie_rate"] == 0.0 and t["max_multiplicity"] == 1
            assert t["d2_mse_min"] == t["d2_mse_max"] == t["d2_mse_pick"]
        counts_only = pair.tie_exposure(True, point_to_plane=False)
        assert counts_only["tied_queries"] == 0 and "d2_mse_min" not in counts_only


def test_far_outliers_count_with_their_own_projection():
    """Queries whose ball is not enumerated (far outliers: more than 4096 cells, or a winner found by the exact rescan) enter all
    three sums with the library's own pick, so that d2_mse_pick is the reported GeoMSE(point_to_plane=True) and the interval
    still contains it."""
    import open_pcc_metric_amd.metric as m
    from open_pcc_metric_amd.calculator import MetricCalculator
    rng = np.random.default_rng(8)
    n = 30000
    a, b = rng.random((n, 3), dtype=np.float32), rng.random((n, 3), dtype=np.float32)
    a[:25] += np.float32(40.0) * rng.standard_normal((25, 3)).astype(np.float32)       # strays far from everything
    na, nb = rng.standard_normal((n, 3)).astype(np.float32), rng.standard_normal((n, 3)).astype(np.float32)
    with CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], nn_engine="grid") as pair:
        t = pair.tie_exposure(True, point_to_plane=True)
        mse = float(MetricCalculator(pair).calculate([m.GeoMSE(True, True)]).as_dict()[("GeoMSE", True, True)])
        assert t["not_enumerated"] > 0
        assert np.isclose(t["d2_mse_pick"], mse, rtol=1e-11, atol=0)
        assert t["d2_mse_min"] <= mse * (1 + 1e-12) and mse <= t["d2_mse_max"] * (1 + 1e-12)
