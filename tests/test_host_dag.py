"""Host logic of open_pcc_metric_amd (device columns, metric DAG, calculator, options, report
text) against the golden vectors made by the reference's own NumPy code.  The GPU engine is
replaced by the oracle-backed test double, so this runs on a CPU-only box; tests/test_gpu_*.py
repeat the same checks through libpccm.so on the MI355X."""
import numpy as np
import pytest

import open_pcc_metric_amd.metric as opmm
from conftest import same_bits
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair, DeviceColumn
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from oracle_engine import OracleEngine


def make_pair(g, **kw):
    a = PointCloud(g["a"], g["na"])
    b = PointCloud(g["b"], g["nb"])
    return CloudPair(a, b, extent=g["extent"], _engine=OracleEngine(), **kw)


def test_getters_match_reference(golden):
    pair = make_pair(golden)
    assert np.array_equal(np.asarray(pair.get_left_neighbour_distances()), golden["left_d2"])
    assert np.array_equal(np.asarray(pair.get_right_neighbour_distances()), golden["right_d2"])
    assert np.array_equal(np.asarray(pair.get_left_error_vector()), golden["left_err"])
    assert np.array_equal(np.asarray(pair.get_right_error_vector()), golden["right_err"])
    assert np.array_equal(np.asarray(pair.get_boundary_sqrt_distances()), golden["boundary"])
    assert np.array_equal(pair.get_extent(), golden["extent"])


@pytest.mark.parametrize("tag", ["h0p0", "h0p1", "h1p0", "h1p1"])
def test_full_report_matches_reference(golden, tag):
    hd, p2p = tag[1] == "1", tag[3] == "1"
    pair = make_pair(golden)
    calc = MetricCalculator(pair)
    metrics = transform_options(CalculateOptions(color=None, hausdorff=hd, point_to_plane=p2p))
    if golden["meta"]["raises"].get(tag) == "IndexError":
        with pytest.raises(IndexError):                 # reference quirk Q1 (metric.py:148-152)
            calc.calculate(metrics)
        return
    with np.errstate(divide="ignore"):
        res = calc.calculate(metrics)
    want = golden["meta"]["results"][tag]
    got = res.as_dict()
    assert [tuple(k) for k, _ in want] == list(got.keys())
    for key, val in want:
        assert same_bits(got[tuple(key)], val), (key, got[tuple(key)], val)
    df = res.as_df()
    assert df.to_string() == golden["meta"]["texts"][tag]["string"]
    assert df.to_csv() == golden["meta"]["texts"][tag]["csv"]
    assert str(res) == str(df)


def test_reductions_are_fused_not_materialised(golden):
    pair = make_pair(golden)
    eng = pair._engine
    with np.errstate(divide="ignore"):
        MetricCalculator(pair).calculate(transform_options(CalculateOptions(hausdorff=True)))
    assert any(c[0] == "reduce" for c in eng.calls)
    col = pair.get_left_neighbour_distances()
    assert isinstance(col, DeviceColumn) and col._host is None
    assert np.sum(col, axis=0) == np.sum(np.asarray(col), axis=0)
    assert np.max(col) == np.max(np.asarray(col)) and np.min(col) == np.min(np.asarray(col))


def test_point_to_plane_columns(golden):
    pair = make_pair(golden)
    for side, is_left in (("left", True), ("right", False)):
        calc = MetricCalculator(pair)
        m = opmm.ErrorVector(is_left=is_left, point_to_plane=True)
        if golden["meta"]["raises"].get(side + "_proj") == "IndexError":
            with pytest.raises(IndexError):
                np.asarray(calc._metric_recursive_calculate(m).value)
            continue
        v = calc._metric_recursive_calculate(m).value
        assert isinstance(v, DeviceColumn)
        assert np.array_equal(np.asarray(v), golden[side + "_proj"])
        sq = np.square(v)
        assert isinstance(sq, DeviceColumn)
        assert np.array_equal(np.asarray(sq), np.square(golden[side + "_proj"]))


def test_neighbour_normal_mode_differs_from_row_mode():
    rng = np.random.default_rng(0)
    a, b = rng.random((300, 3)), rng.random((200, 3))
    na, nb = rng.standard_normal((300, 3)), rng.standard_normal((200, 3))
    pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1, 1, 1], normal_index="neighbour",
                     _engine=OracleEngine())
    res = MetricCalculator(pair).calculate(transform_options(CalculateOptions(point_to_plane=True))).as_dict()
    idx = pair._neighbour_index(0)
    e = a - b[idx]
    want = np.sum(np.square(np.einsum("ij,ij->i", e, nb[idx]))) / 300
    assert res[("GeoMSE", True, True)] == pytest.approx(want, rel=1e-12)   # works although nA > nB


def test_missing_normals_raise_for_point_to_plane_only():
    rng = np.random.default_rng(1)
    pair = CloudPair(PointCloud(rng.random((50, 3))), PointCloud(rng.random((50, 3))), extent=[1, 1, 1],
                     _engine=OracleEngine())
    MetricCalculator(pair).calculate(transform_options(CalculateOptions(hausdorff=True)))
    with pytest.raises(ValueError, match="normals"):
        MetricCalculator(pair).calculate(transform_options(CalculateOptions(point_to_plane=True)))


def test_memo_is_per_calculator():
    rng = np.random.default_rng(2)
    p1 = CloudPair(PointCloud(rng.random((40, 3))), PointCloud(rng.random((40, 3))), extent=[1, 1, 1], _engine=OracleEngine())
    p2 = CloudPair(PointCloud(rng.random((40, 3))), PointCloud(rng.random((40, 3))), extent=[1, 1, 1], _engine=OracleEngine())
    m1 = MetricCalculator(p1).calculate([opmm.GeoMSE(True, False)]).as_dict()
    m2 = MetricCalculator(p2).calculate([opmm.GeoMSE(True, False)]).as_dict()
    assert m1 != m2            # the reference's class-level memo would return m1 twice (quirk Q2)


def test_inputs_are_not_mutated():
    rng = np.random.default_rng(3)
    a, b = PointCloud(rng.random((30, 3))), PointCloud(rng.random((30, 3)))
    CloudPair(a, b, extent=[1, 1, 1], _engine=OracleEngine())
    assert not a.has_normals() and not b.has_normals()      # the reference estimates them in place (Q5)


# ---- the reference's own known-answer tests, tests/unit/test_metric.py:30-70 -------------------
@pytest.mark.parametrize("is_left", [True, False])
def test_default_error_vector(is_left):
    error_vector = opmm.ErrorVector(is_left=is_left, point_to_plane=False)
    primary = opmm.PrimaryErrorVector(is_left=is_left)
    primary.value = np.ones(shape=(5, 3), dtype="float64")
    error_vector.calculate(primary)
    assert np.allclose(error_vector.value, np.sqrt(3) * np.ones(shape=(5,)))


@pytest.mark.parametrize("is_left,point_to_plane", [(True, False), (False, False), (True, True), (False, True)])
def test_default_euclidean_distance(is_left, point_to_plane):
    euclidean_distance = opmm.EuclideanDistance(is_left=is_left, point_to_plane=point_to_plane)
    primary = opmm.PrimaryErrorVector(is_left=is_left)
    primary.value = 2 * np.ones(shape=(5,))
    neighbour_distances = opmm.NeighbourDistances(is_left=is_left)
    neighbour_distances.value = 4 * np.ones(shape=(5,))
    euclidean_distance.calculate(neighbour_distances, primary)
    assert np.allclose(neighbour_distances.value, euclidean_distance.value)


def test_error_vector_plain_arrays_follow_reference_loop():
    rng = np.random.default_rng(4)
    ev = opmm.ErrorVector(is_left=True, point_to_plane=True)
    p, n = opmm.PrimaryErrorVector(True), opmm.CloudNormals(False)
    p.value, n.value = rng.random((6, 3)), rng.random((6, 3))
    ev.calculate(p, n)
    assert np.array_equal(ev.value, np.array([np.dot(p.value[i], n.value[i]) for i in range(6)]))
    n.value = n.value[:4]
    with pytest.raises(IndexError):
        ev.calculate(p, n)


def test_keys_and_symmetric_validation():
    assert opmm.GeoMSE(True, False)._key() == ("GeoMSE", True, False)
    assert opmm.MinSqrtDistance()._key() == ("MinSqrtDistance",)
    assert opmm.ColorMSE(False, "ycc")._key() == ("ColorMSE", False, "ycc")
    s = opmm.SymmetricMetric((opmm.GeoPSNR(True, True), opmm.GeoPSNR(False, True)), True)
    assert s._key() == ("SymmetricMetric", "GeoPSNR", True, True, "GeoPSNR", False, True)
    with pytest.raises(ValueError):
        opmm.SymmetricMetric((opmm.GeoPSNR(True, True),), True)
    with pytest.raises(ValueError):
        opmm.SymmetricMetric((opmm.GeoPSNR(True, True), opmm.GeoMSE(False, True)), True)
    assert str(opmm.MinSqrtDistance.__mro__[1].__name__) == "_BoundaryPick"


def test_option_row_order():
    names = [type(m).__name__ for m in transform_options(CalculateOptions("ycc", True, True))]
    assert len(names) == 2 + 6 + 6 + 6 + 6 + 6
    assert names[:2] == ["MinSqrtDistance", "MaxSqrtDistance"]
    assert names[8:14] == ["ColorMSE", "ColorMSE", "SymmetricMetric", "ColorPSNR", "ColorPSNR", "SymmetricMetric"]
    assert names[-6:] == ["GeoHausdorffDistance", "GeoHausdorffDistance", "GeoHausdorffDistancePSNR",
                          "GeoHausdorffDistancePSNR", "SymmetricMetric", "SymmetricMetric"]
