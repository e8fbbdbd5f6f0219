"""The CPU oracle against the golden vectors produced by the reference's own NumPy code
(tests/golden/make_golden.py) and against the reference's known answers.  CPU only."""
import numpy as np
import pytest

from conftest import same_bits
from oracle import oracle as orc


@pytest.mark.parametrize("method", ["brute", "kdtree"])
def test_per_point_arrays_match_reference(golden, method):
    p = orc.OraclePair(golden["a"], golden["b"], golden["na"], golden["nb"], method=method)
    assert np.array_equal(p.neighbour_distances(True), golden["left_d2"])
    assert np.array_equal(p.neighbour_distances(False), golden["right_d2"])
    assert np.array_equal(p.error_vector(True), golden["left_err"])
    assert np.array_equal(p.error_vector(False), golden["right_err"])
    assert np.array_equal(p.boundary_sqrt_distances(), golden["boundary"])


def test_point_to_plane_matches_reference(golden):
    p = orc.OraclePair(golden["a"], golden["b"], golden["na"], golden["nb"])
    for side, is_left in (("left", True), ("right", False)):
        k = 0 if is_left else 1
        if golden["meta"]["raises"].get(side + "_proj") == "IndexError":
            with pytest.raises(IndexError):        # reference quirk Q1, metric.py:148-152
                orc.point_to_plane(p.points[k], p.points[1 - k], p.nn_idx[k], p.normals[1 - k])
            continue
        proj = orc.point_to_plane(p.points[k], p.points[1 - k], p.nn_idx[k], p.normals[1 - k])
        assert np.array_equal(proj, golden[side + "_proj"])


def test_report_scalars_match_reference(golden):
    p = orc.OraclePair(golden["a"], golden["b"], golden["na"], golden["nb"])
    peak = float(np.max(golden["extent"]))
    assert golden["meta"]["results"], "no result rows in the golden file"
    for tag, rows in golden["meta"]["results"].items():
        if tag.startswith("c"):
            continue                                  # colour rows: tests/test_io_color.py
        rep = p.report(hausdorff=tag[1] == "1", point_to_plane_=tag[3] == "1", peak=peak)
        assert [tuple(k) for k, _ in rows] == list(rep.keys())      # options.py row order
        for key, val in rows:
            assert same_bits(rep[tuple(key)], val), (tag, key)


def test_reference_fixture_known_answers():
    # tests/unit/test_metric.py:13-26: eye(3) vs eye(3) + [.1, .2, .3]; analytic answers
    a = np.eye(3)
    b = a + 1e-1 * np.linspace(1.0, 3, 3)
    nz = np.tile([[0.0, 0.0, 1.0]], (3, 1))
    p = orc.OraclePair(a, b, nz, nz)
    assert np.allclose(p.neighbour_distances(True), 0.14) and np.allclose(p.neighbour_distances(False), 0.14)
    assert p.geo_mse(True, False) == pytest.approx(0.14, rel=1e-15)
    assert p.geo_hausdorff(False, False) == pytest.approx(0.14, rel=1e-15)
    mn, mx = p.min_max_sqrt()
    assert mn == mx == np.sqrt(2.0)
    assert p.geo_mse(True, True) == pytest.approx(0.09, rel=1e-15)
    assert p.psnr(mx, p.geo_hausdorff(True, False)) == pytest.approx(11.54901959985743, rel=1e-14)


def test_kdtree_equals_brute_on_random_and_tied_data():
    rng = np.random.default_rng(3)
    for n, m, lattice in ((3000, 2500, False), (2000, 2000, True), (1, 50, False), (50, 1, False)):
        if lattice:
            a = rng.integers(0, 12, (n, 3)).astype(np.float64)
            b = rng.integers(0, 12, (m, 3)).astype(np.float64)
        else:
            a = rng.random((n, 3))
            b = rng.random((m, 3))
        ib, db = orc.nn(a, b, method="brute")
        ik, dk = orc.nn(a, b, method="kdtree")
        assert np.array_equal(ib, ik) and np.array_equal(db, dk)
        ib, db = orc.nn(a, a, method="brute", skip_same_index=True)
        ik, dk = orc.nn(a, a, method="kdtree", skip_same_index=True)
        assert np.array_equal(ib, ik) and np.array_equal(db, dk)


def test_empty_and_single():
    i, d = orc.nn(np.zeros((0, 3)), np.zeros((5, 3)))
    assert i.shape == (0,) and d.shape == (0,)
    i, d = orc.nn(np.zeros((1, 3)), np.zeros((1, 3)), skip_same_index=True)
    assert i[0] == -1 and d[0] == 0.0
