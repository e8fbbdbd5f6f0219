"""GPU normal estimation (the stand-in for Open3D's estimate_normals, cloud_pair.py:61-64) against the
oracle's restatement.  Open3D itself cannot be pinned, so this is the one place with a
tolerance: normals are eigenvectors, compared up to sign where the eigen-gap makes them well defined
(|cos| >= 1 - 1e-9), and the D2 report computed from them must agree to 1e-9 relative."""
import numpy as np
import pytest

from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def surface(n, seed, noise=0.002):
    rng = np.random.default_rng(seed)
    u, v = rng.random(n) * 2 - 1, rng.random(n) * 2 - 1
    z = 0.3 * np.sin(2 * u) * np.cos(3 * v) + rng.normal(0, noise, n)
    return np.stack([u, v, z], 1)


def check_normals(pts, k=30):
    e = nat.Engine(0)
    e.set_cloud(0, pts)
    e.set_cloud(1, pts[: max(1, len(pts) // 2)])
    e.estimate_normals(0, k)
    got = e.get_normals(0)
    e.close()
    want, w = orc.estimate_normals(pts, k)
    assert got.shape == want.shape
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-12)
    lead = got[np.arange(len(got)), np.argmax(np.abs(got), axis=1)]
    assert np.all(lead > 0)                                   # documented sign convention
    gap_ok = (w[:, 1] - w[:, 0]) > 1e-6 * np.maximum(w[:, 2], 1e-300)
    cos = np.abs(np.sum(got * want, axis=1))
    assert gap_ok.mean() > 0.9
    assert np.all(cos[gap_ok] >= 1 - 1e-9), float(cos[gap_ok].min())
    return got


@pytest.mark.parametrize("n,seed", [(3000, 1), (20000, 2)])
def test_normals_of_a_noisy_surface(n, seed):
    got = check_normals(surface(n, seed))
    assert np.mean(np.abs(got[:, 2]) > 0.5) > 0.9             # the sheet is roughly horizontal


def test_normals_uniform_volume_and_outliers():
    rng = np.random.default_rng(3)
    pts = rng.random((5000, 3))
    pts[:5] += 40.0                                           # isolated points: exact full-scan path
    check_normals(pts)


def test_normals_on_voxelised_surfaces_with_exact_ties():
    """Integer coordinates: the k-th neighbour distance is tied almost everywhere, so the (d2, row) order of the
    selection decides the neighbour set."""
    rng = np.random.default_rng(8)
    v = rng.standard_normal((40000, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    check_normals(np.unique(np.round(64 + 50 * v * [1.0, 0.7, 0.5]), axis=0))
    u = rng.integers(0, 120, (15000, 2)).astype(np.float64)
    sheet = np.unique(np.column_stack([u, np.round(0.3 * u[:, 0] + 4 * np.sin(u[:, 1] / 9.0))]), axis=0)
    check_normals(sheet)


def test_normals_tiny_and_degenerate_clouds():
    e = nat.Engine(0)
    two = np.array([[0.0, 0, 0], [1, 1, 1]])
    e.set_cloud(0, two)
    e.set_cloud(1, two)
    e.estimate_normals(0, 30)
    assert np.array_equal(e.get_normals(0), [[0, 0, 1], [0, 0, 1]])      # fewer than 3 points: Open3D's default
    plane = np.random.default_rng(4).random((200, 3))
    plane[:, 2] = 5.0
    e.set_cloud(0, plane)
    e.estimate_normals(0, 30)
    assert np.allclose(np.abs(e.get_normals(0)), [0, 0, 1], atol=1e-12)
    e.close()


def test_point_to_plane_report_with_estimated_normals():
    a = surface(6000, 5)
    b = a + np.random.default_rng(6).normal(0, 0.003, a.shape)
    pair = CloudPair(PointCloud(a), PointCloud(b), extent=[2, 2, 1])          # no normals given
    res = MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, True, True))).as_dict()
    assert not pair.clouds[0].has_normals() and not pair.clouds[1].has_normals()   # inputs untouched (quirk Q5 not copied)
    na, _ = orc.estimate_normals(a, 30)
    nb, _ = orc.estimate_normals(b, 30)
    want = orc.OraclePair(a, b, na, nb, method="kdtree").report(hausdorff=True, point_to_plane_=True, peak=2.0)
    for key, val in want.items():
        assert res[key] == pytest.approx(val, rel=1e-9), key
    strict = CloudPair(PointCloud(a), PointCloud(b), extent=[2, 2, 1], estimate_normals=False)
    with pytest.raises(ValueError, match="normals"):
        MetricCalculator(strict).calculate(transform_options(CalculateOptions(None, False, True)))
