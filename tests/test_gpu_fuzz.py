"""Seeded random differential tests: odd sizes x hostile shapes x engines, searches and whole reports, against the
oracle, bit for bit.  (scripts/dev_fuzz.py and dev_fuzz_report.py run the same generators open-ended; several
thousand cases have passed there.)"""
import numpy as np
import pytest

from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

KINDS = ["uniform32", "uniform64", "offset64", "lattice", "surface", "clusters", "outliers", "planar", "dups"]


def make(rng, n, kind):
    if kind == "uniform32":
        return rng.random((n, 3), dtype=np.float32).astype(np.float64)
    if kind == "uniform64":
        return rng.random((n, 3)) * rng.choice([1.0, 1e-3, 1e4])
    if kind == "offset64":
        return rng.random((n, 3)) * 50 + rng.choice([1e3, 1e5, 4e6]) * rng.random(3)
    if kind == "lattice":
        return rng.integers(0, rng.choice([4, 16, 64, 1024]), (n, 3)).astype(np.float64)
    if kind == "surface":
        v = rng.standard_normal((n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True) + 1e-30
        p = 100 + 60 * v
        return np.round(p) if rng.random() < 0.5 else p.astype(np.float32).astype(np.float64)
    if kind == "clusters":
        c = rng.random((max(1, n // 200), 3)) * 100
        return (c[rng.integers(0, len(c), n)] + rng.normal(0, 0.05, (n, 3))).astype(np.float32).astype(np.float64)
    if kind == "outliers":
        p = rng.random((n, 3), dtype=np.float32).astype(np.float64)
        k = max(1, n // 500)
        p[:k] = (rng.random((k, 3)) - 0.5) * rng.choice([1e2, 1e4, 1e6])
        return p
    if kind == "planar":
        p = rng.random((n, 3)); p[:, rng.integers(0, 3)] = 0.5
        return p
    base = rng.random((max(1, n // 50), 3), dtype=np.float32).astype(np.float64)     # "dups"
    return base[rng.integers(0, len(base), n)]


def test_random_searches_match_the_oracle():
    e = nat.Engine(0)
    for case in range(90):
        rng = np.random.default_rng(9000 + case)
        na = int(rng.choice([1, 2, 3, 17, 64, 65, 300, 1000, 4097, 20000]))
        nb = int(rng.choice([1, 2, 5, 64, 129, 777, 1000, 8192, 8193, 30000]))
        ka, kb = KINDS[case % len(KINDS)], str(rng.choice(KINDS))
        a, b = make(rng, na, ka), make(rng, nb, kb)
        if rng.random() < 0.3:
            b = b + a[rng.integers(0, na)] - b[0]
        eng = ["auto", "grid", "brute"][case % 3]
        e.set_cloud(0, a); e.set_cloud(1, b)
        e.nn_pair(eng)
        e.nn(nat.DIR_SELF, eng)
        for d, (q, r, skip) in enumerate(((a, b, False), (b, a, False), (a, a, True))):
            idx, d2 = e.fetch_nn(d)
            if skip and na < 2:
                assert np.all(idx == -1) and np.all(d2 == 0)
                continue
            oi, od = orc.nn(q, r, skip_same_index=skip, method="kdtree")
            assert np.array_equal(d2, od) and np.array_equal(idx, oi), (case, eng, ka, na, kb, nb, d)
    e.close()


def test_random_reports_match_the_oracle():
    opts = transform_options(CalculateOptions(None, True, True))
    for case in range(12):
        rng = np.random.default_rng(7000 + case)
        n = int(rng.choice([300, 8192, 8193, 20000, 40000]))
        a, b = make(rng, n, KINDS[case % len(KINDS)]), make(rng, n, str(rng.choice(KINDS)))
        na, nb = rng.standard_normal((n, 3)), rng.standard_normal((n, 3))
        graph = case % 2 == 0
        pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 2.0, 0.5], use_graph=graph)
        with np.errstate(divide="ignore"):
            want = orc.OraclePair(a, b, na, nb, method="kdtree").report(hausdorff=True, point_to_plane_=True, peak=2.0)
            for _ in range(3 if graph else 1):
                got = MetricCalculator(pair).calculate(opts).as_dict()
                assert list(got) == list(want)
                for k in want:
                    x, y = np.float64(got[k]), np.float64(want[k])
                    assert x == y or (np.isnan(x) and np.isnan(y)), (case, k, x, y)
                pair.recompute()
        pair.close()
