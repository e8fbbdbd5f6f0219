"""Parity of the HIP path (through the C ABI of libpccm.so) with the CPU oracle and with the
golden vectors made by the reference's NumPy code.  Needs an MI355X: run with ``-m gpu``.

Bars: integer outputs (neighbour rows) and every fp64 output (d2, error vectors, projections,
sums, maxima, PSNR) are compared BIT-EXACTLY; no tolerance is used anywhere in this file."""
import numpy as np
import pytest

from conftest import same_bits
from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ENGINES = ["grid", "brute"]


def clouds(kind, na, nb, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform32":
        return rng.random((na, 3), dtype=np.float32), rng.random((nb, 3), dtype=np.float32)
    if kind == "f64":
        a = rng.random((na, 3))
        return a, a[rng.integers(0, na, nb)] + rng.normal(0, 1e-3, (nb, 3))
    if kind == "lattice":
        return rng.integers(0, 16, (na, 3)).astype(np.float64), rng.integers(0, 16, (nb, 3)).astype(np.float64)
    if kind == "voxel10":
        a = np.floor(rng.random((na, 3)) * 1024)
        return a, (a[rng.integers(0, na, nb)] + rng.normal(0, .3, (nb, 3))).astype(np.float32)
    if kind == "dup":
        a = rng.random((na, 3), dtype=np.float32)
        a[na // 2:] = a[: na - na // 2]
        return a, a[rng.permutation(na)[:nb]].copy()
    if kind == "surface":       # voxelised closed surface (the shape of MPEG 8i content): integer coordinates, exact ties
        u = rng.random(na * 2) * 2 * np.pi
        v = np.arccos(2 * rng.random(na * 2) - 1)
        r = 180 + 20 * np.sin(3 * u) * np.sin(5 * v)
        p = np.stack([256 + r * np.sin(v) * np.cos(u), 256 + 0.6 * r * np.sin(v) * np.sin(u), 256 + r * np.cos(v)], 1)
        a = np.unique(np.round(p).astype(np.float32), axis=0)[:na]
        b = np.unique(np.round(a + rng.normal(0, 0.7, a.shape)).astype(np.float32), axis=0)[:nb]
        return a, b
    if kind == "disjoint":      # two clouds far apart plus a few stragglers: ring search cannot settle, exact full scans
        a = rng.random((na, 3))
        b = rng.random((nb, 3)) * np.array([1.0, 1.0, 0.02]) + np.array([30.0, -12.0, 55.0])
        a[:5] += 400.0
        return a, b
    if kind == "far":
        return (rng.random((na, 3)) * 1e6 + 1e9), (rng.random((nb, 3)) * 1e6 + 1e9)
    raise ValueError(kind)


def unit_normals(n, seed):
    g = np.random.default_rng(seed).standard_normal((n, 3), dtype=np.float32)
    return (g / np.linalg.norm(g, axis=1, keepdims=True)).astype(np.float32)


@pytest.fixture(scope="module")
def engine():
    e = nat.Engine(0)
    yield e
    e.close()


CASES = [("uniform32", 5, 7), ("uniform32", 1, 1), ("uniform32", 2, 1), ("uniform32", 1000, 1000),
         ("uniform32", 1023, 1025), ("uniform32", 4097, 2049), ("f64", 3000, 2500), ("lattice", 2000, 2000),
         ("voxel10", 5000, 4000), ("dup", 600, 500), ("far", 800, 900), ("uniform32", 70000, 65537),
         ("surface", 60000, 60000), ("disjoint", 3000, 2500)]


@pytest.mark.parametrize("engine_name", ENGINES)
@pytest.mark.parametrize("kind,na,nb", CASES)
def test_nn_bit_exact_all_directions(engine, engine_name, kind, na, nb):
    a, b = clouds(kind, na, nb, seed=na * 7 + nb)
    engine.set_cloud(0, a)
    engine.set_cloud(1, b)
    a64, b64 = np.asarray(a, np.float64), np.asarray(b, np.float64)
    for d, (q, r, skip) in enumerate(((a64, b64, False), (b64, a64, False), (a64, a64, True))):
        engine.nn(d, engine_name)
        idx, d2 = engine.fetch_nn(d)
        if skip and na < 2:
            assert np.all(idx == -1) and np.all(d2 == 0.0)      # Open3D returns zeros for < 2 points
            continue
        oi, od = orc.nn(q, r, skip_same_index=skip, method="kdtree")
        assert np.array_equal(d2, od), f"d2 differs in direction {d}"
        assert np.array_equal(idx, oi), f"neighbour rows differ in direction {d}"
        ev = engine.error_vectors(d)
        assert np.array_equal(ev, q - r[oi])


@pytest.mark.parametrize("engine_name", ENGINES)
def test_golden_vectors_through_the_full_stack(golden, engine_name):
    a, b = PointCloud(golden["a"], golden["na"]), PointCloud(golden["b"], golden["nb"])
    pair = CloudPair(a, b, extent=golden["extent"], nn_engine=engine_name)
    assert np.array_equal(np.asarray(pair.get_left_neighbour_distances()), golden["left_d2"])
    assert np.array_equal(np.asarray(pair.get_right_neighbour_distances()), golden["right_d2"])
    assert np.array_equal(np.asarray(pair.get_left_error_vector()), golden["left_err"])
    assert np.array_equal(np.asarray(pair.get_right_error_vector()), golden["right_err"])
    assert np.array_equal(np.asarray(pair.get_boundary_sqrt_distances()), golden["boundary"])
    for side, is_left in (("left", True), ("right", False)):
        if golden["meta"]["raises"].get(side + "_proj") == "IndexError":
            with pytest.raises(IndexError):
                np.asarray(pair.point_to_plane_column(is_left))
        else:
            assert np.array_equal(np.asarray(pair.point_to_plane_column(is_left)), golden[side + "_proj"])
    for tag in ("h0p0", "h0p1", "h1p0", "h1p1"):
        calc = MetricCalculator(pair)
        metrics = transform_options(CalculateOptions(None, tag[1] == "1", tag[3] == "1"))
        if golden["meta"]["raises"].get(tag) == "IndexError":
            with pytest.raises(IndexError):
                calc.calculate(metrics)
            continue
        with np.errstate(divide="ignore"):
            res = calc.calculate(metrics)
        for key, val in golden["meta"]["results"][tag]:
            assert same_bits(res.as_dict()[tuple(key)], val), (tag, key)
        assert res.as_df().to_string() == golden["meta"]["texts"][tag]["string"]


def _report_vs_oracle(n, engine_name, kind="uniform32", normal_index="row"):
    a, b = clouds(kind, n, n, seed=1234)
    na, nb = unit_normals(n, 4321), unit_normals(n, 8765)
    pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], nn_engine=engine_name,
                     normal_index=normal_index)
    res = MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, True, True))).as_dict()
    o = orc.OraclePair(a, b, na, nb, method="kdtree", normal_index=normal_index)
    want = o.report(hausdorff=True, point_to_plane_=True, peak=1.0)
    assert list(res.keys()) == list(want.keys())
    for k in want:
        assert same_bits(res[k], want[k]), (k, res[k], want[k])
    return pair, o


@pytest.mark.parametrize("engine_name", ENGINES)
def test_config0_10k_full_report_bit_exact(engine_name):
    _report_vs_oracle(10_000, engine_name)


@pytest.mark.parametrize("engine_name", ENGINES)
def test_100k_neighbour_normals_full_report_bit_exact(engine_name):
    _report_vs_oracle(100_000, engine_name, kind="voxel10", normal_index="neighbour")


@pytest.mark.parametrize("engine_name", ENGINES)
def test_config1_config2_1m_bit_exact_and_properties(engine_name):
    # BASELINE.json configs[1] and [2]: 1M vs 1M uniform fp32, D1 + Hausdorff, D2 with normals
    pair, o = _report_vs_oracle(1_000_000, engine_name)
    # size-independent properties at full size:
    col = pair.get_left_neighbour_distances()
    host = np.asarray(col)
    assert np.sum(col, axis=0) == np.sum(host, axis=0)             # fused sum == NumPy's pairwise sum
    assert np.max(col, axis=0) == host.max() and np.min(col) == host.min()
    idx = pair._neighbour_index(0)
    assert np.array_equal(idx, o.nn_idx[0])
    # neighbour relation is a valid witness: d2 recomputed from the returned rows, in fp64
    a64, b64 = o.points
    d = a64 - b64[idx]
    assert np.array_equal((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2], host)
    # determinism: a second pair on the same data gives identical bits
    pair2 = CloudPair(pair.clouds[0], pair.clouds[1], extent=[1, 1, 1], nn_engine=engine_name)
    assert np.array_equal(np.asarray(pair2.get_right_neighbour_distances()), np.asarray(pair.get_right_neighbour_distances()))
    assert np.array_equal(pair2._neighbour_index(1), pair._neighbour_index(1))


@pytest.mark.parametrize("engine_name", ENGINES)
@pytest.mark.parametrize("n,world", [(20_000, 2), (50_001, 4), (300, 8)])
def test_query_axis_shards_reassemble_bit_exactly(engine, engine_name, n, world):
    a, b = clouds("uniform32", n, n + 13, seed=n)
    nb = unit_normals(n + 13, 5)
    engine.set_cloud(0, a)
    engine.set_cloud(1, b)
    engine.set_normals(1, nb)
    engine.set_shard(0, 1)
    engine.nn(0, engine_name)
    full_idx, full_d2 = engine.fetch_nn(0)
    want = {}
    for metric in (nat.METRIC_D1, nat.METRIC_D2):
        xvec, mn, mx = engine.reduce(0, metric)
        want[metric] = (nat.finish_sum(xvec, n), mn, mx)
        assert want[metric][0] == np.sum(engine.point_metric(0, metric))
    parts_idx, parts_d2 = [], []
    acc = {m: [np.zeros(nat.xvec_len(n)), np.inf, -np.inf] for m in want}
    for r in range(world):
        engine.set_shard(r, world)
        engine.nn(0, engine_name)
        i, d = engine.fetch_nn(0)
        parts_idx.append(i)
        parts_d2.append(d)
        for m in want:
            xvec, mn, mx = engine.reduce(0, m)
            acc[m][0] += xvec                      # what the RCCL all-reduce(SUM) does: x + 0 is exact
            acc[m][1] = min(acc[m][1], mn)
            acc[m][2] = max(acc[m][2], mx)
    engine.set_shard(0, 1)
    assert np.array_equal(np.concatenate(parts_idx), full_idx)
    assert np.array_equal(np.concatenate(parts_d2), full_d2)
    for m in want:
        assert (nat.finish_sum(acc[m][0], n), acc[m][1], acc[m][2]) == want[m]


def test_bad_inputs_fail_loudly(engine):
    good = np.random.default_rng(0).random((10, 3))
    with pytest.raises(ValueError):
        engine.set_cloud(0, np.zeros((0, 3)))
    bad = good.copy()
    bad[3, 1] = np.nan
    with pytest.raises(ValueError):
        engine.set_cloud(0, bad)
    bad[3, 1] = 1e16
    with pytest.raises(ValueError):
        engine.set_cloud(0, bad)
    with pytest.raises(ValueError):
        engine.set_cloud(2, good)
    engine.set_cloud(0, good)
    engine.set_cloud(1, good)
    with pytest.raises(RuntimeError):
        engine.fetch_nn(0)                          # nn has not run for these clouds
    engine.nn(0, "auto")
    with pytest.raises(RuntimeError):
        engine.point_metric(0, nat.METRIC_D2)       # no normals


def test_device_resident_inputs(engine):
    torch = pytest.importorskip("torch")
    a, b = clouds("uniform32", 3000, 3100, seed=9)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    engine.set_cloud(0, ta)
    engine.set_cloud(1, tb)
    engine.nn(1, "auto")
    idx, d2 = engine.fetch_nn(1)
    oi, od = orc.nn(b.astype(np.float64), a.astype(np.float64), method="kdtree")
    assert np.array_equal(idx, oi) and np.array_equal(d2, od)
