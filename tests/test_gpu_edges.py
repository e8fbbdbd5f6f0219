"""Edge cases of the GPU path: degenerate geometry, skewed sizes, large offsets, duplicates, hipGraph
capture/replay semantics.  Everything is compared bit for bit with the CPU oracle."""
import numpy as np
import pytest

from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    e = nat.Engine(0)
    yield e
    e.close()


def assert_all_directions(engine, a, b, engines=("grid", "brute")):
    a64, b64 = np.asarray(a, np.float64), np.asarray(b, np.float64)
    engine.set_cloud(0, a)
    engine.set_cloud(1, b)
    for name in engines:
        for d, (q, r, skip) in enumerate(((a64, b64, False), (b64, a64, False), (a64, a64, True))):
            engine.nn(d, name)
            idx, d2 = engine.fetch_nn(d)
            if skip and len(a64) < 2:
                assert np.all(idx == -1) and np.all(d2 == 0)
                continue
            oi, od = orc.nn(q, r, skip_same_index=skip, method="kdtree")
            assert np.array_equal(d2, od), (name, d)
            assert np.array_equal(idx, oi), (name, d)


def test_planar_and_collinear_clouds(engine):
    rng = np.random.default_rng(1)
    plane_a = rng.random((4000, 3)); plane_a[:, 2] = 0.25
    plane_b = rng.random((3500, 3)); plane_b[:, 2] = 0.25
    assert_all_directions(engine, plane_a, plane_b)
    line_a = np.zeros((3000, 3)); line_a[:, 0] = rng.random(3000)
    line_b = np.zeros((2000, 3)); line_b[:, 0] = rng.random(2000); line_b[:, 1] = 1e-3
    assert_all_directions(engine, line_a, line_b)


def test_all_points_identical_and_heavy_duplicates(engine):
    same = np.tile(np.array([[1.5, -2.0, 3.25]]), (2000, 1))
    assert_all_directions(engine, same, same[:1500])
    rng = np.random.default_rng(2)
    base = rng.random((50, 3), dtype=np.float32)
    a = base[rng.integers(0, 50, 5000)]                       # 100 copies of each of 50 points
    b = base[rng.integers(0, 50, 4000)] + np.float32(1e-3)
    assert_all_directions(engine, a, b)


def test_skewed_sizes(engine):
    rng = np.random.default_rng(3)
    assert_all_directions(engine, rng.random((200000, 3), dtype=np.float32), rng.random((37, 3), dtype=np.float32), engines=("grid",))
    assert_all_directions(engine, rng.random((1, 3)), rng.random((50000, 3)), engines=("grid",))


def test_large_offsets_and_scales(engine):
    rng = np.random.default_rng(4)
    a = rng.random((6000, 3)) * 1e-6 + 1e9                    # fp64 detail far below fp32 resolution
    b = a[rng.integers(0, 6000, 5000)] + rng.normal(0, 1e-7, (5000, 3))
    assert_all_directions(engine, a, b)
    a = (rng.random((5000, 3)) - 0.5) * 1e12
    b = (rng.random((5000, 3)) - 0.5) * 1e12
    assert_all_directions(engine, a, b)
    a = rng.random((5000, 3)) * 1e-30
    b = rng.random((5000, 3)) * 1e-30
    assert_all_directions(engine, a, b)


def test_clusters_with_voids(engine):
    rng = np.random.default_rng(5)
    centers = rng.random((6, 3)) * 100
    a = np.concatenate([c + rng.normal(0, 0.05, (2000, 3)) for c in centers])
    b = np.concatenate([c + rng.normal(0, 0.05, (1500, 3)) for c in centers[:4]])   # two clusters have no partner
    assert_all_directions(engine, a, b)


def test_graph_capture_replay_and_staleness(engine):
    rng = np.random.default_rng(6)
    a, b = rng.random((30000, 3), dtype=np.float32), rng.random((31000, 3), dtype=np.float32)
    engine.set_cloud(0, a)
    engine.set_cloud(1, b)
    engine.graph_begin()
    with pytest.raises(RuntimeError):                         # nothing ran yet: capture cannot allocate
        engine.nn_pair("grid")
    with pytest.raises(RuntimeError):                         # ... and the capture reports it was abandoned
        engine.graph_end()
    engine.drop_caches(); engine.nn_pair("grid"); engine.reduce_prefetch_many([(0, 0), (1, 0)])
    want = [engine.reduce_total(d, 0) for d in (0, 1)]
    engine.graph_begin()
    engine.drop_caches(); engine.nn_pair("grid"); engine.reduce_prefetch_many([(0, 0), (1, 0)])
    gid = engine.graph_end()
    assert [engine.reduce_total(d, 0) for d in (0, 1)] == want
    for _ in range(3):
        engine.graph_launch(gid)
        assert [engine.reduce_total(d, 0) for d in (0, 1)] == want
        idx, d2 = engine.fetch_nn(1)
        oi, od = orc.nn(b.astype(np.float64), a.astype(np.float64), method="kdtree")
        assert np.array_equal(idx, oi) and np.array_equal(d2, od)
    engine.graph_begin()
    with pytest.raises(RuntimeError):                         # fetch is not a capturable call
        engine.fetch_nn(0)
    engine.graph_abort()
    engine.set_cloud(1, a[:1000])                             # new input: the recorded graph is stale
    with pytest.raises((RuntimeError, ValueError)):
        engine.graph_launch(gid)
    engine.nn_pair("grid")                                    # and the context still works
    idx, d2 = engine.fetch_nn(0)
    oi, od = orc.nn(a.astype(np.float64), a[:1000].astype(np.float64), method="kdtree")
    assert np.array_equal(idx, oi) and np.array_equal(d2, od)


def test_recompute_with_graph_matches_eager():
    rng = np.random.default_rng(7)
    a, b = rng.random((20000, 3), dtype=np.float32), rng.random((20000, 3), dtype=np.float32)
    na, nb = rng.standard_normal((20000, 3)), rng.standard_normal((20000, 3))
    opts = CalculateOptions(None, True, True)
    eager = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1, 1, 1])
    want = MetricCalculator(eager).calculate(transform_options(opts)).as_dict()
    pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1, 1, 1], use_graph=True)
    for _ in range(5):                                        # eager, capture, replays
        pair.recompute()
        got = MetricCalculator(pair).calculate(transform_options(opts)).as_dict()
        assert got == want
    assert pair._graph_id is not None
    MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, False, False)))   # a different report
    pair.recompute()
    assert MetricCalculator(pair).calculate(transform_options(opts)).as_dict() == want


# ---- hostile distributions: automatic engine choice, trimmed grid box, local-origin fp32 ------------------------
def _hostile(name, n):
    u = lambda s: np.random.default_rng(s).random((n, 3), dtype=np.float32).astype(np.float64)
    if name == "one_outlier":
        a, b = u(1), u(2); a[0] = [1e6, 1e6, 1e6]
    elif name == "outliers_inexact":                 # not fp32-representable: the SHIFT kernel + per-query slack
        a, b = u(1), u(2); a[:5] *= 1e4; b[:5] *= -1e4; a += 1e-9
    elif name == "utm_offset":                       # geo-referenced fp64 coordinates at mm noise
        a = u(11) * 100 + np.array([5.0e5, 5.6e6, 300.0]); b = a + np.random.default_rng(12).normal(0, 0.01, a.shape)
    elif name == "half_overlap":
        a, b = u(15), u(16) + np.array([0.5, 0, 0])
    elif name == "disjoint":
        a, b = u(13), u(14) + np.array([3.0, 0, 0])
    elif name == "gauss_clump":
        g = np.random.default_rng(8); a, b = g.normal(0, 1, (n, 3)) ** 3, g.normal(0, 1, (n, 3)) ** 3
    elif name == "two_clusters":
        a, b = u(6) * 0.01, u(7) * 0.01; a[n // 2:] += 1000; b[n // 2:] += 1000
    elif name == "heavy_tail_box":                   # 1 % of the points far outside: the box is trimmed, they clamp
        a, b = u(20), u(21); a[: n // 100] = a[: n // 100] * 50 - 25; b[: n // 100] = b[: n // 100] * 50 - 25
    else:
        raise KeyError(name)
    return a, b


@pytest.mark.parametrize("name", ["one_outlier", "outliers_inexact", "utm_offset", "half_overlap", "disjoint", "gauss_clump",
                                  "two_clusters", "heavy_tail_box"])
@pytest.mark.parametrize("eng", ["auto", "grid"])
def test_hostile_distributions_stay_exact(engine, name, eng):
    a, b = _hostile(name, 30000)
    engine.set_cloud(0, a)
    engine.set_cloud(1, b)
    engine.nn_pair(eng)
    for d, (q, r) in enumerate(((a, b), (b, a))):
        idx, d2 = engine.fetch_nn(d)
        oi, od = orc.nn(q, r, method="kdtree")
        assert np.array_equal(d2, od), (name, eng, d)
        assert np.array_equal(idx, oi), (name, eng, d)
    engine.nn(nat.DIR_SELF, eng)
    idx, d2 = engine.fetch_nn(nat.DIR_SELF)
    oi, od = orc.nn(a, a, skip_same_index=True, method="kdtree")
    assert np.array_equal(d2, od) and np.array_equal(idx, oi), (name, eng, "self")


def test_automatic_engine_choice(engine):
    """PCCM_ENGINE_AUTO: the grid for ordinary pairs; the brute-force engine (stats['pairs'] = nq * nr) when the
    clouds overlap only in part -- the decision is taken once per pair and survives pccm_drop_caches."""
    n = 30000
    a, b = _hostile("half_overlap", n)
    engine.set_cloud(0, a); engine.set_cloud(1, b)
    for _ in range(2):
        engine.drop_caches()
        engine.nn_pair("auto")
        assert engine.nn_stats(0)["pairs"] == n * n
    engine.nn_pair("grid")
    assert engine.nn_stats(0)["pairs"] == 0
    a, b = _hostile("one_outlier", n)                 # trimmed box: stays on the grid, one exact rescan
    engine.set_cloud(0, a); engine.set_cloud(1, b)
    engine.nn_pair("auto")
    assert engine.nn_stats(0)["pairs"] == 0 and engine.nn_stats(0)["fallback_queries"] <= 2


def test_graph_replay_on_a_pair_that_takes_the_brute_engine():
    """use_graph with a partly overlapping pair (automatic choice: brute force): every recompute() -- eager, captured,
    replayed -- must give the oracle's report."""
    from oracle.oracle import OraclePair
    n = 20000
    a, b = _hostile("half_overlap", n)
    rng = np.random.default_rng(5)
    na = rng.standard_normal((n, 3)); nb = rng.standard_normal((n, 3))
    opts = transform_options(CalculateOptions(None, True, True))
    want = OraclePair(a, b, na, nb).report(hausdorff=True, point_to_plane_=True, peak=1.0)
    pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1, 1, 1], use_graph=True)
    assert pair._engine.nn_stats(0)["pairs"] == n * n            # the brute engine ran
    for _ in range(4):
        got = MetricCalculator(pair).calculate(opts).as_dict()
        assert list(got) == list(want)
        for k in want:
            assert np.array_equal(np.float64(got[k]), np.float64(want[k])), k
        pair.recompute()


def test_inherited_decisions_never_change_results(engine):
    """A pair that looks like the previous one (point counts, bounding boxes, coordinate kind) inherits its grid
    decisions; if the look deceives -- same box, utterly different content -- only speed may suffer."""
    n = 30000
    rng = np.random.default_rng(21)
    corners = np.array([[x, y, z] for x in (0.0, 1.0) for y in (0.0, 1.0) for z in (0.0, 1.0)])

    def boxed(p):
        p = p.astype(np.float32).astype(np.float64)
        p[:8] = corners
        return p

    uniform = lambda: boxed(rng.random((n, 3)))
    clumps = lambda: boxed(0.5 + 0.001 * rng.standard_normal((n, 3)))
    halves = lambda lo: boxed(np.column_stack([lo + 0.4 * rng.random(n), rng.random(n), rng.random(n)]))
    pairs = [(uniform(), uniform()), (clumps(), clumps()), (halves(0.0), halves(0.6)), (uniform(), clumps())]
    for a, b in pairs:
        engine.set_cloud(0, a); engine.set_cloud(1, b)
        engine.nn_pair("auto")
        for d, (q, r) in enumerate(((a, b), (b, a))):
            idx, d2 = engine.fetch_nn(d)
            oi, od = orc.nn(q, r, method="kdtree")
            assert np.array_equal(d2, od) and np.array_equal(idx, oi)


def test_pooled_context_starts_clean():
    """CloudPair borrows its context from a pool (_native.acquire_engine): whatever the previous pair left behind --
    normals, colours, shard, graphs -- must be gone, and results must not depend on what ran before."""
    rng = np.random.default_rng(31)
    a, b = rng.random((5000, 3)), rng.random((5000, 3))
    na, nb = rng.standard_normal((5000, 3)), rng.standard_normal((5000, 3))
    opts = transform_options(CalculateOptions(None, True, True))
    nat.drain_pool()
    with CloudPair(PointCloud(a, na, a), PointCloud(b, nb, b), extent=[1, 1, 1], use_graph=True) as first:
        want = MetricCalculator(first).calculate(opts).as_dict()
        first.recompute(); first.recompute()                       # leaves a captured graph behind
        ctx_id = first._engine._ctx.value
    with CloudPair(PointCloud(a), PointCloud(b), extent=[1, 1, 1], estimate_normals=False) as bare:
        assert bare._engine._ctx.value == ctx_id                   # the same context, reused
        with pytest.raises(ValueError, match="normals"):           # ... without the first pair's normals
            MetricCalculator(bare).calculate(transform_options(CalculateOptions(None, False, True)))
        with pytest.raises(RuntimeError):                          # ... and without its colours
            bare._engine.color_reduce(nat.DIR_LEFT, "rgb")
    with CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1, 1, 1]) as again:
        got = MetricCalculator(again).calculate(opts).as_dict()
    assert list(got) == list(want) and all(np.float64(got[k]) == np.float64(want[k]) for k in want)


def test_threads_sharing_a_context_are_serialised(engine):
    """One context is meant for one caller at a time; threads that share it anyway (ctypes releases the GIL inside every
    call) queue on the context's mutex instead of corrupting it."""
    import threading
    rng = np.random.default_rng(41)
    a, b = rng.random((20000, 3)), rng.random((20000, 3))
    engine.set_cloud(0, a); engine.set_cloud(1, b)
    oi, od = orc.nn(a, b, method="kdtree")
    errors = []

    def worker():
        try:
            for _ in range(20):
                engine.drop_caches()
                engine.nn(nat.DIR_LEFT, "grid")
                idx, d2 = engine.fetch_nn(nat.DIR_LEFT)
                if not (np.array_equal(idx, oi) and np.array_equal(d2, od)):
                    errors.append("mismatch")
        except Exception as ex:            # noqa: BLE001
            errors.append(repr(ex))

    threads = [threading.Thread(target=worker) for _ in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_two_pairs_in_two_threads_run_concurrently():
    """Different contexts are independent: two host threads, one CloudPair each, reports in parallel."""
    import threading
    results, errors = {}, []

    def worker(seed):
        try:
            opts = transform_options(CalculateOptions(None, True, True))      # metric objects carry values: one set per thread
            rng = np.random.default_rng(seed)
            a, b = rng.random((15000, 3)), rng.random((15000, 3))
            na, nb = rng.standard_normal((15000, 3)), rng.standard_normal((15000, 3))
            want = orc.OraclePair(a, b, na, nb, method="kdtree").report(hausdorff=True, point_to_plane_=True, peak=1.0)
            for _ in range(5):
                with CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1, 1, 1]) as pair:
                    got = MetricCalculator(pair).calculate(opts).as_dict()
                if any(np.float64(got[k]) != np.float64(want[k]) for k in want):
                    errors.append(f"mismatch in thread {seed}")
            results[seed] = True
        except Exception as ex:            # noqa: BLE001
            errors.append(repr(ex))

    threads = [threading.Thread(target=worker, args=(s,)) for s in (51, 52)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors and len(results) == 2, errors[:3]
