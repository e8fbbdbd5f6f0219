"""GPU-side behaviour added in round 2: the projection fused into the search (pccm_nn_fuse) against the separate
point kernel, result records against the plain columns, per-direction shards at the C ABI, the LDS-brick kernel's
fall-backs (clumped bricks, leftovers), error propagation."""
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


@pytest.fixture()
def engine():
    e = nat.Engine(0)
    yield e
    e.close()


def _unit(n, seed):
    g = np.random.default_rng(seed).standard_normal((n, 3), dtype=np.float32)
    return (g / np.linalg.norm(g, axis=1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("mode", ["row", "neighbour"])
@pytest.mark.parametrize("n,m", [(60_000, 60_000), (60_000, 75_000), (500, 700)])
def test_fused_projection_equals_the_separate_pass(engine, mode, n, m):
    """pccm_nn_fuse is purely an optimisation: D2 sums / maxima / columns are bit-identical with it on and off, and equal
    the oracle's (the larger cloud iterates in "row" mode only where the reference does not raise, quirk Q1)."""
    rng = np.random.default_rng(n + m)
    a, b = rng.random((n, 3), dtype=np.float32), rng.random((m, 3), dtype=np.float32)
    na, nb = _unit(n, 1), _unit(m, 2)
    engine.set_cloud(0, a); engine.set_cloud(1, b)
    engine.set_normals(0, na); engine.set_normals(1, nb)
    dirs = [0, 1] if mode == "neighbour" or n == m else [0 if n <= m else 1]     # rows in range only
    got = {}
    for fuse in (False, True):
        for d in (0, 1):
            engine.nn_fuse(d, mode if fuse else None)
        engine.drop_caches(); engine.nn_pair("grid")
        got[fuse] = {d: (engine.reduce_total(d, nat.METRIC_D2, mode), engine.reduce_total(d, nat.METRIC_D1, mode),
                         engine.point_metric(d, nat.METRIC_D2, mode), engine.point_metric(d, nat.METRIC_PROJ, mode)) for d in dirs}
    for d in dirs:
        assert got[True][d][0] == got[False][d][0] and got[True][d][1] == got[False][d][1]
        assert np.array_equal(got[True][d][2], got[False][d][2]) and np.array_equal(got[True][d][3], got[False][d][3])
        it, se, nrm = (a, b, nb) if d == 0 else (b, a, na)
        idx, d2 = orc.nn(it.astype(np.float64), se.astype(np.float64), method="kdtree")
        proj = orc.point_to_plane(it.astype(np.float64), se.astype(np.float64), idx, nrm.astype(np.float64), normal_index=mode)
        assert np.array_equal(got[True][d][3], proj)
        assert same_bits(got[True][d][0][0], np.sum(np.square(proj))) and same_bits(got[True][d][0][2], np.max(np.square(proj)))
        assert same_bits(got[True][d][1][0], np.sum(d2))
        fi, fd = engine.fetch_nn(d)
        assert np.array_equal(fi, idx) and np.array_equal(fd, d2)            # result records unpack to the plain columns


@pytest.mark.parametrize("mode", ["row", "neighbour"])
def test_normals_in_fp32_words_or_fp64_rows_same_projection(engine, mode):
    """The brick kernel gathers fp32-exact normals as one aligned 16-byte word and anything else as three fp64 values:
    float32 input, float64 input that happens to be fp32-exact, and float64 input that is not all give the oracle's
    projections bit for bit -- and one inexact component is enough to leave the short form."""
    n = 50_000
    rng = np.random.default_rng(77)
    a, b = rng.random((n, 3), dtype=np.float32), rng.random((n, 3), dtype=np.float32)
    na32, nb32 = _unit(n, 5), _unit(n, 6)
    noisy_a = na32.astype(np.float64) + rng.normal(0, 1e-9, (n, 3))
    one_off_b = nb32.astype(np.float64)
    one_off_b[n // 2, 1] += 2.0 ** -40
    want = {}
    for name, (xa, xb) in {"f32": (na32, nb32), "f64 exact": (na32.astype(np.float64), nb32.astype(np.float64)),
                           "f64 inexact": (noisy_a, one_off_b)}.items():
        engine.set_cloud(0, a); engine.set_cloud(1, b)
        engine.set_normals(0, xa); engine.set_normals(1, xb)
        for d in (0, 1):
            engine.nn_fuse(d, mode)
        engine.drop_caches(); engine.nn_pair("grid")
        for d in (0, 1):
            it, se, nrm = (a, b, xb) if d == 0 else (b, a, xa)
            key = (d, name if name == "f64 inexact" else "exact")
            if key not in want:
                idx, _ = orc.nn(it.astype(np.float64), se.astype(np.float64), method="kdtree")
                want[key] = orc.point_to_plane(it.astype(np.float64), se.astype(np.float64), idx, np.asarray(nrm, dtype=np.float64),
                                               normal_index=mode)
            proj = want[key]
            assert np.array_equal(engine.point_metric(d, nat.METRIC_PROJ, mode), proj), (name, d)
            total = engine.reduce_total(d, nat.METRIC_D2, mode)
            assert same_bits(total[0], np.sum(np.square(proj))) and same_bits(total[2], np.max(np.square(proj))), (name, d)
    assert not np.array_equal(want[(1, "exact")], want[(1, "f64 inexact")])       # the perturbation is visible in the result


@pytest.mark.parametrize("want_idx", [False, True])
def test_reduction_batches_of_every_shape(engine, want_idx):
    """The reduction kernel is instantiated per shape of a batch (record stride x which fields feed which columns) and falls
    back to the general kernel for batches whose jobs differ or that reduce the signed projection: D1 alone, D2 alone, both,
    D1 of one direction with D2 of the other, and the signed projections all equal NumPy on the materialised columns."""
    rng = np.random.default_rng(41)
    n, m = 70_000, 66_000
    a, b = rng.random((n, 3), dtype=np.float32), rng.random((m, 3), dtype=np.float32)
    engine.set_cloud(0, a); engine.set_cloud(1, b)
    engine.set_normals(0, _unit(n, 1)); engine.set_normals(1, _unit(m, 2))
    engine.nn_want_idx(want_idx)                     # 32- or 16-byte result records
    for d in (0, 1):
        engine.nn_fuse(d, "neighbour")
    cols = {}
    engine.nn_pair("grid")
    for d in (0, 1):
        cols[(d, nat.METRIC_D1)] = engine.fetch_nn(d, want_idx=False)[1]
        cols[(d, nat.METRIC_D2)] = engine.point_metric(d, nat.METRIC_D2, "neighbour")
        cols[(d, nat.METRIC_PROJ)] = engine.point_metric(d, nat.METRIC_PROJ, "neighbour")
    batches = [[(0, nat.METRIC_D1)], [(1, nat.METRIC_D2)], [(0, nat.METRIC_D1), (0, nat.METRIC_D2), (1, nat.METRIC_D1), (1, nat.METRIC_D2)],
               [(0, nat.METRIC_D1), (1, nat.METRIC_D2)], [(0, nat.METRIC_PROJ), (1, nat.METRIC_PROJ)],
               [(0, nat.METRIC_D1), (0, nat.METRIC_PROJ)], [(1, nat.METRIC_D1), (1, nat.METRIC_D1)]]
    for batch in batches:
        engine.drop_caches(); engine.nn_pair("grid")                       # fresh results, no slot left over
        got = engine.reduce_total_many(batch, "neighbour")
        for (d, met), (total, mn, mx) in zip(batch, got):
            col = cols[(d, met)]
            assert same_bits(total, np.sum(col)) and mn == np.min(col) and mx == np.max(col), (batch, d, met)
    engine.nn_want_idx(True)


def test_results_without_rows_and_rows_on_demand(engine):
    """pccm_nn_want_idx off: 16-byte result records, identical reductions; the first caller that asks for the matched rows
    gets them from a repeated search of that direction -- same bits as with the rows on from the start."""
    rng = np.random.default_rng(21)
    n, m = 90_000, 80_000
    a, b = rng.random((n, 3), dtype=np.float32), rng.random((m, 3), dtype=np.float32)
    engine.set_cloud(0, a); engine.set_cloud(1, b)
    engine.set_normals(0, _unit(n, 1)); engine.set_normals(1, _unit(m, 2))
    for d in (0, 1):
        engine.nn_fuse(d, "neighbour")
    engine.nn_want_idx(True)
    engine.nn_pair("grid")
    want = {d: (engine.reduce_total(d, nat.METRIC_D1, "neighbour"), engine.reduce_total(d, nat.METRIC_D2, "neighbour"), engine.fetch_nn(d),
                engine.error_vectors(d)) for d in (0, 1)}
    engine.nn_want_idx(False)
    engine.drop_caches(); engine.nn_pair("grid")
    for d in (0, 1):
        assert engine.reduce_total(d, nat.METRIC_D1, "neighbour") == want[d][0]
        assert engine.reduce_total(d, nat.METRIC_D2, "neighbour") == want[d][1]
        _, d2 = engine.fetch_nn(d, want_idx=False)                  # distances alone: no second search
        assert np.array_equal(d2, want[d][2][1])
    for d in (0, 1):
        idx, d2 = engine.fetch_nn(d)                                # rows on demand
        assert np.array_equal(idx, want[d][2][0]) and np.array_equal(d2, want[d][2][1])
        assert np.array_equal(engine.error_vectors(d), want[d][3])
        assert engine.reduce_total(d, nat.METRIC_D2, "neighbour") == want[d][1]
    engine.nn_want_idx(True)


def test_row_mode_out_of_range_is_not_fused_and_raises(engine):
    rng = np.random.default_rng(3)
    a, b = rng.random((3000, 3), dtype=np.float32), rng.random((2000, 3), dtype=np.float32)
    engine.set_cloud(0, a); engine.set_cloud(1, b)
    engine.set_normals(0, _unit(3000, 1)); engine.set_normals(1, _unit(2000, 2))
    for d in (0, 1):
        engine.nn_fuse(d, "row")
    engine.nn_pair("grid")
    with pytest.raises(IndexError):                                # A iterates 3000 rows over B's 2000 normals (quirk Q1)
        engine.reduce_total(0, nat.METRIC_D2, "row")
    total, _, _ = engine.reduce_total(1, nat.METRIC_D2, "row")     # the other direction is fine (and fused)
    idx, _ = orc.nn(b.astype(np.float64), a.astype(np.float64), method="kdtree")
    proj = orc.point_to_plane(b.astype(np.float64), a.astype(np.float64), idx, _unit(3000, 1).astype(np.float64))
    assert same_bits(total, np.sum(np.square(proj)))


def test_per_direction_shards_at_the_c_abi(engine):
    """pccm_set_shard_dir: four 'ranks' -- two share the left direction, two the right one -- reassemble the unsharded
    result bit for bit; a rank builds only the cloud it searches."""
    rng = np.random.default_rng(11)
    n, m = 40_000, 37_000
    a, b = rng.random((n, 3), dtype=np.float32), rng.random((m, 3), dtype=np.float32)
    engine.set_cloud(0, a); engine.set_cloud(1, b)
    engine.set_normals(0, _unit(n, 1)); engine.set_normals(1, _unit(m, 2))
    for d in (0, 1):
        engine.nn_fuse(d, "neighbour")
    engine.nn_pair("grid")
    want = {d: (engine.fetch_nn(d), engine.reduce(d, nat.METRIC_D2, "neighbour")) for d in (0, 1)}
    plan = {0: [(0, 2), (1, 2), (0, 0), (0, 0)], 1: [(0, 0), (0, 0), (0, 2), (1, 2)]}
    acc = {d: [[], [], np.zeros(nat.xvec_len(n if d == 0 else m)), np.inf, -np.inf] for d in (0, 1)}
    for rank in range(4):
        for d in (0, 1, 2):
            engine.set_shard_dir(d, *(plan[d][rank] if d < 2 else (0, 0)))
        engine.drop_caches(); engine.nn_pair("grid")
        for d in (0, 1):
            b0, e0 = engine.shard_range(d)
            assert (e0 > b0) == (plan[d][rank][1] > 0)
            idx, d2 = engine.fetch_nn(d)
            xvec, mn, mx = engine.reduce(d, nat.METRIC_D2, "neighbour")
            acc[d][0].append(idx); acc[d][1].append(d2)
            acc[d][2] += xvec
            acc[d][3], acc[d][4] = min(acc[d][3], mn), max(acc[d][4], mx)
    for d in (0, 1, 2):
        engine.set_shard_dir(d, 0, 1)
    for d in (0, 1):
        (idx, d2), (xvec, mn, mx) = want[d]
        assert np.array_equal(np.concatenate(acc[d][0]), idx) and np.array_equal(np.concatenate(acc[d][1]), d2)
        assert np.array_equal(acc[d][2], xvec) and acc[d][3] == mn and acc[d][4] == mx


def test_chunk_vectors_of_four_ranks_give_the_unsharded_sums(engine):
    """pccm_reduce_chunks_many / pccm_finish_chunks: shards of clouds with a chunk for every rank start on 8192-row chunks,
    and the ranks' chunk vectors (one number per chunk + the raw tail), summed, finish to np.sum of the whole column --
    bit for bit, for D1 and D2 of both directions in one call."""
    rng = np.random.default_rng(12)
    n, m = 70_000, 45_111
    a, b = rng.random((n, 3), dtype=np.float32), rng.random((m, 3), dtype=np.float32)
    engine.set_cloud(0, a); engine.set_cloud(1, b)
    engine.set_normals(0, _unit(n, 1)); engine.set_normals(1, _unit(m, 2))
    for d in (0, 1):
        engine.nn_fuse(d, "neighbour")
    engine.nn_pair("grid")
    reqs = [(0, nat.METRIC_D1), (0, nat.METRIC_D2), (1, nat.METRIC_D1), (1, nat.METRIC_D2)]
    want = engine.reduce_total_many(reqs, "neighbour")
    world = 4
    acc, lens, mins, maxs = None, None, [np.inf] * 4, [-np.inf] * 4
    for rank in range(world):
        engine.set_shard(rank, world)
        for d in (0, 1):
            b0, e0 = engine.shard_range(d)
            assert b0 % 8192 == 0 and (e0 % 8192 == 0 or e0 == (n if d == 0 else m))
        engine.drop_caches(); engine.nn_pair("grid")
        buf, lens, mms = engine.reduce_chunks_many(reqs, "neighbour")
        acc = buf.copy() if acc is None else acc + buf
        mins = [min(x, mm[0]) for x, mm in zip(mins, mms)]
        maxs = [max(x, mm[1]) for x, mm in zip(maxs, mms)]
    engine.set_shard(0, 1)
    assert lens == [nat.cvec_len(n)] * 2 + [nat.cvec_len(m)] * 2 and nat.cvec_len(n) == n // 8192 + n % 8192
    pos = 0
    for k, ((d, _), ln) in enumerate(zip(reqs, lens)):
        total = engine.finish_chunks(acc[pos:pos + ln], n if d == 0 else m)
        assert same_bits(total, want[k][0]) and mins[k] == want[k][1] and maxs[k] == want[k][2]
        pos += ln
    engine.set_shard(0, 3)                                   # 45 111 rows have no chunk for each of 6 ranks: 128-row leaves
    engine.set_shard(5, 6)
    b0, e0 = engine.shard_range(1)
    assert b0 % 128 == 0 and b0 % 8192 != 0
    engine.drop_caches(); engine.nn_pair("grid")
    with pytest.raises(nat.PccmStateError):
        engine.reduce_chunks_many([(1, nat.METRIC_D1)], "neighbour")
    engine.set_shard(0, 1)


def test_clumped_bricks_and_leftovers_stay_exact(engine, monkeypatch):
    """The LDS-brick kernel hands bricks that exceed its LDS budget to the general kernels and loops over leftover queries
    when a brick holds more than a workgroup: a dense clump inside uniform data exercises both."""
    rng = np.random.default_rng(12)
    a = rng.random((200_000, 3), dtype=np.float32)
    b = rng.random((200_000, 3), dtype=np.float32)
    a[:30_000] = 0.5 + 0.01 * rng.standard_normal((30_000, 3)).astype(np.float32)      # 15 % of A in one small clump
    b[:5_000] = 0.5 + 0.01 * rng.standard_normal((5_000, 3)).astype(np.float32)
    monkeypatch.setenv("PCCM_GRID_COOP", "1")                      # keep the brick kernel whatever the occupancy rule says
    engine.set_cloud(0, a); engine.set_cloud(1, b)
    engine.nn_pair("grid"); engine.nn(2, "grid")
    for d, (q, s, skip) in enumerate(((a, b, False), (b, a, False), (a, a, True))):
        idx, d2 = engine.fetch_nn(d)
        oi, od = orc.nn(q.astype(np.float64), s.astype(np.float64), skip_same_index=skip, method="kdtree")
        assert np.array_equal(d2, od) and np.array_equal(idx, oi)


def test_long_runs_inside_the_lds_budget_stay_exact(engine, monkeypatch):
    """A staged x-run of the brick kernel is copied 64 + 21 records at a time: thin dense lines along x inside uniform
    data give runs of a few hundred records in bricks whose total still fits the LDS budget (the clump test above overflows
    it instead).  Projections ride along (fp32-exact normals: the 16-byte gather)."""
    rng = np.random.default_rng(21)
    n = 250_000
    a = rng.random((n, 3), dtype=np.float32)
    b = rng.random((n, 3), dtype=np.float32)
    for k, (y, z) in enumerate(((0.31, 0.42), (0.77, 0.18), (0.52, 0.93))):             # three lines of 1500 points each, both clouds
        for c in (a, b):
            rows = slice(2000 * k, 2000 * k + 1500)
            c[rows, 0] = rng.random(1500, dtype=np.float32)
            c[rows, 1] = np.float32(y) + np.float32(2e-4) * rng.standard_normal(1500).astype(np.float32)
            c[rows, 2] = np.float32(z) + np.float32(2e-4) * rng.standard_normal(1500).astype(np.float32)
    na, nb = _unit(n, 3), _unit(n, 4)
    monkeypatch.setenv("PCCM_GRID_COOP", "1")
    engine.set_cloud(0, a); engine.set_cloud(1, b)
    engine.set_normals(0, na); engine.set_normals(1, nb)
    for d in (0, 1):
        engine.nn_fuse(d, "row")
    engine.nn_pair("grid"); engine.nn(2, "grid")
    for d, (q, s, nrm, skip) in enumerate(((a, b, nb, False), (b, a, na, False), (a, a, None, True))):
        idx, d2 = engine.fetch_nn(d)
        oi, od = orc.nn(q.astype(np.float64), s.astype(np.float64), skip_same_index=skip, method="kdtree")
        assert np.array_equal(d2, od) and np.array_equal(idx, oi)
        if nrm is not None:
            proj = orc.point_to_plane(q.astype(np.float64), s.astype(np.float64), oi, nrm.astype(np.float64), normal_index="row")
            assert np.array_equal(engine.point_metric(d, nat.METRIC_PROJ, "row"), proj)
            assert same_bits(engine.reduce_total(d, nat.METRIC_D2, "row")[0], np.sum(np.square(proj)))


@pytest.mark.parametrize("margin,note", [(0.012, "several thousand tails: more than one entry per wave at both ends of the wave range"),
                                         (0.05, "tens of thousands: the thread-per-query path")])
def test_many_tail_queries_stay_exact(engine, monkeypatch, margin, note):
    """Queries in a rim where the other cloud has no points cannot be settled by ring 1: they go through k_grid_tail
    (one wave per entry, the two directions' lists handed out from opposite ends) or, past 2^18 entries, through the
    per-thread search; a few fall through to the exact rescan.  Every row is still the oracle's."""
    rng = np.random.default_rng(31)
    n = 300_000
    a = (rng.random((n, 3), dtype=np.float32) * np.float32(1.0 + margin)).astype(np.float32)   # A pokes out of B's box
    b = rng.random((n, 3), dtype=np.float32)
    b[:, 2] = b[:, 2] * np.float32(1.0 + margin)                                               # ... and B out of A's along z only
    a[:, 2] = np.minimum(a[:, 2], np.float32(1.0))
    na, nb = _unit(n, 7), _unit(n, 8)
    monkeypatch.setenv("PCCM_GRID_COOP", "1")
    engine.set_cloud(0, a); engine.set_cloud(1, b)
    engine.set_normals(0, na); engine.set_normals(1, nb)
    for d in (0, 1):
        engine.nn_fuse(d, "row")
    engine.nn_pair("grid")
    for d, (q, s, nrm) in enumerate(((a, b, nb), (b, a, na))):
        idx, d2 = engine.fetch_nn(d)
        oi, od = orc.nn(q.astype(np.float64), s.astype(np.float64), method="kdtree")
        assert np.array_equal(d2, od) and np.array_equal(idx, oi), note
        proj = orc.point_to_plane(q.astype(np.float64), s.astype(np.float64), oi, nrm.astype(np.float64), normal_index="row")
        assert np.array_equal(engine.point_metric(d, nat.METRIC_PROJ, "row"), proj), note
        assert same_bits(engine.reduce_total(d, nat.METRIC_D2, "row")[0], np.sum(np.square(proj))), note


def test_cloud_pair_report_identical_with_and_without_fusion(monkeypatch):
    rng = np.random.default_rng(13)
    n = 120_000
    a, b = rng.random((n, 3), dtype=np.float32), rng.random((n, 3), dtype=np.float32)
    na, nb = _unit(n, 5), _unit(n, 6)
    opts = CalculateOptions(None, True, True)
    with CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1, 1, 1]) as pair:
        fused = MetricCalculator(pair).calculate(transform_options(opts)).as_dict()
        assert "point" not in {k for k in nat.KERNEL_CLASSES if pair._engine.profile_get(k)[1]}
    monkeypatch.setenv("PCCM_NO_FUSE", "1")
    import subprocess, sys, json, os
    code = ("import numpy as np, json, sys; sys.path.insert(0, %r)\n"
            "from open_pcc_metric_amd.calculator import MetricCalculator\nfrom open_pcc_metric_amd.cloud_pair import CloudPair\n"
            "from open_pcc_metric_amd.options import CalculateOptions, transform_options\nfrom open_pcc_metric_amd.point_cloud import PointCloud\n"
            "rng = np.random.default_rng(13); n = %d\n"
            "a, b = rng.random((n, 3), dtype=np.float32), rng.random((n, 3), dtype=np.float32)\n"
            "def unit(n, seed):\n    g = np.random.default_rng(seed).standard_normal((n, 3), dtype=np.float32)\n"
            "    return (g / np.linalg.norm(g, axis=1, keepdims=True)).astype(np.float32)\n"
            "pair = CloudPair(PointCloud(a, unit(n, 5)), PointCloud(b, unit(n, 6)), extent=[1, 1, 1])\n"
            "res = MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, True, True))).as_dict()\n"
            "print(json.dumps([[list(map(str, k)), float(v).hex()] for k, v in res.items()]))\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), n)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True, env=dict(os.environ, PCCM_NO_FUSE="1")).stdout
    unfused = json.loads(out.strip().splitlines()[-1])
    assert unfused == [[list(map(str, k)), float(v).hex()] for k, v in fused.items()]


def test_device_errors_are_not_retried(monkeypatch):
    """CloudPair.recompute() falls back to eager launches only on PCCM_E_STATE (a stale graph); any other
    failure of the replay surfaces."""
    rng = np.random.default_rng(14)
    a, b = rng.random((5000, 3), dtype=np.float32), rng.random((5000, 3), dtype=np.float32)
    pair = CloudPair(PointCloud(a), PointCloud(b), extent=[1, 1, 1], use_graph=True)
    opts = transform_options(CalculateOptions(None, True, False))
    for _ in range(3):
        MetricCalculator(pair).calculate(opts)
        pair.recompute()
    assert pair._graph_id is not None
    eng = pair._engine
    calls = {"eager": 0}
    real_nn_pair = eng.nn_pair
    monkeypatch.setattr(eng, "nn_pair", lambda *a_, **k: (calls.__setitem__("eager", calls["eager"] + 1), real_nn_pair(*a_, **k))[1])
    monkeypatch.setattr(eng, "graph_launch", lambda gid: (_ for _ in ()).throw(nat.PccmDeviceError("libpccm error -3: hipGraphLaunch: injected")))
    with pytest.raises(nat.PccmDeviceError):
        pair.recompute()
    assert calls["eager"] == 0                                      # not retried eagerly, not hidden
    monkeypatch.setattr(eng, "graph_launch", lambda gid: (_ for _ in ()).throw(nat.PccmStateError("libpccm error -5: graph is stale")))
    pair.recompute()                                                # a stale graph: eager rerun, capture again later
    assert calls["eager"] == 1 and pair._graph_id is None
    pair.close()
