"""Colour metrics on the GPU: pccm_set_colors / pccm_color_reduce / pccm_color_rows
through the C ABI against the reference's golden values, the oracle, and plain NumPy."""
import numpy as np
import pytest

from conftest import load_golden, same_bits
import open_pcc_metric_amd.metric as opmm
from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

COLOUR_CLASSES = ("ColorMSE", "ColorPSNR", "ColorHausdorffDistance", "ColorHausdorffDistancePSNR")


@pytest.fixture(scope="module")
def engine():
    e = nat.Engine(0)
    yield e
    e.close()


# ---- the summation kernel on its own ---------------------------------------------------------------------
def _columns(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return rng.random((n, 3)) ** 2
    if kind == "wide":                       # 60 binades of dynamic range: crossings everywhere
        return rng.random((n, 3)) * np.exp2(rng.integers(-40, 20, (n, 3)))
    if kind == "ties":                       # few significant bits: exact half-way cases against the running sum
        return rng.integers(0, 4, (n, 3)) * np.exp2(rng.integers(-54, -48, (n, 3))) + (rng.random((n, 3)) < 0.01)
    if kind == "zeros":
        a = np.zeros((n, 3))
        a[n // 2:, 1] = rng.random(n - n // 2)
        a[-1, 2] = 1e-300
        return a
    if kind == "allzero":                    # identical colours on both sides: nothing but +0 (any binade takes a run of zeros)
        return np.zeros((n, 3))
    if kind == "sparse":                     # long runs of zeros between a few values (mostly lossless colour)
        a = np.zeros((n, 3))
        hit = rng.random((n, 3)) < 0.003
        a[hit] = (rng.integers(1, 6, hit.sum()) / 255.0) ** 2
        return a
    if kind == "tiny":                       # subnormal and near-subnormal sums
        return rng.random((n, 3)) * 1e-310
    if kind == "huge":                       # overflow to inf part of the way through
        return rng.random((n, 3)) * 1e306
    if kind == "nan":
        a = rng.random((n, 3))
        a[n // 3, 0] = np.nan
        a[n // 2, 1] = np.inf
        return a
    if kind == "quantised":                  # squares of k/255 differences, the real thing
        return ((rng.integers(0, 256, (n, 3)) - rng.integers(0, 256, (n, 3))) / 255.0) ** 2
    raise KeyError(kind)


@pytest.mark.parametrize("kind", ["uniform", "wide", "ties", "zeros", "allzero", "sparse", "tiny", "huge", "nan", "quantised"])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 8191, 8192, 8193, 100003])
def test_seq_colsum_is_numpys_axis0_sum(engine, kind, n):
    a = _columns(kind, n, 17 * n + len(kind))
    with np.errstate(over="ignore", invalid="ignore"):
        want = np.add.reduce(a, axis=0)
    assert same_bits(engine.seq_colsum(a), want), (kind, n)


def test_seq_colsum_one_million_rows(engine):
    a = _columns("quantised", 1_000_000, 5)
    a[:, 1] = _columns("uniform", 1_000_000, 6)[:, 1]
    a[:, 2] = _columns("wide", 1_000_000, 7)[:, 2]
    assert same_bits(engine.seq_colsum(a), np.add.reduce(a, axis=0))


def test_seq_colsum_beyond_the_chunk_records_kept_in_lds(engine):
    """More than 1536 chunks of 8192 rows: the walk reads the later chunk records from memory."""
    n = 12_700_000
    a = _columns("quantised", n, 11)
    a[:, 1] = _columns("uniform", n, 12)[:, 1] * 1e-3
    a[n // 2:, 2] = 0.0                                   # a long run of zeros behind the LDS-resident part's end
    assert same_bits(engine.seq_colsum(a), np.add.reduce(a, axis=0))


def test_seq_colsum_zero_columns_take_no_serial_walk(engine):
    """A column of zeros (a pair with identical colours) used to be summed one add after the other: 25 ms per 0.8M rows."""
    import time
    a = np.zeros((800_000, 3))
    a[:, 2] = _columns("sparse", 800_000, 3)[:, 2]
    assert same_bits(engine.seq_colsum(a), np.add.reduce(a, axis=0))
    t0 = time.perf_counter()
    engine.seq_colsum(a)
    assert time.perf_counter() - t0 < 0.02        # (upload of 19 MB included; the serial walk alone took longer than this)


# ---- the metrics -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["fixture_eye3_color", "uniform_300_color"])
@pytest.mark.parametrize("scheme", ["rgb", "ycc"])
def test_colour_report_matches_reference(name, scheme):
    g = load_golden(name)
    pair = CloudPair(PointCloud(g["a"], g["na"], g["ca"]), PointCloud(g["b"], g["nb"], g["cb"]), extent=g["extent"])
    res = MetricCalculator(pair).calculate(transform_options(CalculateOptions(color=scheme)))
    got = res.as_dict()
    for key, val in g["meta"]["results"]["c" + scheme]:
        assert same_bits(np.atleast_1d(got[tuple(key)]), np.asarray(val)), key
    assert res.as_df().to_string() == g["meta"]["texts"]["c" + scheme]["string"]


@pytest.mark.parametrize("name", ["fixture_eye3_color", "uniform_300_color"])
@pytest.mark.parametrize("scheme", ["rgb", "ycc", "yuv"])
def test_every_colour_metric_matches_reference(name, scheme):
    g = load_golden(name)
    pair = CloudPair(PointCloud(g["a"], g["na"], g["ca"]), PointCloud(g["b"], g["nb"], g["cb"]), extent=g["extent"])
    with np.errstate(divide="ignore"):
        for is_left in (True, False):
            side = "left" if is_left else "right"
            for cls in COLOUR_CLASSES:
                m = MetricCalculator(pair)._metric_recursive_calculate(getattr(opmm, cls)(is_left=is_left, color_scheme=scheme))
                assert same_bits(m.value, g[f"{cls}_{side}_{scheme}"]), (cls, side, scheme)


def _coloured(n, seed, dtype=np.float64):
    rng = np.random.default_rng(seed)
    a = rng.random((n, 3), dtype=np.float32).astype(np.float64)
    b = (a + rng.normal(0, 2e-3, (n, 3))).astype(np.float32).astype(np.float64)
    ca = (rng.integers(0, 256, (n, 3)) / 255.0).astype(dtype)
    cb = (rng.integers(0, 256, (n, 3)) / 255.0).astype(dtype)
    return a, b, ca, cb


@pytest.mark.parametrize("n,m", [(20000, 20000), (5000, 7777)])
@pytest.mark.parametrize("scheme", ["rgb", "ycc", "yuv"])
def test_colour_metrics_against_oracle(n, m, scheme):
    a, _, ca, _ = _coloured(n, 1)
    _, b, _, cb = _coloured(m, 2)
    pair = CloudPair(PointCloud(a, None, ca), PointCloud(b, None, cb), extent=[1, 1, 1])
    for is_left, own, other, oc, rc in ((True, a, b, ca, cb), (False, b, a, cb, ca)):
        idx, _ = orc.nn(own, other)
        mse = MetricCalculator(pair)._metric_recursive_calculate(opmm.ColorMSE(is_left=is_left, color_scheme=scheme)).value
        hd = MetricCalculator(pair)._metric_recursive_calculate(opmm.ColorHausdorffDistance(is_left=is_left, color_scheme=scheme)).value
        assert same_bits(mse, orc.color_mse(oc, rc, idx, scheme))
        assert same_bits(hd, orc.color_hausdorff(oc, rc, idx, scheme))


def test_colour_rows_materialise_like_numpy():
    a, b, ca, cb = _coloured(30000, 3)
    pair = CloudPair(PointCloud(a, None, ca), PointCloud(b, None, cb), extent=[1, 1, 1])
    idx = pair._neighbour_index(nat.DIR_RIGHT)
    neigh = pair.get_right_neighbour_colors()
    assert np.array_equal(np.asarray(neigh), np.take(ca, idx, axis=0))
    for scheme in ("rgb", "ycc", "yuv"):
        diff = neigh.in_scheme(scheme)
        want = opmm.transform_colors(cb, "rgb", scheme) - opmm.transform_colors(np.take(ca, idx, axis=0), "rgb", scheme)
        assert np.array_equal(np.asarray(diff), want)
        assert np.array_equal(np.asarray((255 * diff) ** 2), (255 * want) ** 2)
        assert same_bits(np.mean(diff ** 2, axis=0), np.mean(want ** 2, axis=0))
        assert same_bits(np.max(np.square(diff), axis=0), np.max(want ** 2, axis=0))


def test_colour_fp32_input_and_one_million_points():
    n = 1_000_000
    a, b, ca, cb = _coloured(n, 4, np.float32)
    pair = CloudPair(PointCloud(a, None, ca), PointCloud(b, None, cb), extent=[1, 1, 1])
    res = MetricCalculator(pair).calculate(transform_options(CalculateOptions(color="ycc"))).as_dict()
    idx = pair._neighbour_index(nat.DIR_LEFT)
    own = opmm.transform_colors(ca.astype(np.float64), "rgb", "ycc")
    other = opmm.transform_colors(np.take(cb.astype(np.float64), idx, axis=0), "rgb", "ycc")
    want = np.mean((own - other) ** 2, axis=0)
    assert same_bits(res[("ColorMSE", True, "ycc")], want)
    assert same_bits(res[("ColorPSNR", True, "ycc")], 10 * np.log10(1.0 / want))


def test_colour_rows_override_and_errors(engine):
    a, b, ca, cb = _coloured(4096, 9)
    engine.set_cloud(0, a)
    engine.set_cloud(1, b)
    with pytest.raises(RuntimeError):
        engine.color_reduce(nat.DIR_LEFT, "rgb")                 # no colours yet
    engine.set_colors(0, ca)
    engine.set_colors(1, cb)
    with pytest.raises(RuntimeError):
        engine.color_reduce(nat.DIR_LEFT, "rgb")                 # no search yet
    with pytest.raises(ValueError):
        engine.set_colors(1, cb[:100])                            # row count must match the cloud
    engine.nn(nat.DIR_LEFT)
    idx, _ = engine.fetch_nn(nat.DIR_LEFT, want_d2=False)
    s0, m0 = engine.color_reduce(nat.DIR_LEFT, "ycc")
    s1, m1 = engine.color_reduce(nat.DIR_LEFT, "ycc", rows=idx)  # what a sharded pair passes after its gather
    assert same_bits(s0, s1) and same_bits(m0, m1)
    bad = idx.copy()
    bad[7] = 4096
    with pytest.raises(IndexError):
        engine.color_reduce(nat.DIR_LEFT, "ycc", rows=bad)
    with pytest.raises(ValueError):
        engine.color_reduce(nat.DIR_LEFT, "ycc", rows=idx[:10])
    with pytest.raises(ValueError):
        engine.color_reduce(nat.DIR_SELF, "ycc")


def test_both_directions_share_the_launches_and_stay_fresh(engine):
    """pccm_color_reduce answers the other direction from the same launches; what it keeps must never outlive a search, new
    colours or another scheme."""
    a, b, ca, cb = _coloured(30000, 21)
    fresh = nat.Engine(0)

    def separately(direction, scheme, colours_b, cloud_b):
        fresh.set_cloud(0, a)
        fresh.set_cloud(1, cloud_b)
        fresh.set_colors(0, ca)
        fresh.set_colors(1, colours_b)
        fresh.nn(direction)                                       # only this direction has a result: nothing rides along
        return fresh.color_reduce(direction, scheme)

    engine.set_cloud(0, a)
    engine.set_cloud(1, b)
    engine.set_colors(0, ca)
    engine.set_colors(1, cb)
    engine.nn_pair()
    for scheme in ("ycc", "rgb"):                                 # (the second scheme must not be answered from the first's memo)
        left = engine.color_reduce(nat.DIR_LEFT, scheme)
        right = engine.color_reduce(nat.DIR_RIGHT, scheme)        # from the memo
        again = engine.color_reduce(nat.DIR_RIGHT, scheme)        # recomputed (the memo answers once)
        for got, d in ((left, nat.DIR_LEFT), (right, nat.DIR_RIGHT), (again, nat.DIR_RIGHT)):
            want = separately(d, scheme, cb, b)
            assert same_bits(got[0], want[0]) and same_bits(got[1], want[1]), (scheme, d)
    # new colours between the two calls
    engine.color_reduce(nat.DIR_LEFT, "ycc")
    cb2 = np.ascontiguousarray(cb[::-1])
    engine.set_colors(1, cb2)
    got, want = engine.color_reduce(nat.DIR_RIGHT, "ycc"), separately(nat.DIR_RIGHT, "ycc", cb2, b)
    assert same_bits(got[0], want[0]) and same_bits(got[1], want[1])
    # a new search between the two calls
    engine.color_reduce(nat.DIR_LEFT, "ycc")
    b2 = np.ascontiguousarray(b[::-1])
    engine.set_cloud(1, b2)
    engine.set_colors(1, cb2)
    engine.nn_pair()
    got, want = engine.color_reduce(nat.DIR_RIGHT, "ycc"), separately(nat.DIR_RIGHT, "ycc", cb2, b2)
    assert same_bits(got[0], want[0]) and same_bits(got[1], want[1])
    fresh.close()


def test_uchar_colours_widen_on_the_device_like_on_the_host(engine):
    a, b, _, _ = _coloured(5000, 12)
    rng = np.random.default_rng(13)
    ua, ub = rng.integers(0, 256, (5000, 3)).astype(np.uint8), rng.integers(0, 256, (5000, 3)).astype(np.uint8)
    engine.set_cloud(0, a); engine.set_cloud(1, b)
    engine.nn(nat.DIR_LEFT)
    engine.set_colors(0, ua / 255.0); engine.set_colors(1, ub / 255.0)
    want = [engine.color_reduce(nat.DIR_LEFT, s, 255.0 if s == "rgb" else 1.0) for s in ("rgb", "ycc", "yuv")]
    engine.set_colors_u8(0, ua); engine.set_colors_u8(1, ub)
    got = [engine.color_reduce(nat.DIR_LEFT, s, 255.0 if s == "rgb" else 1.0) for s in ("rgb", "ycc", "yuv")]
    for (s0, m0), (s1, m1) in zip(want, got):
        assert same_bits(s0, s1) and same_bits(m0, m1)
    assert np.array_equal(engine.color_rows(nat.DIR_LEFT, "rgb", nat.COLOR_OWN), ua / 255.0)
    with pytest.raises(ValueError):
        engine.set_colors_u8(0, ua[:10])
