"""The voxel-brick search (open_pcc_metric_amd/csrc/pccm_vox.hip): voxelised pairs, distances only
(cloud_pair.py:102-106 -> metric.py:213-247, 353-386 of the reference) or -- round 4 -- with the matched rows
(cloud_pair.py:34-40).  Against the oracle, bit for bit:

* squared distances of both directions and of the self search (the intrinsic resolution) on content with duplicates
  inside a cloud (the self search's "another point at distance 0"), negative coordinates, points farther than the
  8 voxels the bricks vouch for (the tail kernels take those) and clouds of different sizes;
* whoever asks for matched rows (nn indices, error vectors, point-to-plane with neighbour normals, colours) gets them from
  the same bricks: the voxels at exactly the nearest distance are enumerated and the smallest row among their points wins,
  the tie rule of every other kernel (include/pccm.h); a pair that was first read for distances only is searched again with
  the rows' table built;
* PCCM_VOX=0 (the per-thread lattice search) gives the same numbers: tests/test_gpu_ab_paths.py."""
import numpy as np
import pytest

from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def shell(n, seed, centre, radius, jitter=0.0, dup=0, far=0, reach=1.4):
    """Integer points on a sphere shell, optionally jittered, with `dup` repeated points and `far` stray points."""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    p = np.round(np.asarray(centre) + radius * v + rng.normal(0, jitter, (n, 3)))
    p = np.unique(p, axis=0)
    if dup:
        p = np.concatenate([p, p[rng.integers(0, len(p), dup)]])
    if far:
        p = np.concatenate([p, np.round(np.asarray(centre) + rng.uniform(-reach, reach, (far, 3)) * radius)])
    return np.ascontiguousarray(p[rng.permutation(len(p))].astype(np.float32))


CASES = {
    "shell_dups_strays": (shell(90_000, 1, (40, -15, 7), 120, dup=500, far=40), shell(70_000, 2, (41, -15, 6), 121, 0.6, dup=300, far=25)),
    # outliers that more than double the bounding box: the grid covers the quantile box (decide_scale), which the bricks do not
    # support -- the per-thread lattice kernel then
    "boxed_outliers": (shell(60_000, 11, (0, 0, 0), 100, far=30, reach=2.6), shell(50_000, 12, (1, 0, 0), 100, 0.6, far=20, reach=2.6)),
    "small_unequal": (shell(3_000, 3, (0, 0, 0), 30, dup=10), shell(900, 4, (1, 1, -1), 33, 0.8)),
    "far_apart": (shell(5_000, 5, (0, 0, 0), 25), shell(5_000, 6, (90, 0, 0), 25)),       # nothing in common: the brute-force engine
    "holes": (shell(40_000, 9, (0, 0, 0), 80), shell(40_000, 10, (0, 0, 0), 80)[:16_000]),  # most of B missing: tails beyond 8 voxels
    "one_cell": (shell(200, 7, (3, 3, 3), 3, dup=20), shell(150, 8, (3, 3, 3), 3, 0.5)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_vox_distances_match_oracle(name):
    a, b = CASES[name]
    want_l, dl = orc.nn(a, b, method="kdtree")
    _, dr = orc.nn(b, a, method="kdtree")
    _, ds = orc.nn(a, a, skip_same_index=True, method="kdtree")
    with CloudPair(PointCloud(a), PointCloud(b), extent=[300.0, 300.0, 300.0], device=0) as pair:
        for _ in range(2):                                   # a first search, then a rebuild after pccm_drop_caches
            pair.recompute()
            eng = pair._engine
            eng.nn(2)                                        # the self search (the intrinsic resolution, metric.py:187-188)
            # the grid these searches ran on is the voxel-brick one: cells of 8 x 8 x 8 voxels over the pair's bounding box
            # (clouds with nothing in common are the exception: PCCM_ENGINE_AUTO hands such a pair to the brute-force engine)
            if name not in ("far_apart", "boxed_outliers"):
                lo, hi = np.minimum(a.min(0), b.min(0)), np.maximum(a.max(0), b.max(0))
                assert eng.nn_stats(0)["splits"] == int(np.prod(np.floor((hi - lo) / 8.0) + 1))
            for direction, want in ((0, dl), (1, dr), (2, ds)):
                _, got = eng.fetch_nn(direction, want_idx=False)
                assert np.array_equal(got, want), f"{name}: direction {direction}: {int(np.sum(got != want))} of {len(want)} squared distances differ"
        # ... and now somebody wants the rows: same distances, indices of equidistant-or-not neighbours that reproduce them
        idx, got = eng.fetch_nn(0, want_idx=True)
        assert np.array_equal(got, dl)
        assert np.array_equal(np.sum((a.astype(np.float64) - b.astype(np.float64)[idx]) ** 2, axis=1), dl)
        assert np.array_equal(idx, want_l), "ties go to the smallest row (include/pccm.h)"


def test_vox_report_then_d2():
    """A distances-only report through the voxel bricks, then point-to-plane on the same pair (normals arrive late): the D2 rows
    need the matched rows / error vectors, the library searches again."""
    a, b = CASES["shell_dups_strays"]
    rng = np.random.default_rng(9)
    na = rng.standard_normal(a.shape).astype(np.float32)
    nb = rng.standard_normal(b.shape).astype(np.float32)
    want1 = orc.OraclePair(a, b, None, None, method="kdtree").report(hausdorff=True, point_to_plane_=False, peak=300.0)
    with CloudPair(PointCloud(a), PointCloud(b), extent=[300.0, 300.0, 300.0], device=0) as pair:
        pair.recompute()
        got1 = MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, True, False))).as_dict()
        for key, val in want1.items():
            assert got1[key] == val, key
    want2 = orc.OraclePair(a, b, na, nb, method="kdtree", normal_index="neighbour").report(hausdorff=True, point_to_plane_=True, peak=300.0)
    with CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[300.0, 300.0, 300.0], device=0, normal_index="neighbour") as pair:
        pair.recompute()
        got2 = MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, True, True))).as_dict()
        for key, val in want2.items():
            assert got2[key] == val, key


@pytest.mark.parametrize("name", ["shell_dups_strays", "small_unequal", "holes", "one_cell"])
def test_vox_rows_match_oracle(name):
    """Matched rows straight from the voxel-brick search (pccm_nn_want_idx on before the first search): indices, distances and
    error vectors of both directions equal the oracle's (smallest row among equidistant nearest neighbours)."""
    a, b = CASES[name]
    from open_pcc_metric_amd import _native as nat
    eng = nat.Engine(0)
    try:
        eng.set_cloud(0, a)
        eng.set_cloud(1, b)
        eng.nn_want_idx(True)
        for _ in range(2):                                   # first build, then a rebuild behind pccm_drop_caches
            eng.drop_caches()
            eng.nn_pair("grid")
            lo, hi = np.minimum(a.min(0), b.min(0)), np.maximum(a.max(0), b.max(0))
            assert eng.nn_stats(0)["splits"] == int(np.prod(np.floor((hi - lo) / 8.0) + 1)), "not the voxel-brick grid"
            for direction, (q, r) in ((0, (a, b)), (1, (b, a))):
                want_idx, want_d2 = orc.nn(q, r, method="kdtree")
                idx, d2 = eng.fetch_nn(direction)
                assert np.array_equal(d2, want_d2)
                assert np.array_equal(idx, want_idx), f"{name}: direction {direction}: {int(np.sum(idx != want_idx))} rows differ"
                err = eng.error_vectors(direction)
                assert np.array_equal(err, q.astype(np.float64) - r.astype(np.float64)[want_idx])
    finally:
        eng.close()


@pytest.mark.parametrize("use_graph", [False, True])
def test_vox_same_pair_distances_then_normals_then_d2(use_graph):
    """ONE pair: a distances-only report through the voxel bricks, then normals arrive (set on the resident clouds) and the
    point-to-plane rows are asked of the same pair -- row-indexed normals (the reference's D2) on clouds of equal size, eagerly and
    through a hipGraph replay."""
    rng = np.random.default_rng(5)
    a = shell(60_000, 21, (10, 20, -5), 90, dup=0)
    b = np.ascontiguousarray((a + np.rint(rng.normal(0, 0.6, a.shape))).astype(np.float32))      # same size: row-indexed normals are legal
    na = rng.standard_normal(a.shape).astype(np.float32)
    nb = rng.standard_normal(b.shape).astype(np.float32)
    want1 = orc.OraclePair(a, b, None, None, method="kdtree").report(hausdorff=True, point_to_plane_=False, peak=300.0)
    want2 = orc.OraclePair(a, b, na, nb, method="kdtree").report(hausdorff=True, point_to_plane_=True, peak=300.0)
    ca, cb = PointCloud(a), PointCloud(b)
    with CloudPair(ca, cb, extent=[300.0, 300.0, 300.0], device=0, use_graph=use_graph, estimate_normals=False) as pair:
        for _ in range(3):
            pair.recompute()
            got1 = MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, True, False))[2:]).as_dict()
            for key in got1:
                assert got1[key] == want1[key], key
        # the normals arrive
        ca.normals, cb.normals = na, nb
        pair._engine.set_normals(0, na)
        pair._engine.set_normals(1, nb)
        pair._update_fusion()
        for _ in range(3):                                   # eager, capture, replay
            pair.recompute()
            got2 = MetricCalculator(pair).calculate(transform_options(CalculateOptions(None, True, True))[2:]).as_dict()
            for key in got2:
                assert got2[key] == want2[key], key


def test_vox_colours_and_neighbour_normals_stay_on_the_bricks():
    """The configs[4]-shaped request -- colour metrics and point-to-plane with the neighbour's normal on clouds of different sizes --
    is served by one voxel-brick grid (no rebuild with other cells), and equals the oracle."""
    a, b = CASES["shell_dups_strays"]
    rng = np.random.default_rng(19)
    na = rng.standard_normal(a.shape).astype(np.float32)
    nb = rng.standard_normal(b.shape).astype(np.float32)
    cola, colb = rng.integers(0, 256, a.shape).astype(np.float64) / 255.0, rng.integers(0, 256, b.shape).astype(np.float64) / 255.0
    o = orc.OraclePair(a, b, na, nb, method="kdtree", normal_index="neighbour")
    want = o.report(hausdorff=True, point_to_plane_=True, peak=300.0)
    with CloudPair(PointCloud(a, na, cola), PointCloud(b, nb, colb), extent=[300.0, 300.0, 300.0], device=0, normal_index="neighbour") as pair:
        got = MetricCalculator(pair).calculate(transform_options(CalculateOptions("ycc", True, True))).as_dict()
        for key, val in want.items():
            assert got[key] == val, key
        assert np.array_equal(got[("ColorMSE", True, "ycc")], orc.color_mse(cola, colb, o.nn_idx[0], "ycc"))
        assert np.array_equal(got[("ColorMSE", False, "ycc")], orc.color_mse(colb, cola, o.nn_idx[1], "ycc"))
        lo, hi = np.minimum(a.min(0), b.min(0)), np.maximum(a.max(0), b.max(0))
        assert pair._engine.nn_stats(0)["splits"] == int(np.prod(np.floor((hi - lo) / 8.0) + 1)), "the report left the voxel-brick grid"
