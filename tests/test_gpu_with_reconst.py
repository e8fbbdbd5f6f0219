"""One reference cloud, several decoded clouds (BASELINE.json configs[4]'s literal shape): ``CloudPair.with_reconst`` keeps what
belongs to the origin cloud -- upload, estimated normals (cloud_pair.py:61-64 of the reference), ``get_extent()``
(cloud_pair.py:111-112), the self search (cloud_pair.py:108-109), colours -- and every report equals the one a fresh pair gives,
bit for bit.  The reference runs its command line once per decoded cloud (handler.py:57-66) and repeats all of it."""
import numpy as np
import pytest
from click.testing import CliRunner

import open_pcc_metric_amd.cloud_pair as cpm
from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.handler import cli
from open_pcc_metric_amd.io import write_point_cloud
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from open_pcc_metric_amd.sequence import evaluate_pairs

pytestmark = pytest.mark.gpu


def content(seed=3, n=60_000):
    """A voxelised surface with uchar colours, no normals, and three 'decoded' versions of different sizes."""
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    a = np.unique(np.round(300 + 140 * v), axis=0).astype(np.float32)
    a = a[rng.permutation(len(a))]

    def cols(p):
        return np.clip(np.rint(128 + 90 * np.sin(p / 31.0) + rng.normal(0, 5, p.shape)), 0, 255).astype(np.uint8) / 255.0

    decoded = []
    for step, drop in ((1, 0.0), (2, 0.03), (4, 0.05)):
        q = a + np.rint(rng.normal(0, 0.5, a.shape)) if step == 1 else np.round(a / step) * step
        q = np.unique(q.astype(np.float32), axis=0)
        q = q[rng.random(len(q)) >= drop]
        decoded.append(PointCloud(q, None, cols(q)))
    return PointCloud(a, None, cols(a)), decoded


def report(pair, options):
    with np.errstate(divide="ignore"):
        return MetricCalculator(pair).calculate(transform_options(options)).as_dict()


def same(a, b):
    return list(a) == list(b) and all(np.array_equal(np.asarray(a[k]), np.asarray(b[k]), equal_nan=True) for k in a)


def test_with_reconst_rows_equal_fresh_pairs_and_origin_work_is_done_once(monkeypatch):
    origin, decoded = content()
    options = CalculateOptions(color="ycc", hausdorff=True, point_to_plane=True)
    fresh = []
    for d in decoded:
        with CloudPair(origin, d, normal_index="neighbour") as pair:
            fresh.append(report(pair, options))
    calls = {"extent": 0, "normals": [], "self": 0}
    real_extent, real_est, real_nn = cpm.minimal_obb_extent, nat.Engine.estimate_normals, nat.Engine.nn

    def count_extent(*a, **k):
        calls["extent"] += 1
        return real_extent(*a, **k)

    def count_est(self, which, knn=30):
        calls["normals"].append(which)
        return real_est(self, which, knn)

    def count_nn(self, direction, engine="auto"):
        calls["self"] += direction == nat.DIR_SELF
        return real_nn(self, direction, engine)

    monkeypatch.setattr(cpm, "minimal_obb_extent", count_extent)
    monkeypatch.setattr(nat.Engine, "estimate_normals", count_est)
    monkeypatch.setattr(nat.Engine, "nn", count_nn)
    pair = CloudPair(origin, decoded[0], normal_index="neighbour")
    chained = [report(pair, options)]
    for d in decoded[1:]:
        old, pair = pair, pair.with_reconst(d)
        assert old.__dict__.get("_engine") is None, "the GPU context moved to the new pair"
        chained.append(report(pair, options))
    pair.close()
    for got, want in zip(chained, fresh):
        assert same(got, want)
    assert calls["extent"] == 1, "the origin's minimal OBB is computed once"
    assert calls["normals"].count(0) == 1 and calls["normals"].count(1) == 3, "the origin's normals are estimated once, every decoded cloud's once"
    assert calls["self"] == 1, "the origin's self search runs once"


def test_evaluate_pairs_chains_shared_origins(monkeypatch):
    origin, decoded = content(seed=4, n=30_000)
    other_origin, other_decoded = content(seed=5, n=20_000)
    options = CalculateOptions(color=None, hausdorff=True, point_to_plane=False)
    items = [(origin, d) for d in decoded] + [(other_origin, other_decoded[0])] + [(origin, decoded[1])]
    want = []
    for o, d in items:
        with CloudPair(o, d) as pair:
            want.append(report(pair, options))
    made = []
    real = nat.Engine.set_cloud

    def count(self, which, points):
        made.append(which)
        return real(self, which, points)

    monkeypatch.setattr(nat.Engine, "set_cloud", count)
    got = evaluate_pairs(items, options, workers=1)
    assert len(got) == len(want) and all(same(g, w) for g, w in zip(got, want))
    # three chains: (origin x 3 decoded), (other origin), (origin again): cloud 0 uploaded three times, cloud 1 five times
    assert made.count(0) == 3 and made.count(1) == 5


def test_cli_with_several_processed_clouds_prints_the_separate_reports(tmp_path):
    origin, decoded = content(seed=6, n=30_000)
    ref = str(tmp_path / "ref.ply")
    write_point_cloud(ref, origin, coord_dtype="float")
    paths = []
    for k, d in enumerate(decoded):
        paths.append(str(tmp_path / f"dec{k}.ply"))
        write_point_cloud(paths[-1], d, coord_dtype="float")
    flags = ["--color", "ycc", "--hausdorff", "--point-to-plane", "--normal-index", "neighbour"]
    with np.errstate(divide="ignore"):
        separate = [CliRunner().invoke(cli, ["--ocloud", ref, "--pcloud", p] + flags) for p in paths]
        args = ["--ocloud", ref] + [x for p in paths for x in ("--pcloud", p)] + flags
        together = CliRunner().invoke(cli, args)
    assert all(r.exit_code == 0 for r in separate) and together.exit_code == 0, together.output
    assert together.output == "".join(r.output for r in separate)
