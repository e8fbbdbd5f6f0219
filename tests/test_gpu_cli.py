"""The command line on the GPU: files in, the reference's report text out (handler.py:44-71)."""
import numpy as np
import pytest
from click.testing import CliRunner

from conftest import load_golden
from open_pcc_metric_amd.handler import cli
from open_pcc_metric_amd.io import write_point_cloud
from open_pcc_metric_amd.point_cloud import PointCloud

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["uniform_1000", "voxel10_noise_600", "lattice_ties_400", "noisy_f64_500"])
@pytest.mark.parametrize("csv", [False, True])
def test_cli_prints_the_reference_report(tmp_path, name, csv):
    g = load_golden(name)
    pa, pb = str(tmp_path / "a.ply"), str(tmp_path / "b.ply")
    write_point_cloud(pa, PointCloud(g["a"], g["na"]), coord_dtype="double")
    write_point_cloud(pb, PointCloud(g["b"], g["nb"]), coord_dtype="double")
    args = ["--ocloud", pa, "--pcloud", pb, "--hausdorff", "--point-to-plane", "--extent"] + [repr(float(x)) for x in g["extent"]]
    if csv:
        args.append("--csv")
    with np.errstate(divide="ignore"):
        out = CliRunner().invoke(cli, args)
    assert out.exit_code == 0, out.output
    want = g["meta"]["texts"]["h1p1"]["csv" if csv else "string"]
    assert out.output == want + "\n"


def test_cli_unequal_sizes_reproduce_the_reference_failure(tmp_path):
    g = load_golden("unequal_300_200")
    pa, pb = str(tmp_path / "a.ply"), str(tmp_path / "b.ply")
    write_point_cloud(pa, PointCloud(g["a"], g["na"]))
    write_point_cloud(pb, PointCloud(g["b"], g["nb"]))
    base = ["--ocloud", pa, "--pcloud", pb, "--extent", "1", "1", "1"]
    ok = CliRunner().invoke(cli, base + ["--hausdorff"])
    assert ok.exit_code == 0 and ok.output == g["meta"]["texts"]["h1p0"]["string"] + "\n"
    bad = CliRunner().invoke(cli, base + ["--point-to-plane"])
    assert isinstance(bad.exception, IndexError)          # quirk Q1: row-indexed normals
    fixed = CliRunner().invoke(cli, base + ["--point-to-plane", "--normal-index", "neighbour"])
    assert fixed.exit_code == 0 and "GeoPSNR" in fixed.output


def test_cli_color_rows(tmp_path):
    g = load_golden("uniform_300_color")
    pa, pb = str(tmp_path / "a.ply"), str(tmp_path / "b.ply")
    write_point_cloud(pa, PointCloud(g["a"], g["na"], g["ca"]), coord_dtype="double")
    write_point_cloud(pb, PointCloud(g["b"], g["nb"], g["cb"]), coord_dtype="double")
    out = CliRunner().invoke(cli, ["--ocloud", pa, "--pcloud", pb, "--color", "ycc", "--extent", "1.0", "0.9", "0.8"])
    assert out.exit_code == 0, out.output
    assert out.output == g["meta"]["texts"]["cycc"]["string"] + "\n"


def test_cli_reads_pcd_like_ply(tmp_path):
    g = load_golden("uniform_1000")
    pa, pb = str(tmp_path / "a.ply"), str(tmp_path / "b.ply")
    write_point_cloud(pa, PointCloud(g["a"], g["na"]), coord_dtype="float")
    write_point_cloud(pb, PointCloud(g["b"], g["nb"]), coord_dtype="float")

    def pcd(path, pts, nrm):
        rec = np.empty(len(pts), dtype=np.dtype([(k, "<f4") for k in ("x", "y", "z", "normal_x", "normal_y", "normal_z")]))
        for k, col in zip(rec.dtype.names, np.hstack([pts, nrm]).T):
            rec[k] = col
        head = ("VERSION 0.7\nFIELDS x y z normal_x normal_y normal_z\nSIZE 4 4 4 4 4 4\nTYPE F F F F F F\nCOUNT 1 1 1 1 1 1\n"
                f"WIDTH {len(pts)}\nHEIGHT 1\nPOINTS {len(pts)}\nDATA binary\n").encode()
        with open(path, "wb") as fh:
            fh.write(head + rec.tobytes())

    qa, qb = str(tmp_path / "a.pcd"), str(tmp_path / "b.pcd")
    pcd(qa, g["a"], g["na"])
    pcd(qb, g["b"], g["nb"])
    args = ["--hausdorff", "--point-to-plane", "--extent", "1", "0.9", "0.8"]
    with np.errstate(divide="ignore"):
        ply = CliRunner().invoke(cli, ["--ocloud", pa, "--pcloud", pb] + args)
        pcd_out = CliRunner().invoke(cli, ["--ocloud", qa, "--pcloud", qb] + args)
    assert ply.exit_code == 0 and pcd_out.exit_code == 0, pcd_out.output
    assert ply.output == pcd_out.output
