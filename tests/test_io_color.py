"""PLY/XYZ reader (handler.py:57 uses o3d.io.read_point_cloud) and the colour rows of the report
(options.py:58-82, metric.py:250-350) against golden vectors made by the reference.  CPU only."""
import numpy as np
import pytest

from conftest import load_golden, same_bits
from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.io import read_point_cloud, write_point_cloud
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from oracle_engine import OracleEngine


@pytest.mark.parametrize("binary", [True, False])
@pytest.mark.parametrize("coord", ["float", "double"])
def test_ply_round_trip(tmp_path, binary, coord):
    rng = np.random.default_rng(1)
    pts = rng.random((257, 3)) * 100
    if coord == "float":
        pts = pts.astype(np.float32).astype(np.float64)
    nrm = rng.standard_normal((257, 3))
    if coord == "float":
        nrm = nrm.astype(np.float32).astype(np.float64)
    col = np.round(rng.random((257, 3)) * 255) / 255.0
    path = str(tmp_path / "c.ply")
    write_point_cloud(path, PointCloud(pts, nrm, col), binary=binary, coord_dtype=coord)
    back = read_point_cloud(path)
    assert np.array_equal(back.points, pts) and back.points.dtype == np.float64
    assert np.array_equal(back.normals, nrm)
    assert np.array_equal(back.colors, col)          # uchar / 255.0, as Open3D does
    assert back.has_normals() and back.has_colors()


def test_ply_big_endian_ints_and_extra_elements(tmp_path):
    pts = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], dtype=">i2")
    path = tmp_path / "be.ply"
    header = ("ply\nformat binary_big_endian 1.0\ncomment made by a test\nelement vertex 3\n"
              "property short x\nproperty short y\nproperty short z\nelement face 0\n"
              "property list uchar int vertex_indices\nend_header\n")
    path.write_bytes(header.encode() + pts.tobytes())
    back = read_point_cloud(str(path))
    assert np.array_equal(back.points, pts.astype(np.float64))
    assert not back.has_normals() and not back.has_colors()


def test_xyz_and_errors(tmp_path):
    p = tmp_path / "a.xyz"
    p.write_text("0 0 0 0 0 1\n1 2 3 0 1 0\n")
    c = read_point_cloud(str(p))
    assert c.points.shape == (2, 3) and c.has_normals()
    bad = tmp_path / "bad.ply"
    bad.write_text("not a ply\n")
    with pytest.raises(ValueError):
        read_point_cloud(str(bad))
    with pytest.raises(ValueError):
        read_point_cloud(str(tmp_path / "cloud.obj"))


def test_color_transform_is_rowwise_matmul():
    rng = np.random.default_rng(3)
    c = rng.random((500, 3))
    m = np.array([[0.2126, 0.7152, 0.0722], [-0.1146, -0.3854, 0.5], [0.5, -0.4542, -0.0458]])
    assert np.array_equal(nat.color_transform(c, "ycc"), np.stack([np.matmul(m, row) for row in c]))
    m = np.array([[0.25, 0.5, 0.25], [1, 0, -1], [-0.5, 1, -0.5]])
    assert np.array_equal(nat.color_transform(c, "yuv"), np.stack([np.matmul(m, row) for row in c]))


@pytest.mark.parametrize("name", ["fixture_eye3_color", "uniform_300_color"])
@pytest.mark.parametrize("scheme", ["rgb", "ycc"])
def test_color_rows_match_reference(name, scheme):
    g = load_golden(name)
    pair = CloudPair(PointCloud(g["a"], g["na"], g["ca"]), PointCloud(g["b"], g["nb"], g["cb"]), extent=g["extent"],
                     _engine=OracleEngine())
    res = MetricCalculator(pair).calculate(transform_options(CalculateOptions(color=scheme)))
    want = g["meta"]["results"]["c" + scheme]
    got = res.as_dict()
    assert [tuple(k) for k, _ in want] == list(got.keys())
    for key, val in want:
        assert same_bits(np.atleast_1d(got[tuple(key)]), np.asarray(val)), (key, got[tuple(key)], val)
    assert res.as_df().to_string() == g["meta"]["texts"]["c" + scheme]["string"]
    assert res.as_df().to_csv() == g["meta"]["texts"]["c" + scheme]["csv"]


COLOUR_CLASSES = ("ColorMSE", "ColorPSNR", "ColorHausdorffDistance", "ColorHausdorffDistancePSNR")


@pytest.mark.parametrize("name", ["fixture_eye3_color", "uniform_300_color"])
@pytest.mark.parametrize("scheme", ["rgb", "ycc", "yuv"])
def test_oracle_colour_restatement_matches_reference(name, scheme):
    """oracle.color_mse / color_hausdorff against the reference's ColorMSE / ColorHausdorffDistance values
    (metric.py:302-333, 389-427), both directions; pins the oracle the GPU colour path is checked with."""
    from oracle import oracle as orc
    g = load_golden(name)
    for side, own, other, oc, rc in (("left", g["a"], g["b"], g["ca"], g["cb"]), ("right", g["b"], g["a"], g["cb"], g["ca"])):
        idx, _ = orc.nn(own, other)
        assert same_bits(orc.color_mse(oc, rc, idx, scheme), g[f"ColorMSE_{side}_{scheme}"])
        assert same_bits(orc.color_hausdorff(oc, rc, idx, scheme), g[f"ColorHausdorffDistance_{side}_{scheme}"])


@pytest.mark.parametrize("name", ["fixture_eye3_color", "uniform_300_color"])
@pytest.mark.parametrize("scheme", ["rgb", "ycc", "yuv"])
def test_every_colour_metric_through_the_dag(name, scheme):
    """The colour metrics no CLI option reaches (ColorHausdorffDistance[PSNR], metric.py:389-443) and "yuv"."""
    import open_pcc_metric_amd.metric as opmm
    g = load_golden(name)
    eng = OracleEngine()
    pair = CloudPair(PointCloud(g["a"], g["na"], g["ca"]), PointCloud(g["b"], g["nb"], g["cb"]), extent=g["extent"], _engine=eng)
    with np.errstate(divide="ignore"):
        for is_left in (True, False):
            side = "left" if is_left else "right"
            for cls in COLOUR_CLASSES:
                m = MetricCalculator(pair)._metric_recursive_calculate(getattr(opmm, cls)(is_left=is_left, color_scheme=scheme))
                assert same_bits(m.value, g[f"{cls}_{side}_{scheme}"]), (cls, side, scheme)
    assert any(c[0] == "color_reduce" for c in eng.calls)       # answered by the engine, not by host NumPy


def test_colour_device_rows_materialise_like_numpy():
    """Anything the fused path does not cover falls back to rows fetched from the engine: same values
    as the reference's host expressions (np.take, transform_colors, np.subtract)."""
    import open_pcc_metric_amd.metric as opmm
    g = load_golden("uniform_300_color")
    pair = CloudPair(PointCloud(g["a"], g["na"], g["ca"]), PointCloud(g["b"], g["nb"], g["cb"]), extent=g["extent"],
                     _engine=OracleEngine())
    idx = pair._neighbour_index(nat.DIR_LEFT)
    neigh = pair.get_left_neighbour_colors()
    assert np.array_equal(np.asarray(neigh), np.take(g["cb"], idx, axis=0))
    diff = neigh.in_scheme("ycc")
    want = opmm.transform_colors(g["ca"], "rgb", "ycc") - opmm.transform_colors(np.take(g["cb"], idx, axis=0), "rgb", "ycc")
    assert np.array_equal(np.asarray(diff), want)
    assert np.array_equal(np.asarray(255 * diff), 255 * want)
    assert np.array_equal(np.asarray(diff ** 2), want ** 2)
    assert same_bits(np.mean(diff ** 2, axis=0), np.mean(want ** 2, axis=0))
    assert same_bits(np.max((255 * diff) ** 2, axis=0), np.max((255 * want) ** 2, axis=0))
    assert same_bits(np.sum(diff ** 2), np.sum(want ** 2))       # not fused: materialises


# ---- PCD / xyzrgb / pts (the other formats o3d.io.read_point_cloud picks by extension) -----------------------------
def _lzf_literals(data: bytes) -> bytes:
    """A valid (if useless) LZF stream: literal runs only."""
    out = bytearray()
    for i in range(0, len(data), 32):
        chunk = data[i:i + 32]
        out.append(len(chunk) - 1)
        out += chunk
    return bytes(out)


def test_lzf_decoder_handles_back_references():
    # "abcabcabcabc!": literal "abc", then a back reference of length 9 at distance 3 (overlapping copy), then "!"
    stream = bytes([2]) + b"abc" + bytes([(7 << 5) | 0, 0, 2]) + bytes([0]) + b"!"
    assert nat.lzf_decompress(stream, 64) == b"abcabcabcabc!"
    assert nat.lzf_decompress(_lzf_literals(bytes(range(200))), 200) == bytes(range(200))
    with pytest.raises(ValueError):
        nat.lzf_decompress(bytes([5, 1, 2]), 64)                       # literal run past the end
    with pytest.raises(ValueError):
        nat.lzf_decompress(bytes([(1 << 5) | 0, 9]), 64)               # reference before the start


def _pcd(tmp_path, mode, pts, nrm, rgb, name="c.pcd"):
    n = len(pts)
    packed = ((rgb[:, 0].astype(np.uint32) << 16) | (rgb[:, 1].astype(np.uint32) << 8) | rgb[:, 2].astype(np.uint32))
    rec = np.empty(n, dtype=np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("normal_x", "<f4"), ("normal_y", "<f4"),
                                      ("normal_z", "<f4"), ("rgb", "<u4")]))
    for k, col in zip(("x", "y", "z"), pts.T):
        rec[k] = col
    for k, col in zip(("normal_x", "normal_y", "normal_z"), nrm.T):
        rec[k] = col
    rec["rgb"] = packed
    head = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z normal_x normal_y normal_z rgb\n"
            "SIZE 4 4 4 4 4 4 4\nTYPE F F F F F F U\nCOUNT 1 1 1 1 1 1 1\n"
            f"WIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA {mode}\n").encode()
    path = tmp_path / name
    if mode == "ascii":
        rows = "\n".join(" ".join([repr(float(v)) for v in (r["x"], r["y"], r["z"], r["normal_x"], r["normal_y"], r["normal_z"])]
                                  + [str(int(r["rgb"]))]) for r in rec)
        path.write_bytes(head + rows.encode() + b"\n")
    elif mode == "binary":
        path.write_bytes(head + rec.tobytes())
    else:
        soa = b"".join(np.ascontiguousarray(rec[k]).tobytes() for k in rec.dtype.names)
        comp = _lzf_literals(soa)
        path.write_bytes(head + np.array([len(comp), len(soa)], dtype="<u4").tobytes() + comp)
    return str(path)


@pytest.mark.parametrize("mode", ["ascii", "binary", "binary_compressed"])
def test_pcd_reader(tmp_path, mode):
    rng = np.random.default_rng(5)
    pts = rng.random((257, 3), dtype=np.float32)
    nrm = rng.standard_normal((257, 3)).astype(np.float32)
    rgb = rng.integers(0, 256, (257, 3))
    cloud = read_point_cloud(_pcd(tmp_path, mode, pts, nrm, rgb))
    assert np.array_equal(np.asarray(cloud.points), pts.astype(np.float64))
    assert np.array_equal(np.asarray(cloud.normals), nrm.astype(np.float64))
    assert np.array_equal(np.asarray(cloud.colors), rgb / 255.0)


def test_pcd_float_packed_rgb_and_errors(tmp_path):
    pts = np.array([[0.0, 1.0, 2.0], [3.0, 4.0, 5.0]], dtype=np.float32)
    packed = np.array([(10 << 16) | (20 << 8) | 30, (255 << 16) | 7], dtype="<u4")
    rec = np.empty(2, dtype=np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgb", "<f4")]))
    rec["x"], rec["y"], rec["z"] = pts.T
    rec["rgb"] = packed.view("<f4")                                     # PCL's classic float-packed colour
    head = b"VERSION .7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 2\nHEIGHT 1\nPOINTS 2\nDATA binary\n"
    p = tmp_path / "f.pcd"
    p.write_bytes(head + rec.tobytes())
    cloud = read_point_cloud(str(p))
    assert np.array_equal(np.asarray(cloud.colors), np.array([[10, 20, 30], [255, 0, 7]]) / 255.0)
    assert not cloud.has_normals()
    (tmp_path / "bad.pcd").write_bytes(head + rec.tobytes()[:10])
    with pytest.raises(ValueError):
        read_point_cloud(str(tmp_path / "bad.pcd"))
    (tmp_path / "noxyz.pcd").write_bytes(b"FIELDS a b\nSIZE 4 4\nTYPE F F\nCOUNT 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n1 2\n")
    with pytest.raises(ValueError):
        read_point_cloud(str(tmp_path / "noxyz.pcd"))


def test_xyzrgb_and_pts_readers(tmp_path):
    rng = np.random.default_rng(6)
    pts, col = rng.random((50, 3)), rng.random((50, 3))
    a = tmp_path / "a.xyzrgb"
    a.write_text("\n".join(" ".join(repr(float(v)) for v in row) for row in np.hstack([pts, col])) + "\n")
    cloud = read_point_cloud(str(a))
    assert np.array_equal(np.asarray(cloud.points), pts) and np.array_equal(np.asarray(cloud.colors), col)
    rgb = rng.integers(0, 256, (50, 3))
    b = tmp_path / "b.pts"
    b.write_text("50\n" + "\n".join(" ".join([repr(float(v)) for v in p] + ["-1000"] + [str(int(c)) for c in k])
                                      for p, k in zip(pts, rgb)) + "\n")
    cloud = read_point_cloud(str(b))
    assert np.array_equal(np.asarray(cloud.points), pts) and np.array_equal(np.asarray(cloud.colors), rgb / 255.0)
    (tmp_path / "short.pts").write_text("3\n0 0 0\n1 1 1\n")
    with pytest.raises(ValueError):
        read_point_cloud(str(tmp_path / "short.pts"))
    with pytest.raises(ValueError):
        read_point_cloud(str(tmp_path / "cloud.obj"))


def test_readers_keep_the_uchar_colours_for_the_gpu(tmp_path):
    rng = np.random.default_rng(8)
    pts, rgb = rng.random((40, 3)), rng.integers(0, 256, (40, 3))
    p = str(tmp_path / "c.ply")
    write_point_cloud(p, PointCloud(pts, None, rgb / 255.0))
    cloud = read_point_cloud(p)
    assert cloud.colors_u8 is not None and cloud.colors_u8.dtype == np.uint8
    assert np.array_equal(cloud.colors_u8, rgb) and np.array_equal(np.asarray(cloud.colors), rgb / 255.0)
    cloud.colors = np.asarray(cloud.colors) * 0.5          # new colours: the companion no longer applies
    assert cloud.colors_u8 is None
    assert PointCloud(pts, None, rgb / 255.0).colors_u8 is None
