"""Point-cloud file IO: what ``o3d.io.read_point_cloud`` does for the reference CLI (handler.py:57).

PLY (ascii, binary_little_endian, binary_big_endian) with vertex properties ``x y z`` (any scalar
type), optional ``nx ny nz`` and ``red green blue`` (uchar -> [0, 1] like Open3D; float colours are
taken as they are); PCL's PCD (ascii, binary, binary_compressed; x y z, normal_*, packed rgb/rgba);
whitespace-separated ``.xyz`` / ``.xyzn`` / ``.txt`` (x y z [nx ny nz]), ``.xyzrgb`` and ``.pts``.  Coordinates are
returned as float64 -- Open3D holds ``Vector3d`` -- so a float PLY yields fp32-representable doubles.
"""
from __future__ import annotations

import numpy as np

from .point_cloud import PointCloud

_PLY_TYPES = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
    "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
    "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}


def _read_ply(path: str) -> PointCloud:
    with open(path, "rb") as fh:
        if fh.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements, current = None, [], None
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                current = {"name": tok[1], "count": int(tok[2]), "props": []}
                elements.append(current)
            elif tok[0] == "property":
                if tok[1] == "list":
                    current["props"].append(("list", tok[2], tok[3], tok[4]))
                else:
                    current["props"].append((tok[-1], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if not elements or elements[0]["name"] != "vertex":
            raise ValueError(f"{path}: the first PLY element must be 'vertex'")
        vert = elements[0]
        if any(p[0] == "list" for p in vert["props"]):
            raise ValueError(f"{path}: list properties on vertices are not supported")
        names = [p[0] for p in vert["props"]]
        n = vert["count"]
        if fmt == "ascii":
            import pandas as pd
            table = pd.read_csv(fh, sep=r"\s+", header=None, nrows=n, engine="c",
                                float_precision="round_trip").to_numpy(dtype=np.float64)
            if table.shape[0] != n or table.shape[1] < len(names):
                raise ValueError(f"{path}: truncated vertex list")
            cols = {name: table[:, k] for k, name in enumerate(names)}
            raw_dtypes = {name: np.dtype(t) for name, t in vert["props"]}
        elif fmt in ("binary_little_endian", "binary_big_endian"):
            order = "<" if fmt == "binary_little_endian" else ">"
            dt = np.dtype([(name, order + t) for name, t in vert["props"]])
            data = np.fromfile(fh, dtype=dt, count=n)
            if data.shape[0] != n:
                raise ValueError(f"{path}: truncated vertex list")
            cols = {name: data[name] for name in names}
            raw_dtypes = {name: np.dtype(t) for name, t in vert["props"]}
        else:
            raise ValueError(f"{path}: unknown PLY format {fmt!r}")

    def stack(keys):
        # three adjacent fields of one type in a binary record: ONE strided view of the file's bytes, one pass to float64
        # (column by column and a stack behind it: three passes and a copy -- 4 of the 11 ms a 0.8M-point file took)
        if fmt != "ascii":
            spec = [data.dtype.fields[k] for k in keys]
            t, size = spec[0][0], spec[0][0].itemsize
            if all(f[0] == t for f in spec) and [f[1] for f in spec] == [spec[0][1] + j * size for j in range(3)]:
                view = np.ndarray((n, 3), dtype=t, buffer=data, offset=spec[0][1], strides=(data.dtype.itemsize, size))
                return view.astype(np.float64)
        return np.stack([np.asarray(cols[k], dtype=np.float64) for k in keys], axis=1)

    if not all(k in cols for k in "xyz"):
        raise ValueError(f"{path}: vertices need x, y and z")
    cloud = PointCloud(stack("xyz"))
    if all(k in cols for k in ("nx", "ny", "nz")):
        cloud.normals = stack(("nx", "ny", "nz"))
    rgb = ("red", "green", "blue") if "red" in cols else ("r", "g", "b")
    if all(k in cols for k in rgb):
        integral = raw_dtypes[rgb[0]].kind in "ui"
        bytes_ = integral and all(raw_dtypes[k] == np.dtype("u1") for k in rgb)
        if bytes_:
            # the file's bytes as they are (2.4 MB for 0.8M points) and their quotients: uint8 / 255.0 is float64(k) / 255.0
            raw = np.stack([np.asarray(cols[k], dtype=np.uint8) for k in rgb], axis=1)
            colors = raw / 255.0             # Open3D: uchar colours -> [0, 1]
        else:
            colors = stack(rgb)
            if integral:
                colors = colors / 255.0
        cloud.colors = colors
        if bytes_:
            cloud.attach_colors_u8(raw)
    return cloud


def _read_xyz(path: str) -> PointCloud:
    import pandas as pd
    table = pd.read_csv(path, sep=r"\s+", header=None, comment="#", engine="c",
                        float_precision="round_trip").to_numpy(dtype=np.float64)
    if table.ndim != 2 or table.shape[1] < 3:
        raise ValueError(f"{path}: expected at least three columns")
    cloud = PointCloud(np.ascontiguousarray(table[:, :3]))
    if table.shape[1] >= 6:
        cloud.normals = np.ascontiguousarray(table[:, 3:6])
    return cloud


def _read_xyzrgb(path: str) -> PointCloud:
    """x y z r g b, colours already in [0, 1] (Open3D's .xyzrgb)."""
    import pandas as pd
    table = pd.read_csv(path, sep=r"\s+", header=None, comment="#", engine="c",
                        float_precision="round_trip").to_numpy(dtype=np.float64)
    if table.ndim != 2 or table.shape[1] < 6:
        raise ValueError(f"{path}: expected six columns (x y z r g b)")
    return PointCloud(np.ascontiguousarray(table[:, :3]), None, np.ascontiguousarray(table[:, 3:6]))


def _read_pts(path: str) -> PointCloud:
    """Leica PTS: a count line, then x y z [intensity [r g b]] with colours 0..255 (Open3D's .pts)."""
    import pandas as pd
    with open(path, "rb") as fh:
        first = fh.readline().split()
    skip = 1 if len(first) == 1 else 0
    table = pd.read_csv(path, sep=r"\s+", header=None, skiprows=skip, engine="c",
                        float_precision="round_trip").to_numpy(dtype=np.float64)
    if table.ndim != 2 or table.shape[1] < 3:
        raise ValueError(f"{path}: expected at least three columns")
    if skip and int(first[0]) != table.shape[0]:
        raise ValueError(f"{path}: header announces {int(first[0])} points, file holds {table.shape[0]}")
    cloud = PointCloud(np.ascontiguousarray(table[:, :3]))
    if table.shape[1] >= 7:
        cloud.colors = np.ascontiguousarray(table[:, 4:7]) / 255.0
    elif table.shape[1] == 6:
        cloud.colors = np.ascontiguousarray(table[:, 3:6]) / 255.0
    return cloud


_PCD_TYPES = {("F", 4): "f4", ("F", 8): "f8", ("U", 1): "u1", ("U", 2): "u2", ("U", 4): "u4", ("U", 8): "u8",
              ("I", 1): "i1", ("I", 2): "i2", ("I", 4): "i4", ("I", 8): "i8"}


def _read_pcd(path: str) -> PointCloud:
    """PCL's PCD v0.7: DATA ascii | binary | binary_compressed; fields x y z, optional normal_x normal_y normal_z and
    rgb / rgba (packed 0x00RRGGBB in a float or uint32 -> [0, 1], as Open3D unpacks it)."""
    with open(path, "rb") as fh:
        meta = {}
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PCD header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0].startswith("#"):
                continue
            meta[tok[0].upper()] = tok[1:]
            if tok[0].upper() == "DATA":
                break
        body = fh.read()
    try:
        names = meta["FIELDS"]
        sizes = [int(s) for s in meta["SIZE"]]
        types = meta["TYPE"]
        counts = [int(c) for c in meta.get("COUNT", ["1"] * len(names))]
        n = int(meta["POINTS"][0]) if "POINTS" in meta else int(meta["WIDTH"][0]) * int(meta["HEIGHT"][0])
        mode = meta["DATA"][0].lower()
    except (KeyError, ValueError, IndexError) as exc:
        raise ValueError(f"{path}: malformed PCD header") from exc
    if not (len(names) == len(sizes) == len(types) == len(counts)):
        raise ValueError(f"{path}: FIELDS / SIZE / TYPE / COUNT disagree")
    fields = []
    for name, size, typ, count in zip(names, sizes, types, counts):
        code = _PCD_TYPES.get((typ.upper(), size))
        if code is None:
            raise ValueError(f"{path}: unsupported PCD field type {typ}{size}")
        for k in range(count):
            fields.append((name if count == 1 else f"{name}_{k}", "<" + code))
    dtype = np.dtype(fields)
    if mode == "ascii":
        import pandas as pd
        if n == 0:
            raise ValueError(f"{path}: empty cloud")
        import io as _io
        table = pd.read_csv(_io.BytesIO(body), sep=r"\s+", header=None, engine="c", float_precision="round_trip", nrows=n)
        if table.shape[0] != n or table.shape[1] != len(fields):
            raise ValueError(f"{path}: expected {n} rows of {len(fields)} values")
        cols = {}
        for j, (name, code) in enumerate(fields):
            raw = table.iloc[:, j].to_numpy()
            # a packed colour written as an integer must not pass through a float column
            cols[name] = raw.astype(np.dtype(code)) if np.dtype(code).kind != "f" or raw.dtype.kind == "f" else raw.astype(np.float64).astype(np.dtype(code))
    elif mode == "binary":
        if len(body) < n * dtype.itemsize:
            raise ValueError(f"{path}: truncated PCD body")
        rec = np.frombuffer(body, dtype=dtype, count=n)
        cols = {name: rec[name] for name, _ in fields}
    elif mode == "binary_compressed":
        if len(body) < 8:
            raise ValueError(f"{path}: truncated PCD body")
        clen, ulen = np.frombuffer(body, dtype="<u4", count=2)
        from . import _native
        raw = _native.lzf_decompress(body[8:8 + int(clen)], int(ulen))
        if len(raw) != n * dtype.itemsize:
            raise ValueError(f"{path}: compressed body has {len(raw)} bytes, header asks for {n * dtype.itemsize}")
        cols, off = {}, 0
        for name, code in fields:                       # the compressed layout is field by field
            width = np.dtype(code).itemsize
            cols[name] = np.frombuffer(raw, dtype=code, count=n, offset=off)
            off += width * n
    else:
        raise ValueError(f"{path}: unknown PCD DATA mode {mode!r}")
    if not all(k in cols for k in ("x", "y", "z")):
        raise ValueError(f"{path}: no x y z fields")
    cloud = PointCloud(np.stack([cols[k].astype(np.float64) for k in ("x", "y", "z")], axis=1))
    if all(k in cols for k in ("normal_x", "normal_y", "normal_z")):
        cloud.normals = np.stack([cols[k].astype(np.float64) for k in ("normal_x", "normal_y", "normal_z")], axis=1)
    packed = cols.get("rgb", cols.get("rgba"))
    if packed is not None:
        bits = np.ascontiguousarray(packed).view(np.uint32) if packed.dtype.itemsize == 4 else packed.astype(np.uint32)
        u8 = np.stack([(bits >> 16) & 255, (bits >> 8) & 255, bits & 255], axis=1).astype(np.uint8)
        cloud.colors = u8.astype(np.float64) / 255.0
        cloud.attach_colors_u8(u8)
    return cloud


def read_point_cloud(path: str) -> PointCloud:
    """What ``o3d.io.read_point_cloud(path)`` yields for the formats Open3D picks by extension."""
    low = str(path).lower()
    if low.endswith(".ply"):
        return _read_ply(path)
    if low.endswith(".pcd"):
        return _read_pcd(path)
    if low.endswith(".xyzrgb"):
        return _read_xyzrgb(path)
    if low.endswith(".pts"):
        return _read_pts(path)
    if low.endswith((".xyz", ".xyzn", ".txt")):
        return _read_xyz(path)
    raise ValueError(f"{path}: unsupported point cloud format (ply, pcd, xyz, xyzn, xyzrgb, pts are)")


def write_point_cloud(path: str, cloud, *, binary: bool = True, coord_dtype: str = "double") -> None:
    """Write a PLY (little endian or ascii).  ``coord_dtype``: 'float' or 'double'."""
    pts = np.asarray(cloud.points, dtype=np.float64)
    has_n = getattr(cloud, "has_normals", lambda: False)()
    has_c = getattr(cloud, "has_colors", lambda: False)()
    t = "f4" if coord_dtype == "float" else "f8"
    fields = [("x", t), ("y", t), ("z", t)]
    if has_n:
        fields += [("nx", t), ("ny", t), ("nz", t)]
    if has_c:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
    data = np.empty(pts.shape[0], dtype=np.dtype([(k, "<" + v) for k, v in fields]))
    for k, col in zip("xyz", pts.T):
        data[k] = col
    if has_n:
        for k, col in zip(("nx", "ny", "nz"), np.asarray(cloud.normals, dtype=np.float64).T):
            data[k] = col
    if has_c:
        rgb = np.clip(np.rint(np.asarray(cloud.colors, dtype=np.float64) * 255.0), 0, 255)
        for k, col in zip(("red", "green", "blue"), rgb.T):
            data[k] = col
    ply_name = {"f4": "float", "f8": "double", "u1": "uchar"}
    header = ["ply", "format " + ("binary_little_endian 1.0" if binary else "ascii 1.0"),
              f"element vertex {pts.shape[0]}"]
    header += [f"property {ply_name[v]} {k}" for k, v in fields] + ["end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(header) + "\n").encode("ascii"))
        if binary:
            data.tofile(fh)
        else:
            for row in data:
                fh.write((" ".join(repr(float(x)) if isinstance(x, (np.floating, float)) else str(int(x)) for x in row) + "\n").encode())
