"""Point-cloud file IO: what ``o3d.io.read_point_cloud`` does for the reference CLI (handler.py:57).

PLY (ascii, binary_little_endian, binary_big_endian) with vertex properties ``x y z`` (any scalar
type), optional ``nx ny nz`` and ``red green blue`` (uchar -> [0, 1] like Open3D; float colours are
taken as they are), and whitespace-separated ``.xyz`` / ``.txt`` (x y z [nx ny nz]).  Coordinates are
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
        return np.stack([np.asarray(cols[k], dtype=np.float64) for k in keys], axis=1)

    if not all(k in cols for k in "xyz"):
        raise ValueError(f"{path}: vertices need x, y and z")
    cloud = PointCloud(stack("xyz"))
    if all(k in cols for k in ("nx", "ny", "nz")):
        cloud.normals = stack(("nx", "ny", "nz"))
    rgb = ("red", "green", "blue") if "red" in cols else ("r", "g", "b")
    if all(k in cols for k in rgb):
        colors = stack(rgb)
        if raw_dtypes[rgb[0]].kind in "ui":
            colors = colors / 255.0          # Open3D: uchar colours -> [0, 1]
        cloud.colors = colors
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


def read_point_cloud(path: str) -> PointCloud:
    low = str(path).lower()
    if low.endswith(".ply"):
        return _read_ply(path)
    if low.endswith((".xyz", ".xyzn", ".txt")):
        return _read_xyz(path)
    raise ValueError(f"{path}: unsupported point cloud format (PLY and XYZ are)")


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
