"""CPU oracle for the open-pcc-metric hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker / CPU baseline.  ``open_pcc_metric_amd`` never
imports it and has no CPU fallback.

It restates, in NumPy over the C searches of ``pccm_oracle.c``, what the reference computes
on the ``CloudPair`` / ``MetricCalculator`` path (paths relative to /root/reference):

* ``cloud_pair.py:10-42, 54-124`` -- directional exact 1-NN, cached d2, error vectors,
  self-search boundary distances;
* ``metric.py:124-247, 353-386, 446-485`` -- ErrorVector (D2 projection), EuclideanDistance,
  GeoMSE, GeoPSNR, GeoHausdorffDistance(+PSNR), Min/MaxSqrtDistance, SymmetricMetric;
* ``options.py:32-174`` -- which metrics a set of options yields, in which order.

Parity pins: the NumPy half is pinned bit-for-bit by golden vectors produced
by the reference's own ``metric.py``/``calculator.py``/``options.py`` (tests/golden/, made by
tests/golden/make_golden.py) and by the reference's known-answer tests
(tests/unit/test_metric.py:30-70 and the fixture :13-26).  The kNN itself is Open3D code that
is not in /root/reference: exact NN distances are mathematically unique, so d2, MSE and
Hausdorff are pinned by the brute-force definition; tie-broken indices are NOT pinned.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "pccm_oracle.c"))
    ):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is not None:
        return _lib
    build()
    lib = ctypes.CDLL(_LIB_PATH)
    c_dp = ctypes.POINTER(ctypes.c_double)
    c_ip = ctypes.POINTER(ctypes.c_int64)
    lib.orc_num_threads.restype = ctypes.c_int
    lib.orc_nn_brute.argtypes = [c_dp, ctypes.c_int64, c_dp, ctypes.c_int64, ctypes.c_int, c_ip, c_dp]
    lib.orc_nn_brute.restype = ctypes.c_int
    lib.orc_kdtree_build.argtypes = [c_dp, ctypes.c_int64]
    lib.orc_kdtree_build.restype = ctypes.c_void_p
    lib.orc_kdtree_free.argtypes = [ctypes.c_void_p]
    lib.orc_kdtree_free.restype = None
    lib.orc_kdtree_query.argtypes = [ctypes.c_void_p, c_dp, ctypes.c_int64, ctypes.c_int, c_ip, c_dp,
                                     ctypes.c_int]
    lib.orc_kdtree_query.restype = ctypes.c_int
    lib.orc_point_to_plane.argtypes = [c_dp, ctypes.c_int64, c_dp, c_ip, c_dp, ctypes.c_int64,
                                       ctypes.c_int, c_dp]
    lib.orc_point_to_plane.restype = ctypes.c_int
    lib.orc_color_columns.argtypes = [c_dp, ctypes.c_int64, c_dp, ctypes.c_int64, c_ip, ctypes.c_int, ctypes.c_double,
                                      c_dp, c_dp, c_dp]
    lib.orc_color_columns.restype = ctypes.c_int
    lib.orc_knn_brute.argtypes = [c_dp, ctypes.c_int64, c_dp, ctypes.c_int64, ctypes.c_int, c_ip]
    lib.orc_knn_brute.restype = ctypes.c_int
    _lib = lib
    return lib


def _f64(a) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("expected an (N, 3) array")
    return a


def _dp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _ip(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


def num_threads() -> int:
    return int(_load().orc_num_threads())


def nn(iter_pts, search_pts, *, skip_same_index: bool = False, method: str = "auto",
       threads: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Exact 1-NN of every row of ``iter_pts`` in ``search_pts`` (cloud_pair.py:10-42).

    Returns ``(idx int64, d2 float64)``; exact ties go to the smallest row index.
    ``method``: "brute" (definition), "kdtree", or "auto" (brute up to 4e7 pairs).
    """
    lib = _load()
    q = _f64(iter_pts)
    r = _f64(search_pts)
    idx = np.empty(q.shape[0], dtype=np.int64)
    d2 = np.empty(q.shape[0], dtype=np.float64)
    if method == "auto":
        method = "brute" if q.shape[0] * max(r.shape[0], 1) <= 40_000_000 else "kdtree"
    if method == "brute":
        rc = lib.orc_nn_brute(_dp(q), q.shape[0], _dp(r), r.shape[0], int(skip_same_index), _ip(idx), _dp(d2))
    elif method == "kdtree":
        tree = lib.orc_kdtree_build(_dp(r), r.shape[0])
        try:
            rc = lib.orc_kdtree_query(tree, _dp(q), q.shape[0], int(skip_same_index), _ip(idx), _dp(d2),
                                      int(threads))
        finally:
            lib.orc_kdtree_free(tree)
    else:
        raise ValueError(method)
    if rc != 0:
        raise RuntimeError(f"oracle nn failed rc={rc}")
    return idx, d2


class KDTree:
    """A kd-tree kept across calls, queried ONE point at a time: the reference's calling pattern (one
    ``search_knn_vector_3d`` call per point out of ``np.apply_along_axis``, cloud_pair.py:16-32).  bench.py times it."""

    def __init__(self, points):
        self._lib = _load()
        self._pts = _f64(points)
        self._tree = self._lib.orc_kdtree_build(_dp(self._pts), self._pts.shape[0])
        self._idx = np.empty(1, dtype=np.int64)
        self._d2 = np.empty(1, dtype=np.float64)

    def search_1nn(self, point) -> Tuple[int, float]:
        q = np.ascontiguousarray(point, dtype=np.float64).reshape(1, 3)
        rc = self._lib.orc_kdtree_query(self._tree, _dp(q), 1, 0, _ip(self._idx), _dp(self._d2), 1)
        if rc != 0:
            raise RuntimeError(f"oracle kd-tree query failed rc={rc}")
        return int(self._idx[0]), float(self._d2[0])

    def close(self):
        if self._tree:
            self._lib.orc_kdtree_free(self._tree)
            self._tree = None

    __del__ = close


def point_to_plane(iter_pts, search_pts, nn_idx, other_normals, *, normal_index: str = "row") -> np.ndarray:
    """ErrorVector(point_to_plane=True).value, metric.py:146-153.

    ``normal_index="row"`` reproduces the reference (row i of the other cloud's normals,
    metric.py:130,148-152) and raises IndexError like it when that row does not exist;
    ``"neighbour"`` uses the matched point's normal.
    """
    lib = _load()
    q = _f64(iter_pts)
    r = _f64(search_pts)
    nrm = _f64(other_normals)
    idx = np.ascontiguousarray(nn_idx, dtype=np.int64)
    out = np.empty(q.shape[0], dtype=np.float64)
    rc = lib.orc_point_to_plane(_dp(q), q.shape[0], _dp(r), _ip(idx), _dp(nrm), nrm.shape[0],
                                int(normal_index == "neighbour"), _dp(out))
    if rc == -2:
        raise IndexError("index out of bounds for the other cloud's normals (reference quirk Q1)")
    if rc != 0:
        raise RuntimeError(f"oracle point_to_plane failed rc={rc}")
    return out


def minimal_obb_extent(points) -> np.ndarray:
    """CloudExtent, cloud_pair.py:111-112: extents of the minimal oriented bounding box as Open3D 0.18 searches
    for it (Qhull hull; for every hull triangle the box of the hull vertices in the triangle's frame -- x along
    its first edge, z along its normal; smallest volume, first one on ties).  NumPy, O(H x T): small clouds only."""
    from scipy.spatial import ConvexHull

    pts = _f64(points)
    hull = ConvexHull(pts)
    verts, tri = pts[hull.vertices], pts[hull.simplices]
    best_vol, best_ext = np.inf, None
    for a, b, c in tri:
        u, v = b - a, c - a
        w = np.cross(u, v)
        v = np.cross(w, u)
        with np.errstate(invalid="ignore", divide="ignore"):
            frame = np.stack([x / np.sqrt(np.sum(x * x)) for x in (u, v, w)])
        loc = (verts - a) @ frame.T
        ext = loc.max(axis=0) - loc.min(axis=0)
        vol = ext.prod()
        if np.isfinite(vol) and vol < best_vol:
            best_vol, best_ext = float(vol), ext
    if best_ext is None:
        raise RuntimeError("degenerate convex hull")
    return best_ext


COLOR_SCHEMES = {"rgb": 0, "ycc": 1, "yuv": 2}


def color_columns(own_rgb, other_rgb, nn_idx, scheme: str, scale: float = 1.0):
    """-> (sq, sums, maxs): ``sq = (scale * (T(own) - T(other[nn])))**2`` rows, their column sums in
    ``np.mean(..., axis=0)``'s row order and their column maxima (metric.py:261-333, 389-427)."""
    lib = _load()
    a, b = _f64(own_rgb), _f64(other_rgb)
    idx = np.ascontiguousarray(nn_idx, dtype=np.int64)
    sq = np.empty((a.shape[0], 3), dtype=np.float64)
    sums, maxs = np.empty(3), np.empty(3)
    rc = lib.orc_color_columns(_dp(a), a.shape[0], _dp(b), b.shape[0], _ip(idx), COLOR_SCHEMES[scheme], float(scale),
                               _dp(sq), _dp(sums), _dp(maxs))
    if rc == -2:
        raise IndexError("neighbour row outside the other cloud")
    if rc != 0:
        raise RuntimeError(f"oracle color_columns failed rc={rc}")
    return sq, sums, maxs


def color_mse(own_rgb, other_rgb, nn_idx, scheme: str) -> np.ndarray:
    """ColorMSE.value, metric.py:302-333."""
    _, sums, _ = color_columns(own_rgb, other_rgb, nn_idx, scheme)
    return sums / len(own_rgb)


def color_hausdorff(own_rgb, other_rgb, nn_idx, scheme: str) -> np.ndarray:
    """ColorHausdorffDistance.value, metric.py:389-427 (differences scaled by 255 in "rgb")."""
    return color_columns(own_rgb, other_rgb, nn_idx, scheme, 255.0 if scheme == "rgb" else 1.0)[2]


def knn(points, k: int) -> np.ndarray:
    """(n, k) rows of the k nearest points of the cloud to each of its points (itself included)."""
    p = _f64(points)
    idx = np.empty((p.shape[0], k), dtype=np.int64)
    rc = _load().orc_knn_brute(_dp(p), p.shape[0], _dp(p), p.shape[0], int(k), _ip(idx))
    if rc != 0:
        raise RuntimeError(f"oracle knn failed rc={rc}")
    return idx


def estimate_normals(points, k: int = 30):
    """Restatement of Open3D's PointCloud.estimate_normals() defaults (cloud_pair.py:61-64): covariance of
    the k nearest points (self included), eigenvector of the smallest eigenvalue.  Not pinned by the
    reference (Open3D is absent); returns (normals, eigenvalues ascending) with unspecified sign."""
    p = _f64(points)
    n = p.shape[0]
    kk = min(k, n)
    nbr = knn(p, kk)
    normals = np.tile(np.array([0.0, 0.0, 1.0]), (n, 1))
    evals = np.zeros((n, 3))
    if kk < 3:
        return normals, evals
    d = p[nbr] - p[:, None, :]                              # (n, k, 3), relative to the query point
    mean = d.mean(axis=1)
    cov = np.einsum("nki,nkj->nij", d, d) / kk - mean[:, :, None] * mean[:, None, :]
    w, v = np.linalg.eigh(cov)
    normals = v[:, :, 0]
    return normals, w


class OraclePair:
    """Restatement of CloudPair (cloud_pair.py:45-124) + the geometry metric DAG.

    ``clouds[0]`` = origin (A), ``clouds[1]`` = reconstructed (B); "left" iterates A and
    searches B (cloud_pair.py:67-72), "right" the other way (cloud_pair.py:73-78).
    """

    def __init__(self, a_points, b_points, a_normals=None, b_normals=None, *, method: str = "auto",
                 threads: int = 0, normal_index: str = "row"):
        self.points = (_f64(a_points), _f64(b_points))
        self.normals = (None if a_normals is None else _f64(a_normals),
                        None if b_normals is None else _f64(b_normals))
        self.normal_index = normal_index
        self._method = method
        self._threads = threads
        li, ld = nn(self.points[0], self.points[1], method=method, threads=threads)
        ri, rd = nn(self.points[1], self.points[0], method=method, threads=threads)
        self.nn_idx = (li, ri)
        self.nn_d2 = (ld, rd)
        self._boundary = None

    # cloud_pair.py:90-100
    def error_vector(self, is_left: bool) -> np.ndarray:
        k = 0 if is_left else 1
        return np.subtract(self.points[k], np.take(self.points[1 - k], self.nn_idx[k], axis=0))

    # cloud_pair.py:102-106
    def neighbour_distances(self, is_left: bool) -> np.ndarray:
        return self.nn_d2[0 if is_left else 1]

    # cloud_pair.py:108-109 (Open3D compute_nearest_neighbor_distance)
    def boundary_sqrt_distances(self) -> np.ndarray:
        if self._boundary is None:
            a = self.points[0]
            if a.shape[0] < 2:
                self._boundary = np.zeros(a.shape[0])
            else:
                _, d2 = nn(a, a, skip_same_index=True, method=self._method, threads=self._threads)
                self._boundary = np.sqrt(d2)
        return self._boundary

    # metric.py:124-153 (p2plane branch) + :156-179
    def euclidean_distance(self, is_left: bool, point_to_plane_: bool) -> np.ndarray:
        k = 0 if is_left else 1
        if not point_to_plane_:
            return self.nn_d2[k]                       # metric.py:175-177
        other = self.normals[1 - k]                    # metric.py:130 CloudNormals(not is_left)
        if other is None:
            raise ValueError("point-to-plane needs normals on both clouds")
        proj = point_to_plane(self.points[k], self.points[1 - k], self.nn_idx[k], other,
                              normal_index=self.normal_index)
        return np.square(proj)                         # metric.py:179

    def geo_mse(self, is_left: bool, p2p: bool):       # metric.py:226-228
        v = self.euclidean_distance(is_left, p2p)
        return np.sum(v, axis=0) / v.shape[0]

    def geo_hausdorff(self, is_left: bool, p2p: bool):  # metric.py:366
        return np.max(self.euclidean_distance(is_left, p2p), axis=0)

    def min_max_sqrt(self):                            # metric.py:187-188
        d = self.boundary_sqrt_distances()
        return np.min(d), np.max(d)

    @staticmethod
    def psnr(peak, mse):                               # metric.py:246-247 / :384-386
        with np.errstate(divide="ignore"):
            return 10 * np.log10(peak ** 2 / mse)

    @staticmethod
    def symmetric(left, right, is_proportional: bool):  # metric.py:475-485
        vals = [left, right]
        return min(vals, key=np.linalg.norm) if is_proportional else max(vals, key=np.linalg.norm)

    def report(self, *, hausdorff: bool = False, point_to_plane_: bool = False,
               peak: Optional[float] = None) -> Dict[tuple, object]:
        """Values keyed exactly like CalculateResult.as_dict() (calculator.py:21-25) for the
        geometry rows of transform_options (options.py:35-56, 84-172).  ``peak`` stands for
        ``max(CloudExtent)`` (metric.py:246); rows needing it are omitted when it is None."""
        out: Dict[tuple, object] = {}
        mn, mx = self.min_max_sqrt()
        out[("MinSqrtDistance",)] = mn
        out[("MaxSqrtDistance",)] = mx

        def sym(name, p2p, lv, rv, prop):
            out[("SymmetricMetric", name, True, p2p, name, False, p2p)] = self.symmetric(lv, rv, prop)

        def block(p2p):
            ml, mr = self.geo_mse(True, p2p), self.geo_mse(False, p2p)
            out[("GeoMSE", True, p2p)] = ml
            out[("GeoMSE", False, p2p)] = mr
            sym("GeoMSE", p2p, ml, mr, False)
            if peak is not None:
                pl, pr = self.psnr(peak, ml), self.psnr(peak, mr)
                out[("GeoPSNR", True, p2p)] = pl
                out[("GeoPSNR", False, p2p)] = pr
                sym("GeoPSNR", p2p, pl, pr, True)

        def hblock(p2p):
            hl, hr = self.geo_hausdorff(True, p2p), self.geo_hausdorff(False, p2p)
            pl, pr = self.psnr(mx, hl), self.psnr(mx, hr)
            out[("GeoHausdorffDistance", True, p2p)] = hl
            out[("GeoHausdorffDistance", False, p2p)] = hr
            if not p2p:                                 # options.py:106-138
                sym("GeoHausdorffDistance", p2p, hl, hr, False)
            out[("GeoHausdorffDistancePSNR", True, p2p)] = pl
            out[("GeoHausdorffDistancePSNR", False, p2p)] = pr
            if p2p:                                     # options.py:140-172: both symmetric rows last
                sym("GeoHausdorffDistance", p2p, hl, hr, False)
            sym("GeoHausdorffDistancePSNR", p2p, pl, pr, True)

        block(False)
        if point_to_plane_:
            block(True)
        if hausdorff:
            hblock(False)
            if point_to_plane_:
                hblock(True)
        return out
