"""CloudExtent: extents of the minimal oriented bounding box of cloud A.

Stands under ``CloudPair.get_extent()``, cloud_pair.py:111-112
(``get_minimal_oriented_bounding_box().extent``).  That is Open3D 0.18 code (Qhull convex hull;
for every hull triangle the axis-aligned box of the hull in the triangle's frame; keep the smallest
volume) which is not in the reference checkout, so this restatement is NOT parity-pinned; pass
``extent=`` to ``CloudPair`` to inject the value a codec test bench already knows (e.g. the voxel
grid size).  The hull is Qhull on the host (SciPy), like Open3D's -- but only over the points that can
still be hull vertices: the GPU picks ~1000 extreme points of the resident cloud, a tiny Qhull run
turns them into an inner polytope, and the GPU reports the points that are not strictly inside it
(exact thinning: a point inside the hull of other points of the cloud is no hull vertex).  The search over
the hull's H vertices x T triangles -- seconds of NumPy for a rounded shape -- runs on the GPU as well
(``pccm_obb_frames``, csrc/pccm_obb.hip).
"""
from __future__ import annotations

import numpy as np

_THIN_ABOVE = 20000          # points; below that Qhull is instantaneous anyway
_DIRECTIONS = 1000


def _directions(k: int) -> np.ndarray:
    """k quasi-uniform unit vectors (Fibonacci lattice) plus the six axis directions."""
    i = np.arange(k) + 0.5
    phi = np.arccos(1.0 - 2.0 * i / k)
    theta = np.pi * (1.0 + 5.0 ** 0.5) * i
    d = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)
    axes = np.concatenate([np.eye(3), -np.eye(3)])
    return np.concatenate([axes, d]).astype(np.float32)


def hull_candidates(points: np.ndarray, engine, which: int = 0) -> np.ndarray:
    """Rows of ``points`` (= the engine's resident cloud ``which``) that may be vertices of its convex hull."""
    from scipy.spatial import ConvexHull, QhullError

    n = points.shape[0]
    if n < _THIN_ABOVE or not hasattr(engine, "rows_outside"):
        return np.arange(n)
    seeds = np.unique(engine.extreme_rows(which, _directions(_DIRECTIONS)))
    try:
        inner = ConvexHull(points[seeds])
    except (QhullError, ValueError):
        return np.arange(n)                       # flat or tiny spread: let the full run report it
    scale = float(np.max(np.abs(points[seeds]))) + 1.0
    outside = engine.rows_outside(which, inner.equations, 1e-9 * scale)
    return np.union1d(seeds, outside)


def convex_hull(points, engine=None, which: int = 0):
    """-> (hull vertices (H, 3), hull triangles as coordinates (T, 3, 3)) of an (N, 3) cloud."""
    from scipy.spatial import ConvexHull     # Qhull, the library Open3D uses as well

    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64))
    if pts.ndim != 2 or pts.shape[1] != 3 or pts.shape[0] < 4:
        raise ValueError("the minimal oriented bounding box needs at least 4 non-coplanar points")
    if engine is not None:
        pts = pts[hull_candidates(pts, engine, which)]
    hull = ConvexHull(pts)
    return pts[hull.vertices], pts[hull.simplices]


def minimal_obb_extent(points, engine) -> np.ndarray:
    """Extents of the smallest box among the hull-face frames; ``engine``: the pair's ``_native.Engine``, whose
    cloud 0 is ``points``."""
    verts, tri = convex_hull(points, engine, 0)
    ext, _ = engine.obb_frames(verts, tri)
    return ext
