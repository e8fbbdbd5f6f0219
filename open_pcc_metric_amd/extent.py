"""CloudExtent on the host: extents of the minimal oriented bounding box of cloud A.

Stands under ``CloudPair.get_extent()``, cloud_pair.py:111-112
(``get_minimal_oriented_bounding_box().extent``).  That is Open3D 0.18 code (Qhull convex hull;
for every hull triangle the axis-aligned box in the triangle's frame; keep the smallest volume)
which is not in the reference checkout, so this restatement is NOT parity-pinned; pass
``extent=`` to ``CloudPair`` to inject the value a codec test bench already knows (e.g. the
voxel grid size).  CPU work, O(N log N) in Qhull -- not part of the GPU hot path.
"""
from __future__ import annotations

import numpy as np


def minimal_obb_extent(points) -> np.ndarray:
    from scipy.spatial import ConvexHull     # Qhull, the library Open3D uses as well

    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64))
    if pts.ndim != 2 or pts.shape[1] != 3 or pts.shape[0] < 4:
        raise ValueError("the minimal oriented bounding box needs at least 4 non-coplanar points")
    hull = ConvexHull(pts)
    verts = pts[hull.vertices]                                   # (H, 3)
    tri = pts[hull.simplices]                                    # (T, 3, 3)
    a = tri[:, 0]
    u = tri[:, 1] - a
    v = tri[:, 2] - a
    w = np.cross(u, v)
    v = np.cross(w, u)
    with np.errstate(invalid="ignore", divide="ignore"):
        frame = np.stack([x / np.linalg.norm(x, axis=1, keepdims=True) for x in (u, v, w)], axis=1)  # (T, 3, 3)
    best_vol, best_ext = np.inf, None
    step = max(1, (1 << 22) // max(1, len(verts)))
    for s in range(0, len(tri), step):
        loc = np.einsum("tij,thj->thi", frame[s:s + step], verts[None, :, :] - a[s:s + step, None, :])
        ext = loc.max(axis=1) - loc.min(axis=1)                  # (t, 3)
        vol = ext.prod(axis=1)
        vol[~np.isfinite(vol)] = np.inf
        k = int(np.argmin(vol))
        if vol[k] < best_vol:
            best_vol, best_ext = float(vol[k]), ext[k].copy()
    if best_ext is None:
        raise RuntimeError("degenerate convex hull")
    return best_ext
