"""CloudExtent: extents of the minimal oriented bounding box of cloud A.

Stands under ``CloudPair.get_extent()``, cloud_pair.py:111-112
(``get_minimal_oriented_bounding_box().extent``).  That is Open3D 0.18 code (Qhull convex hull;
for every hull triangle the axis-aligned box of the hull in the triangle's frame; keep the smallest
volume) which is not in the reference checkout, so this restatement is NOT parity-pinned; pass
``extent=`` to ``CloudPair`` to inject the value a codec test bench already knows (e.g. the voxel
grid size).  The hull is Qhull on the host (SciPy), like Open3D's; the search over the hull's
H vertices x T triangles -- seconds of NumPy for a rounded shape -- runs on the GPU
(``pccm_obb_frames``, csrc/pccm_obb.hip).
"""
from __future__ import annotations

import numpy as np


def convex_hull(points):
    """-> (hull vertices (H, 3), hull triangles as coordinates (T, 3, 3)) of an (N, 3) cloud."""
    from scipy.spatial import ConvexHull     # Qhull, the library Open3D uses as well

    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64))
    if pts.ndim != 2 or pts.shape[1] != 3 or pts.shape[0] < 4:
        raise ValueError("the minimal oriented bounding box needs at least 4 non-coplanar points")
    hull = ConvexHull(pts)
    return pts[hull.vertices], pts[hull.simplices]


def minimal_obb_extent(points, engine) -> np.ndarray:
    """Extents of the smallest box among the hull-face frames; ``engine``: the pair's ``_native.Engine``."""
    verts, tri = convex_hull(points)
    ext, _ = engine.obb_frames(verts, tri)
    return ext
