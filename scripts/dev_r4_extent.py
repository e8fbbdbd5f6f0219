"""Round-4 developer scratch: where the minimal-OBB time of the content cloud goes (host Qhull runs vs GPU thinning)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from open_pcc_metric_amd import _native as nat  # noqa: E402
from open_pcc_metric_amd import extent as ex  # noqa: E402
from scipy.spatial import ConvexHull  # noqa: E402

ca, cb = bench.synth_content()
pts = np.ascontiguousarray(ca, dtype=np.float64)
e = nat.Engine(0)
e.set_cloud(0, ca)
for rep in range(3):
    t0 = time.perf_counter()
    seeds = np.unique(e.extreme_rows(0, ex._directions(ex._DIRECTIONS)))
    t1 = time.perf_counter()
    inner = ConvexHull(pts[seeds])
    t2 = time.perf_counter()
    scale = float(np.max(np.abs(pts[seeds]))) + 1.0
    outside = e.rows_outside(0, inner.equations, 1e-9 * scale)
    t3 = time.perf_counter()
    cand = np.union1d(seeds, outside)
    hull = ConvexHull(pts[cand])
    t4 = time.perf_counter()
    verts, tri = pts[cand][hull.vertices], pts[cand][hull.simplices]
    ext, _ = e.obb_frames(verts, tri)
    t5 = time.perf_counter()
    print(f"n {len(pts)} seeds {len(seeds)} inner facets {len(inner.equations)} outside {len(outside)} cand {len(cand)} hull verts {len(verts)} tris {len(tri)} | "
          f"extreme {1e3 * (t1 - t0):.2f} inner hull {1e3 * (t2 - t1):.2f} outside {1e3 * (t3 - t2):.2f} hull {1e3 * (t4 - t3):.2f} frames {1e3 * (t5 - t4):.2f} ms; extent {ext}")

for rep in range(3):
    t0 = time.perf_counter()
    got = ex.minimal_obb_extent(pts, e)
    print("minimal_obb_extent", 1e3 * (time.perf_counter() - t0), "ms", got, "candidates", len(ex.hull_candidates(pts, e, 0)))

