"""dev: where get_extent() spends its time on the 0.8M-point voxelised surrogate.
Round 3, one MI355X box: candidates 29 258 of 800 000 (3.5 ms: extreme rows, 805-point inner hull, rows outside it), Qhull on
them 18-20 ms (2687 hull vertices), frame search 0.6-1.5 ms.  A second, finer polytope (5 x 1000 directions) thins to 10 596
candidates and Qhull to 12 ms but costs 6 ms itself, and the smallest box then comes from another of the hull's near-equal
frames (extent 528.4 x 521.3 instead of 522 x 528.2): not kept."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from open_pcc_metric_amd import _native as nat, extent
from scipy.spatial import ConvexHull

a, _ = bench.synth_content()
pts = a.astype(np.float64)
eng = nat.Engine(0)
eng.set_cloud(0, a); eng.set_cloud(1, a[:1000])
for _ in range(3):
    t0 = time.perf_counter(); cand = extent.hull_candidates(pts, eng, 0); t1 = time.perf_counter()
    hull = ConvexHull(pts[cand]); t2 = time.perf_counter()
    ext, _ = eng.obb_frames(pts[cand][hull.vertices], pts[cand][hull.simplices]); t3 = time.perf_counter()
    print(f"{len(cand)} candidates ({(t1-t0)*1e3:.1f} ms), hull {len(hull.vertices)} vertices ({(t2-t1)*1e3:.1f} ms), "
          f"frames ({(t3-t2)*1e3:.1f} ms), total {(t3-t0)*1e3:.1f} ms, extent {ext}")
