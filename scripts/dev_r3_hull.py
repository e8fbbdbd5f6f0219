"""dev: where get_extent() spends its time on the 0.8M-point voxelised surrogate (two-round thinning vs one)."""
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
for refine in (10**9, 8000, 8000):
    extent._REFINE_ABOVE = refine
    t0 = time.perf_counter(); cand = extent.hull_candidates(pts, eng, 0); t1 = time.perf_counter()
    hull = ConvexHull(pts[cand]); t2 = time.perf_counter()
    ext, _ = eng.obb_frames(pts[cand][hull.vertices], pts[cand][hull.simplices]); t3 = time.perf_counter()
    print(f"refine above {refine}: {len(cand)} candidates ({(t1-t0)*1e3:.1f} ms), hull {len(hull.vertices)} vertices ({(t2-t1)*1e3:.1f} ms), "
          f"frames ({(t3-t2)*1e3:.1f} ms), total {(t3-t0)*1e3:.1f} ms, extent {ext}")
