"""dev: where get_extent() spends its time on the 0.8M-point voxelised surrogate."""
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
for rep in range(2):
    t0 = time.perf_counter(); seeds = np.unique(eng.extreme_rows(0, extent._directions(1000))); t1 = time.perf_counter()
    inner = ConvexHull(pts[seeds]); t2 = time.perf_counter()
    scale = float(np.max(np.abs(pts[seeds]))) + 1.0
    outside = eng.rows_outside(0, inner.equations, 1e-9 * scale); t3 = time.perf_counter()
    cand = np.union1d(seeds, outside); t4 = time.perf_counter()
    hull = ConvexHull(pts[cand]); t5 = time.perf_counter()
    verts, tri = pts[cand][hull.vertices], pts[cand][hull.simplices]
    ext, _ = eng.obb_frames(verts, tri); t6 = time.perf_counter()
    print(f"seeds {len(seeds)} ({(t1-t0)*1e3:.1f} ms) inner hull {len(inner.equations)} facets ({(t2-t1)*1e3:.1f}) outside {len(outside)} ({(t3-t2)*1e3:.1f}) "
          f"union ({(t4-t3)*1e3:.1f}) final hull of {len(cand)}: {len(hull.vertices)} vertices {len(hull.simplices)} facets ({(t5-t4)*1e3:.1f}) frames ({(t6-t5)*1e3:.1f}) total {(t6-t0)*1e3:.1f} ms ext {ext}")
