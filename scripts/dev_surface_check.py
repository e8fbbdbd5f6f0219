"""Developer scratch: voxelised-surface clouds (the shape of MPEG 8i content) through the grid engine."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat
from oracle import oracle as orc
rng = np.random.default_rng(7)
n = int(os.environ.get("N", 800000))
# a bumpy closed surface sampled densely, snapped to a 10-bit voxel grid, duplicates removed
u = rng.random(n * 2) * 2 * np.pi; v = np.arccos(2 * rng.random(n * 2) - 1)
r = 380 + 40 * np.sin(3 * u) * np.sin(5 * v)
p = np.stack([512 + r * np.sin(v) * np.cos(u), 512 + 0.6 * r * np.sin(v) * np.sin(u), 512 + r * np.cos(v)], 1)
a = np.unique(np.round(p).astype(np.float32), axis=0)[:n]
b = np.unique(np.round(a + rng.normal(0, 0.7, a.shape)).astype(np.float32), axis=0)
print("A", a.shape, "B", b.shape)
e = nat.Engine(0); e.set_cloud(0, a); e.set_cloud(1, b)
for eng in ("grid",):
    e.drop_caches()
    t = time.perf_counter(); e.nn_pair(eng); e.nn(2, eng); e.sync(); dt = time.perf_counter() - t
    print(eng, "first pass ms", dt * 1e3)
    e.profile(True); e.profile_reset()
    for _ in range(5):
        e.drop_caches(); e.nn_pair(eng)
    e.sync()
    print({k: e.profile_get(k) for k in ("grid_build", "grid_query", "fallback")}, [e.nn_stats(d) for d in (0, 1)])
    e.profile(False)
ok = True
for d, (q, s, skip) in enumerate(((a, b, False), (b, a, False), (a, a, True))):
    idx, d2 = e.fetch_nn(d)
    oi, od = orc.nn(q.astype(np.float64), s.astype(np.float64), skip_same_index=skip, method="kdtree")
    good = np.array_equal(d2, od) and np.array_equal(idx, oi)
    print("dir", d, "bit-exact" if good else f"MISMATCH idx {(idx != oi).sum()} d2 {(d2 != od).sum()}")
    ok &= good
print("SURFACE OK" if ok else "SURFACE FAIL")
