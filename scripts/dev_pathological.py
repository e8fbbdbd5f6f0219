"""Timing + exactness of the grid engine on hostile point distributions (developer scratch)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat
from oracle import oracle as orc

def cases(n):
    rng = np.random.default_rng(0)
    u = lambda s: np.random.default_rng(s).random((n, 3), dtype=np.float32).astype(np.float64)
    out = {}
    a, b = u(1), u(2); a[0] = [1e6, 1e6, 1e6]; out["one_outlier"] = (a, b)
    a, b = u(1), u(2); a[:5] *= 1e4; b[:5] *= -1e4; out["few_outliers"] = (a, b)
    def lidar(s):
        g = np.random.default_rng(s)
        r = 1 + 99 * g.random(n); th = 2 * np.pi * g.random(n); ring = g.integers(0, 64, n)
        el = np.deg2rad(-25 + ring * 0.4)
        p = np.stack([r * np.cos(th) * np.cos(el), r * np.sin(th) * np.cos(el), r * np.sin(el)], 1)
        return (p + g.normal(0, 0.01, p.shape)).astype(np.float32).astype(np.float64)
    out["lidar"] = (lidar(3), lidar(4))
    out["identical_points"] = (np.full((n, 3), 0.5), np.full((n, 3), 0.25))
    t = np.random.default_rng(5).random(n); out["collinear"] = (np.stack([t, 2 * t, 3 * t], 1), np.stack([t[::-1], 2 * t[::-1] + 1e-3, 3 * t[::-1]], 1))
    a, b = u(6) * 0.01, u(7) * 0.01; a[n // 2:] += 1000; b[n // 2:] += 1000; out["two_clusters"] = (a, b)
    g = np.random.default_rng(8); out["gauss_clump"] = (g.normal(0, 1, (n, 3)) ** 3, g.normal(0, 1, (n, 3)) ** 3)
    a = u(11) * 100 + np.array([5.0e5, 5.6e6, 300.0]); out["utm_offset"] = (a, a + np.random.default_rng(12).normal(0, 0.01, a.shape))
    out["disjoint"] = (u(13), u(14) + np.array([3.0, 0, 0]))
    out["half_overlap"] = (u(15), u(16) + np.array([0.5, 0, 0]))
    out["dup_heavy"] = (np.floor(u(9) * 4), np.floor(u(10) * 4))
    return out

ENGINE = os.environ.get("ENG", "auto")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
only = sys.argv[2:] 
e = nat.Engine(0)
for name, (a, b) in cases(n).items():
    if only and name not in only: continue
    e.set_cloud(0, a); e.set_cloud(1, b)
    e.drop_caches()
    t = time.perf_counter(); e.nn_pair(ENGINE); e.sync(); dt = time.perf_counter() - t
    e.drop_caches()
    t = time.perf_counter(); e.nn_pair(ENGINE); e.sync(); dt2 = time.perf_counter() - t
    idx, d2 = e.fetch_nn(0)
    oi, od = orc.nn(a, b)
    print(f"{name:18s} n={n} first {dt*1e3:9.2f} ms  again {dt2*1e3:9.2f} ms  exact d2 {bool((d2 == od).all())} idx {bool((idx == oi).all())} stats {e.nn_stats(0)}", flush=True)
