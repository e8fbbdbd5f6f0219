"""Developer scratch: cooperative vs per-thread grid query on clouds of different shapes (run with PCCM_GRID_COOP=1 and =0)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat

def shapes(n):
    rng = np.random.default_rng(0)
    f32 = lambda x: x.astype(np.float32)
    out = {}
    out["cube"] = (f32(rng.random((n, 3))), f32(rng.random((n, 3))))
    for t in (2, 8, 32, 128):
        a = rng.random((n, 3)); b = rng.random((n, 3)); a[:, 2] /= t; b[:, 2] /= t
        out[f"slab_1/{t}"] = (f32(a), f32(b))
    def sphere(m, noise, voxel):
        v = rng.standard_normal((m, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        p = 512 + 400 * v + rng.normal(0, noise, v.shape)
        return np.unique(np.round(p), axis=0).astype(np.float32) if voxel else f32(p)
    out["sphere_float"] = (sphere(n, 0.0, False), sphere(n, 0.5, False))
    out["sphere_voxel"] = (sphere(int(n * 1.3), 0.0, True), sphere(int(n * 1.3), 0.7, True))
    out["cube_voxel"] = (np.unique(np.floor(rng.random((n, 3)) * 128), axis=0).astype(np.float32),
                         np.unique(np.floor(rng.random((n, 3)) * 128), axis=0).astype(np.float32))
    th = rng.random(n) * 2 * np.pi; r = 1 + 99 * rng.random(n); el = np.deg2rad(-25 + rng.integers(0, 64, n) * 0.4)
    lid = lambda s: f32(np.stack([r * np.cos(th + s) * np.cos(el), r * np.sin(th + s) * np.cos(el), r * np.sin(el)], 1))
    out["lidar"] = (lid(0.0), lid(1e-3))
    return out

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
only = sys.argv[2:]
e = nat.Engine(0)
def more(n):
    rng = np.random.default_rng(1)
    out = {}
    for sig in (3, 10, 30, 100):
        def shell(s):
            v = rng.standard_normal((n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
            return (512 + (400 + rng.normal(0, sig, (n, 1))) * v).astype(np.float32)
        out[f"shell_sigma{sig}"] = (shell(0), shell(1))
    return out
allshapes = dict(shapes(n)); allshapes.update(more(n))
for name, (a, b) in allshapes.items():
    if only and name not in only: continue
    e.set_cloud(0, a); e.set_cloud(1, b)
    e.nn_pair("grid"); e.sync()
    e.profile(True); e.profile_reset()
    for _ in range(5):
        e.drop_caches(); e.nn_pair("grid")
    e.sync()
    q = (e.profile_get("grid_query")[0] + e.profile_get("grid_finish")[0]) / 5
    bld = e.profile_get("grid_build")[0] / 5
    e.profile(False)
    print(f"RESULT {name:14s} nA {len(a):8d} nB {len(b):8d} query_us {q*1e3:9.1f} build_us {bld*1e3:9.1f}", flush=True)
