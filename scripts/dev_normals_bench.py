import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rng = np.random.default_rng(0)
kind = sys.argv[2] if len(sys.argv) > 2 else "uniform"
if kind == "uniform":
    a = rng.random((n, 3), dtype=np.float32)
else:   # voxelised sphere surface
    v = rng.standard_normal((n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    a = np.unique(np.round(v * 500 + 512), axis=0).astype(np.float32)
e = nat.Engine(0)
e.set_cloud(0, a); e.set_cloud(1, a[: len(a) // 2])
for _ in range(3):
    t = time.perf_counter(); e.estimate_normals(0, 30); e.sync(); dt = time.perf_counter() - t
    print(kind, len(a), "estimate_normals ms", round(dt * 1e3, 2))
