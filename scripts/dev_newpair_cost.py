"""Developer scratch: cost of the once-per-pair decisions (trim box, cell size, kernel, isolation) for fresh pairs."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat
n = 1000000
rng = np.random.default_rng(0)
kinds = {"uniform": (rng.random((n, 3), dtype=np.float32), rng.random((n, 3), dtype=np.float32))}
v = rng.standard_normal((n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
s = np.unique(np.round(512 + 400 * v), axis=0).astype(np.float32)
kinds["voxel sphere"] = (s, np.unique(np.round(s + rng.normal(0, 0.7, s.shape)), axis=0).astype(np.float32))
e = nat.Engine(0)
for name, (a, b) in kinds.items():
    for rep in range(4):
        t0 = time.perf_counter(); e.set_cloud(0, a); e.set_cloud(1, b); e.sync()
        t1 = time.perf_counter(); e.nn_pair("auto"); e.sync()
        t2 = time.perf_counter(); e.drop_caches(); e.nn_pair("auto"); e.sync()
        t3 = time.perf_counter()
        print(f"{name:14s} rep {rep}: upload+ingest {1e3*(t1-t0):6.2f} ms | first search of the pair {1e3*(t2-t1):6.2f} ms | repeat {1e3*(t3-t2):6.2f} ms")
