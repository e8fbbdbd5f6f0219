"""Developer scratch: average time of the grid query (dir 0, grids cached) and of a grid build."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat
n = int(os.environ.get("N", 1000000))
a = np.random.default_rng(1234).random((n, 3), dtype=np.float32); b = np.random.default_rng(5678).random((n, 3), dtype=np.float32)
e = nat.Engine(0); e.set_cloud(0, a); e.set_cloud(1, b)
e.nn(0, "grid"); e.nn(2, "grid"); e.sync()
e.profile(True); e.profile_reset()
for _ in range(20): e.nn(0, "grid")
q0 = e.profile_get("grid_query"); e.profile_reset()
for _ in range(20): e.nn(2, "grid")
q2 = e.profile_get("grid_query"); e.profile_reset()
for _ in range(10): e.drop_caches(); e.nn(0, "grid")
bld = e.profile_get("grid_build")
print(f"RESULT query_left {q0[0]/q0[1]*1e3:.1f} us   query_self {q2[0]/q2[1]*1e3:.1f} us   build(2 clouds) {bld[0]/bld[1]*1e3:.1f} us")
