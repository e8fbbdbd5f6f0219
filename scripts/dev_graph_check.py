import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat
n = 200000
a = np.random.default_rng(1).random((n, 3), dtype=np.float32); b = np.random.default_rng(2).random((n, 3), dtype=np.float32)
e = nat.Engine(0); e.set_cloud(0, a); e.set_cloud(1, b)
e.profile(True)
e.drop_caches(); e.nn(0); e.nn(1); e.reduce_prefetch(0, 0); x = e.reduce(0, 0)
print("eager", e.profile_get("grid_query"), e.profile_get("grid_build"))
e.profile_reset()
e.graph_begin(); e.drop_caches(); e.nn(0); e.nn(1); e.reduce_prefetch(0, 0); gid = e.graph_end()
x1 = e.reduce(0, 0)
print("after capture", e.profile_get("grid_query"), e.profile_get("grid_build"))
for i in range(3):
    e.graph_launch(gid); x2 = e.reduce(0, 0)
print("after 3 replays", e.profile_get("grid_query"), e.profile_get("grid_build"), e.profile_get("reduce"))
print("same result", np.array_equal(x[0], x2[0]), x[1] == x2[1], x[2] == x2[2])
