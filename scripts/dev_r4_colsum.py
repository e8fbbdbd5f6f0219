import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat
n = int(os.environ.get("N", 800000))
rng = np.random.default_rng(1)
cols = (rng.random((n, 3)) * 0.05) ** 2
e = nat.Engine(0)
got = e.seq_colsum(cols)
want = np.add.reduce(cols, axis=0)
print("equal", np.array_equal(got, want))
e.profile(True); e.profile_reset()
for _ in range(5): e.seq_colsum(cols)
print("reduce class us per call", e.profile_get("reduce")[0] / 5 * 1e3)
