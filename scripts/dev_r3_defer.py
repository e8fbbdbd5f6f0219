import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat
n = int(os.environ.get("N", 150000))
a = np.random.default_rng(11).random((n, 3), dtype=np.float32)
b = np.random.default_rng(12).random((n, 3), dtype=np.float32)
na = np.random.default_rng(13).standard_normal((n, 3), dtype=np.float32)
nb = np.random.default_rng(14).standard_normal((n, 3), dtype=np.float32)
e = nat.Engine(0)
e.set_cloud(0, a); e.set_normals(0, na); e.set_cloud(1, b); e.set_normals(1, nb)
e.nn_fuse(0, "row"); e.nn_fuse(1, "row"); e.nn_want_idx(False)
e.nn_pair("auto")
tot = e.reduce_total_many([(0, 0), (0, 1), (1, 0), (1, 1)])
print("deferred totals:", [float(t[0]) for t in tot])
idx, d2 = e.fetch_nn(0)
err = a.astype(np.float64) - b.astype(np.float64)[idx]
p = np.einsum("ij,ij->i", err, nb.astype(np.float64))
print("numpy: d1", d2.sum(), "d2", (p * p).sum(), " only-x:", ((err[:, 0] * nb[:, 0].astype(np.float64)) ** 2).sum())
tot2 = e.reduce_total_many([(0, 0), (0, 1)])
print("after fetch (rows on):", [float(t[0]) for t in tot2])
