"""round-3 dev: persistent brick kernel vs oracle on one direction; prints mismatch statistics."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat
from oracle import oracle as orc
na, nb = int(os.environ.get("NA", 70000)), int(os.environ.get("NB", 65537))
rng = np.random.default_rng(na * 7 + nb)
a = rng.random((na, 3), dtype=np.float32); b = rng.random((nb, 3), dtype=np.float32)
e = nat.Engine(0)
e.set_cloud(0, a); e.set_cloud(1, b)
for d, (q, r) in enumerate(((a, b), (b, a))):
    e.nn(d, "grid")
    idx, d2 = e.fetch_nn(d)
    oi, od = orc.nn(q.astype(np.float64), r.astype(np.float64), skip_same_index=False, method="kdtree")
    bad = np.flatnonzero(d2 != od)
    print("dir", d, "n", len(q), "mismatch d2", len(bad), "idx", int(np.sum(idx != oi)), "stats", e.nn_stats(d))
    if len(bad):
        k = bad[:8]
        print("  rows", k, "\n  got d2", d2[k], "\n  want  ", od[k], "\n  got idx", idx[k], "want", oi[k])
        print("  got > want:", int(np.sum(d2[bad] > od[bad])), " got < want:", int(np.sum(d2[bad] < od[bad])), " got==0:", int(np.sum(d2[bad] == 0)))
