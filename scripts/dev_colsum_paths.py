"""Developer scratch: cost of the fast and the slow path of k_color_colsum."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat
n = 1_000_000
rng = np.random.default_rng(0)
e = nat.Engine(0); e.profile(True)
cases = {"typical (k/255 squares)": ((rng.integers(0, 256, (n, 3)) - rng.integers(0, 256, (n, 3))) / 255.0) ** 2}
x = cases["typical (k/255 squares)"].copy(); x[0] = 3.0e5; cases["one binade (big first element)"] = x
cases["wide (60 binades)"] = rng.random((n, 3)) * np.exp2(rng.integers(-40, 20, (n, 3)))
for name, a in cases.items():
    for _ in range(3):
        e.profile_reset(); s = e.seq_colsum(a)
    ms = e.profile_get("reduce")[0]
    print(f"{name:34s} kernel {ms*1e3:8.1f} us  exact {bool((s == np.add.reduce(a, axis=0)).all())}")
