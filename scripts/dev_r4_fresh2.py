"""Round-4 developer scratch: where the staged fresh-array loop's every-other-pair 2.5 ms go (upload calls, search, report)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from open_pcc_metric_amd import _native as nat
a, b, na, nb = bench.synth(1_000_000)
e = nat.Engine(0)
e.set_io_staged(os.environ.get("STAGED", "1") == "1")
req = [(0, nat.METRIC_D1), (0, nat.METRIC_D2), (1, nat.METRIC_D1), (1, nat.METRIC_D2)]
keep = None
for it in range(10):
    cl = (np.array(a), np.array(na), np.array(b), np.array(nb))      # the next arrays exist before the previous ones go
    keep = cl
    t = [time.perf_counter()]
    e.set_cloud(0, cl[0]); t.append(time.perf_counter())
    e.set_cloud(1, cl[2]); t.append(time.perf_counter())
    e.set_normals(0, cl[1]); t.append(time.perf_counter())
    e.set_normals(1, cl[3]); t.append(time.perf_counter())
    e.drop_caches(); e.nn_pair("grid"); e.reduce_total_many(req, "row"); t.append(time.perf_counter())
    print(it, "ms:", " ".join(f"{(y - x) * 1e3:.2f}" for x, y in zip(t, t[1:])), "| addr", hex(cl[0].ctypes.data))
