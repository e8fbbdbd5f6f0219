import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat
n = int(os.environ.get("N", 1000000))
a = np.random.default_rng(1234).random((n, 3), dtype=np.float32); b = np.random.default_rng(5678).random((n, 3), dtype=np.float32)
e = nat.Engine(0); e.set_cloud(0, a); e.set_cloud(1, b)
e.nn(0, "brute"); e.sync()
e.profile(True); e.profile_reset()
for _ in range(3): e.nn(0, "brute")
ms, k = e.profile_get("scan")
print(f"RESULT scan {ms/k:.2f} ms  {n*n*8/(ms/k*1e-3)/1e12:.1f} TFLOP/s  ({n*n*8/(ms/k*1e-3)/1e12/157.3*100:.1f}% of fp32 peak)  splits {e.nn_stats(0)['splits']}")
