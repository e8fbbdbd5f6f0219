"""Round-3 experiment: what would a spatially coherent row order buy?  Same synthetic pair as bench.py, rows permuted on the
host before upload: random (as generated), Morton (10 bits per axis), cell-linear (x fastest, ~1.4 points per cell),
brick-major.  Prints per-kernel-class us per step (eager, HIP events) and ms per step (hipGraph)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import open_pcc_metric_amd.metric as m
from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud

n = int(os.environ.get("N", 1000000))
a, b, na, nb = bench.synth(n)

def morton(p, bits=10):
    q = np.minimum((p.astype(np.float64) * (1 << bits)).astype(np.uint64), (1 << bits) - 1)
    key = np.zeros(len(p), dtype=np.uint64)
    for bit in range(bits):
        for ax in range(3):
            key |= ((q[:, ax] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + ax)
    return np.argsort(key, kind="stable")

def cell_linear(p, npts, ppc=1.4):
    d = int(round((npts / ppc) ** (1 / 3)))
    c = np.minimum((p.astype(np.float64) * d).astype(np.int64), d - 1)
    return np.argsort((c[:, 2] * d + c[:, 1]) * d + c[:, 0], kind="stable")

def blocks(p, npts, bx=46, by=4, bz=2):
    d = int(round((npts / 1.4) ** (1 / 3)))
    c = np.minimum((p.astype(np.float64) * d).astype(np.int64), d - 1)
    nbx, nby = -(-d // bx), -(-d // by)
    brick = ((c[:, 2] // bz) * nby + c[:, 1] // by) * nbx + c[:, 0] // bx
    inner = ((c[:, 2] % bz) * by + c[:, 1] % by) * bx + c[:, 0] % bx
    return np.argsort(brick * (bx * by * bz) + inner, kind="stable")

opts = CalculateOptions(color=None, hausdorff=False, point_to_plane=True)
def metrics():
    return transform_options(opts)[2:] + [m.GeoHausdorffDistance(True, False), m.GeoHausdorffDistance(False, False)]

for name in os.environ.get("ORDERS", "random,morton,cell,coarse,brick").split(","):
    if name == "random":
        pa, pb = np.arange(n), np.arange(n)
    elif name == "morton":
        pa, pb = morton(a), morton(b)
    elif name == "cell":
        pa, pb = cell_linear(a, n), cell_linear(b, n)
    elif name == "coarse":
        pa, pb = cell_linear(a, n, 4.0), cell_linear(b, n, 4.0)     # a cloud-intrinsic coarse (z, y, x) order: 64^3-ish at 1M
    else:
        pa, pb = blocks(a, n), blocks(b, n)
    A, B = np.ascontiguousarray(a[pa]), np.ascontiguousarray(b[pb])
    # quirk Q1: query row i of A reads row i of B's normals -> B's normals travel in A's order (and vice versa)
    NB, NA = np.ascontiguousarray(nb[pa]), np.ascontiguousarray(na[pb])
    with CloudPair(PointCloud(A, NA), PointCloud(B, NB), extent=[1.0, 1.0, 1.0]) as pair:
        eng = pair._engine
        def step():
            pair.recompute()
            return MetricCalculator(pair).calculate(metrics()).as_dict()
        for _ in range(5):
            r = step()
        eng.sync()
        K = 100
        t = time.perf_counter()
        for _ in range(K):
            r = step()
        eng.sync()
        ms = (time.perf_counter() - t) / K * 1e3
        pair._use_graph = False
        eng.profile(True); eng.profile_reset()
        for _ in range(10):
            step()
        eng.sync()
        prof = {k: round(eng.profile_get(k)[0] / 10 * 1e3, 1) for k in nat.KERNEL_CLASSES if eng.profile_get(k)[1]}
        eng.profile(False)
        print(json.dumps({"order": name, "n": n, "ms_per_step_graph": round(ms, 4), "kernel_us": prof,
                          "mse_d1": float(r[("SymmetricMetric", "GeoMSE", True, False, "GeoMSE", False, False)])}), flush=True)
