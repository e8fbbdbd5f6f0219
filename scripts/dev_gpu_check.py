"""Developer scratch: engine vs oracle on a few sizes + raw timing (run on the GPU box)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat
from oracle import oracle as orc

def check(na, nb, kind, seed, engine="brute"):
    rng = np.random.default_rng(seed)
    if kind == "uniform32":
        a = rng.random((na, 3), dtype=np.float32); b = rng.random((nb, 3), dtype=np.float32)
    elif kind == "f64":
        a = rng.random((na, 3)); b = a[rng.integers(0, na, nb)] + rng.normal(0, 1e-3, (nb, 3))
    elif kind == "lattice":
        a = rng.integers(0, 16, (na, 3)).astype(np.float64); b = rng.integers(0, 16, (nb, 3)).astype(np.float64)
    elif kind == "voxel":
        a = np.floor(rng.random((na, 3)) * 1024); b = (a[rng.integers(0, na, nb)] + rng.normal(0, .3, (nb, 3))).astype(np.float32)
    e = nat.Engine(0)
    e.set_cloud(0, a); e.set_cloud(1, b)
    ok = True
    for d, (q, r, skip) in enumerate(((a, b, False), (b, a, False), (a, a, True))):
        e.nn(d, engine)
        idx, d2 = e.fetch_nn(d)
        oi, od = orc.nn(np.asarray(q, np.float64), np.asarray(r, np.float64), skip_same_index=skip, method="kdtree")
        good = np.array_equal(idx, oi) and np.array_equal(d2, od)
        st = e.nn_stats(d)
        print(f"  {kind} {na}x{nb} dir{d}: {'OK' if good else 'MISMATCH'} idx_mis={(idx!=oi).sum()} d2_mis={(d2!=od).sum()} {st}")
        ok &= good
    e.close()
    return ok

if __name__ == "__main__":
    print("devices", nat.device_count())
    allok = True
    ENG = os.environ.get("ENG", "brute")
    for args in [(5, 7, "uniform32", 1), (1000, 1000, "uniform32", 2), (3000, 2500, "f64", 3), (2000, 2000, "lattice", 4),
                 (5000, 4000, "voxel", 5), (70000, 65537, "uniform32", 6)]:
        allok &= check(*args, engine=ENG)
    print("ALL OK" if allok else "FAILURES")
    # timing
    n = int(os.environ.get("N", 1000000))
    rng = np.random.default_rng(1234)
    a = rng.random((n, 3), dtype=np.float32); b = np.random.default_rng(5678).random((n, 3), dtype=np.float32)
    e = nat.Engine(0); e.set_cloud(0, a); e.set_cloud(1, b)
    e.profile(True)
    for rep in range(2):
        e.drop_caches()
        t = time.perf_counter(); e.nn(0, ENG); e.sync(); dt = time.perf_counter() - t
        print(f"nn left {n}x{n}: {dt*1e3:.2f} ms  -> {n*n/dt/1e12*8:.1f} TFLOP/s(8 flop/pair)")
    for d in (1, 2, 0):
        t = time.perf_counter(); e.nn(d, ENG); e.sync(); print(f"dir {d} (grids cached): {(time.perf_counter()-t)*1e3:.3f} ms", e.nn_stats(d))
    for k in ("scan", "refine", "fallback", "grid_build", "grid_query"):
        print(k, e.profile_get(k))
    print(e.nn_stats(0))
    idx, d2 = e.fetch_nn(0)
    oi, od = orc.nn(a.astype(np.float64), b.astype(np.float64), method="kdtree")
    print("1M parity:", np.array_equal(idx, oi), np.array_equal(d2, od))
