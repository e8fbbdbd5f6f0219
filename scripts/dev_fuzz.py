"""Developer scratch: randomised differential test of the search engines against the oracle (GPU box).

    python scripts/dev_fuzz.py SECONDS [SEED]
"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_pcc_metric_amd import _native as nat
from oracle import oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def make(rng, n, kind):
    if kind == "uniform32":
        return rng.random((n, 3), dtype=np.float32).astype(np.float64)
    if kind == "uniform64":
        return rng.random((n, 3)) * rng.choice([1.0, 1e-3, 1e4])
    if kind == "offset64":
        return rng.random((n, 3)) * 50 + rng.choice([1e3, 1e5, 4e6]) * rng.random(3)
    if kind == "lattice":
        return rng.integers(0, rng.choice([4, 16, 64, 1024]), (n, 3)).astype(np.float64)
    if kind == "surface":
        v = rng.standard_normal((n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True) + 1e-30
        p = 100 + 60 * v
        return np.round(p) if rng.random() < 0.5 else p.astype(np.float32).astype(np.float64)
    if kind == "clusters":
        c = rng.random((max(1, n // 200), 3)) * 100
        return (c[rng.integers(0, len(c), n)] + rng.normal(0, 0.05, (n, 3))).astype(np.float32).astype(np.float64)
    if kind == "outliers":
        p = rng.random((n, 3), dtype=np.float32).astype(np.float64)
        k = max(1, n // 500)
        p[:k] = (rng.random((k, 3)) - 0.5) * rng.choice([1e2, 1e4, 1e6])
        return p
    if kind == "planar":
        p = rng.random((n, 3)); p[:, rng.integers(0, 3)] = 0.5
        return p
    if kind == "dups":
        base = rng.random((max(1, n // 50), 3), dtype=np.float32).astype(np.float64)
        return base[rng.integers(0, len(base), n)]
    raise KeyError(kind)


KINDS = ["uniform32", "uniform64", "offset64", "lattice", "surface", "clusters", "outliers", "planar", "dups"]
if os.environ.get("FUZZ_KINDS"):                 # e.g. FUZZ_KINDS=lattice,surface: voxelised pairs (voxel bricks, lattice kernel)
    KINDS = os.environ["FUZZ_KINDS"].split(",")
if __name__ != "__main__":
    budget = 0.0
e = nat.Engine(0) if __name__ == "__main__" else None
t_end = time.time() + budget
it = fails = 0
while __name__ == "__main__" and time.time() < t_end:
    rng = np.random.default_rng(seed0 * 1000003 + it)
    it += 1
    na = int(rng.choice([1, 2, 3, 17, 64, 65, 300, 1000, 4097, 20000, 60000, 150000, 300000]))
    nb = int(rng.choice([1, 2, 5, 64, 129, 777, 1000, 8192, 8193, 30000, 50000, 160000, 280000]))
    ka, kb = rng.choice(KINDS), rng.choice(KINDS)
    a, b = make(rng, na, ka), make(rng, nb, kb)
    if rng.random() < 0.3:
        b = b + a[rng.integers(0, na)] - b[0]            # make the clouds meet somewhere
    eng = str(rng.choice(["auto", "grid", "grid", "brute"]))
    if max(na, nb) > 100000 and eng == "brute":
        eng = "grid"
    mode = str(rng.choice(["none", "row", "neighbour"]))           # fused projection (round 2)
    nrm = None
    if mode != "none":
        nrm = []
        for m_ in (na, nb):
            g_ = rng.standard_normal((m_, 3))
            g_ /= np.linalg.norm(g_, axis=1, keepdims=True) + 1e-30
            nrm.append(g_.astype(np.float32).astype(np.float64) if rng.random() < 0.6 else g_)
    world = int(rng.choice([1, 1, 2, 3, 4, 5]))                      # sharded contexts, one "rank" after the other (round 2)
    if world > 1:
        try:
            e.set_cloud(0, a); e.set_cloud(1, b)
            e.nn_want_idx(bool(rng.random() < 0.5))
            for d in (0, 1):
                e.nn_fuse(d, None)
            pieces = {0: [], 1: []}
            vec = {0: None, 1: None}
            chunked = {d: (na if d == 0 else nb) >= world * 8192 for d in (0, 1)}
            from open_pcc_metric_amd.cloud_pair import shard_plan
            mode_s = str(rng.choice(["rows", "direction"]))
            plan = shard_plan(world, mode_s)
            sub = {d: max(w_ for _, w_ in plan[d]) for d in (0, 1)}
            chunked = {d: (na if d == 0 else nb) >= sub[d] * 8192 for d in (0, 1)}
            for rank in range(world):
                for d in (0, 1, 2):
                    e.set_shard_dir(d, *plan[d][rank])
                e.drop_caches(); e.nn_pair("grid" if eng == "brute" else eng)
                for d in (0, 1):
                    b0, e0 = e.shard_range(d)
                    if e0 <= b0:
                        continue
                    pieces[d].append(e.fetch_nn(d))
                    if chunked[d]:
                        buf, lens, mms = e.reduce_chunks_many([(d, nat.METRIC_D1)], "row")
                    else:
                        buf = e.reduce(d, nat.METRIC_D1, "row")[0]
                    vec[d] = buf.copy() if vec[d] is None else vec[d] + buf
            for d in (0, 1, 2):
                e.set_shard_dir(d, 0, 1)
            for d, (q, r) in enumerate(((a, b), (b, a))):
                oi, od = orc.nn(q, r, method="kdtree")
                idx = np.concatenate([p_[0] for p_ in pieces[d]]); d2 = np.concatenate([p_[1] for p_ in pieces[d]])
                tot = (e.finish_chunks if chunked[d] else e.finish_sum)(vec[d], len(q))
                if not (np.array_equal(d2, od) and np.array_equal(idx, oi) and np.float64(tot).tobytes() == np.float64(np.sum(od)).tobytes()):
                    fails += 1
                    print(f"MISMATCH (sharded x{world}) it={it} seed={seed0} dir={d} eng={eng} A={ka}:{na} B={kb}:{nb}", flush=True)
        except Exception as ex:                               # noqa: BLE001
            fails += 1
            for d in (0, 1, 2):
                e.set_shard_dir(d, 0, 1)
            print(f"ERROR (sharded x{world}) it={it} seed={seed0} eng={eng} A={ka}:{na} B={kb}:{nb}: {type(ex).__name__}: {ex}", flush=True)
        if it % 50 == 0:
            print(f"... {it} cases, {fails} failures", flush=True)
        continue
    try:
        e.set_cloud(0, a); e.set_cloud(1, b)
        e.nn_want_idx(bool(rng.random() < 0.5))
        for d in (0, 1):
            e.nn_fuse(d, None if mode == "none" else mode)
        if nrm is not None:
            e.set_normals(0, nrm[0]); e.set_normals(1, nrm[1])
        e.nn_pair(eng)
        e.nn(nat.DIR_SELF, eng)
        for d, (q, r, skip) in enumerate(((a, b, False), (b, a, False), (a, a, True))):
            _, d2_only = e.fetch_nn(d, want_idx=False)      # distances first: what a search without rows (voxel bricks) left
            idx, d2 = e.fetch_nn(d)
            if not np.array_equal(d2_only, d2):
                fails += 1
                print(f"MISMATCH (distances-only read) it={it} seed={seed0} dir={d} eng={eng} mode={mode} A={ka}:{na} B={kb}:{nb}", flush=True)
            if skip and na < 2:
                ok = bool(np.all(idx == -1) and np.all(d2 == 0))
            else:
                oi, od = orc.nn(q, r, skip_same_index=skip, method="kdtree")
                ok = np.array_equal(d2, od) and np.array_equal(idx, oi)
            if ok and nrm is not None and d < 2 and (mode == "neighbour" or len(q) <= len(r)):
                proj = orc.point_to_plane(q, r, oi, nrm[1 - d], normal_index=mode)
                tot = e.reduce_total(d, nat.METRIC_D2, mode)
                sq = np.square(proj)
                ok = (np.float64(tot[0]).tobytes() == np.float64(np.sum(sq)).tobytes() and tot[2] == np.max(sq)
                      and np.array_equal(e.point_metric(d, nat.METRIC_PROJ, mode), proj))
            if not ok:
                fails += 1
                print(f"MISMATCH it={it} seed={seed0} dir={d} eng={eng} mode={mode} A={ka}:{na} B={kb}:{nb}", flush=True)
    except Exception as ex:                               # noqa: BLE001
        fails += 1
        print(f"ERROR it={it} seed={seed0} eng={eng} A={ka}:{na} B={kb}:{nb}: {type(ex).__name__}: {ex}", flush=True)
    if it % 50 == 0:
        print(f"... {it} cases, {fails} failures", flush=True)
if __name__ == "__main__":
    print(f"FUZZ DONE: {it} cases, {fails} failures")
