"""Developer scratch: randomised differential test of whole reports (D1/D2/Hausdorff/colour) against the oracle.

    python scripts/dev_fuzz_report.py SECONDS [SEED]
"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import open_pcc_metric_amd.metric as opmm
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from oracle import oracle as orc
from dev_fuzz import make, KINDS          # noqa: E402  (same directory)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t_end = time.time() + budget
it = fails = 0


def same(x, y):
    x, y = np.atleast_1d(np.asarray(x, dtype=np.float64)), np.atleast_1d(np.asarray(y, dtype=np.float64))
    return x.shape == y.shape and bool(np.all((x == y) | (np.isnan(x) & np.isnan(y))))


while time.time() < t_end:
    rng = np.random.default_rng(seed0 * 7919 + it)
    it += 1
    na = int(rng.choice([3, 64, 300, 1000, 8192, 8193, 20000, 70000]))
    nb = na if rng.random() < 0.6 else int(rng.choice([5, 129, 1000, 9000, 30000]))
    a, b = make(rng, na, rng.choice(KINDS)), make(rng, nb, rng.choice(KINDS))
    nrm_a, nrm_b = rng.standard_normal((na, 3)), rng.standard_normal((nb, 3))
    ca, cb = rng.integers(0, 256, (na, 3)) / 255.0, rng.integers(0, 256, (nb, 3)) / 255.0
    mode = "row" if (na == nb or rng.random() < 0.3) else "neighbour"
    scheme = str(rng.choice(["rgb", "ycc", "yuv"]))
    use_graph = bool(rng.random() < 0.3)
    try:
        want = orc.OraclePair(a, b, nrm_a, nrm_b, normal_index=mode, method="kdtree")
        pair = CloudPair(PointCloud(a, nrm_a, ca), PointCloud(b, nrm_b, cb), extent=[1.0, 2.0, 0.5], normal_index=mode,
                         use_graph=use_graph)
        opts = transform_options(CalculateOptions(None, True, True))
        expect_error = mode == "row" and na != nb
        for rep in range(3 if use_graph else 1):
            try:
                with np.errstate(divide="ignore"):
                    got = MetricCalculator(pair).calculate(opts).as_dict()
            except IndexError:
                if not expect_error:
                    raise
                got = None
            if expect_error:
                if got is not None:
                    fails += 1
                    print(f"NO-INDEXERROR it={it} seed={seed0} na={na} nb={nb}", flush=True)
                break
            with np.errstate(divide="ignore"):
                ref = want.report(hausdorff=True, point_to_plane_=True, peak=2.0)
            bad = [k for k in ref if not same(got[k], ref[k])]
            if bad or list(got) != list(ref):
                fails += 1
                print(f"MISMATCH it={it} seed={seed0} na={na} nb={nb} mode={mode} graph={use_graph} rep={rep}: {bad[:3]}", flush=True)
                break
            pair.recompute()
        if not expect_error:
            for is_left, own, other, oc, rc in ((True, a, b, ca, cb), (False, b, a, cb, ca)):
                idx, _ = orc.nn(own, other, method="kdtree")
                with np.errstate(divide="ignore"):
                    mse = MetricCalculator(pair)._metric_recursive_calculate(opmm.ColorMSE(is_left=is_left, color_scheme=scheme)).value
                    hd = MetricCalculator(pair)._metric_recursive_calculate(opmm.ColorHausdorffDistance(is_left=is_left, color_scheme=scheme)).value
                if not (same(mse, orc.color_mse(oc, rc, idx, scheme)) and same(hd, orc.color_hausdorff(oc, rc, idx, scheme))):
                    fails += 1
                    print(f"COLOUR MISMATCH it={it} seed={seed0} na={na} nb={nb} scheme={scheme} left={is_left}", flush=True)
        pair.close()
    except Exception as ex:                               # noqa: BLE001
        fails += 1
        print(f"ERROR it={it} seed={seed0} na={na} nb={nb} mode={mode}: {type(ex).__name__}: {ex}", flush=True)
    if it % 25 == 0:
        print(f"... {it} reports, {fails} failures", flush=True)
print(f"REPORT FUZZ DONE: {it} reports, {fails} failures")
