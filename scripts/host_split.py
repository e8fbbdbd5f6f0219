"""Where the host's share of a headline step goes (dev tool, GPU box only).

Times, over many steps of the bench's own step(), (a) the hipGraph launch (CloudPair.recompute), (b) planning the metric
program, (c) the wait for the reductions + finishing the totals, (d) the Python metric chain behind them -- and the same
step at a size where the GPU's share is negligible (the host floor)."""
import argparse, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=300)
    args = ap.parse_args()
    import bench
    from open_pcc_metric_amd.cloud_pair import CloudPair
    from open_pcc_metric_amd.point_cloud import PointCloud
    from open_pcc_metric_amd.calculator import MetricCalculator
    from open_pcc_metric_amd.options import CalculateOptions, transform_options
    from open_pcc_metric_amd import metric as m
    a, b, na, nb = bench.synth(args.points)
    pair = CloudPair(PointCloud(a, na), PointCloud(b, nb), extent=[1.0, 1.0, 1.0], use_graph=True)
    eng = pair._engine
    options = CalculateOptions(color=None, hausdorff=False, point_to_plane=True)

    def metrics():
        return transform_options(options)[2:] + [m.GeoHausdorffDistance(True, False), m.GeoHausdorffDistance(False, False)]

    for _ in range(5):
        pair.recompute(); MetricCalculator(pair).calculate(metrics()).as_dict()
    import gc; gc.collect(); gc.freeze()
    eng.sync()
    t = [0.0] * 5
    T0 = time.perf_counter()
    for _ in range(args.steps):
        t0 = time.perf_counter()
        pair.recompute()
        t1 = time.perf_counter()
        ms = metrics()
        calc = MetricCalculator(pair)
        t2 = time.perf_counter()
        eng.sync()                       # the GPU's share, seen from the host (launch latency + kernels + wake-up)
        t3 = time.perf_counter()
        res = calc.calculate(ms)
        t4 = time.perf_counter()
        res.as_dict()
        t5 = time.perf_counter()
        for k, (x, y) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))):
            t[k] += y - x
    total = time.perf_counter() - T0
    n = args.steps
    print(f"points {args.points}: step {total / n * 1e6:.1f} us = launch {t[0] / n * 1e6:.1f} + plan {t[1] / n * 1e6:.1f} + "
          f"wait {t[2] / n * 1e6:.1f} + calculate {t[3] / n * 1e6:.1f} + as_dict {t[4] / n * 1e6:.1f}")


if __name__ == "__main__":
    main()
