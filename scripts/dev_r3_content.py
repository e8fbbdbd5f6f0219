"""round-3 dev: per-step cost of a voxelised-surface pair (configs[4]-class content): 0.8M vs ~0.7M points, no normals fused."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import open_pcc_metric_amd.metric as m
from open_pcc_metric_amd import _native as nat
from open_pcc_metric_amd.calculator import MetricCalculator
from open_pcc_metric_amd.cloud_pair import CloudPair
from open_pcc_metric_amd.options import CalculateOptions, transform_options
from open_pcc_metric_amd.point_cloud import PointCloud
from test_gpu_config4_surrogate import voxel_surface, decode

pts, cols = voxel_surface()
for step_, drop in ((1, 0.1), (2, 0.0)):
    q, qc = decode(pts, cols, step_, drop, 5)
    pair = CloudPair(PointCloud(pts), PointCloud(q), extent=[511.0, 322.0, 505.0])
    eng = pair._engine
    opts = CalculateOptions(color=None, hausdorff=True, point_to_plane=False)
    def step():
        pair.recompute()
        return MetricCalculator(pair).calculate(transform_options(opts)[2:]).as_dict()
    for _ in range(5): r = step()
    eng.sync()
    K = 50
    t = time.perf_counter()
    for _ in range(K): r = step()
    eng.sync()
    ms = (time.perf_counter() - t) / K * 1e3
    pair._use_graph = False
    eng.profile(True); eng.profile_reset()
    for _ in range(10): step()
    eng.sync()
    prof = {k: round(eng.profile_get(k)[0] / 10 * 1e3, 1) for k in nat.KERNEL_CLASSES if eng.profile_get(k)[1]}
    eng.profile(False)
    print(json.dumps({"pair": f"{len(pts)} vs {len(q)} (step {step_})", "ms_per_step": round(ms, 4), "kernel_us": prof,
                      "cells": eng.nn_stats(0)["splits"]}), flush=True)
    pair.close()
