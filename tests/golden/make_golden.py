#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own NumPy code.

Run only in the authoring container (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden.py [case ...]

What runs for real: the reference's ``open_pcc_metric.cloud_pair`` glue (cloud_pair.py:10-124),
every metric node (metric.py), the DAG executor and result formatting (calculator.py) and
``transform_options`` (options.py) -- imported unmodified from /root/reference.

What cannot run: ``open3d`` (pinned 0.18.0, requirements.txt:33) is not installed and cannot be.
The eight Open3D symbols the reference touches are provided by the
in-memory stand-in below, which is OUR code and deliberately independent of oracle/:
its kNN is a dense NumPy distance matrix with nanoflann's accumulation order and a stable
argsort (ties -> smallest index).  ``get_minimal_oriented_bounding_box().extent`` (Qhull-based
Open3D code) is not restated here at all: each case carries the extent as an INPUT.

Outputs are data only (inputs + the reference's outputs); no reference source is stored.
"""
import io
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"


# --------------------------------------------------------------------------- open3d stand-in
def _d2_matrix(q, p):
    d = q[:, None, :] - p[None, :, :]
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


class _Box:
    def __init__(self, extent):
        self.extent = np.asarray(extent, dtype=np.float64)


class PointCloud:
    def __init__(self):
        self.points = np.zeros((0, 3))
        self.normals = np.zeros((0, 3))
        self.colors = np.zeros((0, 3))
        self._extent = None

    def has_normals(self):
        return len(self.normals) > 0

    def has_colors(self):
        return len(self.colors) > 0

    def estimate_normals(self):
        raise RuntimeError("golden cases always carry normals (normal estimation is Open3D code)")

    def compute_nearest_neighbor_distance(self):
        p = np.asarray(self.points, dtype=np.float64)
        if p.shape[0] < 2:
            return np.zeros(p.shape[0])
        d2 = np.sort(_d2_matrix(p, p), axis=1)     # k=2 search of the cloud in itself
        return np.sqrt(d2[:, 1])

    def get_minimal_oriented_bounding_box(self):
        return _Box(self._extent)


class KDTreeFlann:
    def __init__(self, cloud):
        self.p = np.asarray(cloud.points, dtype=np.float64)

    def search_knn_vector_3d(self, q, k):
        q = np.asarray(q, dtype=np.float64).reshape(1, 3)
        d2 = _d2_matrix(q, self.p)[0]
        order = np.argsort(d2, kind="stable")[:k]
        return k, [int(i) for i in order], [float(x) for x in d2[order]]


def _install_standin():
    o3d = types.ModuleType("open3d")
    o3d.geometry = types.SimpleNamespace(PointCloud=PointCloud, KDTreeFlann=KDTreeFlann)
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.array(a, dtype=np.float64))
    sys.modules["open3d"] = o3d


# --------------------------------------------------------------------------- cases
def _unit(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def _uniform(n, seed):
    # float32 values held in float64: what a float PLY gives the reference through Open3D
    return np.random.default_rng(seed).random((n, 3), dtype=np.float32).astype(np.float64)


def _normals(n, seed):
    g = np.random.default_rng(seed).standard_normal((n, 3), dtype=np.float32)
    return _unit(g.astype(np.float64)).astype(np.float32).astype(np.float64)


def cases():
    out = {}
    # (1) the reference's only geometry fixture, tests/unit/test_metric.py:13-26, plus z normals
    a = np.eye(3, dtype="float64")
    b = a + 1e-1 * np.linspace(1.0, 3, 3)
    nz = np.tile(np.array([[0.0, 0.0, 1.0]]), (3, 1))
    out["fixture_eye3"] = dict(a=a, b=b, na=nz, nb=nz, extent=[1.5, 1.0, 0.5])
    # the same fixture with its colours (tests/unit/test_metric.py:16,19,23,25: colours = points)
    out["fixture_eye3_color"] = dict(a=a, b=b, na=nz, nb=nz, extent=[1.5, 1.0, 0.5], ca=a.copy(), cb=b.copy())
    rc = np.random.default_rng(77)
    out["uniform_300_color"] = dict(a=_uniform(300, 41), b=_uniform(300, 42), na=_normals(300, 43), nb=_normals(300, 44),
                                    extent=[1.0, 0.9, 0.8], ca=np.round(rc.random((300, 3)) * 255) / 255.0,
                                    cb=np.round(rc.random((300, 3)) * 255) / 255.0)
    # (2) seeded uniform fp32 clouds of equal size (BASELINE.json configs, scaled down)
    for n in (1, 2, 3, 64, 257, 1000):
        out[f"uniform_{n}"] = dict(a=_uniform(n, 1234 + n), b=_uniform(n, 5678 + n),
                                   na=_normals(n, 4321 + n), nb=_normals(n, 8765 + n),
                                   extent=[1.0, 0.9, 0.8])
    # (3) codec-like: B = A + small noise, fp64 coordinates that are NOT fp32-representable
    a = np.random.default_rng(7).random((500, 3))
    b = a + np.random.default_rng(99).normal(0.0, 1e-3, (500, 3))
    out["noisy_f64_500"] = dict(a=a, b=b, na=_normals(500, 1), nb=_normals(500, 2), extent=[1.0, 1.0, 1.0])
    # (4) unequal sizes: D2 in "row" mode raises IndexError in the reference for the bigger cloud
    out["unequal_300_200"] = dict(a=_uniform(300, 11), b=_uniform(200, 12),
                                  na=_normals(300, 13), nb=_normals(200, 14), extent=[1.0, 1.0, 1.0])
    # (5) voxelised integer lattice with exact ties and duplicated points
    rng = np.random.default_rng(5)
    a = rng.integers(0, 8, (400, 3)).astype(np.float64)
    b = rng.integers(0, 8, (400, 3)).astype(np.float64)
    out["lattice_ties_400"] = dict(a=a, b=b, na=_normals(400, 15), nb=_normals(400, 16), extent=[7.0, 7.0, 7.0])
    # (6) identical clouds: MSE = 0, PSNR = inf (metric.py:247, no guard)
    a = _uniform(100, 21)
    out["identical_100"] = dict(a=a, b=a.copy(), na=_normals(100, 22), nb=_normals(100, 22), extent=[1.0, 1.0, 1.0])
    # (7) large coordinates (10-bit voxel range) with sub-voxel noise: stresses fp32 cancellation
    a = np.floor(np.random.default_rng(31).random((600, 3)) * 1024.0)
    b = (a + np.random.default_rng(32).normal(0, 0.3, (600, 3))).astype(np.float32).astype(np.float64)
    out["voxel10_noise_600"] = dict(a=a, b=b, na=_normals(600, 33), nb=_normals(600, 34), extent=[1023.0, 1023.0, 1023.0])
    return out


def _cloud(pts, nrm, extent, colors=None):
    c = PointCloud()
    c.points = np.array(pts, dtype=np.float64)
    c.normals = np.array(nrm, dtype=np.float64)
    if colors is not None:
        c.colors = np.array(colors, dtype=np.float64)
    c._extent = extent
    return c


def run_case(name, spec):
    import open_pcc_metric.metric as rm
    from open_pcc_metric.cloud_pair import CloudPair
    from open_pcc_metric.calculator import MetricCalculator
    from open_pcc_metric.options import CalculateOptions, transform_options

    rec = {k: np.asarray(v, dtype=np.float64) for k, v in spec.items()}
    pair = CloudPair(_cloud(spec["a"], spec["na"], spec["extent"], spec.get("ca")),
                     _cloud(spec["b"], spec["nb"], spec["extent"], spec.get("cb")))
    rec["left_d2"] = np.asarray(pair.get_left_neighbour_distances())
    rec["right_d2"] = np.asarray(pair.get_right_neighbour_distances())
    rec["left_err"] = np.asarray(pair.get_left_error_vector())
    rec["right_err"] = np.asarray(pair.get_right_error_vector())
    rec["boundary"] = np.asarray(pair.get_boundary_sqrt_distances())

    raises = {}
    results = {}
    texts = {}
    for hd in (False, True):
        for p2p in (False, True):
            MetricCalculator._calculated_metrics.clear()      # class-level memo, calculator.py:60 (quirk Q2)
            opts = CalculateOptions(color=None, hausdorff=hd, point_to_plane=p2p)
            tag = f"h{int(hd)}p{int(p2p)}"
            try:
                res = MetricCalculator(pair).calculate(transform_options(opts))
            except IndexError:
                raises[tag] = "IndexError"
                continue
            d = res.as_dict()
            results[tag] = [[list(k), float(v)] for k, v in d.items()]
            df = res.as_df()
            texts[tag] = {"string": df.to_string(), "csv": df.to_csv()}
    if "ca" in spec:            # colour rows (options.py:58-82), transform_colors / ColorMSE / ColorPSNR
        for scheme in ("rgb", "ycc"):
            MetricCalculator._calculated_metrics.clear()
            opts = CalculateOptions(color=scheme, hausdorff=False, point_to_plane=False)
            res = MetricCalculator(pair).calculate(transform_options(opts))
            rows = []
            for k, v in res.as_dict().items():
                rows.append([list(k), [float(x) for x in np.atleast_1d(v)]])
            results["c" + scheme] = rows
            df = res.as_df()
            texts["c" + scheme] = {"string": df.to_string(), "csv": df.to_csv()}
    if "ca" in spec:            # colour metrics no option reaches (metric.py:389-443) and the "yuv" scheme
        for scheme in ("rgb", "ycc", "yuv"):
            for is_left in (True, False):
                side = "left" if is_left else "right"
                for cls in (rm.ColorMSE, rm.ColorPSNR, rm.ColorHausdorffDistance, rm.ColorHausdorffDistancePSNR):
                    MetricCalculator._calculated_metrics.clear()
                    m = MetricCalculator(pair)._metric_recursive_calculate(cls(is_left=is_left, color_scheme=scheme))
                    rec[f"{cls.__name__}_{side}_{scheme}"] = np.asarray(m.value, dtype=np.float64)
    # per-point D2 vectors, each direction on its own (one may raise, quirk Q1)
    for is_left in (True, False):
        side = "left" if is_left else "right"
        MetricCalculator._calculated_metrics.clear()
        try:
            m = MetricCalculator(pair)._metric_recursive_calculate(rm.ErrorVector(is_left=is_left, point_to_plane=True))
            rec[f"{side}_proj"] = np.asarray(m.value)
        except IndexError:
            raises[f"{side}_proj"] = "IndexError"
    MetricCalculator._calculated_metrics.clear()
    rec["meta"] = np.frombuffer(json.dumps({"results": results, "raises": raises, "texts": texts}).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **rec)
    print(f"{name}: nA={len(spec['a'])} nB={len(spec['b'])} raises={raises}")


def main():
    if not os.path.isdir(REFERENCE):
        sys.exit("needs /root/reference (authoring container only)")
    _install_standin()
    sys.path.insert(0, REFERENCE)
    import logging
    logging.disable(logging.CRITICAL)
    only = set(sys.argv[1:])                     # optional: regenerate just the named cases
    for name, spec in cases().items():
        if not only or name in only:
            run_case(name, spec)


if __name__ == "__main__":
    main()
