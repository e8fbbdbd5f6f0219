"""The reference's import lines work unchanged: ``open_pcc_metric`` is an alias package whose
submodules are the modules of ``open_pcc_metric_amd``.  The two known-answer tests of the reference's own suite
(/root/reference/tests/unit/test_metric.py:30-70, restated here as values, not copied) run through the alias."""
import importlib
import subprocess
import sys

import numpy as np
import pytest


def test_reference_import_lines_resolve():
    # /root/reference/tests/unit/test_metric.py:4-8 and open_pcc_metric/handler.py:53-55
    import open_pcc_metric.metric as opmm
    from open_pcc_metric.calculator import MetricCalculator
    from open_pcc_metric.cloud_pair import CloudPair
    from open_pcc_metric.options import CalculateOptions, transform_options
    import open_pcc_metric_amd.calculator as c
    import open_pcc_metric_amd.cloud_pair as cp
    import open_pcc_metric_amd.metric as m
    import open_pcc_metric_amd.options as o
    assert opmm is m and opmm.CloudPair is cp.CloudPair and CloudPair is cp.CloudPair
    assert MetricCalculator is c.MetricCalculator
    assert CalculateOptions is o.CalculateOptions and transform_options is o.transform_options
    for name in ("cloud_pair", "metric", "calculator", "options", "handler", "logger"):
        assert importlib.import_module(f"open_pcc_metric.{name}") is importlib.import_module(f"open_pcc_metric_amd.{name}")


@pytest.mark.parametrize("is_left", [True, False])
def test_reference_kat_error_vector(is_left):
    """ErrorVector D1 on ones((5, 3)) -> sqrt(3) per row (reference test_default_error_vector)."""
    import open_pcc_metric.metric as opmm
    ev = opmm.ErrorVector(is_left=is_left, point_to_plane=False)
    prim = opmm.PrimaryErrorVector(is_left=is_left)
    prim.value = np.ones((5, 3), dtype="float64")
    ev.calculate(prim)
    assert np.allclose(ev.value, np.sqrt(3) * np.ones(5))


@pytest.mark.parametrize("is_left,point_to_plane", [(True, False), (False, False), (True, True), (False, True)])
def test_reference_kat_euclidean_distance(is_left, point_to_plane):
    """EuclideanDistance: D1 passes the neighbour distances through, D2 squares the projection (2 -> 4)
    (reference test_default_euclidean_distance)."""
    import open_pcc_metric.metric as opmm
    ed = opmm.EuclideanDistance(is_left=is_left, point_to_plane=point_to_plane)
    prim = opmm.PrimaryErrorVector(is_left=is_left)
    prim.value = 2 * np.ones(5)
    nd = opmm.NeighbourDistances(is_left=is_left)
    nd.value = 4 * np.ones(5)
    ed.calculate(nd, prim)
    assert np.allclose(nd.value, ed.value)


def test_module_entry_point_shows_the_reference_flags():
    out = subprocess.run([sys.executable, "-m", "open_pcc_metric", "--help"], capture_output=True, text=True, check=True).stdout
    for flag in ("--ocloud", "--pcloud", "--color", "--hausdorff", "--point-to-plane", "--csv"):
        assert flag in out
