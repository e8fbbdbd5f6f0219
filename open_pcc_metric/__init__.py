"""``open_pcc_metric`` -- the reference's import names over the MI355X engine.

Code written against the reference keeps its imports (handler.py:52-56, tests/unit/test_metric.py:1-8 there):

    import open_pcc_metric.metric as opmm
    from open_pcc_metric.cloud_pair import CloudPair
    from open_pcc_metric.calculator import MetricCalculator
    from open_pcc_metric.options import CalculateOptions, transform_options

Every submodule of the reference package (``cloud_pair``, ``metric``, ``calculator``, ``options``, ``handler``,
``logger``) is the module of the same name in :mod:`open_pcc_metric_amd` -- the same objects, not copies, so
``open_pcc_metric.metric.GeoMSE is open_pcc_metric_amd.metric.GeoMSE``.  There is no CPU implementation behind
either name: without libpccm.so or without an MI355X, constructing a ``CloudPair`` raises.
"""
import importlib
import sys

import open_pcc_metric_amd as _impl

_SUBMODULES = ("cloud_pair", "metric", "calculator", "options", "handler", "logger", "io", "point_cloud")

for _name in _SUBMODULES:
    _mod = importlib.import_module(f"{_impl.__name__}.{_name}")
    sys.modules[f"{__name__}.{_name}"] = _mod
    setattr(sys.modules[__name__], _name, _mod)

del importlib, sys, _name, _mod
