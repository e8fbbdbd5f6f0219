"""DAG executor and result table -- drop-in for ``open_pcc_metric.calculator``.

Reference: open_pcc_metric/calculator.py:15-108.  Same depth-first resolution (primaries get the
``CloudPair``, secondaries get their resolved dependencies as keyword arguments named by the
keys of ``_get_dependencies()``), same ``as_dict`` keys and ``as_df`` columns.  One deliberate
difference: the memo is per ``MetricCalculator`` instance; the reference keeps it on the class
(calculator.py:60), so a second pair evaluated in the same process silently returns the first
pair's numbers.  For the first pair of a process both behave identically.
"""
from __future__ import annotations

import typing

import pandas as pd

from .cloud_pair import CloudPair
from .metric import (AbstractMetric, BoundarySqrtDistances, EuclideanDistance, PrimaryMetric, SecondaryMetric,
                     SymmetricMetric)

_COLUMNS = ("label", "is_left", "point-to-plane", "value")


class CalculateResult:
    def __init__(self, metrics: typing.List[AbstractMetric], keys: typing.Optional[typing.List[typing.Tuple]] = None):
        self._metrics = metrics
        self._keys = keys          # the keys the calculator already derived while planning (same as m._key())

    def as_dict(self) -> typing.Dict[typing.Tuple, typing.Any]:
        if self._keys is not None:
            return {k: m.value for k, m in zip(self._keys, self._metrics)}
        return {m._key(): m.value for m in self._metrics}

    def as_df(self) -> pd.DataFrame:
        rows: typing.Dict[str, list] = {c: [] for c in _COLUMNS}
        for m in self._metrics:
            if isinstance(m, SymmetricMetric):
                label = type(m.metrics[0]).__name__ + "(symmetric)"
            else:
                label = type(m).__name__
            rows["label"].append(label)
            rows["is_left"].append(getattr(m, "is_left", ""))
            rows["point-to-plane"].append(getattr(m, "point_to_plane", ""))
            rows["value"].append(str(m.value))
        return pd.DataFrame(rows)

    def __str__(self) -> str:
        return str(self.as_df())


class MetricCalculator:
    def __init__(self, cloud_pair: CloudPair):
        self._cloud_pair = cloud_pair
        self._calculated_metrics: typing.Dict[typing.Tuple, AbstractMetric] = {}

    def _metric_recursive_calculate(self, metric: AbstractMetric) -> AbstractMetric:
        key = metric._key()
        done = self._calculated_metrics.get(key)
        if done is not None:
            return done
        role = getattr(metric, "_pccm_role", 0) or (1 if isinstance(metric, PrimaryMetric) else
                                                    2 if isinstance(metric, SecondaryMetric) else 0)
        if role == 1:
            metric.calculate(self._cloud_pair)
        elif role == 2:
            resolved = {name: self._metric_recursive_calculate(dep)
                        for name, dep in metric._get_dependencies().items()}
            metric.calculate(**resolved)
        else:
            raise RuntimeError(f"Metric of unknown AbstractMetric subclass {type(metric).__name__}")
        self._calculated_metrics[key] = metric
        return metric

    def _visit(self, metric, early, late, planned, wanted):
        """Post-order walk below ``metric`` (a method, not a closure: a recursive closure is a reference cycle that
        would keep the calculator -- and with it the CloudPair and its GPU context -- alive until the cyclic GC runs).
        Returns (the metric standing for this key, its key, whether it has to wait for the GPU)."""
        key = metric._key()
        known = self._calculated_metrics.get(key)
        if known is not None:
            return known, key, False
        known = planned.get(key)
        if known is not None:
            return known
        role = getattr(metric, "_pccm_role", 0) or (1 if isinstance(metric, PrimaryMetric) else
                                                    2 if isinstance(metric, SecondaryMetric) else 0)
        waits = bool(getattr(metric, "_pccm_waits", False))
        if role == 1:
            if isinstance(metric, BoundarySqrtDistances):
                wanted.append("boundary")
            (late if waits else early).append((metric, None, key))
        elif role == 2:
            if isinstance(metric, EuclideanDistance):
                wanted.append((metric.is_left, metric.point_to_plane))
            resolved = {}
            for name, dep in metric._get_dependencies().items():
                resolved[name], _, dep_waits = self._visit(dep, early, late, planned, wanted)
                waits = waits or dep_waits
            (late if waits else early).append((metric, resolved, key))
        else:
            raise RuntimeError(f"Metric of unknown AbstractMetric subclass {type(metric).__name__}")
        planned[key] = (metric, key, waits)
        return planned[key]

    def _plan(self, metrics_list: typing.List[AbstractMetric]):
        """Resolve the dependency DAG of the request WITHOUT evaluating anything.

        Returns the evaluation program: ``(metric, resolved dependencies or None, key)`` in exactly the order
        and with exactly the memoisation of ``_metric_recursive_calculate`` (post-order, first object of a
        key wins) -- split into the nodes that only pass device columns along and those that read a reduction
        back from the GPU or hang off one that does --, plus the metric standing for every requested one.
        Meanwhile the CloudPair is told which GPU reductions the request contains (``prefetch_reductions``), so
        they are enqueued -- or, with ``use_graph``, already running -- while this Python bookkeeping happens; the
        blocking part of a report is then only the late ``calculate()`` calls themselves."""
        early, late, planned, wanted = [], [], {}, []
        requested = [self._visit(m, early, late, planned, wanted) for m in metrics_list]
        prefetch = getattr(self._cloud_pair, "prefetch_reductions", None)
        if prefetch is not None and wanted:
            prefetch(sorted(set(wanted), key=str))
        return early, late, [r[0] for r in requested], [r[1] for r in requested]

    def calculate(self, metrics_list: typing.List[AbstractMetric]) -> CalculateResult:
        early, late, requested, keys = self._plan(metrics_list)
        pair, done = self._cloud_pair, self._calculated_metrics
        # Two passes over the program (calculator.py:85-95, unrolled): first every node that only passes device
        # columns along -- none of them waits for the GPU -- then, in the original order, the reducers and
        # whatever hangs off them.  The first reducer is where the host blocks; nothing is left to do in Python
        # after it that could have been done before.
        for program in (early, late):
            for metric, resolved, key in program:
                if resolved is None:
                    metric.calculate(pair)
                else:
                    metric.calculate(**resolved)
                done[key] = metric
        return CalculateResult(requested, keys)
