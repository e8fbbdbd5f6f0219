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
    def __init__(self, metrics: typing.List[AbstractMetric]):
        self._metrics = metrics

    def as_dict(self) -> typing.Dict[typing.Tuple, typing.Any]:
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

    def _plan(self, metrics_list: typing.List[AbstractMetric]) -> None:
        """Walk the dependency DAG of the request (no evaluation) and let the CloudPair enqueue every
        GPU reduction it contains; evaluation below then finds the results already on their way."""
        prefetch = getattr(self._cloud_pair, "prefetch_reductions", None)
        if prefetch is None:
            return
        wanted, seen, stack = [], set(), list(metrics_list)
        while stack:
            m = stack.pop()
            key = m._key()
            if key in seen or key in self._calculated_metrics:
                continue
            seen.add(key)
            if isinstance(m, EuclideanDistance):
                wanted.append((m.is_left, m.point_to_plane))
            elif isinstance(m, BoundarySqrtDistances):
                wanted.append("boundary")
            if getattr(m, "_pccm_role", 0) == 2 or isinstance(m, SecondaryMetric):
                stack.extend(m._get_dependencies().values())
        if wanted:
            prefetch(sorted(wanted, key=str))

    def calculate(self, metrics_list: typing.List[AbstractMetric]) -> CalculateResult:
        self._plan(metrics_list)
        return CalculateResult([self._metric_recursive_calculate(m) for m in metrics_list])
