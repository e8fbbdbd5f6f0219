"""Options -> ordered metric list; same order as ``open_pcc_metric.options.transform_options``.

Reference: open_pcc_metric/options.py:16-174.  The order of the returned list is the row order
of the CLI report, so it is part of the contract:

  always                 MinSqrt, MaxSqrt, GeoMSE L/R/sym, GeoPSNR L/R/sym          (options.py:35-56)
  color                  ColorMSE L/R/sym, ColorPSNR L/R/sym                        (options.py:58-82)
  point_to_plane         GeoMSE L/R/sym, GeoPSNR L/R/sym with point_to_plane=True   (options.py:84-104)
  hausdorff              Hausdorff L/R/sym, HausdorffPSNR L/R/sym (D1)              (options.py:106-138)
  hausdorff & p2plane    Hausdorff L/R, HausdorffPSNR L/R, then both symmetric rows (options.py:140-172)
"""
from __future__ import annotations

import typing

from .metric import (AbstractMetric, ColorMSE, ColorPSNR, GeoHausdorffDistance, GeoHausdorffDistancePSNR,
                     GeoMSE, GeoPSNR, MaxSqrtDistance, MinSqrtDistance, SymmetricMetric)


class CalculateOptions:
    def __init__(self, color: typing.Optional[str] = None, hausdorff: bool = False,
                 point_to_plane: bool = False):
        self.color = color
        self.hausdorff = hausdorff
        self.point_to_plane = point_to_plane


def _sides(cls, **kw):
    return [cls(is_left=True, **kw), cls(is_left=False, **kw)]


def _sym(cls, higher_is_better: bool, **kw) -> SymmetricMetric:
    return SymmetricMetric(metrics=tuple(_sides(cls, **kw)), is_proportional=higher_is_better)


def _error_then_psnr(err_cls, psnr_cls, **kw) -> typing.List[AbstractMetric]:
    return (_sides(err_cls, **kw) + [_sym(err_cls, False, **kw)]
            + _sides(psnr_cls, **kw) + [_sym(psnr_cls, True, **kw)])


def transform_options(options: CalculateOptions) -> typing.List[AbstractMetric]:
    metrics: typing.List[AbstractMetric] = [MinSqrtDistance(), MaxSqrtDistance()]
    metrics += _error_then_psnr(GeoMSE, GeoPSNR, point_to_plane=False)
    if options.color is not None:
        metrics += _error_then_psnr(ColorMSE, ColorPSNR, color_scheme=options.color)
    if options.point_to_plane:
        metrics += _error_then_psnr(GeoMSE, GeoPSNR, point_to_plane=True)
    if options.hausdorff:
        metrics += _error_then_psnr(GeoHausdorffDistance, GeoHausdorffDistancePSNR, point_to_plane=False)
    if options.hausdorff and options.point_to_plane:
        kw = dict(point_to_plane=True)
        metrics += (_sides(GeoHausdorffDistance, **kw) + _sides(GeoHausdorffDistancePSNR, **kw)
                    + [_sym(GeoHausdorffDistance, False, **kw), _sym(GeoHausdorffDistancePSNR, True, **kw)])
    return metrics
