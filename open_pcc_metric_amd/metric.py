"""Metric nodes -- same classes, keys and ``calculate`` signatures as ``open_pcc_metric.metric``.

Reference: open_pcc_metric/metric.py:14-485.  The classes are thin: every N-sized quantity they
pass around is a device column of :mod:`open_pcc_metric_amd.cloud_pair`, so ``np.sum`` in
``GeoMSE``, ``np.max`` in ``GeoHausdorffDistance``, ``np.square`` in ``EuclideanDistance`` and
``np.min``/``np.max`` in ``BoundarySqrtDistances`` run as fused GPU reductions while the code
below reads like the reference's NumPy.  Values injected by hand (as the reference's unit tests
do, tests/unit/test_metric.py:30-70) are plain ndarrays and take the same code path through
NumPy itself.

``_key()`` tuples are both the memo keys of the calculator and the keys of
``CalculateResult.as_dict()`` (metric.py:17-18, 59-60, 70-71, 257-258, 472-473).
"""
from __future__ import annotations

import abc
import math
import typing

import numpy as np

from .cloud_pair import CloudPair, CloudNormalsView, DeviceRows


# --------------------------------------------------------------------------- base classes
class AbstractMetric(abc.ABC):                       # metric.py:14-29
    value: typing.Any

    def _key(self) -> typing.Tuple:
        return (type(self).__name__,)

    @abc.abstractmethod
    def calculate(self, cloud_pair: CloudPair, **kwargs) -> None:
        raise NotImplementedError("calculate is not implemented")

    def __str__(self) -> str:
        return f"{self._key()}: {self.value}"


class PrimaryMetric(AbstractMetric):                 # metric.py:32-38 -- reads the CloudPair
    _pccm_role = 1          # what the calculator dispatches on (an isinstance on an ABC costs ~1 us)

    @abc.abstractmethod
    def calculate(self, cloud_pair: CloudPair) -> None:
        raise NotImplementedError("calculate is not implemented")


class SecondaryMetric(AbstractMetric):               # metric.py:41-50 -- pure function of its deps
    _pccm_role = 2

    def _get_dependencies(self) -> typing.Dict[str, "AbstractMetric"]:
        return {}

    @abc.abstractmethod
    def calculate(self, **kwargs) -> None:
        raise NotImplementedError("calculate is not implemented")


class DirectionalMetric(AbstractMetric):             # metric.py:53-60
    is_left: bool

    def __init__(self, is_left: bool):
        self.is_left = is_left

    def _key(self) -> typing.Tuple:
        return (type(self).__name__, self.is_left)


class PointToPlaneable(DirectionalMetric):           # metric.py:63-71
    point_to_plane: bool

    def __init__(self, is_left: bool, point_to_plane: bool):
        super().__init__(is_left)
        self.point_to_plane = point_to_plane

    def _key(self) -> typing.Tuple:
        return (type(self).__name__, self.is_left, self.point_to_plane)


def _side(metric: DirectionalMetric, left, right):
    return left() if metric.is_left else right()


# --------------------------------------------------------------------------- primaries
class PrimaryErrorVector(PrimaryMetric, DirectionalMetric):      # metric.py:74-80
    def calculate(self, cloud_pair: CloudPair) -> None:
        self.value = _side(self, cloud_pair.get_left_error_vector, cloud_pair.get_right_error_vector)


class NeighbourDistances(PrimaryMetric, DirectionalMetric):      # metric.py:83-89
    def calculate(self, cloud_pair: CloudPair) -> None:
        self.value = _side(self, cloud_pair.get_left_neighbour_distances,
                           cloud_pair.get_right_neighbour_distances)


class CloudNormals(PrimaryMetric, DirectionalMetric):            # metric.py:92-98
    def calculate(self, cloud_pair: CloudPair) -> None:
        which = 0 if self.is_left else 1
        if hasattr(cloud_pair, "get_normals"):
            self.value = cloud_pair.get_normals(which)
        else:
            self.value = np.asarray(cloud_pair.clouds[which].normals)


class CloudExtent(PrimaryMetric):                                # metric.py:101-103
    def calculate(self, cloud_pair: CloudPair) -> None:
        self.value = cloud_pair.get_extent()


class CloudColors(PrimaryMetric, DirectionalMetric):             # metric.py:106-112
    def calculate(self, cloud_pair: CloudPair) -> None:
        self.value = _side(self, cloud_pair.get_left_colors, cloud_pair.get_right_colors)


class NeighbourColors(PrimaryMetric, DirectionalMetric):         # metric.py:115-121
    def calculate(self, cloud_pair: CloudPair) -> None:
        self.value = _side(self, cloud_pair.get_left_neighbour_colors,
                           cloud_pair.get_right_neighbour_colors)


class BoundarySqrtDistances(PrimaryMetric):                      # metric.py:182-188
    _pccm_waits = True      # reads a reduction back from the GPU: the calculator evaluates these last
    def calculate(self, cloud_pair: CloudPair) -> None:
        spacing = cloud_pair.get_boundary_sqrt_distances()
        self.value = (np.min(spacing), np.max(spacing))


# --------------------------------------------------------------------------- geometry secondaries
class ErrorVector(SecondaryMetric, PointToPlaneable):            # metric.py:124-153
    def _get_dependencies(self) -> typing.Dict[str, AbstractMetric]:
        deps: typing.Dict[str, AbstractMetric] = {"primary_error_vector": PrimaryErrorVector(is_left=self.is_left)}
        if self.point_to_plane:
            # the OTHER cloud's normals, indexed by the iterating row (reference quirk Q1)
            deps["cloud_normals"] = CloudNormals(is_left=not self.is_left)
        return deps

    def calculate(self, primary_error_vector: PrimaryErrorVector,
                  cloud_normals: typing.Optional[CloudNormals] = None) -> None:
        err = primary_error_vector.value
        if not self.point_to_plane:
            err = np.asarray(err)
            # row-wise Euclidean norm (metric.py:138-144); not consumed by any shipped metric
            self.value = np.sqrt(np.einsum("ij,ij->i", err, err))
            return
        nrm = cloud_normals.value
        origin = getattr(nrm, "_pccm_origin", None)
        if isinstance(err, DeviceRows) and isinstance(nrm, CloudNormalsView) and origin == (
                id(err._pair), 1 if self.is_left else 0):
            # both operands live in HBM of the same pair: fused gather + projection on the GPU
            self.value = err._pair.point_to_plane_column(self.is_left)
            return
        # hand-injected arrays: the reference's own row loop, metric.py:146-153
        err, nrm = np.asarray(err), np.asarray(nrm)
        out = np.zeros(shape=(err.shape[0],))
        for i in range(err.shape[0]):
            out[i] = np.dot(err[i], nrm[i])
        self.value = out


class EuclideanDistance(SecondaryMetric, PointToPlaneable):      # metric.py:156-179
    def _get_dependencies(self) -> typing.Dict[str, AbstractMetric]:
        if self.point_to_plane:
            return {"error_vector": ErrorVector(is_left=self.is_left, point_to_plane=True)}
        return {"neighbour_distances": NeighbourDistances(is_left=self.is_left)}

    def calculate(self, neighbour_distances: typing.Optional[NeighbourDistances] = None,
                  error_vector: typing.Optional[ErrorVector] = None) -> None:
        if self.point_to_plane:
            self.value = np.square(error_vector.value)      # stays a device column when it is one
        else:
            self.value = neighbour_distances.value          # the search's own d2, verbatim


class _BoundaryPick(SecondaryMetric):
    _slot = 0

    def _get_dependencies(self) -> typing.Dict[str, AbstractMetric]:
        return {"boundary_metric": BoundarySqrtDistances()}

    def calculate(self, boundary_metric: BoundarySqrtDistances) -> None:
        self.value = boundary_metric.value[self._slot]


class MinSqrtDistance(_BoundaryPick):                            # metric.py:191-199
    _slot = 0


class MaxSqrtDistance(_BoundaryPick):                            # metric.py:202-210
    _slot = 1


class _OverEuclidean(SecondaryMetric, PointToPlaneable):
    def _get_dependencies(self) -> typing.Dict[str, AbstractMetric]:
        return {"euclidean_distance": EuclideanDistance(is_left=self.is_left, point_to_plane=self.point_to_plane)}


class GeoMSE(_OverEuclidean):                                    # metric.py:213-228
    _pccm_waits = True      # reads a reduction back from the GPU: the calculator evaluates these last
    def calculate(self, euclidean_distance: EuclideanDistance) -> None:
        column = euclidean_distance.value
        fused = getattr(column, "_reduced", None)             # a device column: what np.sum would dispatch to, called directly
        total = fused()[0] if fused is not None else None
        self.value = (np.sum(column, axis=0) if total is None else total) / column.shape[0]


class GeoHausdorffDistance(_OverEuclidean):                      # metric.py:353-366 (a SQUARED distance)
    _pccm_waits = True      # reads a reduction back from the GPU: the calculator evaluates these last
    def calculate(self, euclidean_distance: EuclideanDistance) -> None:
        column = euclidean_distance.value
        fused = getattr(column, "_reduced", None)             # a device column: what np.max would dispatch to, called directly
        self.value = fused()[2] if fused is not None else np.max(column, axis=0)


def _peak_of(cloud_extent):
    """np.max(cloud_extent.value), evaluated once per extent array (a report asks for it four times)."""
    extent = cloud_extent.value
    memo = cloud_extent.__dict__.get("_peak_memo")
    if memo is None or memo[0] is not extent:
        memo = cloud_extent._peak_memo = (extent, np.max(extent))
    return memo[1]


def _psnr(peak, distortion):
    return 10 * np.log10(peak ** 2 / distortion)                 # metric.py:247, 350, 384-386, 443


class GeoPSNR(SecondaryMetric, PointToPlaneable):                # metric.py:231-247
    def _get_dependencies(self) -> typing.Dict[str, AbstractMetric]:
        return {"cloud_extent": CloudExtent(),
                "geo_mse": GeoMSE(is_left=self.is_left, point_to_plane=self.point_to_plane)}

    def calculate(self, cloud_extent: CloudExtent, geo_mse: GeoMSE) -> None:
        self.value = _psnr(_peak_of(cloud_extent), geo_mse.value)


class GeoHausdorffDistancePSNR(SecondaryMetric, PointToPlaneable):   # metric.py:369-386
    def _get_dependencies(self) -> typing.Dict[str, AbstractMetric]:
        return {"max_sqrt": MaxSqrtDistance(),
                "hausdorff_distance": GeoHausdorffDistance(is_left=self.is_left,
                                                           point_to_plane=self.point_to_plane)}

    def calculate(self, max_sqrt: MaxSqrtDistance, hausdorff_distance: GeoHausdorffDistance) -> None:
        self.value = _psnr(max_sqrt.value, hausdorff_distance.value)


# --------------------------------------------------------------------------- colour secondaries
class ColorMetric(DirectionalMetric):                            # metric.py:250-258
    color_scheme: str

    def __init__(self, is_left: bool, color_scheme: str):
        super().__init__(is_left)
        self.color_scheme = color_scheme

    def _key(self) -> typing.Tuple:
        return (type(self).__name__, self.is_left, self.color_scheme)


_FROM_RGB = {                                                    # metric.py:270-281
    "ycc": np.array([[0.2126, 0.7152, 0.0722], [-0.1146, -0.3854, 0.5], [0.5, -0.4542, -0.0458]]),
    "yuv": np.array([[0.25, 0.5, 0.25], [1, 0, -1], [-0.5, 1, -0.5]]),
}
_PEAKS = {"rgb": 255.0, "ycc": 1.0, "yuv": 1.0}                  # metric.py:293-299


def transform_colors(colors: np.ndarray, source_scheme: str, target_scheme: str) -> np.ndarray:
    """metric.py:261-290: rows are mapped by the 3x3 matrix of the target scheme."""
    if source_scheme == target_scheme:
        return colors
    if source_scheme != "rgb" or target_scheme not in _FROM_RGB:
        raise TypeError(f"no transform from {source_scheme!r} to {target_scheme!r}")   # the reference fails here too
    colors = np.asarray(colors)
    if not len(colors):
        return colors
    from . import _native
    return _native.color_transform(colors, target_scheme)      # == np.matmul(matrix, row) for every row, in C


def get_color_peak(color_scheme: str) -> np.float64:
    return _PEAKS[color_scheme]


class _ColorPairMetric(SecondaryMetric, ColorMetric):
    def _get_dependencies(self) -> typing.Dict[str, AbstractMetric]:
        return {"origin_cloud_colors": CloudColors(is_left=self.is_left),
                "neighbour_cloud_colors": NeighbourColors(is_left=self.is_left)}

    def _difference(self, origin_cloud_colors, neighbour_cloud_colors) -> np.ndarray:
        own_rows, other_rows = origin_cloud_colors.value, neighbour_cloud_colors.value
        pair_of = getattr(other_rows, "_pair", None)
        if (pair_of is not None and hasattr(other_rows, "in_scheme")
                and getattr(own_rows, "_pccm_origin", None) == (id(pair_of), 0 if self.is_left else 1)):
            if self.color_scheme not in _PEAKS:
                raise TypeError(f"no transform from 'rgb' to {self.color_scheme!r}")
            return other_rows.in_scheme(self.color_scheme)      # gather, transform and subtract on the GPU
        own = transform_colors(np.copy(origin_cloud_colors.value), "rgb", self.color_scheme)
        other = transform_colors(np.copy(neighbour_cloud_colors.value), "rgb", self.color_scheme)
        return np.subtract(own, other)


class ColorMSE(_ColorPairMetric):                                # metric.py:302-333
    _pccm_waits = True      # reads a reduction back from the GPU: the calculator evaluates these last
    def calculate(self, origin_cloud_colors: CloudColors, neighbour_cloud_colors: NeighbourColors) -> None:
        diff = self._difference(origin_cloud_colors, neighbour_cloud_colors)
        self.value = np.mean(diff ** 2, axis=0)


class ColorHausdorffDistance(_ColorPairMetric):                  # metric.py:389-427
    _pccm_waits = True      # reads a reduction back from the GPU: the calculator evaluates these last
    def calculate(self, origin_cloud_colors: CloudColors, neighbour_cloud_colors: NeighbourColors) -> None:
        diff = self._difference(origin_cloud_colors, neighbour_cloud_colors)
        if self.color_scheme == "rgb":
            diff = 255 * diff                                    # metric.py:422-425
        self.value = np.max(diff ** 2, axis=0)


class ColorPSNR(SecondaryMetric, ColorMetric):                   # metric.py:336-350
    def _get_dependencies(self) -> typing.Dict[str, AbstractMetric]:
        return {"color_mse": ColorMSE(is_left=self.is_left, color_scheme=self.color_scheme)}

    def calculate(self, color_mse: ColorMSE) -> None:
        self.value = _psnr(get_color_peak(self.color_scheme), color_mse.value)


class ColorHausdorffDistancePSNR(SecondaryMetric, ColorMetric):  # metric.py:430-443
    def _get_dependencies(self) -> typing.Dict[str, AbstractMetric]:
        return {"hausdorff_distance": ColorHausdorffDistance(is_left=self.is_left,
                                                             color_scheme=self.color_scheme)}

    def calculate(self, hausdorff_distance: ColorHausdorffDistance) -> None:
        self.value = _psnr(get_color_peak(self.color_scheme), hausdorff_distance.value)


# --------------------------------------------------------------------------- symmetric
class SymmetricMetric(SecondaryMetric):                          # metric.py:446-485
    is_proportional: bool
    metrics: typing.Sequence[DirectionalMetric]

    def __init__(self, metrics: typing.Sequence[DirectionalMetric], is_proportional: bool):
        if len(metrics) != 2:
            raise ValueError("Must be exactly two metrics")
        if type(metrics[0]) is not type(metrics[1]):
            raise ValueError(f"Metrics must be of same class, got: {type(metrics[0])}, {type(metrics[1])}")
        self.metrics = metrics
        self.is_proportional = is_proportional

    def _get_dependencies(self) -> typing.Dict[str, AbstractMetric]:
        return {"lmetric": self.metrics[0], "rmetric": self.metrics[1]}

    def _key(self) -> typing.Tuple:
        return (type(self).__name__,) + self.metrics[0]._key() + self.metrics[1]._key()

    def calculate(self, lmetric: AbstractMetric, rmetric: AbstractMetric) -> None:
        # quality-like metrics (PSNR) report the worse = smaller side, error-like the larger;
        # the left value wins ties in both cases (Python's min/max keep the first extreme).
        # (written out: min / max keep the first extreme, i.e. the right value replaces the left one only when it is strictly
        # smaller / larger -- with NaN keys nothing is, exactly as min([l, r], key=...) behaves)
        left, right = lmetric.value, rmetric.value
        kl, kr = _norm(left), _norm(right)
        if self.is_proportional:
            self.value = right if kr < kl else left
        else:
            self.value = right if kr > kl else left


def _norm(value):
    """np.linalg.norm(value), the reference's comparison key (metric.py:481-485).  For the scalar metrics that is
    sqrt(x . x) = sqrt(x * x) -- evaluated here without NumPy's dispatch, same bits; arrays go to NumPy."""
    if type(value) in (float, np.float64):
        v = float(value)
        return math.sqrt(v * v) if v == v else v
    return np.linalg.norm(value)
