"""NumPy-backed point cloud: the few attributes of ``o3d.geometry.PointCloud`` this path reads.

The reference touches ``.points``, ``.normals``, ``.colors``, ``has_normals()`` and ``has_colors()``
(cloud_pair.py:35-40, 61-64, 114-124; metric.py:95-98).  ``CloudPair`` accepts anything that
duck-types those -- including real Open3D clouds -- and this class when Open3D is absent.
"""
from __future__ import annotations

import numpy as np


def _rows(a, name):
    if a is None:
        return np.zeros((0, 3), dtype=np.float64)
    if hasattr(a, "is_cuda"):          # torch tensor: kept as is (device arrays go straight to the engine)
        if a.dim() != 2 or a.shape[1] != 3:
            raise ValueError(f"{name} must have shape (N, 3)")
        return a
    arr = np.asarray(a)
    if arr.dtype != np.float32:
        arr = arr.astype(np.float64, copy=False)
    if arr.ndim != 2 or arr.shape[1] != 3:
        raise ValueError(f"{name} must have shape (N, 3)")
    return arr


class PointCloud:
    def __init__(self, points=None, normals=None, colors=None):
        self.points = points
        self.normals = normals
        self.colors = colors

    points = property(lambda self: self._points, lambda self, v: setattr(self, "_points", _rows(v, "points")))
    normals = property(lambda self: self._normals, lambda self, v: setattr(self, "_normals", _rows(v, "normals")))
    def _set_colors(self, v):
        self._colors = _rows(v, "colors")
        self.colors_u8 = None          # see attach_colors_u8

    colors = property(lambda self: self._colors, _set_colors)

    def attach_colors_u8(self, u8) -> None:
        """The (N, 3) uint8 array ``colors`` was made from as ``u8 / 255.0`` (file readers call this): lets the GPU path
        ship 3 bytes per point instead of 24 and redo the same division on the device.  Dropped when colors change."""
        u8 = np.ascontiguousarray(u8, dtype=np.uint8)
        if u8.shape != self._colors.shape:
            raise ValueError("colors_u8 must match colors")
        self.colors_u8 = u8
        # the two views must not drift apart: with the bytes attached, colors can only change through the setter
        # (which drops the bytes); an in-place edit raises instead of leaving stale bytes for the GPU
        if isinstance(self._colors, np.ndarray):
            self._colors.setflags(write=False)

    def has_points(self) -> bool:
        return len(self._points) > 0

    def has_normals(self) -> bool:
        return len(self._normals) > 0

    def has_colors(self) -> bool:
        return len(self._colors) > 0

    def __repr__(self) -> str:
        return f"PointCloud with {len(self._points)} points."
