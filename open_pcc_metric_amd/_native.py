"""ctypes binding of libpccm.so (include/pccm.h) -- the only door to the GPU.

There is no CPU implementation behind this module: if the shared library is missing, or no
MI355X is visible when an :class:`Engine` is created, it raises.  Build the library with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C open_pcc_metric_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
import sys
import threading
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PCCM_LIB: another build of the same library (tests load a diagnostic build -- make DIAG=1 BUILD=<dir> -- through it)
LIB_PATH = os.environ.get("PCCM_LIB") or os.path.join(_HERE, "csrc", "libpccm.so")

OK, E_ARG, E_NODEV, E_HIP, E_OOM, E_STATE, E_RANGE = 0, -1, -2, -3, -4, -5, -6
F32, F64 = 0, 1
DIR_LEFT, DIR_RIGHT, DIR_SELF = 0, 1, 2
ENGINES = {"auto": 0, "brute": 1, "grid": 2}
NORMAL_MODES = {"row": 0, "neighbour": 1}
METRIC_D1, METRIC_D2, METRIC_PROJ = 0, 1, 2
KERNEL_CLASSES = {"ingest": 0, "scan": 1, "refine": 2, "fallback": 3, "point": 4, "reduce": 5,
                  "grid_build": 6, "grid_query": 7, "grid_finish": 8}

# every symbol include/pccm.h declares (tests check that the library exports all of them)
SYMBOLS = (
    "pccm_version", "pccm_last_error", "pccm_device_count", "pccm_ctx_create", "pccm_ctx_destroy", "pccm_ctx_reset",
    "pccm_set_cloud", "pccm_set_normals", "pccm_set_normals_deferred", "pccm_flush_uploads", "pccm_set_io_staged", "pccm_estimate_normals", "pccm_get_normals", "pccm_set_shard", "pccm_set_shard_dir", "pccm_shard_range", "pccm_nn", "pccm_nn_pair", "pccm_nn_fuse", "pccm_nn_want_idx", "pccm_nn_fetch",
    "pccm_error_vectors", "pccm_point_metric", "pccm_tie_exposure", "pccm_xvec_len", "pccm_reduce_prefetch", "pccm_reduce_prefetch_many", "pccm_reduce", "pccm_finish_sum",
    "pccm_reduce_total", "pccm_reduce_total_many", "pccm_cvec_len", "pccm_reduce_chunks_many", "pccm_finish_chunks",
    "pccm_set_colors", "pccm_set_colors_u8", "pccm_color_reduce", "pccm_color_rows", "pccm_seq_colsum", "pccm_obb_frames", "pccm_extreme_rows", "pccm_rows_outside",
    "pccm_color_transform", "pccm_lzf_decompress", "pccm_drop_caches", "pccm_graph_begin", "pccm_graph_end", "pccm_graph_launch", "pccm_graph_destroy",
    "pccm_sync",
    "pccm_profile_enable", "pccm_profile_reset", "pccm_profile_get", "pccm_nn_stats",
)

COLOR_SCHEMES = {"rgb": 0, "ycc": 1, "yuv": 2}
COLOR_OWN, COLOR_NEIGHBOUR, COLOR_DIFF, COLOR_SQUARE = 0, 1, 2, 3

_lib = None


class NativeLibraryMissing(ImportError):
    pass


class PccmStateError(RuntimeError):
    """PCCM_E_STATE: a call came in the wrong order or hit a stale object (a hipGraph captured before buffers
    changed, a reduction before its search).  The context is intact; callers may recover (CloudPair re-captures)."""


class PccmDeviceError(RuntimeError):
    """PCCM_E_HIP / PCCM_E_NODEV: the HIP runtime reported a failure.  Never retried, never swallowed."""


def load() -> ctypes.CDLL:
    """dlopen libpccm.so and declare the prototypes of include/pccm.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            f"{LIB_PATH} is not built: open_pcc_metric_amd has no CPU fallback. "
            "Run `make -C open_pcc_metric_amd/csrc` (needs hipcc, --offload-arch=gfx950).")
    # PyTorch ships its own libamdhip64.so.7; loading it first makes libpccm.so bind to that same
    # runtime (one HIP runtime per process), so torch device tensors and streams can be handed in.
    # PCCM_NO_TORCH=1 (the command line sets it for single-process runs): skip the ~1.5 s import; nothing in this
    # process may import torch afterwards and expect to see the GPU.
    if "torch" in sys.modules or os.environ.get("PCCM_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    dp = ctypes.POINTER(ctypes.c_double)
    lib.pccm_version.restype = i32
    lib.pccm_last_error.restype = ctypes.c_char_p
    lib.pccm_device_count.argtypes = [ctypes.POINTER(i32)]
    lib.pccm_ctx_create.argtypes = [i32, vp, ctypes.POINTER(vp)]
    lib.pccm_ctx_destroy.argtypes = [vp]
    lib.pccm_ctx_reset.argtypes = [vp]
    lib.pccm_set_cloud.argtypes = [vp, i32, vp, i64, i32, i32]
    lib.pccm_set_normals.argtypes = [vp, i32, vp, i64, i32, i32]
    lib.pccm_set_normals_deferred.argtypes = [vp, i32, vp, i64, i32]
    lib.pccm_flush_uploads.argtypes = [vp]
    lib.pccm_set_io_staged.argtypes = [vp, i32]
    lib.pccm_set_shard.argtypes = [vp, i32, i32]
    lib.pccm_set_shard_dir.argtypes = [vp, i32, i32, i32]
    lib.pccm_estimate_normals.argtypes = [vp, i32, i32]
    lib.pccm_get_normals.argtypes = [vp, i32, vp]
    lib.pccm_shard_range.argtypes = [vp, i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    lib.pccm_nn.argtypes = [vp, i32, i32]
    lib.pccm_nn_pair.argtypes = [vp, i32]
    lib.pccm_nn_fuse.argtypes = [vp, i32, i32]
    lib.pccm_nn_want_idx.argtypes = [vp, i32]
    lib.pccm_nn_fetch.argtypes = [vp, i32, vp, vp]
    lib.pccm_error_vectors.argtypes = [vp, i32, vp]
    lib.pccm_point_metric.argtypes = [vp, i32, i32, i32, vp]
    lib.pccm_tie_exposure.argtypes = [vp, i32, i32, dp]
    lib.pccm_xvec_len.argtypes = [i64]
    lib.pccm_xvec_len.restype = i64
    lib.pccm_reduce.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.pccm_reduce_prefetch.argtypes = [vp, i32, i32, i32]
    ip = ctypes.POINTER(i32)
    lib.pccm_reduce_prefetch_many.argtypes = [vp, i32, ip, ip, ip]
    lib.pccm_finish_sum.argtypes = [vp, i64, dp]
    lib.pccm_reduce_total.argtypes = [vp, i32, i32, i32, dp]
    lib.pccm_reduce_total_many.argtypes = [vp, i32, ip, ip, ip, dp]
    lib.pccm_cvec_len.argtypes = [i64]
    lib.pccm_cvec_len.restype = i64
    lib.pccm_reduce_chunks_many.argtypes = [vp, i32, ip, ip, ip, dp, dp]
    lib.pccm_finish_chunks.argtypes = [vp, i64, dp]
    lib.pccm_sync.argtypes = [vp]
    lib.pccm_drop_caches.argtypes = [vp]
    lib.pccm_color_transform.argtypes = [vp, i64, i32, vp]
    lib.pccm_lzf_decompress.argtypes = [vp, i64, vp, i64, ctypes.POINTER(i64)]
    lib.pccm_set_colors.argtypes = [vp, i32, vp, i64, i32, i32]
    lib.pccm_set_colors_u8.argtypes = [vp, i32, vp, i64]
    lib.pccm_color_reduce.argtypes = [vp, i32, i32, ctypes.c_double, vp, i64, dp, dp]
    lib.pccm_seq_colsum.argtypes = [vp, vp, i64, dp]
    lib.pccm_obb_frames.argtypes = [vp, vp, i64, vp, i64, dp, dp]
    lib.pccm_extreme_rows.argtypes = [vp, i32, vp, i32, vp]
    lib.pccm_rows_outside.argtypes = [vp, i32, vp, i32, ctypes.c_double, vp, ctypes.POINTER(i64)]
    lib.pccm_color_rows.argtypes = [vp, i32, i32, ctypes.c_double, i32, vp, i64, vp]
    lib.pccm_graph_begin.argtypes = [vp]
    lib.pccm_graph_end.argtypes = [vp, ctypes.POINTER(i32)]
    lib.pccm_graph_launch.argtypes = [vp, i32]
    lib.pccm_graph_destroy.argtypes = [vp, i32]
    lib.pccm_profile_enable.argtypes = [vp, i32]
    lib.pccm_profile_reset.argtypes = [vp]
    lib.pccm_profile_get.argtypes = [vp, i32, dp, ctypes.POINTER(i64)]
    lib.pccm_nn_stats.argtypes = [vp, i32, ctypes.POINTER(i64)]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("pccm_last_error", "pccm_xvec_len", "pccm_cvec_len"):
            fn.restype = i32
    _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc == OK:
        return
    msg = load().pccm_last_error().decode("utf-8", "replace")
    if rc == E_ARG:
        raise ValueError(msg)
    if rc == E_RANGE:
        raise IndexError(msg)
    if rc == E_OOM:
        raise MemoryError(msg)
    if rc == E_STATE:
        raise PccmStateError(f"libpccm error {rc}: {msg}")
    if rc in (E_HIP, E_NODEV):
        raise PccmDeviceError(f"libpccm error {rc}: {msg}")
    raise RuntimeError(f"libpccm error {rc}: {msg}")


def device_count() -> int:
    n = ctypes.c_int(0)
    _check(load().pccm_device_count(ctypes.byref(n)))
    return int(n.value)


def xvec_len(n: int) -> int:
    return int(load().pccm_xvec_len(int(n)))


def cvec_len(n: int) -> int:
    return int(load().pccm_cvec_len(int(n)))


def finish_chunks(cvec: np.ndarray, n: int) -> float:
    """np.sum of the whole per-point column, from its (all-reduced) chunk vector (pccm_reduce_chunks_many)."""
    cvec = np.ascontiguousarray(cvec, dtype=np.float64)
    if cvec.shape[0] != cvec_len(n):
        raise ValueError("chunk vector has the wrong length")
    out = ctypes.c_double(0.0)
    _check(load().pccm_finish_chunks(cvec.ctypes.data_as(ctypes.c_void_p), int(n), ctypes.byref(out)))
    return np.float64(out.value)


def finish_sum(xvec: np.ndarray, n: int) -> float:
    """np.sum of the whole per-point column, from its (all-reduced) exchange vector."""
    xvec = np.ascontiguousarray(xvec, dtype=np.float64)
    if xvec.shape[0] != xvec_len(n):
        raise ValueError("exchange vector has the wrong length")
    out = ctypes.c_double(0.0)
    _check(load().pccm_finish_sum(xvec.ctypes.data_as(ctypes.c_void_p), int(n), ctypes.byref(out)))
    return np.float64(out.value)


def color_transform(colors: np.ndarray, scheme: str) -> np.ndarray:
    """rgb rows -> "ycc" / "yuv" rows (host helper of libpccm; see pccm_color_transform)."""
    src = np.ascontiguousarray(colors, dtype=np.float64)
    if src.ndim != 2 or src.shape[1] != 3:
        raise ValueError("colors must have shape (N, 3)")
    out = np.empty_like(src)
    _check(load().pccm_color_transform(src.ctypes.data_as(ctypes.c_void_p), src.shape[0], {"ycc": 1, "yuv": 2}[scheme],
                                       out.ctypes.data_as(ctypes.c_void_p)))
    return out


def lzf_decompress(data: bytes, size: int) -> bytes:
    """liblzf decompression (host helper of libpccm; PCD binary_compressed bodies)."""
    out = ctypes.create_string_buffer(max(1, int(size)))
    got = ctypes.c_int64()
    _check(load().pccm_lzf_decompress(ctypes.c_char_p(bytes(data)), len(data), out, int(size), ctypes.byref(got)))
    return out.raw[:got.value]


# ---- engine pool ------------------------------------------------------------------------------------------------
# Creating a context is cheap, tearing one down is not (4 ms of hipFree for a 1M-point pair) and a fresh one starts
# with cold allocations (+0.8 ms) and no grid decisions to inherit.  CloudPair therefore borrows its engine here and
# hands it back when it dies; a caller that evaluates one pair after the other pays the work, not the lifecycle.
_POOL: dict = {}
_POOL_MAX = 2
_POOL_LOCK = threading.Lock()      # evaluate_pairs drives the pool from several host threads; __del__ may run on any


def acquire_engine(device: int = 0) -> "Engine":
    while True:
        with _POOL_LOCK:
            pool = _POOL.get(int(device))
            eng = pool.pop() if pool else None
        if eng is None:
            return Engine(device)
        try:
            eng.reset()
            eng.pairs_served = getattr(eng, "pairs_served", 0) + 1      # (a context on its second pair: a loop over pairs, see CloudPair)
            return eng
        except PccmStateError:                 # a context that cannot be reset is not worth keeping
            eng.close()
        # anything else (a HIP failure in particular) propagates: it is not ours to hide


def release_engine(eng: "Engine") -> None:
    if not getattr(eng, "_ctx", None) or not eng._ctx.value:
        return                                 # closed by hand
    with _POOL_LOCK:
        pool = _POOL.setdefault(eng.device, [])
        if eng in pool:
            return
        keep = len(pool) < _POOL_MAX
        if keep:
            pool.append(eng)
    if not keep:
        eng.close()


def drain_pool() -> None:
    while True:
        with _POOL_LOCK:
            eng = next((pool.pop() for pool in _POOL.values() if pool), None)
        if eng is None:
            return
        eng.close()


import atexit  # noqa: E402

atexit.register(drain_pool)


def _as_rows(a, what: str) -> Tuple[object, int, int, int, object]:
    """-> (pointer, n, dtype code, on_device, keepalive) for an (N, 3) f32/f64 array or CUDA tensor."""
    if hasattr(a, "is_cuda") and hasattr(a, "data_ptr"):      # torch tensor
        import torch
        if not a.is_cuda:
            a = a.detach().cpu().numpy()
        else:
            if a.dtype not in (torch.float32, torch.float64):
                a = a.to(torch.float64)
            a = a.contiguous()
            if a.dim() != 2 or a.shape[1] != 3:
                raise ValueError(f"{what} must have shape (N, 3)")
            torch.cuda.current_stream(a.device).synchronize()
            return ctypes.c_void_p(a.data_ptr()), int(a.shape[0]), F32 if a.dtype == torch.float32 else F64, 1, a
    arr = np.asarray(a)
    if arr.dtype != np.float32:
        arr = arr.astype(np.float64, copy=False)
    arr = np.ascontiguousarray(arr)
    if arr.ndim != 2 or arr.shape[1] != 3:
        raise ValueError(f"{what} must have shape (N, 3)")
    return arr.ctypes.data_as(ctypes.c_void_p), int(arr.shape[0]), F32 if arr.dtype == np.float32 else F64, 0, arr


def _modes(normal_mode, k: int):
    """One normal mode for all k requests, or one per request."""
    if isinstance(normal_mode, str):
        return [NORMAL_MODES[normal_mode]] * k
    modes = [NORMAL_MODES[m] for m in normal_mode]
    if len(modes) != k:
        raise ValueError(f"{len(modes)} normal modes for {k} requests")
    return modes


class Engine:
    """One libpccm context = one GPU.  Method names mirror include/pccm.h."""
    keeps_self_search = True      # pccm_set_cloud(1, ...) leaves cloud 0 and its self search untouched (CloudPair.with_reconst)

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        self._lib = load()
        self._ctx = ctypes.c_void_p()
        _check(self._lib.pccm_ctx_create(int(device), ctypes.c_void_p(stream) if stream else None,
                                         ctypes.byref(self._ctx)))
        self.device = int(device)
        self._n = [0, 0]

    def close(self) -> None:
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.pccm_ctx_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self) -> None:
        """Forget clouds, shard, graphs and profile; keep the device allocations (pccm_ctx_reset)."""
        _check(self._lib.pccm_ctx_reset(self._ctx))
        self._n = [0, 0]

    # -- inputs ---------------------------------------------------------------------------
    def set_cloud(self, which: int, points) -> None:
        ptr, n, dt, dev, keep = _as_rows(points, "points")
        _check(self._lib.pccm_set_cloud(self._ctx, int(which), ptr, n, dt, dev))
        self._n[which] = n

    def set_normals(self, which: int, normals) -> None:
        ptr, n, dt, dev, keep = _as_rows(normals, "normals")
        self.__dict__.setdefault("_deferred", {}).pop(int(which), None)
        _check(self._lib.pccm_set_normals(self._ctx, int(which), ptr, n, dt, dev))

    def set_normals_deferred(self, which: int, normals) -> None:
        """Announce host normals now, upload them when they are needed or at flush_uploads() -- beside the searches, which do
        not read them (include/pccm.h).  The array is kept alive here until then; device arrays are set at once."""
        ptr, n, dt, dev, keep = _as_rows(normals, "normals")
        if dev:
            _check(self._lib.pccm_set_normals(self._ctx, int(which), ptr, n, dt, dev))
            return
        _check(self._lib.pccm_set_normals_deferred(self._ctx, int(which), ptr, n, dt))
        self.__dict__.setdefault("_deferred", {})[int(which)] = keep

    def set_io_staged(self, on: bool) -> None:
        """Large transfers through the context's own pinned buffers (default) or straight from / to the caller's arrays
        (see pccm_set_io_staged in include/pccm.h)."""
        _check(self._lib.pccm_set_io_staged(self._ctx, 1 if on else 0))

    def flush_uploads(self) -> None:
        try:
            _check(self._lib.pccm_flush_uploads(self._ctx))
        finally:
            self.__dict__.get("_deferred", {}).clear()

    def set_colors(self, which: int, colors) -> None:
        ptr, n, dt, dev, keep = _as_rows(colors, "colors")
        _check(self._lib.pccm_set_colors(self._ctx, int(which), ptr, n, dt, dev))

    def set_colors_u8(self, which: int, colors_u8) -> None:
        """Colours as (N, 3) uint8; the device widens them as k / 255.0 (what the file readers do on the host)."""
        c = np.ascontiguousarray(colors_u8, dtype=np.uint8)
        if c.ndim != 2 or c.shape[1] != 3:
            raise ValueError("colors_u8 must have shape (N, 3)")
        _check(self._lib.pccm_set_colors_u8(self._ctx, int(which), c.ctypes.data_as(ctypes.c_void_p), c.shape[0]))

    @staticmethod
    def _rows_arg(rows):
        if rows is None:
            return None, 0, None
        keep = np.ascontiguousarray(rows, dtype=np.int32)
        return keep.ctypes.data_as(ctypes.c_void_p), keep.shape[0], keep

    def color_reduce(self, direction: int, scheme: str, scale: float = 1.0, rows=None):
        """-> (column sums in np.add.reduce(axis=0) order, column maxima) of (scale * colour difference)**2."""
        ptr, n, keep = self._rows_arg(rows)
        sums, maxs = (ctypes.c_double * 3)(), (ctypes.c_double * 3)()
        _check(self._lib.pccm_color_reduce(self._ctx, int(direction), COLOR_SCHEMES[scheme], float(scale), ptr, n, sums, maxs))
        return np.array(sums[:], dtype=np.float64), np.array(maxs[:], dtype=np.float64)

    def extreme_rows(self, which: int, directions) -> np.ndarray:
        """Rows of the (about) farthest points of cloud ``which`` along each of <= 1024 directions."""
        d = np.ascontiguousarray(directions, dtype=np.float32)
        out = np.empty(d.shape[0], dtype=np.int32)
        _check(self._lib.pccm_extreme_rows(self._ctx, int(which), d.ctypes.data_as(ctypes.c_void_p), d.shape[0],
                                           out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def rows_outside(self, which: int, planes, margin: float) -> np.ndarray:
        """Rows of cloud ``which`` not strictly inside the polytope ``n.x + off <= 0`` (planes: (F, 4), Qhull's equations)."""
        p = np.ascontiguousarray(planes, dtype=np.float64)
        out = np.empty(self._n[which], dtype=np.int32)
        cnt = ctypes.c_int64()
        _check(self._lib.pccm_rows_outside(self._ctx, int(which), p.ctypes.data_as(ctypes.c_void_p), p.shape[0], float(margin),
                                           out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(cnt)))
        return out[:cnt.value]

    def obb_frames(self, hull_vertices, hull_triangles):
        """-> (extents, volume) of the smallest box over the frames of the hull's triangles (pccm_obb_frames)."""
        v = np.ascontiguousarray(hull_vertices, dtype=np.float64)
        t = np.ascontiguousarray(hull_triangles, dtype=np.float64)
        if v.ndim != 2 or v.shape[1] != 3 or t.ndim != 3 or t.shape[1:] != (3, 3):
            raise ValueError("hull_vertices must be (H, 3) and hull_triangles (T, 3, 3)")
        ext, vol = (ctypes.c_double * 3)(), ctypes.c_double()
        _check(self._lib.pccm_obb_frames(self._ctx, v.ctypes.data_as(ctypes.c_void_p), v.shape[0], t.ctypes.data_as(ctypes.c_void_p),
                                         t.shape[0], ext, ctypes.byref(vol)))
        return np.array(ext[:], dtype=np.float64), vol.value

    def seq_colsum(self, columns) -> np.ndarray:
        """np.add.reduce(a, axis=0) of a non-negative (N, 3) array, bit for bit, on the device."""
        a = np.asarray(columns, dtype=np.float64)
        if a.ndim != 2 or a.shape[1] != 3 or a.shape[0] == 0:
            raise ValueError("columns must have shape (N, 3), N > 0")
        cols = np.ascontiguousarray(a.T)
        out = (ctypes.c_double * 3)()
        _check(self._lib.pccm_seq_colsum(self._ctx, cols.ctypes.data_as(ctypes.c_void_p), a.shape[0], out))
        return np.array(out[:], dtype=np.float64)

    def color_rows(self, direction: int, scheme: str, what: int, scale: float = 1.0, rows=None) -> np.ndarray:
        """(n_iter, 3) rows: what = COLOR_OWN | COLOR_NEIGHBOUR | COLOR_DIFF | COLOR_SQUARE."""
        ptr, n, keep = self._rows_arg(rows)
        out = np.empty((self.n_iter(direction), 3), dtype=np.float64)
        _check(self._lib.pccm_color_rows(self._ctx, int(direction), COLOR_SCHEMES[scheme], float(scale), int(what), ptr, n,
                                         out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def estimate_normals(self, which: int, knn: int = 30) -> None:
        """Open3D-style normals (k-NN covariance, smallest eigenvector) computed and kept on the device."""
        _check(self._lib.pccm_estimate_normals(self._ctx, int(which), int(knn)))

    def get_normals(self, which: int) -> np.ndarray:
        out = np.empty((self._n[which], 3), dtype=np.float64)
        _check(self._lib.pccm_get_normals(self._ctx, int(which), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def set_shard(self, rank: int, world: int) -> None:
        _check(self._lib.pccm_set_shard(self._ctx, int(rank), int(world)))

    def set_shard_dir(self, direction: int, rank: int, world: int) -> None:
        """This context is number ``rank`` of the ``world`` ranks that share the rows of ``direction``; world 0: it
        owns none of them (pccm_set_shard_dir)."""
        _check(self._lib.pccm_set_shard_dir(self._ctx, int(direction), int(rank), int(world)))

    def shard_range(self, direction: int) -> Tuple[int, int]:
        b, e = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(self._lib.pccm_shard_range(self._ctx, int(direction), ctypes.byref(b), ctypes.byref(e)))
        return int(b.value), int(e.value)

    def n_iter(self, direction: int) -> int:
        return self._n[1] if direction == DIR_RIGHT else self._n[0]

    # -- nearest neighbours -----------------------------------------------------------------
    def nn(self, direction: int, engine: str = "auto") -> None:
        _check(self._lib.pccm_nn(self._ctx, int(direction), ENGINES[engine]))

    def nn_pair(self, engine: str = "auto") -> None:
        """Both directional sweeps of CloudPair.__init__ (cloud_pair.py:67-78) in one call."""
        _check(self._lib.pccm_nn_pair(self._ctx, ENGINES[engine]))

    def nn_fuse(self, direction: int, normal_mode: Optional[str]) -> None:
        """Fuse the D2 projection of ``direction`` into the next searches (pccm_nn_fuse); ``None`` switches it off."""
        _check(self._lib.pccm_nn_fuse(self._ctx, int(direction), -1 if normal_mode is None else NORMAL_MODES[normal_mode]))

    def nn_want_idx(self, on: bool) -> None:
        """Whether the searches store the matched row with every result (pccm_nn_want_idx); off halves the result bytes, and
        whoever asks for the rows later gets them through a repeated search of that direction."""
        _check(self._lib.pccm_nn_want_idx(self._ctx, int(bool(on))))

    def fetch_nn(self, direction: int, want_idx: bool = True, want_d2: bool = True):
        b, e = self.shard_range(direction)
        idx = np.empty(e - b, dtype=np.int32) if want_idx else None
        d2 = np.empty(e - b, dtype=np.float64) if want_d2 else None
        _check(self._lib.pccm_nn_fetch(self._ctx, int(direction),
                                       idx.ctypes.data_as(ctypes.c_void_p) if want_idx else None,
                                       d2.ctypes.data_as(ctypes.c_void_p) if want_d2 else None))
        return idx, d2

    def error_vectors(self, direction: int) -> np.ndarray:
        b, e = self.shard_range(direction)
        out = np.empty((e - b, 3), dtype=np.float64)
        _check(self._lib.pccm_error_vectors(self._ctx, int(direction), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def point_metric(self, direction: int, metric: int, normal_mode: str = "row") -> np.ndarray:
        b, e = self.shard_range(direction)
        out = np.empty(e - b, dtype=np.float64)
        _check(self._lib.pccm_point_metric(self._ctx, int(direction), int(metric), NORMAL_MODES[normal_mode],
                                           out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def tie_exposure(self, direction: int, normal_mode=None) -> dict:
        """Diagnostic of include/pccm.h's pccm_tie_exposure for this shard's rows: how many queries have several
        equidistant nearest neighbours, and the sums of the smallest / largest / chosen squared projection over them
        (``normal_mode`` None: counts only)."""
        out = (ctypes.c_double * 8)()
        _check(self._lib.pccm_tie_exposure(self._ctx, int(direction), -1 if normal_mode is None else NORMAL_MODES[normal_mode], out))
        return {"queries": int(out[0]), "tied": int(out[1]), "sum_min": float(out[2]), "sum_max": float(out[3]), "sum_pick": float(out[4]),
                "not_enumerated": int(out[5]), "max_multiplicity": int(out[6])}

    def reduce_prefetch(self, direction: int, metric: int, normal_mode: str = "row") -> None:
        """Enqueue a reduction without waiting; a later reduce() with the same arguments consumes it."""
        _check(self._lib.pccm_reduce_prefetch(self._ctx, int(direction), int(metric), NORMAL_MODES[normal_mode]))

    def reduce_prefetch_many(self, requests, normal_mode: str = "row") -> None:
        """``requests``: up to 8 ``(direction, metric)`` pairs, evaluated and reduced by two launches in all."""
        k = len(requests)
        if k == 0:
            return
        arr = ctypes.c_int * k
        dirs = arr(*[int(r[0]) for r in requests])
        mets = arr(*[int(r[1]) for r in requests])
        modes = arr(*_modes(normal_mode, k))
        _check(self._lib.pccm_reduce_prefetch_many(self._ctx, k, dirs, mets, modes))

    def reduce(self, direction: int, metric: int, normal_mode: str = "row"):
        """-> (xvec, min, max) of this shard; see pccm_reduce() in include/pccm.h."""
        xvec = np.empty(xvec_len(self.n_iter(direction)), dtype=np.float64)
        mm = np.empty(2, dtype=np.float64)
        _check(self._lib.pccm_reduce(self._ctx, int(direction), int(metric), NORMAL_MODES[normal_mode],
                                     xvec.ctypes.data_as(ctypes.c_void_p), mm.ctypes.data_as(ctypes.c_void_p)))
        return xvec, mm[0], mm[1]

    finish_sum = staticmethod(finish_sum)

    def reduce_total(self, direction: int, metric: int, normal_mode: str = "row"):
        """-> (sum, min, max) of the whole column; only when the context is not sharded."""
        out = (ctypes.c_double * 3)()
        _check(self._lib.pccm_reduce_total(self._ctx, int(direction), int(metric), NORMAL_MODES[normal_mode], out))
        return np.float64(out[0]), np.float64(out[1]), np.float64(out[2])

    def reduce_total_many(self, requests, normal_mode: str = "row"):
        """-> [(sum, min, max)] of up to 8 whole columns ``(direction, metric)`` in one call (unsharded contexts)."""
        k = len(requests)
        arr = ctypes.c_int * k
        out = (ctypes.c_double * (3 * k))()
        _check(self._lib.pccm_reduce_total_many(self._ctx, k, arr(*[int(r[0]) for r in requests]), arr(*[int(r[1]) for r in requests]),
                                                arr(*_modes(normal_mode, k)), out))
        return [(np.float64(out[3 * i]), np.float64(out[3 * i + 1]), np.float64(out[3 * i + 2])) for i in range(k)]

    def reduce_chunks_many(self, requests, normal_mode: str = "row"):
        """-> (one float64 array holding the chunk vectors of up to 8 columns ``(direction, metric)`` one after the other,
        their lengths, [(min, max)] of this shard) -- see pccm_reduce_chunks_many() in include/pccm.h."""
        k = len(requests)
        arr = ctypes.c_int * k
        lens = [cvec_len(self.n_iter(int(r[0]))) for r in requests]
        buf = np.empty(sum(lens), dtype=np.float64)
        mm = (ctypes.c_double * (2 * k))()
        dp = ctypes.POINTER(ctypes.c_double)
        _check(self._lib.pccm_reduce_chunks_many(self._ctx, k, arr(*[int(r[0]) for r in requests]), arr(*[int(r[1]) for r in requests]),
                                                 arr(*([NORMAL_MODES[normal_mode]] * k)), buf.ctypes.data_as(dp), mm))
        return buf, lens, [(mm[2 * i], mm[2 * i + 1]) for i in range(k)]

    finish_chunks = staticmethod(finish_chunks)

    # -- housekeeping -------------------------------------------------------------------------
    def sync(self) -> None:
        _check(self._lib.pccm_sync(self._ctx))

    def drop_caches(self) -> None:
        _check(self._lib.pccm_drop_caches(self._ctx))

    # -- hipGraph capture of {drop_caches, nn, reduce_prefetch}* (see include/pccm.h) ----------------
    def graph_begin(self) -> None:
        _check(self._lib.pccm_graph_begin(self._ctx))

    def graph_end(self) -> int:
        gid = ctypes.c_int(-1)
        _check(self._lib.pccm_graph_end(self._ctx, ctypes.byref(gid)))
        return int(gid.value)

    def graph_abort(self) -> None:
        """Leave capture mode after a captured call failed (the partial graph is discarded)."""
        gid = ctypes.c_int(-1)
        self._lib.pccm_graph_end(self._ctx, ctypes.byref(gid))      # reports the failure; nothing to keep

    def graph_launch(self, graph_id: int) -> None:
        _check(self._lib.pccm_graph_launch(self._ctx, int(graph_id)))

    def graph_destroy(self, graph_id: int) -> None:
        _check(self._lib.pccm_graph_destroy(self._ctx, int(graph_id)))

    def profile(self, on: bool) -> None:
        _check(self._lib.pccm_profile_enable(self._ctx, int(bool(on))))

    def profile_reset(self) -> None:
        _check(self._lib.pccm_profile_reset(self._ctx))

    def profile_get(self, kernel_class: str) -> Tuple[float, int]:
        ms, n = ctypes.c_double(0.0), ctypes.c_int64(0)
        _check(self._lib.pccm_profile_get(self._ctx, KERNEL_CLASSES[kernel_class], ctypes.byref(ms), ctypes.byref(n)))
        return float(ms.value), int(n.value)

    def nn_stats(self, direction: int) -> dict:
        out = (ctypes.c_int64 * 3)()
        _check(self._lib.pccm_nn_stats(self._ctx, int(direction), out))
        stats = {"fallback_queries": int(out[0]), "splits": int(out[1]), "pairs": int(out[2])}
        _check(self._lib.pccm_nn_stats(self._ctx, int(direction) | 0x10, out))      # PCCM_STATS_TAIL
        stats["tail_queries"] = int(out[0])
        return stats
