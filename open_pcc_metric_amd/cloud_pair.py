"""CloudPair on the GPU -- drop-in for ``open_pcc_metric.cloud_pair.CloudPair``.

Reference: open_pcc_metric/cloud_pair.py:45-124.  Same constructor arguments, same attributes
(``clouds``, ``origin_cloud``, ``reconst_cloud``) and the same ten getters; as in the reference
all nearest-neighbour work happens eagerly in ``__init__`` (cloud_pair.py:54-80).  What the
getters return are *device columns*: array-likes that live in HBM, answer ``np.sum`` /
``np.max`` / ``np.min`` / ``np.square`` with fused GPU reductions (bit-identical to what NumPy
returns on the materialised array) and turn into ordinary ``float64`` ndarrays on
``np.asarray``.  The metric DAG above therefore runs unchanged and never copies an N-sized
array to the host unless a caller asks for one.

Differences from the reference, all deliberate:

* inputs are never mutated (quirk Q5).  A cloud without normals gets them estimated on the GPU the
  first time a point-to-plane metric needs them (k = 30 nearest-neighbour covariance, as Open3D's
  ``estimate_normals`` at cloud_pair.py:61-64 does; not bit-pinned, see csrc/pccm_normals.hip) and
  kept inside the pair; ``estimate_normals=False`` raises ``ValueError`` instead;
* ``normal_index="row"`` (default) reproduces the reference's D2, including its ``IndexError``
  when the iterating cloud is larger than the other one (quirk Q1); ``"neighbour"`` uses the
  matched point's normal;
* exact ties go to the smallest row index (Open3D's order is traversal dependent);
* ``extent=`` injects ``get_extent()`` (the minimal-OBB is CPU code and not parity-pinned);
* ``group=`` shards the pair over the ranks of a ``torch.distributed`` group: by direction first (half of the ranks
  search cloud_pair.py:67-72, the other half :73-78, so every rank builds one cloud's search structure), then by
  query rows inside each half (``shard_mode="rows"``: every rank takes rows of both directions, as in round 1);
* ``use_graph=True`` lets ``recompute()`` replay the whole sweep + reductions of the previous report
  as one hipGraph launch (for callers that evaluate the same resident pair repeatedly).
"""
from __future__ import annotations

import os
import typing

import numpy as np

from . import _native as nat
from .collective import Collective
from .extent import minimal_obb_extent

_REDUCERS = {np.sum: "sum", np.max: "max", np.min: "min", np.amax: "max", np.amin: "min"}


def _plain(x):
    return np.asarray(x) if isinstance(x, _DeviceArray) else x


class _DeviceArray(np.lib.mixins.NDArrayOperatorsMixin):
    """Common array-like plumbing; anything not fused falls back to the materialised ndarray."""
    dtype = np.dtype(np.float64)
    _host: typing.Optional[np.ndarray] = None

    def __array__(self, dtype=None, copy=None):
        if self._host is None:
            self._host = self._materialise()
            self._host.setflags(write=False)
        return self._host if dtype is None else self._host.astype(dtype)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        fused = self._fused_ufunc(ufunc, method, inputs, kwargs)
        if fused is not NotImplemented:
            return fused
        if "out" in kwargs:
            kwargs["out"] = tuple(_plain(o) for o in kwargs["out"])
        return getattr(ufunc, method)(*[_plain(i) for i in inputs], **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        fused = self._fused_function(func, args, kwargs)
        if fused is not NotImplemented:
            return fused
        return func(*[_plain(a) for a in args], **{k: _plain(v) for k, v in kwargs.items()})

    def _fused_ufunc(self, ufunc, method, inputs, kwargs):
        return NotImplemented

    def _fused_function(self, func, args, kwargs):
        return NotImplemented

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape))

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, key):
        return np.asarray(self)[key]

    def __iter__(self):
        return iter(np.asarray(self))

    def astype(self, dtype, **kw):
        return np.asarray(self).astype(dtype, **kw)

    def tolist(self):
        return np.asarray(self).tolist()

    def __repr__(self):
        return f"<{type(self).__name__} {self._label} shape={self.shape} on GPU>"


class DeviceColumn(_DeviceArray):
    """A per-point float64 column of one direction: ``kind`` is

    ``"d1"``        squared NN distances (NeighbourDistances / D1 EuclideanDistance),
    ``"proj"``      signed point-to-plane projections (ErrorVector, point_to_plane=True),
    ``"d2"``        their squares (D2 EuclideanDistance),
    ``"boundary"``  sqrt of the self-search distances (get_boundary_sqrt_distances).
    """

    def __init__(self, pair: "CloudPair", direction: int, kind: str):
        self._pair, self._dir, self._kind = pair, direction, kind
        self.shape = (pair._engine.n_iter(direction),)
        self._label = f"{kind}[dir={direction}]"
        self._red = None

    _METRIC = {"d1": nat.METRIC_D1, "boundary": nat.METRIC_D1, "proj": nat.METRIC_PROJ, "d2": nat.METRIC_D2}

    def _materialise(self) -> np.ndarray:
        p = self._pair
        if self._kind in ("d1", "boundary"):
            _, col = p._engine.fetch_nn(self._dir, want_idx=False)
        else:
            col = p._engine.point_metric(self._dir, self._METRIC[self._kind], p.normal_index)
        col = p._gather(self._dir, col)
        return np.sqrt(col) if self._kind == "boundary" else col

    def _reduced(self):
        """(sum, min, max) of the whole column: fused on the GPU, exchanged across ranks."""
        if self._red is None:
            p = self._pair
            if p._fast_totals:                            # unsharded, engine with pccm_reduce_total: the usual case
                total, mn, mx = p._total(self._dir, self._METRIC[self._kind])
            elif p._coll.sharded:
                total, mn, mx = p._sharded_reduction(self._dir, self._METRIC[self._kind])
            else:
                xvec, mn, mx = p._engine.reduce(self._dir, self._METRIC[self._kind], p.normal_index)
                total = p._engine.finish_sum(xvec, self.shape[0])
            if self._kind == "boundary":          # sqrt is monotonic: min/max commute with it
                mn, mx, total = np.sqrt(mn), np.sqrt(mx), None
            self._red = (total, np.float64(mn), np.float64(mx))
        return self._red

    def _fused_function(self, func, args, kwargs):
        which = _REDUCERS.get(func)
        if which is None or len(args) != 1 or args[0] is not self:
            return NotImplemented
        if set(kwargs) - {"axis"} or kwargs.get("axis", None) not in (None, 0):
            return NotImplemented
        total, mn, mx = self._reduced()
        if which == "sum":
            return NotImplemented if total is None else total
        return mx if which == "max" else mn

    def _fused_ufunc(self, ufunc, method, inputs, kwargs):
        if ufunc is np.square and method == "__call__" and not kwargs and self._kind == "proj":
            return DeviceColumn(self._pair, self._dir, "d2")     # metric.py:179 stays on the GPU
        return NotImplemented


class DeviceRows(_DeviceArray):
    """(N, 3) error vectors ``iter[i] - search[nn(i)]`` of one direction (cloud_pair.py:90-100)."""

    def __init__(self, pair: "CloudPair", direction: int):
        self._pair, self._dir = pair, direction
        self.shape = (pair._engine.n_iter(direction), 3)
        self._label = f"error_vector[dir={direction}]"

    def _materialise(self) -> np.ndarray:
        return self._pair._gather(self._dir, self._pair._engine.error_vectors(self._dir))


class DeviceColorRows(_DeviceArray):
    """(N, 3) colour rows of one direction, kept in HBM (metric.py:302-333, 389-427).

    ``what``: ``"neighbour"`` -- colours of the matched points, ``np.take(colors, idx, axis=0)`` of
    cloud_pair.py:120-124;  ``"diff"`` -- ``scale * (T(own) - T(neighbour))`` in ``scheme``;
    ``"square"`` -- its square.  ``np.mean(square, axis=0)`` and ``np.max(square, axis=0)`` are answered
    by pccm_color_reduce (bit-identical to NumPy on the materialised rows); ``255 * diff``,
    ``diff ** 2`` and ``np.square(diff)`` stay on the device; anything else materialises."""
    _WHAT = {"neighbour": nat.COLOR_NEIGHBOUR, "diff": nat.COLOR_DIFF, "square": nat.COLOR_SQUARE}

    def __init__(self, pair: "CloudPair", direction: int, what: str, scheme: str = "rgb", scale: float = 1.0):
        self._pair, self._dir, self._what, self._scheme, self._scale = pair, direction, what, scheme, float(scale)
        self.shape = (pair._engine.n_iter(direction), 3)
        self._label = f"colour {what}[dir={direction}, {scheme}, x{self._scale:g}]"

    def in_scheme(self, scheme: str) -> "DeviceColorRows":
        """own - neighbour in ``scheme`` (the np.subtract of metric.py:326-329 / 417-420)."""
        return DeviceColorRows(self._pair, self._dir, "diff", scheme)

    def _materialise(self) -> np.ndarray:
        p = self._pair
        return p._engine.color_rows(self._dir, self._scheme, self._WHAT[self._what], self._scale, p._colour_rows_arg(self._dir))

    def _fused_ufunc(self, ufunc, method, inputs, kwargs):
        if method != "__call__" or kwargs or self._what != "diff":
            return NotImplemented
        if ufunc is np.square and inputs[0] is self:
            return DeviceColorRows(self._pair, self._dir, "square", self._scheme, self._scale)
        if ufunc is np.power and inputs[0] is self and np.isscalar(inputs[1]) and inputs[1] == 2:
            return DeviceColorRows(self._pair, self._dir, "square", self._scheme, self._scale)
        if ufunc is np.multiply and len(inputs) == 2 and self._scale == 1.0:
            other = inputs[1] if inputs[0] is self else inputs[0]
            if np.isscalar(other) and not isinstance(other, (bool, np.bool_)):
                return DeviceColorRows(self._pair, self._dir, "diff", self._scheme, float(other))
        return NotImplemented

    def _fused_function(self, func, args, kwargs):
        if self._what != "square" or len(args) != 1 or args[0] is not self or set(kwargs) != {"axis"} or kwargs["axis"] != 0:
            return NotImplemented
        if func is np.mean:
            sums, _ = self._pair._colour_reduction(self._dir, self._scheme, self._scale)
            return sums / self.shape[0]                        # np.mean = np.add.reduce(axis=0) / N
        if func in (np.max, np.amax):
            return self._pair._colour_reduction(self._dir, self._scheme, self._scale)[1]
        return NotImplemented


class CloudColorsView(np.ndarray):
    """``np.asarray(cloud.colors)`` that remembers which cloud of which pair it came from."""
    _pccm_origin: typing.Optional[tuple] = None

    def __array_finalize__(self, obj):
        self._pccm_origin = None


class CloudNormalsView(np.ndarray):
    """``np.asarray(cloud.normals)`` that remembers which cloud of which pair it came from, so that
    ErrorVector can keep the projection on the GPU (metric.py:92-98, 146-153)."""
    _pccm_origin: typing.Optional[tuple] = None

    def __array_finalize__(self, obj):
        self._pccm_origin = None      # any derived array is just data


class CloudPair:
    clouds: typing.Tuple[typing.Any, typing.Any]

    def __init__(self, origin_cloud, reconst_cloud, *, device: typing.Optional[int] = None,
                 nn_engine: str = "auto", normal_index: str = "row", extent=None, group=None,
                 use_graph: bool = False, estimate_normals: bool = True, normals_knn: int = 30,
                 shard_mode: str = "direction", _engine=None, _uploads_first: bool = False,
                 staged_io: typing.Optional[bool] = None):
        if normal_index not in nat.NORMAL_MODES:
            raise ValueError("normal_index must be 'row' or 'neighbour'")
        if nn_engine not in nat.ENGINES:
            raise ValueError(f"nn_engine must be one of {sorted(nat.ENGINES)}")
        self.clouds = (origin_cloud, reconst_cloud)
        self.normal_index = normal_index
        self.nn_engine = nn_engine
        self._use_graph = bool(use_graph)
        self._estimate_normals, self._normals_knn = bool(estimate_normals), int(normals_knn)
        self._estimated = [False, False]
        self._xchg, self._xchg_wanted = {}, []
        self._colours_on_device = [False, False]
        self._colour_red = {}
        self._graph_id = None
        self._last_wanted = None
        self._extent = None if extent is None else np.asarray(extent, dtype=np.float64)
        self._coll = Collective(group)
        self._owns_engine = False
        if _engine is None:
            if device is None:
                device = int(os.environ.get("LOCAL_RANK", "0")) if self._coll.sharded else 0
            _engine = nat.acquire_engine(device)  # raises without libpccm.so or without a GPU; pooled contexts are reused
            self._owns_engine = True
            self._coll.device = device            # the nccl exchange is staged on the same GPU
        self._engine = _engine
        # ``staged_io``: the clouds' arrays will be FREED while this context is still in use (a sequence of pairs read from files):
        # their bytes then go through the context's own pinned buffers instead of being handed to the HIP runtime, which pins
        # the caller's pages and keeps the mapping -- and whose tear-down, when such an array is freed, stops every GPU queue of
        # the process for 13-27 ms (pccm_set_io_staged).  None: direct for a context's first pair, staged from its second on.
        # A pooled context that has served a pair before means a LOOP over pairs -- whose clouds are normally read, used and
        # freed one after the other -- and takes the staged path unless told otherwise (a loop over fresh 1M-point pairs: 2.2-5 ms
        # per pair staged; direct, every second pair stands still for 30 ms).
        if staged_io is None and self._owns_engine and getattr(_engine, "pairs_served", 0) >= 1:
            staged_io = True
        if staged_io is not None and hasattr(_engine, "set_io_staged"):
            _engine.set_io_staged(bool(staged_io))
        self._fast_totals = not self._coll.sharded and hasattr(_engine, "reduce_total")   # whole columns finished by one call
        # Points first; normals are announced and cross PCIe behind the searches, which do not read them (the reference's
        # constructor orders nothing between the two: cloud_pair.py:61-80) -- see the flush at the end
        deferred = hasattr(_engine, "set_normals_deferred")
        for k, cloud in enumerate(self.clouds):
            _engine.set_cloud(k, cloud.points)
            if _has_normals(cloud):
                (_engine.set_normals_deferred if deferred else _engine.set_normals)(k, cloud.normals)
        if shard_mode not in ("direction", "rows"):
            raise ValueError("shard_mode must be 'direction' or 'rows'")
        self._plan = shard_plan(self._coll.world, shard_mode if hasattr(_engine, "set_shard_dir") else "rows")
        if self._coll.sharded:
            if hasattr(_engine, "set_shard_dir"):
                for direction in (nat.DIR_LEFT, nat.DIR_RIGHT, nat.DIR_SELF):
                    _engine.set_shard_dir(direction, *self._plan[direction][self._coll.rank])
            else:
                _engine.set_shard(self._coll.rank, self._coll.world)
        self._update_fusion()
        if deferred and _uploads_first:
            _engine.flush_uploads()                       # (A/B for bench.py: everything uploaded before the first search starts)
        self.recompute()
        if deferred:
            _engine.flush_uploads()                       # (the searches are running: the normals' upload runs beside them)

    def with_reconst(self, reconst_cloud) -> "CloudPair":
        """The pair of THIS pair's origin cloud and another reconstructed cloud -- one reference against several decoded
        versions (BASELINE.json configs[4]; the reference runs its command line once per version, handler.py:57-66, and
        repeats everything).  What belongs to the origin cloud alone is kept: its points, normals (given or estimated,
        cloud_pair.py:61-64) and colours in HBM, its spatial order, ``get_extent()`` (cloud_pair.py:111-112) and the self
        search behind ``get_boundary_sqrt_distances()`` (cloud_pair.py:108-109); only the new cloud is uploaded and the two
        directional searches run.  The GPU context moves to the new pair: this one is closed.  Rows are the same, bit for
        bit, as those of a fresh ``CloudPair(origin_cloud, reconst_cloud)``."""
        eng = self.__dict__.get("_engine")
        if eng is None:
            raise RuntimeError("this CloudPair has been closed")
        if self._coll.sharded or not hasattr(eng, "nn_pair"):
            # (a sharded pair keeps nothing that pays: every rank would have to agree on what is resident)
            kw = dict(nn_engine=self.nn_engine, normal_index=self.normal_index, extent=self._extent, use_graph=self._use_graph,
                      estimate_normals=self._estimate_normals, normals_knn=self._normals_knn)
            group, owns = self._coll.group, self._owns_engine
            self.__dict__.pop("_engine")
            if owns:
                nat.release_engine(eng)
                return CloudPair(self.clouds[0], reconst_cloud, group=group, **kw)
            return CloudPair(self.clouds[0], reconst_cloud, group=group, _engine=eng, **kw)
        new = object.__new__(CloudPair)
        new.__dict__.update({k: v for k, v in self.__dict__.items() if k != "_engine"})
        new.clouds = (self.clouds[0], reconst_cloud)
        new._estimated = [self._estimated[0], False]
        new._colours_on_device = [self._colours_on_device[0], False]
        new._xchg, new._xchg_wanted, new._colour_red = {}, [], {}
        new._graph_id, new._last_wanted = None, None
        keep_self = bool(self.__dict__.get("_self_done")) and bool(getattr(eng, "keeps_self_search", False))
        self_total = self.__dict__.get("_totals", {}).get((nat.DIR_SELF, nat.METRIC_D1)) if keep_self else None
        self.__dict__.pop("_engine")                     # moved: this pair is closed, the context is not handed back
        self._owns_engine = False
        new._engine = eng
        if hasattr(eng, "set_io_staged"):
            eng.set_io_staged(True)                              # (the superseded cloud is about to be freed by its owner: see __init__)
        eng.set_cloud(1, reconst_cloud.points)               # (the library keeps cloud 0, everything it owns and its self search)
        deferred = hasattr(eng, "set_normals_deferred")
        if _has_normals(reconst_cloud):
            (eng.set_normals_deferred if deferred else eng.set_normals)(1, reconst_cloud.normals)
        new._update_fusion()
        new.recompute()
        if deferred:
            eng.flush_uploads()
        new._self_done = keep_self
        if self_total is not None:
            new._totals[(nat.DIR_SELF, nat.METRIC_D1)] = self_total
        return new

    def close(self) -> None:
        """Hand the GPU context back (it is reused by the next pair); the pair must not be used afterwards."""
        eng = self.__dict__.pop("_engine", None)
        if eng is not None and self.__dict__.get("_owns_engine"):
            nat.release_engine(eng)

    def __del__(self):
        try:
            self.close()
        except Exception:            # noqa: BLE001 -- interpreter shutdown
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _update_fusion(self) -> None:
        """Let the searches leave the D2 projection next to the distance (pccm_nn_fuse) for every direction whose
        *other* cloud has normals: a report then needs no second pass over the points (metric.py:146-153 fused into
        cloud_pair.py:67-78).  Purely an optimisation; the library ignores what it cannot honour."""
        eng = self._engine
        if not hasattr(eng, "nn_fuse"):
            return
        for direction, other in ((nat.DIR_LEFT, 1), (nat.DIR_RIGHT, 0)):
            eng.nn_fuse(direction, self.normal_index if self._normals_ready(other) else None)
        # the matched rows (cloud_pair.py:34-42) are only read by the colour metrics and the error-vector / neighbour
        # getters: clouds without colours leave them out of the result records (a getter that asks later still gets them)
        if hasattr(eng, "nn_want_idx"):
            eng.nn_want_idx(any(_has_colors(c) for c in self.clouds))

    def recompute(self) -> None:
        """Run both directional sweeps again on the clouds already resident in HBM
        (cloud_pair.py:67-78 does this once, eagerly, in the constructor) and drop cached results."""
        eng = self._engine
        self._idx_cache = {}
        self._xchg = {}
        self._totals = {}
        self._colour_red = {}
        if self._use_graph and self._last_wanted is not None and hasattr(eng, "graph_begin"):
            wants_self = "boundary" in self._last_wanted
            if self._graph_id is not None:
                try:
                    eng.graph_launch(self._graph_id)          # sweeps + the last report's reductions, one launch
                    self._self_done = wants_self
                    return
                except nat.PccmStateError:
                    self._graph_id = None                     # stale (buffers changed): run eagerly, capture again later
                # anything else -- a HIP failure during the replay in particular -- is not retried and not hidden
            else:
                try:
                    eng.graph_begin()
                    self._enqueue_sweeps()
                    self._self_done = False
                    self.prefetch_reductions(self._last_wanted, _remember=False)
                    self._graph_id = eng.graph_end()
                    self._self_done = wants_self
                    return
                except (nat.PccmStateError, ValueError, IndexError):
                    eng.graph_abort()                         # leave capture mode; results were invalidated
                    self._graph_id = None
                    self._use_graph = False                   # capture not possible here: stay eager
                except BaseException:
                    eng.graph_abort()                         # leave capture mode, then let the failure surface
                    self._graph_id = None
                    raise
        self._enqueue_sweeps()
        self._self_done = False

    def _enqueue_sweeps(self) -> None:
        eng = self._engine
        if hasattr(eng, "drop_caches"):
            eng.drop_caches()                                 # search structures are rebuilt, like the KD-trees
        if hasattr(eng, "nn_pair"):
            eng.nn_pair(self.nn_engine)                       # cloud_pair.py:67-78, both directions fused
        else:
            eng.nn(nat.DIR_LEFT, self.nn_engine)
            eng.nn(nat.DIR_RIGHT, self.nn_engine)

    # -- helpers ------------------------------------------------------------------------------
    def _gather(self, direction: int, local: np.ndarray) -> np.ndarray:
        if not self._coll.sharded:
            return local
        n = self._engine.n_iter(direction)
        counts = [_shard_bounds(n, *self._plan[direction][r]) for r in range(self._coll.world)]
        return self._coll.allgather_rows(local, [e - b for b, e in counts])

    def _sharded_reduction(self, direction: int, metric: int):
        """(sum, min, max) of a whole column when the query axis is sharded.

        The exchange vectors of ALL columns the current report asked for (prefetch_reductions) and their
        extrema travel in ONE all-reduce(SUM) per report, however many columns it has.  Every rank calls
        this with the same requests in the same order."""
        key = (direction, metric)
        if key not in self._xchg:
            eng, coll = self._engine, self._coll
            batch = [k for k in self._xchg_wanted if k not in self._xchg]
            if key not in batch:
                batch.append(key)
            parts, ext = [], []
            # shards that start and end on whole 8192-row chunks (every direction of the batch, decided from the row
            # counts and the plan alone: the same answer on every rank) exchange one number per chunk instead of one per
            # 128-row leaf -- 20 KB instead of 0.5 MB per report at 1M points
            chunked = hasattr(eng, "reduce_chunks_many") and len(batch) <= 8 and all(self._chunk_shards(d) for d, _ in batch)
            if chunked:
                buf, lens, mms = eng.reduce_chunks_many(batch, self.normal_index)
                pos = 0
                for ln, (mn, mx) in zip(lens, mms):
                    parts.append(buf[pos:pos + ln])
                    ext += [mx, mn]
                    pos += ln
            else:
                for d, m in batch:
                    xvec, mn, mx = eng.reduce(d, m, self.normal_index)
                    parts.append(xvec)
                    ext += [mx, mn]
            finish = eng.finish_chunks if chunked else eng.finish_sum
            # the extrema ride along: every rank owns one block of slots and leaves the others zero, so the SUM
            # hands every rank all the local extrema unchanged (x + 0 is exact) -- one collective per report
            slots = np.zeros((coll.world, len(ext)), dtype=np.float64)
            slots[coll.rank] = ext
            summed = coll.allreduce(np.concatenate(parts + [slots.ravel()]), "sum")   # x + 0 is exact: bitwise assembly
            ext = summed[len(summed) - slots.size:].reshape(slots.shape)
            pos = 0
            for i, (d, m) in enumerate(batch):
                n = eng.n_iter(d)
                ln = len(parts[i])
                self._xchg[(d, m)] = (finish(summed[pos:pos + ln], n), np.float64(np.min(ext[:, 2 * i + 1])),
                                      np.float64(np.max(ext[:, 2 * i])))
                pos += ln
        return self._xchg[key]

    def _chunk_shards(self, direction: int) -> bool:
        """Do the shards of ``direction`` start and end on 8192-row chunks?  (the rule of pccm_set_shard: whenever the cloud
        has a chunk for every rank that shares the direction)"""
        sub_world = max(w for _, w in self._plan[direction])
        return sub_world >= 1 and self._engine.n_iter(direction) >= sub_world * 8192

    def _total(self, direction: int, metric: int):
        """(sum, min, max) of a whole column, unsharded.  The first column a report asks for brings every column the
        report enqueued (prefetch_reductions) back in ONE call: one wait for the GPU, one trip through ctypes."""
        key = (direction, metric)
        done = self._totals.get(key)
        if done is not None:
            return done
        if True:
            eng = self._engine
            batch = [k for k in self._xchg_wanted if k not in self._totals]
            if key not in batch:
                batch.append(key)
            if hasattr(eng, "reduce_total_many") and len(batch) > 1:
                for k, v in zip(batch[:8], eng.reduce_total_many(batch[:8], self.normal_index)):
                    self._totals[k] = v
            if key not in self._totals:
                self._totals[key] = eng.reduce_total(direction, metric, self.normal_index)
        return self._totals[key]

    def tie_exposure(self, is_left: bool = True, point_to_plane: bool = True) -> dict:
        """Opt-in diagnostic (not part of any report): how far the order of EXACT ties can move the point-to-plane MSE.

        The reference keeps whichever equidistant nearest neighbour nanoflann's traversal meets (cloud_pair.py:22-23),
        this package the smallest row; D1 is identical either way, the projection (metric.py:146-153) is not.  Returns
        for one direction: ``tie_rate`` (queries with >= 2 equidistant nearest neighbours), ``max_multiplicity``, and --
        with ``point_to_plane`` -- ``d2_mse_min <= d2_mse_pick <= d2_mse_max``: the smallest / this package's / the largest
        GeoMSE(point_to_plane=True) any tie rule can produce (the reference's value lies in that interval; on tie-free
        data the three coincide).  Sums are plain fp64 accumulations: a diagnostic, not a bit-pinned metric."""
        direction = nat.DIR_LEFT if is_left else nat.DIR_RIGHT
        mode = None
        if point_to_plane:
            self._require_normals(1 if is_left else 0)
            mode = self.normal_index
        r = self._engine.tie_exposure(direction, mode)      # (the sweeps ran in the constructor / recompute)
        vec = np.array([r["queries"], r["tied"], r["sum_min"], r["sum_max"], r["sum_pick"], r["not_enumerated"]], dtype=np.float64)
        mult = np.array([float(r["max_multiplicity"])])
        if self._coll.sharded:
            vec = self._coll.allreduce(vec, "sum")
            mult = self._coll.allreduce(mult, "max")
        n = self._engine.n_iter(direction)
        out = {"direction": "left" if is_left else "right", "queries": int(vec[0]), "tied_queries": int(vec[1]),
               "tie_rate": float(vec[1] / max(vec[0], 1.0)), "max_multiplicity": int(mult[0]), "not_enumerated": int(vec[5])}
        if point_to_plane:
            out.update(d2_mse_min=float(vec[2] / n), d2_mse_max=float(vec[3] / n), d2_mse_pick=float(vec[4] / n),
                       normal_index=self.normal_index)
        return out

    def _neighbour_index(self, direction: int) -> np.ndarray:
        if direction not in self._idx_cache:
            idx, _ = self._engine.fetch_nn(direction, want_d2=False)
            self._idx_cache[direction] = self._gather(direction, idx).astype(np.int64)
        return self._idx_cache[direction]

    def _normals_ready(self, which: int) -> bool:
        return self._estimated[which] or _has_normals(self.clouds[which])

    def _require_normals(self, which: int) -> None:
        if self._normals_ready(which):
            return
        if not (self._estimate_normals and hasattr(self._engine, "estimate_normals")):
            raise ValueError(
                f"cloud {which} has no normals: point-to-plane metrics need them "
                "(the reference would call Open3D's estimate_normals here, cloud_pair.py:61-64)")
        self._engine.estimate_normals(which, self._normals_knn)      # stays in HBM; inputs are not touched
        self._estimated[which] = True
        self._graph_id = None                                        # device buffers changed
        self._update_fusion()                                        # the next sweeps carry the projection along

    # -- reference surface, cloud_pair.py:82-124 -------------------------------------------------
    @property
    def origin_cloud(self):
        return self.clouds[0]

    @property
    def reconst_cloud(self):
        return self.clouds[1]

    def get_left_error_vector(self):
        return DeviceRows(self, nat.DIR_LEFT)

    def get_right_error_vector(self):
        return DeviceRows(self, nat.DIR_RIGHT)

    def get_left_neighbour_distances(self):
        return DeviceColumn(self, nat.DIR_LEFT, "d1")

    def get_right_neighbour_distances(self):
        return DeviceColumn(self, nat.DIR_RIGHT, "d1")

    def get_boundary_sqrt_distances(self):
        if not self._self_done:
            self._engine.nn(nat.DIR_SELF, self.nn_engine)
            self._self_done = True
        return DeviceColumn(self, nat.DIR_SELF, "boundary")

    def get_extent(self):
        if self._extent is None:
            self._extent = minimal_obb_extent(_host_rows(self.clouds[0].points), self._engine)
        return self._extent

    def get_normals(self, which: int):
        """np.asarray(clouds[which].normals), tagged for the fused projection (metric.py:92-98)."""
        self._require_normals(which)
        if self._estimated[which]:
            host = self._engine.get_normals(which)
        else:
            host = np.asarray(_host_rows(self.clouds[which].normals))
        view = host.view(CloudNormalsView)
        view._pccm_origin = (id(self), which)
        return view

    def _own_colours(self, which: int):
        """cloud_pair.py:114-118; tagged so that the colour metrics can recognise the pair's own rows."""
        view = np.asarray(_host_rows(self.clouds[which].colors)).view(CloudColorsView)
        view._pccm_origin = (id(self), which)
        return view

    def get_left_colors(self):
        return self._own_colours(0)

    def get_right_colors(self):
        return self._own_colours(1)

    def _ensure_colours(self) -> None:
        for k, cloud in enumerate(self.clouds):
            if self._colours_on_device[k]:
                continue
            u8 = getattr(cloud, "colors_u8", None)
            if u8 is not None and hasattr(self._engine, "set_colors_u8"):
                self._engine.set_colors_u8(k, u8)        # file colours: 3 B/point up, the k / 255.0 redone on the device
            else:
                self._engine.set_colors(k, _host_rows(cloud.colors))
            self._colours_on_device[k] = True

    def _colour_rows_arg(self, direction: int):
        """Neighbour rows for the colour kernels: the context's own (None) unless the search was sharded."""
        self._ensure_colours()
        return self._neighbour_index(direction) if self._coll.sharded else None

    def _colour_reduction(self, direction: int, scheme: str, scale: float):
        key = (direction, scheme, scale)
        if key not in self._colour_red:
            self._colour_red[key] = self._engine.color_reduce(direction, scheme, scale, self._colour_rows_arg(direction))
        return self._colour_red[key]

    def get_left_neighbour_colors(self):
        """cloud_pair.py:120-121: the matched points' colours -- gathered on the device when asked for."""
        return DeviceColorRows(self, nat.DIR_LEFT, "neighbour")

    def get_right_neighbour_colors(self):
        return DeviceColorRows(self, nat.DIR_RIGHT, "neighbour")

    def prefetch_reductions(self, wanted, _remember: bool = True) -> None:
        """Enqueue the fused reductions a report is about to ask for, without waiting for any of them.

        ``wanted``: iterable of ``(is_left, point_to_plane)`` pairs and/or the string ``"boundary"``.
        MetricCalculator.calculate() calls this after walking the DAG of the requested metrics, so
        that the host waits for the GPU once per report instead of once per column.  Purely an
        optimisation: columns that were not prefetched are reduced on demand."""
        eng = self._engine
        can_prefetch = hasattr(eng, "reduce_prefetch")
        wanted = list(wanted)
        if _remember and can_prefetch:
            if self._last_wanted is not None and wanted != self._last_wanted and self._graph_id is not None:
                eng.graph_destroy(self._graph_id)             # a different report: capture anew next time
                self._graph_id = None
            self._last_wanted = wanted
        requests = []
        for item in wanted:
            if item == "boundary":
                if self._self_done and (nat.DIR_SELF, nat.METRIC_D1) in self._totals:
                    continue                              # inherited with the origin cloud (with_reconst): nothing to reduce again
                if not self._self_done:
                    eng.nn(nat.DIR_SELF, self.nn_engine)
                    self._self_done = True
                requests.append((nat.DIR_SELF, nat.METRIC_D1))
                continue
            is_left, p2p = item
            direction = nat.DIR_LEFT if is_left else nat.DIR_RIGHT
            if not p2p:
                requests.append((direction, nat.METRIC_D1))
            else:
                other = 1 if is_left else 0
                try:
                    self._require_normals(other)
                except ValueError:
                    continue          # surfaces when the column is evaluated
                n_other = self._engine.n_iter(nat.DIR_RIGHT if other else nat.DIR_LEFT) if self._estimated[other] \
                    else len(self.clouds[other].normals)
                if self.normal_index == "row" and eng.n_iter(direction) > n_other:
                    continue      # row-indexed normals out of range (the WHOLE cloud decides, so that every rank of a
                    #               sharded pair agrees): surfaces, on every rank, where the reference raises
                requests.append((direction, nat.METRIC_D2))
        self._xchg_wanted = list(requests)
        if not can_prefetch:
            return
        if hasattr(eng, "reduce_prefetch_many"):
            eng.reduce_prefetch_many(requests[:8], self.normal_index)
        else:
            for direction, metric in requests:
                eng.reduce_prefetch(direction, metric, self.normal_index)

    # -- fused projection used by metric.ErrorVector ------------------------------------------------
    def point_to_plane_column(self, is_left: bool) -> DeviceColumn:
        self._require_normals(1 if is_left else 0)
        return DeviceColumn(self, nat.DIR_LEFT if is_left else nat.DIR_RIGHT, "proj")


def _has_normals(cloud) -> bool:
    has = getattr(cloud, "has_normals", None)
    if callable(has):
        return bool(has())
    nrm = getattr(cloud, "normals", None)
    return nrm is not None and len(nrm) > 0


def _has_colors(cloud) -> bool:
    has = getattr(cloud, "has_colors", None)
    if callable(has):
        return bool(has())
    col = getattr(cloud, "colors", None)
    return col is not None and len(col) > 0


def _host_rows(a) -> np.ndarray:
    if hasattr(a, "is_cuda"):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


def shard_plan(world: int, mode: str = "direction"):
    """Who searches what: ``plan[direction][rank] = (sub_rank, sub_world)`` -- rank ``rank`` is number ``sub_rank`` of the
    ``sub_world`` ranks that share the rows of ``direction`` (sub_world 0: it owns none of them).

    "direction" (default, world >= 2): the first half of the ranks takes the left direction (A's rows searched in B:
    they build B's grid only), the second half the right direction and the self search of A (both search A).  The
    replicated part of a step -- the search-structure build -- is halved that way and each half shards its rows; the
    all-gathers of materialised columns concatenate ranks in rank order, which is row order inside each half.
    "rows": every rank takes the same row range of every direction (round 1)."""
    if world <= 1 or mode == "rows":
        same = [(r, world) for r in range(world)]
        return {nat.DIR_LEFT: same, nat.DIR_RIGHT: list(same), nat.DIR_SELF: list(same)}
    nl = (world + 1) // 2
    nr = world - nl
    left = [(r, nl) if r < nl else (0, 0) for r in range(world)]
    right = [(r - nl, nr) if r >= nl else (0, 0) for r in range(world)]
    return {nat.DIR_LEFT: left, nat.DIR_RIGHT: right, nat.DIR_SELF: list(right)}


def _shard_bounds(n: int, rank: int, world: int):
    """Rows of an n-row cloud owned by ``rank``: the rule of pccm_set_shard (whole 8192-row chunks of NumPy's sum when every
    rank can have one, else 128-row leaves); world 0 owns nothing."""
    if world <= 0:
        return 0, 0
    unit = 8192 if n >= world * 8192 else 128
    units = (n + unit - 1) // unit
    return min(n, units * rank // world * unit), min(n, units * (rank + 1) // world * unit)
