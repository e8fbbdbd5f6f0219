"""Test double with the interface of ``open_pcc_metric_amd._native.Engine`` backed by the CPU
oracle.  TEST INFRASTRUCTURE: lets the host logic (device columns, metric DAG, sharding and the
cross-rank exchange) run in the CPU-only suite.  The product never constructs it."""
import numpy as np

from open_pcc_metric_amd import _native as nat
from oracle import oracle as orc


def _signed_rows(rgb, _zeros, _idx, scheme):
    """T(rgb) row by row exactly as the reference evaluates it: np.matmul(M, c) (metric.py:283-290)."""
    if scheme == "rgb":
        return np.array(rgb, dtype=np.float64)
    from open_pcc_metric_amd.metric import _FROM_RGB
    m = _FROM_RGB[scheme]
    return np.array([np.matmul(m, c) for c in np.asarray(rgb, dtype=np.float64)]).reshape(-1, 3)


class OracleEngine:
    def __init__(self, method="auto"):
        self.pts = [None, None]
        self.nrm = [None, None]
        self.rgb = [None, None]
        self.rank, self.world = 0, 1
        self.dir_shard = {}            # direction -> (rank, world) set by set_shard_dir (world 0: owns nothing)
        self.res = {}
        self.method = method
        self.calls = []

    # inputs
    def set_cloud(self, which, points):
        p = np.asarray(points, dtype=np.float64)
        if p.ndim != 2 or p.shape[1] != 3 or p.shape[0] == 0:
            raise ValueError("empty cloud")
        self.pts[which] = np.ascontiguousarray(p)
        self.res.clear()

    def set_normals(self, which, normals):
        self.nrm[which] = np.ascontiguousarray(np.asarray(normals, dtype=np.float64))

    def set_colors(self, which, colors):
        self.rgb[which] = np.ascontiguousarray(np.asarray(colors, dtype=np.float64))

    def _colour_operands(self, d, rows):
        it, se = self._clouds(d)
        if rows is None:
            assert self.world == 1, "sharded search: the caller passes the gathered rows"
            rows = self.res[d][0]
        return self.rgb[it], self.rgb[se], np.asarray(rows, dtype=np.int64)

    def color_reduce(self, d, scheme, scale=1.0, rows=None):
        own, other, rows = self._colour_operands(d, rows)
        _, sums, maxs = orc.color_columns(own, other, rows, scheme, scale)
        self.calls.append(("color_reduce", d, scheme, scale))
        return sums, maxs

    def color_rows(self, d, scheme, what, scale=1.0, rows=None):
        own, other, rows = self._colour_operands(d, rows)
        if what == nat.COLOR_SQUARE:
            return orc.color_columns(own, other, rows, scheme, scale)[0]
        zeros = np.zeros((1, 3))
        if what == nat.COLOR_OWN:         # T(own) = sqrt-free: difference against a black row, scale 1
            return _signed_rows(own, zeros, np.zeros(len(own), np.int64), scheme)
        if what == nat.COLOR_NEIGHBOUR:
            return _signed_rows(other[rows], zeros, np.zeros(len(rows), np.int64), scheme)
        return scale * (_signed_rows(own, zeros, np.zeros(len(own), np.int64), scheme)
                        - _signed_rows(other[rows], zeros, np.zeros(len(rows), np.int64), scheme))

    def extreme_rows(self, which, directions):
        return np.argmax(self.pts[which] @ np.asarray(directions, dtype=np.float64).T, axis=0).astype(np.int32)

    def rows_outside(self, which, planes, margin):
        p = np.asarray(planes, dtype=np.float64)
        val = self.pts[which] @ p[:, :3].T + p[:, 3]
        return np.nonzero(np.any(val > -margin, axis=1))[0].astype(np.int32)

    def obb_frames(self, hull_vertices, hull_triangles):
        best_vol, best_ext = np.inf, None
        for a, b, c in np.asarray(hull_triangles, dtype=np.float64):
            u, v = b - a, c - a
            w = np.cross(u, v)
            v = np.cross(w, u)
            with np.errstate(invalid="ignore", divide="ignore"):
                frame = np.stack([x / np.sqrt(np.sum(x * x)) for x in (u, v, w)])
            loc = (np.asarray(hull_vertices) - a) @ frame.T
            ext = loc.max(axis=0) - loc.min(axis=0)
            vol = ext.prod()
            if np.isfinite(vol) and vol < best_vol:
                best_vol, best_ext = float(vol), ext
        return best_ext, best_vol

    def set_shard(self, rank, world):
        self.rank, self.world = rank, world
        self.dir_shard = {}
        self.res.clear()

    def set_shard_dir(self, d, rank, world):
        """pccm_set_shard_dir: per-direction row ownership (world 0: this rank owns no rows of d)."""
        self.dir_shard[d] = (rank, world)
        if world != 1:
            self.world = max(self.world, 2)        # "sharded" for the checks that only ask whether there are peers
        self.res.clear()

    def n_iter(self, d):
        return self.pts[1 if d == nat.DIR_RIGHT else 0].shape[0]

    def shard_range(self, d):
        n = self.n_iter(d)
        rank, world = self.dir_shard.get(d, (self.rank, self.world))
        from open_pcc_metric_amd.cloud_pair import _shard_bounds        # the library's rule (chunks when possible, else leaves)
        return _shard_bounds(n, rank, world)

    def _clouds(self, d):
        return {nat.DIR_LEFT: (0, 1), nat.DIR_RIGHT: (1, 0), nat.DIR_SELF: (0, 0)}[d]

    # nn
    def nn(self, d, engine="auto"):
        it, se = self._clouds(d)
        b, e = self.shard_range(d)
        q = self.pts[it][b:e]
        self.calls.append(("nn", d))
        if e == b:                                     # no rows of this direction on this rank
            self.res[d] = (np.zeros(0, np.int64), np.zeros(0))
            return
        if d == nat.DIR_SELF:
            if self.pts[0].shape[0] < 2:
                self.res[d] = (np.full(e - b, -1, np.int64), np.zeros(e - b))
                return
            # skip_same_index works on global rows: search with the full cloud, then slice
            idx, d2 = orc.nn(self.pts[0], self.pts[0], skip_same_index=True, method=self.method)
            self.res[d] = (idx[b:e], d2[b:e])
        else:
            self.res[d] = orc.nn(q, self.pts[se], method=self.method)

    def fetch_nn(self, d, want_idx=True, want_d2=True):
        idx, d2 = self.res[d]
        return (idx.astype(np.int32) if want_idx else None), (d2.copy() if want_d2 else None)

    def error_vectors(self, d):
        it, se = self._clouds(d)
        b, e = self.shard_range(d)
        return self.pts[it][b:e] - self.pts[se][self.res[d][0]]

    def point_metric(self, d, metric, normal_mode="row"):
        it, se = self._clouds(d)
        b, e = self.shard_range(d)
        idx, d2 = self.res[d]
        if metric == nat.METRIC_D1:
            return d2.copy()
        nrm = self.nrm[se]
        if nrm is None:
            raise RuntimeError("no normals")
        if normal_mode == "row":
            # as libpccm's check_normals: sharded, the WHOLE iterating cloud decides, so that every rank raises
            if (self.n_iter(d) if self.world > 1 else e) > nrm.shape[0]:
                raise IndexError(f"index {nrm.shape[0]} is out of bounds for axis 0 with size {nrm.shape[0]}")
            rows = nrm[b:e]
            proj = orc.point_to_plane(self.pts[it][b:e], self.pts[se], idx, np.ascontiguousarray(rows))
        else:
            proj = orc.point_to_plane(self.pts[it][b:e], self.pts[se], idx, nrm, normal_index="neighbour")
        return proj if metric == nat.METRIC_PROJ else np.square(proj)

    def reduce(self, d, metric, normal_mode="row"):
        col = self.point_metric(d, metric, normal_mode)
        n = self.n_iter(d)
        b, e = self.shard_range(d)
        xvec = np.zeros(nat.xvec_len(n))
        nfull = n // 8192
        full_rows = nfull * 8192
        for row in range(b, min(e, full_rows), 128):
            xvec[row // 128] = np.sum(col[row - b:row - b + 128])     # one NumPy pairwise leaf
        t0 = max(b, full_rows)
        if t0 < e:
            xvec[nfull * 64 + (t0 - full_rows):nfull * 64 + (e - full_rows)] = col[t0 - b:]
        mn = np.min(col) if len(col) else np.inf
        mx = np.max(col) if len(col) else -np.inf
        self.calls.append(("reduce", d, metric))
        return xvec, mn, mx

    finish_sum = staticmethod(nat.finish_sum)

    def reduce_chunks_many(self, requests, normal_mode="row"):
        """pccm_reduce_chunks_many: one number per owned 8192-row chunk (np.sum of the chunk = NumPy's pairwise tree over it)
        + the raw values of the partial last chunk."""
        bufs, lens, mms = [], [], []
        for d, metric in requests:
            col = self.point_metric(d, metric, normal_mode)
            n = self.n_iter(d)
            b, e = self.shard_range(d)
            nfull, tail = n // 8192, n % 8192
            cvec = np.zeros(nfull + tail)
            if e > b:
                assert b % 8192 == 0 and (e % 8192 == 0 or e == n), "shard is not chunk-aligned"
                for row in range(b, min(e, nfull * 8192), 8192):
                    cvec[row // 8192] = np.sum(col[row - b:row - b + 8192])
                t0 = max(b, nfull * 8192)
                if t0 < e:
                    cvec[nfull + (t0 - nfull * 8192):nfull + (e - nfull * 8192)] = col[t0 - b:]
            bufs.append(cvec); lens.append(len(cvec))
            mms.append((np.min(col) if len(col) else np.inf, np.max(col) if len(col) else -np.inf))
            self.calls.append(("reduce_chunks", d, metric))
        return np.concatenate(bufs) if bufs else np.zeros(0), lens, mms

    @staticmethod
    def finish_chunks(cvec, n):
        nfull = n // 8192
        s = None
        for c in range(nfull):
            s = cvec[c] if s is None else s + cvec[c]
        if n % 8192:
            ts = np.sum(np.ascontiguousarray(cvec[nfull:]))
            s = ts if s is None else s + ts
        return np.float64(0.0 if s is None else s)

    def sync(self):
        pass

    def close(self):
        pass
