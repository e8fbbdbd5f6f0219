/*
 * pccm.h -- C ABI of libpccm.so, the MI355X (gfx950) engine under open_pcc_metric_amd.
 *
 * The reference (aaletov/open-pcc-metric v0.1.2) is pure Python over the open3d wheel and has
 * no C ABI / FFI of its own; its "operator interface" for this path is the Python object
 * protocol of open_pcc_metric/cloud_pair.py.  Each entry point below names the reference
 * interface it stands under (paths relative to the reference checkout).  The Python binding a
 * maintainer would add is shown in INTEGRATION.md (ctypes, ~40 lines).
 *
 * Conventions
 *   - every function returns PCCM_OK (0) or a negative PCCM_E_* code; the message of the last
 *     failure on the calling thread is pccm_last_error().  No C++ exception crosses the ABI.
 *   - point/normal arrays are packed row-major [n][3], dtype PCCM_F32 or PCCM_F64, in host
 *     memory (on_device = 0) or device memory of the context's GPU (on_device = 1).  The
 *     library copies what it needs; callers keep ownership of everything they pass in or out.
 *   - cloud 0 = origin cloud ("A"), cloud 1 = reconstructed cloud ("B")  (cloud_pair.py:54-59).
 *   - direction PCCM_DIR_LEFT iterates A and searches B (cloud_pair.py:67-72), PCCM_DIR_RIGHT
 *     iterates B and searches A (cloud_pair.py:73-78), PCCM_DIR_SELF iterates A and searches A
 *     for the nearest point with a different row index (cloud_pair.py:108-109).
 *   - a context serves one caller at a time: every entry point holds the context's (recursive) mutex for its whole
 *     duration, so threads that share one are serialised, not corrupted; different contexts run concurrently.
 *   - one context drives one GPU.  Multi-GPU = one process (and context) per GPU, each with
 *     pccm_set_shard(rank, world); the only cross-rank data are the small vectors documented at
 *     pccm_reduce(), which the host exchanges with an RCCL all-reduce (DESIGN.md section e).
 */
#ifndef PCCM_H
#define PCCM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCCM_VERSION 100 /* 0.1.0 */

/* error codes */
#define PCCM_OK 0
#define PCCM_E_ARG (-1)    /* bad argument (null, size, non-finite or too large coordinates) -> ValueError */
#define PCCM_E_NODEV (-2)  /* no usable HIP device */
#define PCCM_E_HIP (-3)    /* HIP runtime error */
#define PCCM_E_OOM (-4)    /* device or host allocation failed */
#define PCCM_E_STATE (-5)  /* call order: clouds / normals / nn result not set yet */
#define PCCM_E_RANGE (-6)  /* row index outside the other cloud's normals: the reference's IndexError
                              at metric.py:148-152 */

/* dtypes */
#define PCCM_F32 0
#define PCCM_F64 1

/* directions */
#define PCCM_DIR_LEFT 0
#define PCCM_DIR_RIGHT 1
#define PCCM_DIR_SELF 2

/* nearest-neighbour engines (all exact; they differ only in speed) */
#define PCCM_ENGINE_AUTO 0  /* the grid, unless the pair of clouds is one no uniform grid can separate (clumps, partial overlap) */
#define PCCM_ENGINE_BRUTE 1 /* LDS-tiled fp32 scan + fp64 certification/refine */
#define PCCM_ENGINE_GRID 2  /* uniform-grid ring search */

/* which normal row the D2 projection uses */
#define PCCM_NORMAL_ROW 0       /* row i of the other cloud (what the reference does, metric.py:130,148-152) */
#define PCCM_NORMAL_NEIGHBOUR 1 /* row nn(i) of the other cloud */

/* per-point quantities */
#define PCCM_METRIC_D1 0   /* EuclideanDistance(point_to_plane=False): squared NN distance, metric.py:175-177 */
#define PCCM_METRIC_D2 1   /* EuclideanDistance(point_to_plane=True): projection squared, metric.py:179 */
#define PCCM_METRIC_PROJ 2 /* ErrorVector(point_to_plane=True): signed projection, metric.py:146-153 */

/* kernel classes for pccm_profile_get() */
#define PCCM_K_INGEST 0
#define PCCM_K_SCAN 1     /* brute-force fp32 scan (dominant kernel of PCCM_ENGINE_BRUTE) */
#define PCCM_K_REFINE 2   /* fp64 certification + winner refine */
#define PCCM_K_FALLBACK 3 /* exact rescan of flagged queries (uncertified fp32 winners; queries the grid's rings left open) */
#define PCCM_K_POINT 4    /* fused gather + error vector + projection */
#define PCCM_K_REDUCE 5   /* leaf sums / max / min */
#define PCCM_K_GRID_BUILD 6
#define PCCM_K_GRID_QUERY 7  /* k_grid_query_coop: ring 1 of both directions (dominant kernel of PCCM_ENGINE_GRID) */
#define PCCM_K_GRID_FINISH 8 /* k_grid_tail: rings 2..3 of the unsettled queries + the exact rescan of what they leave */
#define PCCM_K_COUNT 9

typedef struct pccm_ctx pccm_ctx;

int pccm_version(void);
const char *pccm_last_error(void);

/* Number of HIP devices visible to the process (0 and PCCM_OK when there is none). */
int pccm_device_count(int *n);

/* Create a context on `device`.  `hip_stream` may be NULL (the library creates its own
 * non-blocking stream) or a hipStream_t the caller owns (e.g. torch's current stream). */
int pccm_ctx_create(int device, void *hip_stream, pccm_ctx **out);
int pccm_ctx_destroy(pccm_ctx *ctx);
/* Back to the state after pccm_ctx_create -- no clouds, shard 0 of 1, no graphs, profiling off -- but with every
 * device allocation (and the grid decisions a look-alike pair may inherit) kept: for callers that run one pair
 * after the other through the same context instead of paying context teardown and fresh allocations per pair. */
int pccm_ctx_reset(pccm_ctx *ctx);

/* Replaces CloudPair.__init__'s capture of the two clouds, cloud_pair.py:54-59.
 * Coordinates must be finite with |x| <= 1e15.  Invalidates earlier nn results. */
int pccm_set_cloud(pccm_ctx *ctx, int which, const void *xyz, int64_t n, int dtype, int on_device);

/* Replaces np.asarray(cloud.normals), metric.py:92-98.  n must equal the cloud's point count
 * for PCCM_NORMAL_NEIGHBOUR; PCCM_NORMAL_ROW only needs the rows it indexes. */
int pccm_set_normals(pccm_ctx *ctx, int which, const void *nrm, int64_t n, int dtype, int on_device);

/* The same for HOST normals, announced now and uploaded later: the library notes the pointer (the array must stay alive and
 * unchanged until pccm_flush_uploads returns or another call replaces the cloud or its normals) and moves the data when a
 * call needs it -- or at pccm_flush_uploads -- on a copy stream of its own.  Searches whose results are matched records never
 * read normals, so a caller that announces, starts the searches and then flushes has the upload run beside them:
 * handler.py:57-58 of the reference reads both files, then cloud_pair.py:61-80 does everything else; nothing in it orders the
 * normals before the searches.  Non-finite normals are reported by the call that uploads them (PCCM_E_ARG), not by this one. */
int pccm_set_normals_deferred(pccm_ctx *ctx, int which, const void *nrm, int64_t n, int dtype);
int pccm_flush_uploads(pccm_ctx *ctx);

/* How arrays of the CALLER cross PCIe (uploads of clouds, normals, colours; downloads of rows, distances, normals ...).
 * off (default; pccm_ctx_reset restores it): handed to the HIP runtime as they are -- which pins the caller's pages and keeps the
 * mapping cached: fastest (43 GB/s), but when the caller later FREES such an array the driver evicts and restores every GPU
 * queue of the process, and a kernel that is running stands still for 13-27 ms (measured).
 * on: through pinned buffers of the context's own, copied by a few host threads (25 GB/s): for callers that load, use and
 * free their clouds in a loop (CloudPair.with_reconst, evaluate_pairs with loader items, the command line with several
 * --pcloud).  No counterpart in the reference. */
int pccm_set_io_staged(pccm_ctx *ctx, int on);

/* Replaces clouds[k].estimate_normals(), cloud_pair.py:61-64 (Open3D EstimateNormals, default
 * KDTreeSearchParamKNN(knn = 30)): per point, the eigenvector of the smallest eigenvalue of the
 * covariance of its knn nearest points of the same cloud (itself included).  Open3D's own arithmetic
 * and sign convention cannot be pinned without it; here the component of largest
 * magnitude is positive.  The normals stay on the device; pccm_get_normals copies [n][3] doubles out. */
int pccm_estimate_normals(pccm_ctx *ctx, int which, int knn);
int pccm_get_normals(pccm_ctx *ctx, int which, double *out);

/* Query-axis shard of this context: rank r of `world` owns, in every direction, the rows
 * [begin, end) of the iterating cloud returned by pccm_shard_range (boundaries are multiples
 * of 8192 rows -- whole chunks of NumPy's sum: pccm_reduce_chunks_many -- when the cloud has a chunk for every rank,
 * else of 128 rows, so that reduction leaves never straddle ranks).  Default: rank 0 of 1. */
int pccm_set_shard(pccm_ctx *ctx, int rank, int world);
/* The same per direction: `world` ranks share the rows of direction `dir` and this context is number `rank` of them;
 * world = 0: this context owns NO rows of that direction (another group of ranks searches it) -- its searches and
 * reductions of `dir` are empty and contribute zeros to the exchange.  Lets the ranks of a node split by DIRECTION
 * first (cloud_pair.py:67-72 on one half, :73-78 on the other), so that every rank builds the search structure of one
 * cloud only. */
int pccm_set_shard_dir(pccm_ctx *ctx, int dir, int rank, int world);
int pccm_shard_range(pccm_ctx *ctx, int dir, int64_t *begin, int64_t *end);

/* Replaces get_neighbour_cloud(), cloud_pair.py:10-42 (and, for PCCM_DIR_SELF, Open3D's
 * compute_nearest_neighbor_distance behind cloud_pair.py:108-109): exact 1-NN of every row
 * of the shard, squared L2 distance d2 = ((dx*dx)+(dy*dy))+(dz*dz) in fp64, exact ties to the
 * smallest row index.  Asynchronous on the context's stream; results stay on the device. */
int pccm_nn(pccm_ctx *ctx, int dir, int engine);

/* PCCM_DIR_LEFT and PCCM_DIR_RIGHT together -- what CloudPair.__init__ does at cloud_pair.py:67-78.
 * Same results as two pccm_nn() calls; the grid engine fuses both directions into the same launches. */
int pccm_nn_pair(pccm_ctx *ctx, int engine);

/* Fuse the point-to-plane projection of direction `dir` (0 or 1) into the search: every settled query then also
 * leaves err . normal_other[row] (normal_mode PCCM_NORMAL_ROW: row i of the searched cloud's normals, what
 * metric.py:146-153 of the reference computes; PCCM_NORMAL_NEIGHBOUR: row nn(i)), so that the D2 reductions
 * (metric.py:179, 226-228, 366) need no second pass over the points.  normal_mode -1 switches it off.  A request
 * that cannot be honoured at search time (no normals set, row-indexed normals shorter than the iterating cloud)
 * is ignored; the reductions then take the separate pass and report the reference's IndexError (PCCM_E_RANGE).
 * Purely an optimisation: results are bit-identical either way.  Takes effect at the next pccm_nn / pccm_nn_pair. */
int pccm_nn_fuse(pccm_ctx *ctx, int dir, int normal_mode);

/* Whether the searches keep the matched row of every point (default: on).  Off, a search leaves 16 bytes per point
 * (squared distance + fused projection) instead of 32 -- half the scattered stores, and columns the reductions read
 * densely -- which is all that GeoMSE / GeoPSNR / Hausdorff need (metric.py:213-247, 353-386).  Whoever needs the rows
 * later (pccm_nn_fetch with idx, pccm_error_vectors, the colour calls: cloud_pair.py:34-42, 90-100, 120-124) gets them
 * anyway: the library repeats the search of that direction with the rows on.  Purely an optimisation. */
int pccm_nn_want_idx(pccm_ctx *ctx, int on);

/* Copy the shard's results to the host (either pointer may be NULL).  idx[i] is the row in
 * the searched cloud, d2[i] the squared distance: the (idxs, sqrdists) of cloud_pair.py:32-33
 * and the value behind get_left/right_neighbour_distances(), cloud_pair.py:102-106. */
int pccm_nn_fetch(pccm_ctx *ctx, int dir, int32_t *idx, double *d2);

/* get_left/right_error_vector(), cloud_pair.py:90-100: out[i][:] = iter[i] - search[nn(i)]. */
int pccm_error_vectors(pccm_ctx *ctx, int dir, double *out);

/* Diagnostic (opt-in, not on the report's path): how much of the point-to-plane result hangs on the ORDER OF EXACT TIES.
 * get_neighbour_cloud() keeps idx[-1] of a one-neighbour nanoflann search (cloud_pair.py:22-23) -- whichever of several
 * equidistant nearest points the tree meets; this library keeps the smallest row.  D1 is the same either way; the projection
 * err . normal (metric.py:146-153) is not.  For the shard's rows of direction `dir` (0 or 1; the search must have run):
 *   out[0] queries, out[1] queries with >= 2 equidistant nearest neighbours, out[2] / out[3] the sum over the queries of the
 *   SMALLEST / LARGEST squared projection over all their nearest neighbours, out[4] the same sum for the library's own picks,
 *   out[5] queries whose tie set was not enumerated (outliers far from the searched cloud: counted with their pick alone),
 *   out[6] the largest tie multiplicity seen.
 * out[2] / n_iter <= any admissible D2 MSE (the reference's included) <= out[3] / n_iter; tie-free data: out[1] = 0 and
 * out[2] = out[3] = out[4].  normal_mode -1: counts only (no normals needed).  Sums are plain fp64 accumulations. */
int pccm_tie_exposure(pccm_ctx *ctx, int dir, int normal_mode, double out[8]);

/* Per-point metric vector of the shard (PCCM_METRIC_*), metric.py:124-179. */
int pccm_point_metric(pccm_ctx *ctx, int dir, int metric, int normal_mode, double *out);

/* Fused reduction of a per-point metric over the shard: the np.sum / np.max of
 * GeoMSE.calculate (metric.py:226-228), GeoHausdorffDistance.calculate (metric.py:366) and
 * the np.min / np.max of BoundarySqrtDistances (metric.py:187-188; apply sqrt to both).
 *
 *   xvec    [pccm_xvec_len(n_iter)] doubles, zero except for this shard's entries:
 *           first 64 * (n_iter / 8192) sums of aligned 128-row leaves, accumulated exactly as
 *           NumPy's pairwise sum does, then the (n_iter % 8192) raw values of the last,
 *           partial 8192-row chunk.  Summing the xvecs of all ranks element-wise (RCCL
 *           all-reduce; x + 0 is exact) gives the full vector; pccm_finish_sum() then
 *           returns bit for bit what np.sum of the whole per-point array returns.
 *   minmax  [2]: min and max over the shard (+inf / -inf for an empty shard).
 *
 * pccm_reduce_prefetch() only enqueues the kernels and the copy of the small result arrays into
 * pinned host memory (no host wait); a later pccm_reduce() with the same arguments consumes it.
 * Prefetching every column a report needs right after pccm_nn() makes the host wait once per pair.
 */
int64_t pccm_xvec_len(int64_t n_iter);
int pccm_reduce_prefetch(pccm_ctx *ctx, int dir, int metric, int normal_mode);
/* n <= 8 columns at once: one kernel evaluates every point-to-plane column, one kernel reduces all of them. */
int pccm_reduce_prefetch_many(pccm_ctx *ctx, int n, const int *dirs, const int *metrics, const int *normal_modes);
int pccm_reduce(pccm_ctx *ctx, int dir, int metric, int normal_mode, double *xvec, double *minmax);
int pccm_finish_sum(const double *xvec, int64_t n_iter, double *sum);
/* Unsharded shortcut (world = 1): out = {np.sum, np.min, np.max} of the whole column in one call. */
int pccm_reduce_total(pccm_ctx *ctx, int dir, int metric, int normal_mode, double out[3]);
/* The same for up to 8 columns in one call (out[k][3]): one wait for the GPU and one trip through the FFI per report instead
 * of one per column -- the np.sum / np.max of every GeoMSE / GeoHausdorffDistance row (metric.py:226-228, 366). */
int pccm_reduce_total_many(pccm_ctx *ctx, int n, const int *dirs, const int *metrics, const int *normal_modes, double *out);

/* Sharded contexts whose rows start and end on whole 8192-row chunks (pccm_set_shard / pccm_set_shard_dir do that
 * whenever every rank can have a chunk): the exchange vector shrinks to ONE number per chunk -- NumPy adds the
 * chunks of a column one after the other, and the GPU has finished each chunk's pairwise tree -- plus the raw values
 * of the last, partial chunk.  cvecs: the columns' vectors one after the other, pccm_cvec_len(n_iter) doubles each,
 * zero except for this shard's entries (SUM them over the ranks); minmax[2k], [2k+1]: this shard's extrema of column k.
 * pccm_finish_chunks() gives np.sum of the whole column from a summed vector.  PCCM_E_STATE when the shard is not
 * chunk-aligned (then use pccm_reduce / pccm_finish_sum).  Stands under metric.py:226-228, 366 like pccm_reduce. */
int64_t pccm_cvec_len(int64_t n_iter);
int pccm_reduce_chunks_many(pccm_ctx *ctx, int n, const int *dirs, const int *metrics, const int *normal_modes, double *cvecs,
                            double *minmax);
int pccm_finish_chunks(const double *cvec, int64_t n_iter, double *sum);

/* Colours of cloud `which` ([n][3] RGB in [0, 1] as Open3D holds them; n = the cloud's point count):
 * replaces np.asarray(cloud.colors) behind get_left/right_colors(), cloud_pair.py:114-118. */
int pccm_set_colors(pccm_ctx *ctx, int which, const void *rgb, int64_t n, int dtype, int on_device);

/* The same from the uchar colours point-cloud files hold: rgb[n][3] bytes, widened on the device as k / 255.0 -- the
 * division o3d.io.read_point_cloud (and io.py) perform on the host, bit for bit. */
int pccm_set_colors_u8(pccm_ctx *ctx, int which, const unsigned char *rgb, int64_t n);

/* Colour metrics of one direction on the device, metric.py:302-333 and :389-427.  Per row i of the
 * iterating cloud: own = T(rgb_own[i]), other = T(rgb_other[nn(i)]) (the gather of
 * get_left/right_neighbour_colors(), cloud_pair.py:120-124; T = transform_colors(), metric.py:261-290,
 * scheme 0 "rgb" | 1 "ycc" | 2 "yuv"), sq = (scale * (own - other))^2.
 *   sum_out[c] = np.add.reduce(sq, axis=0)[c]  -- bit for bit: the left-to-right row order NumPy uses for
 *                axis 0 (ColorMSE = sum / n, i.e. np.mean(diff**2, axis=0));
 *   max_out[c] = np.max(sq, axis=0)[c]          (ColorHausdorffDistance; scale = 255 for "rgb", metric.py:422-425).
 * `rows`: NULL = the neighbour rows of the context's own search of `dir` (which must cover the whole cloud);
 * otherwise `nrows` = n_iter host rows (sharded searches: the ranks' slices gathered by the caller). */
int pccm_color_reduce(pccm_ctx *ctx, int dir, int scheme, double scale, const int32_t *rows, int64_t nrows,
                      double sum_out[3], double max_out[3]);

/* The same rows materialised to the host, out[n_iter][3]: what = 0 own colours in the scheme,
 * 1 neighbour colours in the scheme, 2 scale * (own - other), 3 its square. */
int pccm_color_rows(pccm_ctx *ctx, int dir, int scheme, double scale, int what, const int32_t *rows, int64_t nrows,
                    double *out);

/* The frame search behind CloudPair.get_extent() (cloud_pair.py:111-112, Open3D's minimal oriented bounding box):
 * verts = the nv vertices of the convex hull of cloud A, tri = its nt triangles as vertex coordinates [nt][3][3]
 * (the hull itself is Qhull on the host, as in Open3D).  For every triangle the axis-aligned extents of the hull
 * vertices in the triangle's frame (x along its first edge, z along its normal) are evaluated on the device;
 * ext_out = the extents of the frame with the smallest volume (first one on ties), *vol_out its volume. */
int pccm_obb_frames(pccm_ctx *ctx, const double *verts, int64_t nv, const double *tri, int64_t nt, double ext_out[3], double *vol_out);

/* Thinning cloud `which` before the host's Qhull run (get_minimal_oriented_bounding_box, cloud_pair.py:111-112):
 * pccm_extreme_rows  rows_out[k] = row of (about) the farthest point of the cloud along dirs[k] (ndirs <= 1024, fp32);
 * pccm_rows_outside  the rows (unordered) of all points x with n.x + off > -margin for some plane (n, off) of
 *                    planes[nplanes][4] -- Qhull's convention: n.x + off <= 0 inside.  With the facets of the hull of
 *                    the extreme points as planes, every point NOT reported lies strictly inside the hull of other
 *                    points of the cloud and cannot be a vertex of the cloud's hull.  rows_out must hold n rows. */
int pccm_extreme_rows(pccm_ctx *ctx, int which, const float *dirs, int ndirs, int32_t *rows_out);
int pccm_rows_outside(pccm_ctx *ctx, int which, const double *planes, int nplanes, double margin, int32_t *rows_out, int64_t *count);

/* Utility behind pccm_color_reduce: out[c] = left-to-right fp64 sum of column c of three non-negative
 * host columns cols[3][n] -- what np.add.reduce(a, axis=0) returns for the (n, 3) array a = cols.T --
 * evaluated on the device without the dependent chain (csrc/pccm_color.hip). */
int pccm_seq_colsum(pccm_ctx *ctx, const double *cols, int64_t n, double out[3]);

/* Host-only helper (no GPU) of the PCD reader that stands in for o3d.io.read_point_cloud (handler.py:57):
 * liblzf decompression of a binary_compressed body.  *out_len = bytes written (<= out_cap). */
int pccm_lzf_decompress(const unsigned char *in, int64_t in_len, unsigned char *out, int64_t out_cap, int64_t *out_len);

/* Host-only helper (no GPU): rows of RGB in [n][3] -> the target scheme of transform_colors(),
 * metric.py:261-290 (scheme 1 = "ycc", 2 = "yuv"), bit-compatible with the reference's per-row np.matmul. */
int pccm_color_transform(const double *rgb, int64_t n, int scheme, double *out);

/* Forget the search structures derived from the clouds (the grid engine's cell-sorted copies,
 * the analogue of the KD-trees CloudPair.__init__ builds at cloud_pair.py:65), so that the next
 * pccm_nn() rebuilds them.  bench.py calls it every step: a step pays for its builds. */
int pccm_drop_caches(pccm_ctx *ctx);

/* hipGraph capture.  Between pccm_graph_begin() and pccm_graph_end() only pccm_drop_caches(),
 * pccm_nn() and pccm_reduce_prefetch() may be called; they are recorded on the context's stream
 * instead of executed.  The same sequence must have run once before (capture cannot allocate).
 * pccm_graph_end() instantiates the graph, runs it once and returns its id; pccm_graph_launch()
 * replays the whole sequence -- kernels and host-side bookkeeping -- with a single launch, after which
 * pccm_reduce()/pccm_nn_fetch() read the fresh results.  A graph goes stale (PCCM_E_STATE) when
 * clouds, normals, shard or any buffer it references change.  One report over resident clouds is
 * ~45 small launches, which an eager host cannot issue as fast as the GPU retires them. */
int pccm_graph_begin(pccm_ctx *ctx);
int pccm_graph_end(pccm_ctx *ctx, int *graph_id);
int pccm_graph_launch(pccm_ctx *ctx, int graph_id);
int pccm_graph_destroy(pccm_ctx *ctx, int graph_id);

/* Wait for everything queued on the context's stream. */
int pccm_sync(pccm_ctx *ctx);

/* HIP-event timing of kernel classes on the context's stream (for bench.py's roofline). */
int pccm_profile_enable(pccm_ctx *ctx, int on);
int pccm_profile_reset(pccm_ctx *ctx);
int pccm_profile_get(pccm_ctx *ctx, int kernel_class, double *ms_total, int64_t *launches);

/* Bookkeeping of the last pccm_nn() in `dir`: out[0] = queries sent to the exact fallback
 * rescan, out[1] = ref-axis splits of the brute-force scan / number of cells of the grid the
 * grid engine searched, out[2] = (query, ref) pairs evaluated by the brute-force scan (grid: 0).
 * `dir | PCCM_STATS_TAIL`: out[0] = queries the grid engine's ring-1 kernel left to the tail launch, out[1] = out[2] = 0. */
#define PCCM_STATS_TAIL 0x10
int pccm_nn_stats(pccm_ctx *ctx, int dir, int64_t out[3]);

#ifdef __cplusplus
}
#endif
#endif /* PCCM_H */
