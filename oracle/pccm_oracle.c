/*
 * pccm_oracle.c -- CPU restatement (plain C, fp64) of the open-pcc-metric hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / CPU baseline.  The product path (open_pcc_metric_amd +
 * libpccm.so) never links, imports or falls back to this file.
 *
 * What is restated (reference = aaletov/open-pcc-metric v0.1.2, paths relative to
 * /root/reference):
 *   - exact 1-NN of every point of an "iterating" cloud in a "search" cloud, squared L2
 *     in fp64: open_pcc_metric/cloud_pair.py:10-42 (get_neighbour_cloud, k = n+1 = 1,
 *     keeps idx[-1], dists[-1]) and its two call sites cloud_pair.py:67-78.
 *   - the self search behind get_boundary_sqrt_distances, cloud_pair.py:108-109
 *     (Open3D compute_nearest_neighbor_distance: k = 2 search of the cloud in itself,
 *     sqrt of the 2nd hit; == nearest point with a different row index).
 *   - the D2 point-to-plane projection loop, open_pcc_metric/metric.py:146-153.
 *
 * The arithmetic of the kNN itself lives in a third-party dependency that is NOT in
 * /root/reference: open3d==0.18.0 (requirements.txt:33, pyproject.toml:12), whose
 * KDTreeFlann wraps nanoflann's L2 adaptor over double.  Its published algorithm is
 * restated here: d2(q, p) = ((dx*dx) + (dy*dy)) + (dz*dz), accumulated in that order in
 * fp64 with separately rounded multiplies and adds (no FMA); the neighbour is the point
 * of minimal d2.  Exact-tie order in nanoflann depends on tree traversal and cannot be
 * pinned without Open3D: this oracle (and the product) break exact ties towards the
 * SMALLEST row index.  d2, MSE and Hausdorff are tie-invariant; indices, error vectors
 * and D2 on exactly tied data are not (see DESIGN.md "parity pins").
 *
 * Two search structures are provided so they can check each other:
 *   orc_nn_brute   O(Nq*Nr) scan, the definition itself (small cases)
 *   orc_kdtree_*   exact KD-tree with conservative pruning (large cases, CPU baseline)
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* squared L2 in nanoflann's accumulation order (dim 0, 1, 2; result += diff*diff). */
static inline double orc_d2(const double *q, const double *p)
{
    double dx = q[0] - p[0];
    double dy = q[1] - p[1];
    double dz = q[2] - p[2];
    double r = dx * dx;
    r = r + dy * dy;
    r = r + dz * dz;
    return r;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------
 * Brute force: the definition.  cloud_pair.py:22-23 for every row of iter_cloud.
 * skip_same_index != 0 restates the self search of cloud_pair.py:108-109.
 * idx[i] = -1 and d2[i] = 0 when no candidate exists (Open3D returns 0 for clouds of
 * fewer than 2 points).
 * ---------------------------------------------------------------------------------- */
int orc_nn_brute(const double *q, int64_t nq, const double *r, int64_t nr,
                 int skip_same_index, int64_t *idx, double *d2)
{
    if (nq < 0 || nr < 0) return -1;
    for (int64_t i = 0; i < nq; ++i) {
        double best = INFINITY;
        int64_t bi = -1;
        for (int64_t j = 0; j < nr; ++j) {
            if (skip_same_index && j == i) continue;
            double d = orc_d2(q + 3 * i, r + 3 * j);
            if (d < best) { best = d; bi = j; }   /* strict <: smallest index wins ties */
        }
        if (bi < 0) best = 0.0;
        if (idx) idx[i] = bi;
        if (d2) d2[i] = best;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Exact KD-tree.  Balanced median split on the widest axis, leaves of <= ORC_LEAF points.
 * Pruning uses only the single-axis plane distance, evaluated with the same roundings as
 * orc_d2, and descends into the far side when plane_d2 <= best: because rounding is
 * monotonic, fl(d2(q,p)) >= fl((q_a - s)^2) for every p behind the plane, so no candidate
 * with d2 <= best (equal included, for the smallest-index tie rule) is ever skipped.
 * ---------------------------------------------------------------------------------- */
#define ORC_LEAF 12

typedef struct {
    int32_t dim;      /* -1 = leaf */
    double split;
    int64_t lo, hi;   /* range in perm[] */
    int64_t left, right;
} orc_node;

typedef struct orc_kdtree {
    const double *pts;   /* borrowed: [n][3] */
    double *sorted;      /* points gathered in perm order, [n][3] */
    int64_t *perm;
    int64_t n;
    orc_node *nodes;
    int64_t nnodes, cap;
} orc_kdtree;

static void orc_swap(int64_t *a, int64_t *b) { int64_t t = *a; *a = *b; *b = t; }

/* quickselect on perm[lo,hi) by coordinate dim so that perm[k] is in sorted position */
static void orc_select(const double *pts, int64_t *perm, int64_t lo, int64_t hi, int64_t k, int dim)
{
    while (hi - lo > 1) {
        int64_t mid = lo + (hi - lo) / 2;
        /* median of three */
        double a = pts[3 * perm[lo] + dim], b = pts[3 * perm[mid] + dim], c = pts[3 * perm[hi - 1] + dim];
        int64_t pi = (a < b) ? ((b < c) ? mid : ((a < c) ? hi - 1 : lo)) : ((a < c) ? lo : ((b < c) ? hi - 1 : mid));
        double pv = pts[3 * perm[pi] + dim];
        /* three-way partition: [lo,lt) < pv, [lt,gt) == pv, [gt,hi) > pv */
        int64_t lt = lo, gt = hi, i = lo;
        while (i < gt) {
            double v = pts[3 * perm[i] + dim];
            if (v < pv) { orc_swap(&perm[i], &perm[lt]); ++lt; ++i; }
            else if (v > pv) { --gt; orc_swap(&perm[i], &perm[gt]); }
            else ++i;
        }
        if (k < lt) hi = lt;
        else if (k >= gt) lo = gt;
        else return;
    }
}

static int64_t orc_new_node(orc_kdtree *t)
{
    if (t->nnodes == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->nodes = (orc_node *)realloc(t->nodes, (size_t)t->cap * sizeof(orc_node));
    }
    return t->nnodes++;
}

static int64_t orc_build(orc_kdtree *t, int64_t lo, int64_t hi)
{
    int64_t id = orc_new_node(t);
    orc_node nd;
    nd.lo = lo; nd.hi = hi; nd.left = nd.right = -1; nd.dim = -1; nd.split = 0.0;
    if (hi - lo > ORC_LEAF) {
        double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int64_t i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) {
                double v = t->pts[3 * t->perm[i] + a];
                if (v < mn[a]) mn[a] = v;
                if (v > mx[a]) mx[a] = v;
            }
        int dim = 0;
        if (mx[1] - mn[1] > mx[dim] - mn[dim]) dim = 1;
        if (mx[2] - mn[2] > mx[dim] - mn[dim]) dim = 2;
        if (mx[dim] > mn[dim]) {   /* otherwise all points coincide: keep as one leaf */
            int64_t mid = lo + (hi - lo) / 2;
            orc_select(t->pts, t->perm, lo, hi, mid, dim);
            nd.dim = dim;
            nd.split = t->pts[3 * t->perm[mid] + dim];
            t->nodes[id] = nd;
            int64_t l = orc_build(t, lo, mid);
            int64_t r = orc_build(t, mid, hi);
            t->nodes[id].left = l;
            t->nodes[id].right = r;
            return id;
        }
    }
    t->nodes[id] = nd;
    return id;
}

orc_kdtree *orc_kdtree_build(const double *pts, int64_t n)
{
    orc_kdtree *t = (orc_kdtree *)calloc(1, sizeof(orc_kdtree));
    if (!t) return NULL;
    t->pts = pts; t->n = n;
    t->perm = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    t->sorted = (double *)malloc((size_t)(n > 0 ? n : 1) * 3 * sizeof(double));
    for (int64_t i = 0; i < n; ++i) t->perm[i] = i;
    if (n > 0) orc_build(t, 0, n);
    for (int64_t i = 0; i < n; ++i) memcpy(t->sorted + 3 * i, pts + 3 * t->perm[i], 3 * sizeof(double));
    return t;
}

void orc_kdtree_free(orc_kdtree *t)
{
    if (!t) return;
    free(t->perm); free(t->sorted); free(t->nodes); free(t);
}

static void orc_search(const orc_kdtree *t, int64_t id, const double *q, int64_t skip,
                       double *best, int64_t *bi)
{
    const orc_node *nd = &t->nodes[id];
    if (nd->dim < 0) {
        for (int64_t i = nd->lo; i < nd->hi; ++i) {
            int64_t j = t->perm[i];
            if (j == skip) continue;
            double d = orc_d2(q, t->sorted + 3 * i);
            if (d < *best || (d == *best && j < *bi)) { *best = d; *bi = j; }
        }
        return;
    }
    double diff = q[nd->dim] - nd->split;
    int64_t near = diff < 0.0 ? nd->left : nd->right;
    int64_t far = diff < 0.0 ? nd->right : nd->left;
    orc_search(t, near, q, skip, best, bi);
    if (diff * diff <= *best) orc_search(t, far, q, skip, best, bi);
}

/* nthreads <= 0: all OpenMP threads. */
int orc_kdtree_query(const orc_kdtree *t, const double *q, int64_t nq, int skip_same_index,
                     int64_t *idx, double *d2, int nthreads)
{
    if (!t || nq < 0) return -1;
#ifdef _OPENMP
    int nt = nthreads > 0 ? nthreads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4096) num_threads(nt)
#else
    (void)nthreads;
#endif
    for (int64_t i = 0; i < nq; ++i) {
        double best = INFINITY;
        int64_t bi = INT64_MAX;
        if (t->n > 0) orc_search(t, 0, q + 3 * i, skip_same_index ? i : -1, &best, &bi);
        if (bi == INT64_MAX) { bi = -1; best = 0.0; }
        if (idx) idx[i] = bi;
        if (d2) d2[i] = best;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * D2 projection, metric.py:146-153: plane_errs[i] = dot(err[i], normals[i]) with
 * err[i] = iter[i] - search[nn[i]] (cloud_pair.py:90-100).  np.dot on two length-3
 * float64 vectors goes to the BLAS ddot NumPy links (OpenBLAS); on every FMA-capable x86
 * core its tail loop is dot = fma(x[i], y[i], dot) from dot = 0, i.e.
 * fma(e2, n2, fma(e1, n1, e0*n0)).  That is what the golden vectors made by the reference
 * in the authoring container contain (tests/golden/noisy_f64_500.npz pins it; for
 * fp32-representable inputs the products are exact and the contracted and uncontracted
 * forms agree bit for bit), so that is the form restated here and in the HIP kernel.
 * normals_by_neighbour == 0: row i of the OTHER cloud's normals (what the reference
 * does); != 0: row nn[i] (geometrically meaningful opt-in).
 * ---------------------------------------------------------------------------------- */
int orc_point_to_plane(const double *iter_pts, int64_t n, const double *search_pts,
                       const int64_t *nn, const double *other_normals, int64_t n_normals,
                       int normals_by_neighbour, double *proj)
{
    for (int64_t i = 0; i < n; ++i) {
        int64_t j = nn[i];
        int64_t k = normals_by_neighbour ? j : i;
        if (k < 0 || k >= n_normals) return -2;   /* the reference raises IndexError here */
        double ex = iter_pts[3 * i] - search_pts[3 * j];
        double ey = iter_pts[3 * i + 1] - search_pts[3 * j + 1];
        double ez = iter_pts[3 * i + 2] - search_pts[3 * j + 2];
        const double *nr = other_normals + 3 * k;
        double r = ex * nr[0];
        r = fma(ey, nr[1], r);
        r = fma(ez, nr[2], r);
        proj[i] = r;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Colour metrics, metric.py:261-333 and :389-427.  For row i of the iterating cloud:
 *   own = T(rgb_own[i]), other = T(rgb_other[nn[i]])  (the np.take of cloud_pair.py:120-124),
 *   diff = scale * (own - other)   (scale = 255 only for ColorHausdorffDistance in "rgb", :422-425),
 *   sq[i][c] = diff * diff.
 * T (transform_colors, :261-290) maps every row with np.matmul(M, c); with NumPy 2.2.6 / OpenBLAS on
 * the authoring host that is fma(m2, c2, fma(m0, c0, m1 * c1)) per component -- pinned by
 * tests/golden/fixture_eye3_color.npz and uniform_300_color.npz.  scheme: 0 rgb (identity), 1 ycc, 2 yuv.
 * sum[c] is accumulated row by row, which is the order np.mean(sq, axis=0) uses for a C-contiguous
 * (N, 3) array (no pairwise tree on axis 0); max[c] = np.max(sq, axis=0) (NaN propagates).
 * Any of sq / sum / max may be NULL.
 * ---------------------------------------------------------------------------------- */
int orc_color_columns(const double *rgb_own, int64_t n, const double *rgb_other, int64_t n_other,
                      const int64_t *nn, int scheme, double scale, double *sq, double *sum, double *max)
{
    static const double M[2][9] = {
        {0.2126, 0.7152, 0.0722, -0.1146, -0.3854, 0.5, 0.5, -0.4542, -0.0458},
        {0.25, 0.5, 0.25, 1, 0, -1, -0.5, 1, -0.5},
    };
    if (scheme < 0 || scheme > 2) return -1;
    double s[3] = {0, 0, 0}, m[3] = {0, 0, 0};
    for (int64_t i = 0; i < n; ++i) {
        int64_t j = nn[i];
        if (j < 0 || j >= n_other) return -2;
        const double *a = rgb_own + 3 * i, *b = rgb_other + 3 * j;
        for (int c = 0; c < 3; ++c) {
            double ta = a[c], tb = b[c];
            if (scheme) {
                const double *r = M[scheme - 1] + 3 * c;
                ta = fma(r[2], a[2], fma(r[0], a[0], r[1] * a[1]));
                tb = fma(r[2], b[2], fma(r[0], b[0], r[1] * b[1]));
            }
            double d = scale * (ta - tb);
            double v = d * d;
            if (sq) sq[3 * i + c] = v;
            s[c] = i ? s[c] + v : v;
            if (i == 0) m[c] = v;
            else if (m[c] == m[c] && (v != v || v > m[c])) m[c] = v;      /* np.max: a NaN sticks */
        }
    }
    for (int c = 0; c < 3; ++c) {
        if (sum) sum[c] = s[c];
        if (max) max[c] = m[c];
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Exact k nearest neighbours by brute force (self included), ascending (d2, row); used to
 * restate Open3D's estimate_normals() (cloud_pair.py:61-64; KDTreeSearchParamKNN(30)).
 * idx: [nq][k] (rows padded with -1 when nr < k).
 * ---------------------------------------------------------------------------------- */
int orc_knn_brute(const double *q, int64_t nq, const double *r, int64_t nr, int k, int64_t *idx)
{
    if (k <= 0 || k > 256) return -1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64)
#endif
    for (int64_t i = 0; i < nq; ++i) {
        double bd[256];
        int64_t bi[256];
        int cnt = 0;
        for (int64_t j = 0; j < nr; ++j) {
            double d = orc_d2(q + 3 * i, r + 3 * j);
            if (cnt == k && !(d < bd[k - 1])) continue;       /* j ascends: equal d never displaces */
            int p = cnt < k ? cnt : k - 1;
            while (p > 0 && d < bd[p - 1]) { bd[p] = bd[p - 1]; bi[p] = bi[p - 1]; --p; }
            bd[p] = d; bi[p] = j;
            if (cnt < k) ++cnt;
        }
        for (int c = 0; c < k; ++c) idx[i * k + c] = c < cnt ? bi[c] : -1;
    }
    return 0;
}
