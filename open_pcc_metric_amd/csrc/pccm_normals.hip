// Normal estimation on the GPU.
//
// Stands under `clouds[k].estimate_normals()` in CloudPair.__init__, open_pcc_metric/cloud_pair.py:61-64,
// i.e. Open3D 0.18 PointCloud::EstimateNormals with its defaults (KDTreeSearchParamKNN(knn = 30),
// fast_normal_computation): for every point, the covariance of its 30 nearest points of the same cloud
// (the point itself included) and the eigenvector of the smallest eigenvalue.  Open3D is not in the
// reference checkout, so this is a restatement of the published algorithm and is NOT parity-pinned:
//   - neighbours: exact k-NN, squared distance in fp64 ((dx*dx)+(dy*dy))+(dz*dz), ties to the smaller row;
//   - covariance: E[d d^T] - E[d] E[d]^T with d = p - q (Open3D forms the same matrix from raw moments;
//     shifting by the query point only improves the conditioning);
//   - eigenvector: closed-form eigenvalues of the symmetric 3x3 (trigonometric form on the scaled matrix)
//     and the largest cross product of two rows of (C - lambda I), as in Open3D's FastEigen3x3;
//   - fewer than 3 points in the cloud, or a degenerate covariance: (0, 0, 1) (Open3D's default normal);
//   - sign: Open3D leaves it to the eigen-solver; here the component of largest magnitude is made positive.
//     D2 squares the projection (metric.py:179), so no metric depends on the sign.
// The neighbours come from the grid engine's cell-sorted records: one thread per point scans the cube
// [c-r, c+r]^3 ring by ring, keeping the k best (d2, row) in a sorted private list, until the k-th best is
// provably closer than anything outside the cube (same stop rule as the 1-NN search).  Points that are
// still open after kKnnMaxRing rings (isolated outliers) are finished by an exact block-per-point scan of
// the whole cloud.
#include "pccm_internal.h"

namespace pccm {

constexpr int kKnnMax = 64;        // largest supported k
constexpr int kKnnMaxRing = 6;

struct KnnGeom {
    int dim[3];
    double org[3], h[3], inv_h[3], slack[3];
};

__device__ __forceinline__ int ncell_coord(double v, double org, double inv_h, int dim)
{
    double t = floor(__dmul_rn(__dsub_rn(v, org), inv_h));
    t = t < 0.0 ? 0.0 : t;
    const double top = (double)(dim - 1);
    t = t > top ? top : t;
    return (int)t;
}

__device__ __forceinline__ double nd2(double qx, double qy, double qz, double rx, double ry, double rz)
{
    double dx = __dsub_rn(qx, rx), dy = __dsub_rn(qy, ry), dz = __dsub_rn(qz, rz);
    double d = __dmul_rn(dx, dx);
    d = __dadd_rn(d, __dmul_rn(dy, dy));
    d = __dadd_rn(d, __dmul_rn(dz, dz));
    return d;
}

// sorted insertion of (d, row) into the k best kept in ascending (d, row) order
__device__ __forceinline__ void knn_insert(double *bd, int *bi, int k, int &cnt, double d, int row)
{
    if (cnt == k && !(d < bd[k - 1] || (d == bd[k - 1] && row < bi[k - 1]))) return;
    int p = cnt < k ? cnt : k - 1;
    while (p > 0 && (d < bd[p - 1] || (d == bd[p - 1] && row < bi[p - 1]))) {
        bd[p] = bd[p - 1];
        bi[p] = bi[p - 1];
        --p;
    }
    bd[p] = d;
    bi[p] = row;
    if (cnt < k) ++cnt;
}

// eigenvector of the smallest eigenvalue of the symmetric matrix [a00 a01 a02; a01 a11 a12; a02 a12 a22]
__device__ void smallest_eigenvector(double a00, double a01, double a02, double a11, double a12, double a22, double n[3])
{
    n[0] = 0.0; n[1] = 0.0; n[2] = 1.0;
    double mx = fmax(fmax(fabs(a00), fabs(a11)), fmax(fabs(a22), fmax(fabs(a01), fmax(fabs(a02), fabs(a12)))));
    if (!(mx > 0.0)) return;
    const double s = 1.0 / mx;
    a00 *= s; a01 *= s; a02 *= s; a11 *= s; a12 *= s; a22 *= s;
    const double norm = a01 * a01 + a02 * a02 + a12 * a12;
    double lam;
    if (norm > 0.0) {
        const double q = (a00 + a11 + a22) / 3.0;
        const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
        const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + 2.0 * norm) / 6.0);
        const double c00 = b11 * b22 - a12 * a12, c01 = a01 * b22 - a12 * a02, c02 = a01 * a12 - b11 * a02;
        const double det = (b00 * c00 - a01 * c01 + a02 * c02) / (p * p * p);
        const double half = fmin(fmax(0.5 * det, -1.0), 1.0);
        const double angle = acos(half) / 3.0;
        lam = q + 2.0 * p * cos(angle + 2.0943951023931953);      // smallest root: + 2*pi/3
    } else {
        lam = fmin(a00, fmin(a11, a22));
    }
    // rows of (A - lam I); the eigenvector is orthogonal to all of them: take the best-conditioned cross product
    const double r0[3] = {a00 - lam, a01, a02}, r1[3] = {a01, a11 - lam, a12}, r2[3] = {a02, a12, a22 - lam};
    double c[3][3];
    c[0][0] = r0[1] * r1[2] - r0[2] * r1[1]; c[0][1] = r0[2] * r1[0] - r0[0] * r1[2]; c[0][2] = r0[0] * r1[1] - r0[1] * r1[0];
    c[1][0] = r0[1] * r2[2] - r0[2] * r2[1]; c[1][1] = r0[2] * r2[0] - r0[0] * r2[2]; c[1][2] = r0[0] * r2[1] - r0[1] * r2[0];
    c[2][0] = r1[1] * r2[2] - r1[2] * r2[1]; c[2][1] = r1[2] * r2[0] - r1[0] * r2[2]; c[2][2] = r1[0] * r2[1] - r1[1] * r2[0];
    int best = 0;
    double bl = -1.0;
    for (int k = 0; k < 3; ++k) {
        const double l = c[k][0] * c[k][0] + c[k][1] * c[k][1] + c[k][2] * c[k][2];
        if (l > bl) { bl = l; best = k; }
    }
    if (!(bl > 1.0e-280)) return;                           // (numerically) isotropic or rank-0 spread
    const double inv = 1.0 / sqrt(bl);
    double v0 = c[best][0] * inv, v1 = c[best][1] * inv, v2 = c[best][2] * inv;
    const double m0 = fabs(v0), m1 = fabs(v1), m2 = fabs(v2);
    const double lead = (m0 >= m1 && m0 >= m2) ? v0 : (m1 >= m2 ? v1 : v2);
    if (lead < 0.0) { v0 = -v0; v1 = -v1; v2 = -v2; }
    n[0] = v0; n[1] = v1; n[2] = v2;
}

__device__ void normal_from_neighbours(const double *__restrict__ x64, double qx, double qy, double qz, const int *bi, int cnt,
                                       double *__restrict__ out)
{
    double n[3] = {0.0, 0.0, 1.0};
    if (cnt >= 3) {
        double m0 = 0, m1 = 0, m2 = 0, s00 = 0, s01 = 0, s02 = 0, s11 = 0, s12 = 0, s22 = 0;
        for (int k = 0; k < cnt; ++k) {
            const double *p = x64 + 3 * (int64_t)bi[k];
            const double dx = p[0] - qx, dy = p[1] - qy, dz = p[2] - qz;
            m0 += dx; m1 += dy; m2 += dz;
            s00 += dx * dx; s01 += dx * dy; s02 += dx * dz; s11 += dy * dy; s12 += dy * dz; s22 += dz * dz;
        }
        const double inv = 1.0 / (double)cnt;
        m0 *= inv; m1 *= inv; m2 *= inv;
        smallest_eigenvector(s00 * inv - m0 * m0, s01 * inv - m0 * m1, s02 * inv - m0 * m2, s11 * inv - m1 * m1,
                             s12 * inv - m1 * m2, s22 * inv - m2 * m2, n);
    }
    out[0] = n[0]; out[1] = n[1]; out[2] = n[2];
}

// one thread per point (in cell-sorted order); rings 0..kKnnMaxRing
// `todo` / `todo_count`: positions (within this cloud's slice) the wave kernel handed on; the threads stride over them
__global__ __launch_bounds__(256) void k_knn_normals(const GridRec *__restrict__ recs, int64_t qbase, KnnGeom g,
                                                     const uint32_t *__restrict__ cell_start, const double *__restrict__ x64,
                                                     int k, double *__restrict__ nrm_out, const uint32_t *__restrict__ todo,
                                                     const uint32_t *__restrict__ todo_count, int32_t *__restrict__ open_list,
                                                     uint32_t *__restrict__ open_count)
{
  const int64_t n = *todo_count;
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < n; u += (int64_t)gridDim.x * 256) {
    const int64_t t = todo[u];
    const double4 qa = *reinterpret_cast<const double4 *>(&recs[qbase + t]);   // this cloud's slice of the combined array
    const double qx = qa.x, qy = qa.y, qz = qa.z;
    const int qrow = (int)(__double_as_longlong(qa.w) & 0xffffffffll);
    const int dimx = g.dim[0], dimy = g.dim[1], dimz = g.dim[2];
    const int cx = ncell_coord(qx, g.org[0], g.inv_h[0], dimx);
    const int cy = ncell_coord(qy, g.org[1], g.inv_h[1], dimy);
    const int cz = ncell_coord(qz, g.org[2], g.inv_h[2], dimz);
    double bd[kKnnMax];
    int bi[kKnnMax];
    int cnt = 0;
    bool done = false;
    for (int r = 0; r <= kKnnMaxRing && !done; ++r) {
        const int z0 = max(cz - r, 0), z1 = min(cz + r, dimz - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, dimy - 1);
        const int x0 = max(cx - r, 0), x1 = min(cx + r, dimx - 1);
        for (int z = z0; z <= z1; ++z) {
            const bool zface = (z == cz - r) || (z == cz + r);
            for (int y = y0; y <= y1; ++y) {
                const uint32_t row = ((uint32_t)z * dimy + y) * dimx;
                const bool full = zface || y == cy - r || y == cy + r;
                for (int part = 0; part < (full ? 1 : 2); ++part) {
                    int xa, xb;
                    if (full) { xa = x0; xb = x1; }
                    else if (part == 0) { xa = xb = cx - r; if (xa < 0) continue; }
                    else { xa = xb = cx + r; if (xb > dimx - 1) continue; }
                    const uint32_t s = cell_start[row + xa], e = cell_start[row + xb + 1];
                    for (uint32_t p = s; p < e; ++p) {
                        const double4 a = *reinterpret_cast<const double4 *>(&recs[p]);
                        knn_insert(bd, bi, k, cnt, nd2(qx, qy, qz, a.x, a.y, a.z), (int)(__double_as_longlong(a.w) & 0xffffffffll));
                    }
                }
            }
        }
        // stop rule of the grid engine, applied to the k-th best
        double L = INFINITY;
        const double q[3] = {qx, qy, qz};
        const int c[3] = {cx, cy, cz};
        for (int a = 0; a < 3; ++a) {
            if (c[a] - r > 0) L = fmin(L, (q[a] - (g.org[a] + (double)(c[a] - r) * g.h[a])) - g.slack[a]);
            if (c[a] + r < g.dim[a] - 1) L = fmin(L, ((g.org[a] + (double)(c[a] + r + 1) * g.h[a]) - q[a]) - g.slack[a]);
        }
        if (L == INFINITY) done = true;
        else if (cnt == k && L > 0.0 && bd[k - 1] < L * L * (1.0 - 0x1.0p-30)) done = true;
    }
    if (done) {
        normal_from_neighbours(x64, qx, qy, qz, bi, cnt, nrm_out + 3 * (int64_t)qrow);
    } else {
        open_list[atomicAdd(open_count, 1u)] = qrow;
    }
  }
}

// ---- one wave per point -------------------------------------------------------------------------------
// The per-thread search above keeps its k best in a private sorted list: ~85 insertions of ~15 shifts each per
// point, all through scratch memory (12 ms per million points).  Here a wave takes one point: the lanes own the
// x-runs of the cube [c-r, c+r]^3 (r = 2, then 3), the candidates' distances go to LDS, the k-th smallest is
// found by a wave-wide quickselect (pivot = some staged distance inside the bracket, counted with ballots), ties at
// the k-th distance go to the smaller rows, and the covariance of the selected points is accumulated by all lanes
// and written out; k_normals_from_cov then solves the 3x3 eigenproblems one thread per point.  Same neighbour set
// as the per-thread search (exact k-NN, (d2, row) order); the sums are taken in a different order.
// Points the two cubes cannot settle, or with more than kWCap candidates, are passed on to k_knn_normals.
constexpr int kWCap = 512;

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(256) void k_knn_cov_wave(const GridRec *__restrict__ recs, int64_t qbase, int64_t n, KnnGeom g,
                                                      const uint32_t *__restrict__ cell_start, int k,
                                                      double *__restrict__ cov_out /*[n][6] by row*/, int32_t *__restrict__ cnt_out,
                                                      uint32_t *__restrict__ todo, uint32_t *__restrict__ todo_count)
{
    __shared__ double s_d[4][kWCap];
    __shared__ uint32_t s_p[4][kWCap];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int dimx = g.dim[0], dimy = g.dim[1], dimz = g.dim[2];
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t t = (int64_t)blockIdx.x * 4 + w; t < n; t += nwaves) {
        const double4 qa = *reinterpret_cast<const double4 *>(&recs[qbase + t]);          // wave-uniform
        const double qx = qa.x, qy = qa.y, qz = qa.z;
        const int qrow = (int)(__double_as_longlong(qa.w) & 0xffffffffll);
        const int cx = ncell_coord(qx, g.org[0], g.inv_h[0], dimx);
        const int cy = ncell_coord(qy, g.org[1], g.inv_h[1], dimy);
        const int cz = ncell_coord(qz, g.org[2], g.inv_h[2], dimz);
        bool done = false, giveup = false;
        for (int r = 2; r <= 3 && !done && !giveup; ++r) {
            // the lanes own the (2r+1)^2 x-runs of the cube
            const int side = 2 * r + 1;
            uint32_t s = 0, len = 0;
            if (lane < side * side) {
                const int z = cz + lane / side - r, y = cy + lane % side - r;
                if (z >= 0 && z < dimz && y >= 0 && y < dimy) {
                    const uint32_t row = ((uint32_t)z * dimy + y) * dimx;
                    const int x0 = max(cx - r, 0), x1 = min(cx + r, dimx - 1);
                    s = cell_start[row + x0];
                    len = cell_start[row + x1 + 1] - s;
                }
            }
            uint32_t inc = len;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(inc, off);
                if (lane >= off) inc += o;
            }
            const uint32_t T = __shfl(inc, 63);
            if (T > (uint32_t)kWCap) { giveup = true; break; }
            for (uint32_t u = 0; u < len; ++u) s_p[w][inc - len + u] = s + u;            // flatten the runs
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double dmax = 0.0;
            for (uint32_t i = lane; i < T; i += 64) {
                const double4 a = *reinterpret_cast<const double4 *>(&recs[s_p[w][i]]);
                const double d = nd2(qx, qy, qz, a.x, a.y, a.z);
                s_d[w][i] = d;
                dmax = fmax(dmax, d);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, off));
            // stop rule of the grid engine for this cube
            double L = INFINITY;
            {
                const double q[3] = {qx, qy, qz};
                const int c[3] = {cx, cy, cz};
                for (int a = 0; a < 3; ++a) {
                    if (c[a] - r > 0) L = fmin(L, (q[a] - (g.org[a] + (double)(c[a] - r) * g.h[a])) - g.slack[a]);
                    if (c[a] + r < g.dim[a] - 1) L = fmin(L, ((g.org[a] + (double)(c[a] + r + 1) * g.h[a]) - q[a]) - g.slack[a]);
                }
            }
            const bool whole = (L == INFINITY);                     // the cube covers the grid: these are all the points
            if (!whole && T < (uint32_t)k) continue;
            // k-th smallest distance tau by quickselect over the staged values; bracket: #(d <= lo) < kk <= #(d <= hi)
            const uint32_t kk = T < (uint32_t)k ? T : (uint32_t)k;
            double lo = -1.0, hi = dmax;
            for (;;) {
                double cand = 0.0;
                bool have = false;
                for (uint32_t i = lane; i < T && !have; i += 64) {
                    const double d = s_d[w][i];
                    if (d > lo && d < hi) { cand = d; have = true; }
                }
                const unsigned long long m = __ballot(have);
                if (!m) break;                                      // nothing strictly inside: tau = hi
                const double x = __shfl(cand, __ffsll((long long)m) - 1);
                uint32_t c = 0;
                for (uint32_t i = lane; i < T; i += 64) c += (s_d[w][i] <= x) ? 1u : 0u;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
                if (c >= kk) hi = x; else lo = x;
            }
            const double tau = hi;
            if (!whole && !(L > 0.0 && tau < L * L * (1.0 - 0x1.0p-30))) continue;        // try the next cube
            // ties at tau: the smaller rows win
            uint32_t below = 0, equal = 0;
            for (uint32_t i = lane; i < T; i += 64) {
                const double d = s_d[w][i];
                below += d < tau ? 1u : 0u;
                equal += d == tau ? 1u : 0u;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                below += __shfl_xor(below, off);
                equal += __shfl_xor(equal, off);
            }
            int row_cut = 0x7fffffff;                                // rows <= row_cut among the tied are taken
            if (below + equal > kk) {
                int last = -1;
                for (uint32_t need = kk - below; need > 0; --need) {                       // need-th smallest tied row
                    int best = 0x7fffffff;
                    for (uint32_t i = lane; i < T; i += 64)
                        if (s_d[w][i] == tau) {
                            const int row = recs[s_p[w][i]].idx;
                            if (row > last && row < best) best = row;
                        }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) best = min(best, __shfl_xor(best, off));
                    last = best;
                }
                row_cut = last;
            }
            double m0 = 0, m1 = 0, m2 = 0, s00 = 0, s01 = 0, s02 = 0, s11 = 0, s12 = 0, s22 = 0;
            for (uint32_t i = lane; i < T; i += 64) {
                const double d = s_d[w][i];
                if (d > tau) continue;
                const double4 a = *reinterpret_cast<const double4 *>(&recs[s_p[w][i]]);
                if (d == tau && (int)(__double_as_longlong(a.w) & 0xffffffffll) > row_cut) continue;
                const double dx = a.x - qx, dy = a.y - qy, dz = a.z - qz;
                m0 += dx; m1 += dy; m2 += dz;
                s00 += dx * dx; s01 += dx * dy; s02 += dx * dz; s11 += dy * dy; s12 += dy * dz; s22 += dz * dz;
            }
            m0 = wave_sum_f64(m0); m1 = wave_sum_f64(m1); m2 = wave_sum_f64(m2);
            s00 = wave_sum_f64(s00); s01 = wave_sum_f64(s01); s02 = wave_sum_f64(s02);
            s11 = wave_sum_f64(s11); s12 = wave_sum_f64(s12); s22 = wave_sum_f64(s22);
            if (lane == 0) {
                const double inv = 1.0 / (double)kk;
                m0 *= inv; m1 *= inv; m2 *= inv;
                double *o = cov_out + 6 * (int64_t)qrow;
                o[0] = s00 * inv - m0 * m0; o[1] = s01 * inv - m0 * m1; o[2] = s02 * inv - m0 * m2;
                o[3] = s11 * inv - m1 * m1; o[4] = s12 * inv - m1 * m2; o[5] = s22 * inv - m2 * m2;
                cnt_out[qrow] = (int)kk;
            }
            done = true;
        }
        if (!done && lane == 0) {
            cnt_out[qrow] = -1;                                       // k_knn_normals writes this normal itself
            todo[atomicAdd(todo_count, 1u)] = (uint32_t)t;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");        // LDS is reused by the next point
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ __launch_bounds__(256) void k_normals_from_cov(const double *__restrict__ cov, const int32_t *__restrict__ cnt, int64_t n,
                                                          double *__restrict__ nrm_out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = cnt[i];
    if (c < 0) return;                                               // settled by the per-thread kernel
    double nn[3] = {0.0, 0.0, 1.0};
    const double *a = cov + 6 * i;
    if (c >= 3) smallest_eigenvector(a[0], a[1], a[2], a[3], a[4], a[5], nn);
    nrm_out[3 * i] = nn[0];
    nrm_out[3 * i + 1] = nn[1];
    nrm_out[3 * i + 2] = nn[2];
}

// isolated points: exact k-NN by a full scan, one workgroup per point.  Every thread keeps the k best of its
// stride; the k global best are then extracted one by one with a workgroup-wide lexicographic minimum.
__global__ __launch_bounds__(256) void k_knn_normals_full(const double *__restrict__ x64, int64_t n, int k,
                                                          const int32_t *__restrict__ open_list,
                                                          const uint32_t *__restrict__ open_count,
                                                          double *__restrict__ nrm_out)
{
    __shared__ double s_d[256];
    __shared__ int s_i[256];
    __shared__ int s_sel[kKnnMax];
    const int tid = threadIdx.x;
    const uint32_t count = *open_count;
    for (uint32_t f = blockIdx.x; f < count; f += gridDim.x) {
        const int qrow = open_list[f];
        const double qx = x64[3 * (int64_t)qrow], qy = x64[3 * (int64_t)qrow + 1], qz = x64[3 * (int64_t)qrow + 2];
        double bd[kKnnMax];
        int bi[kKnnMax];
        int cnt = 0;
        for (int64_t j = tid; j < n; j += 256) knn_insert(bd, bi, k, cnt, nd2(qx, qy, qz, x64[3 * j], x64[3 * j + 1], x64[3 * j + 2]), (int)j);
        int head = 0, nsel = 0;
        const int want = n < k ? (int)n : k;
        for (int round = 0; round < want; ++round) {
            s_d[tid] = head < cnt ? bd[head] : INFINITY;
            s_i[tid] = head < cnt ? bi[head] : 0x7fffffff;
            __syncthreads();
            for (int off = 128; off > 0; off >>= 1) {
                if (tid < off) {
                    const double od = s_d[tid + off];
                    const int oi = s_i[tid + off];
                    if (od < s_d[tid] || (od == s_d[tid] && oi < s_i[tid])) { s_d[tid] = od; s_i[tid] = oi; }
                }
                __syncthreads();
            }
            const int win = s_i[0];
            if (head < cnt && bi[head] == win) ++head;      // rows are unique: exactly one thread owns the winner
            if (tid == 0) s_sel[nsel] = win;
            ++nsel;
            __syncthreads();
        }
        if (tid == 0) normal_from_neighbours(x64, qx, qy, qz, s_sel, nsel, nrm_out + 3 * (int64_t)qrow);
        __syncthreads();
    }
}

int estimate_normals(pccm_ctx *ctx, int which, int k)
{
    Cloud &c = ctx->cloud[which];
    if (c.n <= 0) return fail(PCCM_E_STATE, "cloud %d is not set", which);
    if (k < 3 || k > kKnnMax) return fail(PCCM_E_ARG, "k must be in 3..%d", kKnnMax);
    int rc;
    // GridRec (fp64) records of this cloud alone.  The pair's geometry follows the pair's larger cloud: fine for that cloud and for
    // one of similar size, hopeless for a much sparser one (a low rate of a codec: k = 30 neighbours then lie six rings out), which
    // gets cells of its own (grid_ensure_solo: a few histogram passes, cached with the cloud)
    const Cloud &other = ctx->cloud[1 - which];
    if (other.n > 2 * c.n) {
        if ((rc = grid_ensure_solo(ctx, which))) return rc;
    } else if ((rc = grid_ensure(ctx, true, 1 << which))) return rc;
    const Grid &gr = ctx->grid;
    KnnGeom g;
    for (int a = 0; a < 3; ++a) {
        g.dim[a] = gr.dim[a];
        g.org[a] = gr.org[a];
        g.h[a] = gr.h[a];
        g.inv_h[a] = gr.inv_h[a];
        g.slack[a] = (fabs(gr.org[a]) + (gr.dim[a] + 2) * gr.h[a]) * 0x1.0p-48;
    }
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    if ((rc = grow((void **)&c.nrm64, c.cap_nrm, (size_t)c.n * 3 * sizeof(double)))) return rc;
    c.n_nrm = c.n;
    c.nrm_exact32 = false;
    c.nrm_deferred = false;
    c.nrm_host = nullptr;
    for (int d = 0; d < 3; ++d) ctx->nn_gen[d]++;      // pending D2 reductions would use stale normals
    ctx->epoch++;
    if ((rc = ensure(ctx, ctx->g_cell_of, (size_t)c.n * sizeof(uint32_t)))) return rc;   // reused: points left to the full scan
    if ((rc = ensure(ctx, ctx->g_rank, (size_t)c.n * sizeof(uint32_t)))) return rc;      // reused: points left to the per-thread search
    if ((rc = ensure(ctx, ctx->val, (size_t)c.n * (6 * sizeof(double) + sizeof(int32_t))))) return rc;   // covariances + counts
    if ((rc = ensure(ctx, ctx->g_blocksum, 256))) return rc;
    uint32_t *open_count = (uint32_t *)ctx->g_blocksum.p, *todo_count = open_count + 1;
    PCCM_HIP(hipMemsetAsync(open_count, 0, 2 * sizeof(uint32_t), ctx->stream));
    double *cov = (double *)ctx->val.p;
    int32_t *cnt = (int32_t *)(cov + 6 * c.n);
    // cell_start holds positions relative to the cloud's first record
    const uint32_t *cs = (const uint32_t *)gr.cell_start.p + (which ? gr.ncells + 1 : 0);
    const GridRec *crecs = (const GridRec *)gr.recs.p + (which ? gr.n[0] : 0);
    const int64_t qbase = 0;
    const int64_t wblocks = (c.n + 3) / 4;
    hipLaunchKernelGGL(k_knn_cov_wave, dim3((unsigned)(wblocks < 16384 ? wblocks : 16384)), dim3(256), 0, ctx->stream,
                       crecs, qbase, c.n, g, cs, k, cov, cnt, (uint32_t *)ctx->g_rank.p, todo_count);
    hipLaunchKernelGGL(k_normals_from_cov, dim3((unsigned)((c.n + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)cov,
                       (const int32_t *)cnt, c.n, c.nrm64);
    hipLaunchKernelGGL(k_knn_normals, dim3(2048), dim3(256), 0, ctx->stream, crecs, qbase, g, cs,
                       (const double *)c.xyz64, k, c.nrm64, (const uint32_t *)ctx->g_rank.p, (const uint32_t *)todo_count,
                       (int32_t *)ctx->g_cell_of.p, open_count);
    hipLaunchKernelGGL(k_knn_normals_full, dim3(512), dim3(256), 0, ctx->stream, (const double *)c.xyz64, c.n, k,
                       (const int32_t *)ctx->g_cell_of.p, (const uint32_t *)open_count, c.nrm64);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

}  // namespace pccm
