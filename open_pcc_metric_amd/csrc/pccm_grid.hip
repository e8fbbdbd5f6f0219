// Uniform-grid exact 1-NN engine for gfx950.
//
// Replaces the same reference interface as the brute-force engine -- get_neighbour_cloud(),
// open_pcc_metric/cloud_pair.py:10-42, and the KD-tree builds of cloud_pair.py:65 -- with
// O(N) work: the searched cloud is counting-sorted into a uniform grid (the analogue of the
// reference's KDTreeFlann build), every query scans the 3x3x3 cells around its own cell and
// widens ring by ring until the best distance found is provably smaller than the distance to
// anything not scanned yet.  All candidate distances are evaluated in fp64 with the reference's
// arithmetic ((dx*dx)+(dy*dy))+(dz*dz) (no FMA), exact ties go to the smallest original row:
// results are bit-identical to the brute-force engine and to the oracle.
//
// Data layout in HBM: GridRec = {double x, y, z; int32 row; pad} (32 B, two 16-byte loads) sorted
// by cell, cells in x-fastest order, so the 3 cells of one x-run are one contiguous range and a
// 3x3x3 neighbourhood is 9 ranges; cell_start is uint32[ncells + 1].  Queries are processed in
// cell-sorted order too (the iterating cloud's own grid records, or the shard's rows sorted by
// the searched grid's cells), so the lanes of a wave walk the same ranges and their loads
// coalesce in L1/L2.  The kernel is latency/L2-bound: ~54 candidates x 32 B per query, ALU work
// is negligible (DESIGN.md gives the byte model).
//
// Exactness of the stop rule.  cell(x) = clamp(floor((x - org) * inv_h)) is monotonic in x, so a
// point in a cell left of cell c lies below org + c*h up to a few ulps of the grid's size; the
// kernel subtracts that slack (g.slack[a]) from every face distance and compares with a strict
// "<" after shrinking the bound by 2^-30, so a stop is never taken on a rounding coincidence.
// Queries that are still unresolved after kMaxRing rings are handed to the brute engine's exact
// rescan kernel (k2b_fallback) with the best distance found so far as the candidate threshold.
#include "pccm_internal.h"

namespace pccm {

constexpr int kMaxRing = 3;
constexpr uint32_t kTailWaveMax = 16384;   // tails up to this many queries take the wave-per-query kernel
constexpr int kScanItems = 8;                       // per thread in the prefix scan
constexpr int kScanBlock = 256 * kScanItems;

struct GridGeom {
    int dim[3];
    double org[3], h[3], inv_h[3], slack[3];
};

__device__ __forceinline__ int cell_coord(double v, double org, double inv_h, int dim)
{
    double t = floor(__dmul_rn(__dsub_rn(v, org), inv_h));
    t = t < 0.0 ? 0.0 : t;
    const double top = (double)(dim - 1);
    t = t > top ? top : t;
    return (int)t;
}

__device__ __forceinline__ double gdist64(double qx, double qy, double qz, double rx, double ry, double rz)
{
    double dx = __dsub_rn(qx, rx), dy = __dsub_rn(qy, ry), dz = __dsub_rn(qz, rz);
    double d = __dmul_rn(dx, dx);
    d = __dadd_rn(d, __dmul_rn(dy, dy));
    d = __dadd_rn(d, __dmul_rn(dz, dz));
    return d;
}

// ---- build: cell ids + histogram, scan, scatter ------------------------------------------------
__global__ __launch_bounds__(256) void k_grid_cells(const double *__restrict__ x64, int64_t row0, int64_t n,
                                                    GridGeom g, uint32_t *__restrict__ cell_of,
                                                    uint32_t *__restrict__ hist)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *p = x64 + 3 * (row0 + i);
    const int cx = cell_coord(p[0], g.org[0], g.inv_h[0], g.dim[0]);
    const int cy = cell_coord(p[1], g.org[1], g.inv_h[1], g.dim[1]);
    const int cz = cell_coord(p[2], g.org[2], g.inv_h[2], g.dim[2]);
    const uint32_t c = ((uint32_t)cz * g.dim[1] + cy) * g.dim[0] + cx;
    cell_of[i] = c;
    atomicAdd(&hist[c], 1u);
}

__global__ __launch_bounds__(256) void k_grid_scatter(const double *__restrict__ x64, int64_t row0, int64_t n,
                                                      const uint32_t *__restrict__ cell_of,
                                                      uint32_t *__restrict__ cursor, GridRec *__restrict__ recs)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *p = x64 + 3 * (row0 + i);
    const uint32_t pos = atomicAdd(&cursor[cell_of[i]], 1u);
    GridRec r;
    r.x = p[0]; r.y = p[1]; r.z = p[2];
    r.idx = (int32_t)(row0 + i);
    r.pad = 0;
    recs[pos] = r;
}

// exclusive prefix sum of uint32 data[0..m) in place: block scan, scan of block totals, add back
__global__ __launch_bounds__(256) void k_scan_block(uint32_t *__restrict__ data, int64_t m, uint32_t *__restrict__ blocksum)
{
    __shared__ uint32_t wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * kScanBlock + (int64_t)tid * kScanItems;
    uint32_t v[kScanItems], tot = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = (base + k < m) ? data[base + k] : 0u;
        tot += v[k];
    }
    uint32_t inc = tot;                                   // inclusive scan of thread totals in the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum[k];
    uint32_t run = woff + inc - tot;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < m) data[base + k] = run;
        run += v[k];
    }
    if (tid == 255) blocksum[blockIdx.x] = woff + inc;
}

__global__ __launch_bounds__(1024) void k_scan_sums(uint32_t *__restrict__ blocksum, int64_t nb)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t i = base + tid;
        const uint32_t v = i < nb ? blocksum[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t woff = carry_s;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        if (i < nb) blocksum[i] = woff + inc - v;
        __syncthreads();
        if (tid == 1023) carry_s = woff + inc;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_scan_add(uint32_t *__restrict__ data, int64_t m, const uint32_t *__restrict__ blocksum)
{
    const uint32_t add = blocksum[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * kScanBlock + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < m) data[base + k] += add;
}

static int exclusive_scan(pccm_ctx *ctx, uint32_t *data, int64_t m)
{
    const int64_t nb = (m + kScanBlock - 1) / kScanBlock;
    int rc = ensure(ctx, ctx->g_blocksum, (size_t)nb * sizeof(uint32_t));
    if (rc) return rc;
    uint32_t *bs = (uint32_t *)ctx->g_blocksum.p;
    hipLaunchKernelGGL(k_scan_block, dim3((unsigned)nb), dim3(256), 0, ctx->stream, data, m, bs);
    if (nb > 1) {
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, bs, nb);
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)nb), dim3(256), 0, ctx->stream, data, m, bs);
    }
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

// counting sort of rows [row0, row0+n) of x64 by cell of geometry g:
// cell_start (uint32[ncells+1], may be null for a throw-away histogram) and recs[n]
static int sort_by_cell(pccm_ctx *ctx, const double *x64, int64_t row0, int64_t n, const GridGeom &g, int64_t ncells,
                        uint32_t *cell_start, GridRec *recs)
{
    int rc;
    if ((rc = ensure(ctx, ctx->g_cell_of, (size_t)n * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->g_hist, (size_t)(ncells + 1) * sizeof(uint32_t)))) return rc;
    uint32_t *cell_of = (uint32_t *)ctx->g_cell_of.p;
    uint32_t *cursor = (uint32_t *)ctx->g_hist.p;
    uint32_t *hist = cell_start ? cell_start : cursor;
    PCCM_HIP(hipMemsetAsync(hist, 0, (size_t)(ncells + 1) * sizeof(uint32_t), ctx->stream));
    dim3 grid((unsigned)((n + 255) / 256));
    hipLaunchKernelGGL(k_grid_cells, grid, dim3(256), 0, ctx->stream, x64, row0, n, g, cell_of, hist);
    if ((rc = exclusive_scan(ctx, hist, ncells + 1))) return rc;
    if (cell_start)
        PCCM_HIP(hipMemcpyAsync(cursor, cell_start, (size_t)(ncells + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_grid_scatter, grid, dim3(256), 0, ctx->stream, x64, row0, n, cell_of, cursor, recs);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

// ---- query --------------------------------------------------------------------------------------
struct Best {
    double d;
    int idx;
};

__device__ __forceinline__ void consider(const double4 &a, double qx, double qy, double qz, int qrow, bool self, Best &b)
{
    const int row = (int)(__double_as_longlong(a.w) & 0xffffffffll);
    const double d = gdist64(qx, qy, qz, a.x, a.y, a.z);
    bool better = d < b.d || (d == b.d && row < b.idx);
    if (self) better = better && (row != qrow);
    b.d = better ? d : b.d;
    b.idx = better ? row : b.idx;
}

// Scan records [s, e) kBatch at a time: the loads of a batch are independent and issued together
// (one memory round trip per batch instead of one per record); indices past the end are clamped to
// e-1, and re-evaluating a record is harmless because the lexicographic min is idempotent.
constexpr int kBatch = 4;

template <bool SELF>
__device__ __forceinline__ void scan_range(const GridRec *__restrict__ recs, uint32_t s, uint32_t e, double qx, double qy,
                                           double qz, int qrow, Best &b)
{
    for (uint32_t p = s; p < e; p += kBatch) {
        double4 a[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const uint32_t pj = (p + j < e) ? p + j : e - 1;
            a[j] = *reinterpret_cast<const double4 *>(&recs[pj]);   // x y z | (row, pad)
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) consider(a[j], qx, qy, qz, qrow, SELF, b);
    }
}

template <bool SELF>
__global__ __launch_bounds__(256) void k_grid_query(const GridRec *__restrict__ qrecs, int64_t nq_host,
                                                    const uint32_t *__restrict__ nq_dev, GridGeom g,
                                                    const uint32_t *__restrict__ cell_start,
                                                    const GridRec *__restrict__ srecs, int64_t row_base, double slack32,
                                                    int32_t *__restrict__ idx_out, double *__restrict__ d2_out,
                                                    int32_t *__restrict__ flagged, float *__restrict__ flag_thr,
                                                    uint32_t *__restrict__ nflag)
{
    // the query count is a host value (direct launch) or lives on the device (tail of the cooperative kernel)
    int64_t nq = nq_host;
    if (nq_dev) {
        nq = (int64_t)*nq_dev;
        if (nq <= (int64_t)kTailWaveMax) return;        // short tail: k_grid_tail_wave handled it
    }
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < nq; t += (int64_t)gridDim.x * 256) {
    const double4 qa = *reinterpret_cast<const double4 *>(&qrecs[t]);
    const double qx = qa.x, qy = qa.y, qz = qa.z;
    const int qrow = (int)(__double_as_longlong(qa.w) & 0xffffffffll);
    const int cx = cell_coord(qx, g.org[0], g.inv_h[0], g.dim[0]);
    const int cy = cell_coord(qy, g.org[1], g.inv_h[1], g.dim[1]);
    const int cz = cell_coord(qz, g.org[2], g.inv_h[2], g.dim[2]);
    const int dimx = g.dim[0], dimy = g.dim[1], dimz = g.dim[2];

    Best b;
    b.d = INFINITY;
    b.idx = 0x7fffffff;
    bool done = false;
    {
        // ring 1 = the 3x3x3 block: nine x-runs whose bounds are fetched together up front
        const int x0 = max(cx - 1, 0), x1 = min(cx + 1, dimx - 1);
        uint32_t rs[9], re[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int z = cz + k / 3 - 1, y = cy + k % 3 - 1;
            const bool in = z >= 0 && z < dimz && y >= 0 && y < dimy;
            const uint32_t row = in ? ((uint32_t)z * dimy + y) * dimx : 0u;
            const uint32_t a = cell_start[row + x0], c = cell_start[row + x1 + 1];
            rs[k] = in ? a : 0u;
            re[k] = in ? c : 0u;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) scan_range<SELF>(srecs, rs[k], re[k], qx, qy, qz, qrow, b);
    }
    for (int r = 1; r <= kMaxRing && !done; ++r) {
        if (r > 1) {
        const int z0 = max(cz - r, 0), z1 = min(cz + r, dimz - 1);
        const int y0 = max(cy - r, 0), y1 = min(cy + r, dimy - 1);
        const int x0 = max(cx - r, 0), x1 = min(cx + r, dimx - 1);
        for (int z = z0; z <= z1; ++z) {
            const bool zface = (z == cz - r) || (z == cz + r);
            for (int y = y0; y <= y1; ++y) {
                const uint32_t row = ((uint32_t)z * dimy + y) * dimx;
                if (zface || y == cy - r || y == cy + r) {
                    scan_range<SELF>(srecs, cell_start[row + x0], cell_start[row + x1 + 1], qx, qy, qz, qrow, b);
                } else {                                    // interior of the shell: only the two end cells
                    if (cx - r >= 0)
                        scan_range<SELF>(srecs, cell_start[row + cx - r], cell_start[row + cx - r + 1], qx, qy, qz, qrow, b);
                    if (cx + r <= dimx - 1)
                        scan_range<SELF>(srecs, cell_start[row + cx + r], cell_start[row + cx + r + 1], qx, qy, qz, qrow, b);
                }
            }
        }
        }
        // distance from the query to the nearest face of the scanned cube that still has cells behind it
        double L = INFINITY;
        const double q[3] = {qx, qy, qz};
        const int c[3] = {cx, cy, cz};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (c[a] - r > 0) L = fmin(L, (q[a] - (g.org[a] + (double)(c[a] - r) * g.h[a])) - g.slack[a]);
            if (c[a] + r < g.dim[a] - 1) L = fmin(L, ((g.org[a] + (double)(c[a] + r + 1) * g.h[a]) - q[a]) - g.slack[a]);
        }
        if (L == INFINITY) done = true;                    // the cube covers the whole grid
        else if (L > 0.0 && b.d < L * L * (1.0 - 0x1.0p-30)) done = true;
    }
    if (done) {
        if (b.idx == 0x7fffffff) { b.idx = -1; b.d = 0.0; }   // SELF on a one-point cloud (not reached: host handles it)
        idx_out[qrow - row_base] = b.idx;
        d2_out[qrow - row_base] = b.d;
    } else {
        // unresolved: exact rescan by the brute engine's fallback kernel; every possible winner has
        // d32 <= thr(best so far) -- same bound as k2_refine
        double tq = (b.d == INFINITY) ? 1.0e18 : sqrt(b.d) * (1.0 + 0x1.0p-20) + slack32;
        double thr = tq * tq * (1.0 + 0x1.0p-30) + 1.0e-36;
        float tf = thr > 3.0e38 ? 3.0e38f : (float)thr;
        tf = __uint_as_float(__float_as_uint(tf) + 1u);
        const uint32_t pos = atomicAdd(nflag, 1u);
        flagged[pos] = (int32_t)(qrow - row_base);
        flag_thr[pos] = tf;
    }
    }
}

// ---- cooperative ring-1 kernel ---------------------------------------------------------------------
// A wave owns 64 consecutive queries of the cell-sorted list and works on one *segment* at a time: the
// queries that share an x-row of cells (same cy, cz; at most kSegWidth cells wide).  For the nine x-runs
// around that row the wave
//   1. fetches all cell bounds with nine coalesced loads issued together (lane i reads
//      cell_start[x_lo + i]) and hands every lane its own [s, e) per run by shuffle,
//   2. stages the runs' records into LDS as fp32 SoA (x | y | z [| row]) with coalesced 32-byte-per-lane
//      loads -- every record is fetched once per wave instead of once per lane,
//   3. lets every lane scan only its own candidates from LDS in fp32 (same-cell lanes broadcast),
//      tracking best d32, its record position and the second-best d32 -- no fp64, no branches,
//   4. certifies like k2_refine: if the second-best d32 is above thr(best d32) the fp32 winner is the
//      unique fp64 winner, whose exact d2 is then computed once from its fp64 record.
// Queries that cannot be certified (near ties, exact ties on lattices) or whose ring-1 result does not
// satisfy the stop rule go to `tail` and are finished exactly by k_grid_query.
constexpr int kCap = 768;         // fp32 records staged per wave (9 KB); more -> several windows
constexpr int kSegWidth = 61;     // + 3 bounds = 64 lanes
constexpr float kBigF = 3.0e38f;

template <bool SELF>
__global__ __launch_bounds__(256) void k_grid_query_coop(const GridRec *__restrict__ qrecs, int64_t nq, GridGeom g,
                                                         const uint32_t *__restrict__ cell_start,
                                                         const GridRec *__restrict__ srecs, int64_t row_base,
                                                         double slack32, int32_t *__restrict__ idx_out,
                                                         double *__restrict__ d2_out, GridRec *__restrict__ tail,
                                                         uint32_t *__restrict__ tailcount)
{
    __shared__ float lx[4][kCap], ly[4][kCap], lz[4][kCap];
    __shared__ int lrow[SELF ? 4 : 1][SELF ? kCap : 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t chunk = (int64_t)blockIdx.x * 4 + w;
    if (chunk * 64 >= nq) return;                       // wave-uniform
    const int64_t t = chunk * 64 + lane;
    const bool valid = t < nq;
    const double4 qa = *reinterpret_cast<const double4 *>(&qrecs[valid ? t : nq - 1]);
    const double qx = qa.x, qy = qa.y, qz = qa.z;
    const float fx = (float)qx, fy = (float)qy, fz = (float)qz;
    const int qrow = (int)(__double_as_longlong(qa.w) & 0xffffffffll);
    const int dimx = g.dim[0], dimy = g.dim[1], dimz = g.dim[2];
    const int cx = cell_coord(qx, g.org[0], g.inv_h[0], dimx);
    const int cy = cell_coord(qy, g.org[1], g.inv_h[1], dimy);
    const int cz = cell_coord(qz, g.org[2], g.inv_h[2], dimz);
    const int R = cz * dimy + cy;

    unsigned long long pending = __ballot(valid);
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const int Rl = __shfl(R, leader), xa = __shfl(cx, leader);
        const int cyl = __shfl(cy, leader), czl = __shfl(cz, leader);
        const bool inseg = ((pending >> lane) & 1ull) && R == Rl && cx >= xa && (cx - xa) < kSegWidth;
        const unsigned long long seg = __ballot(inseg);
        int xb = inseg ? cx : xa;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) xb = max(xb, __shfl_xor(xb, off));
        const int x_lo = max(xa - 1, 0);
        const int x_hi = min(xb + 2, dimx);             // index of the last bound needed
        const int my_s = (max(cx - 1, 0) - x_lo) & 63, my_e = (min(cx + 2, dimx) - x_lo) & 63;

        uint32_t csv[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int z = czl + k / 3 - 1, y = cyl + k % 3 - 1;
            const bool in = z >= 0 && z < dimz && y >= 0 && y < dimy;          // wave-uniform
            const uint32_t rowbase = in ? ((uint32_t)z * dimy + y) * dimx : 0u;
            csv[k] = in ? cell_start[rowbase + min(x_lo + lane, x_hi)] : 0u;
        }
        uint32_t s[9], e[9], S[9], off[10];
        off[0] = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            s[k] = __shfl(csv[k], my_s);
            e[k] = __shfl(csv[k], my_e);
            S[k] = __shfl(csv[k], 0);
            off[k + 1] = off[k] + (__shfl(csv[k], x_hi - x_lo) - S[k]);       // 0 for rows outside the grid
        }
        const uint32_t T = off[9];

        float best = kBigF, second = kBigF;
        uint32_t bestpos = 0xffffffffu;
        for (uint32_t W0 = 0; W0 < T; W0 += kCap) {
            const uint32_t W1 = W0 + kCap;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const uint32_t a = off[k] > W0 ? off[k] : W0;
                const uint32_t bnd = off[k + 1] < W1 ? off[k + 1] : W1;
                for (uint32_t f = a + lane; f < bnd; f += 64) {
                    const double4 r = *reinterpret_cast<const double4 *>(&srecs[S[k] + (f - off[k])]);
                    lx[w][f - W0] = (float)r.x;
                    ly[w][f - W0] = (float)r.y;
                    lz[w][f - W0] = (float)r.z;
                    if (SELF) lrow[w][f - W0] = (int)(__double_as_longlong(r.w) & 0xffffffffll);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (inseg) {
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const uint32_t fa = off[k] + (s[k] - S[k]), fb = off[k] + (e[k] - S[k]);
                    const uint32_t lo = fa > W0 ? fa : W0, hi = fb < W1 ? fb : W1;
                    const uint32_t delta = S[k] - off[k];
                    for (uint32_t f = lo; f < hi; ++f) {
                        const uint32_t o = f - W0;
                        const float dx = fx - lx[w][o], dy = fy - ly[w][o], dz = fz - lz[w][o];
                        float d = dx * dx;
                        d = __builtin_fmaf(dy, dy, d);
                        d = __builtin_fmaf(dz, dz, d);
                        if (SELF) d = (lrow[w][o] == qrow) ? kBigF : d;
                        second = __builtin_amdgcn_fmed3f(best, second, d);
                        const bool upd = d < best;
                        best = upd ? d : best;
                        bestpos = upd ? f + delta : bestpos;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (inseg) {
            // certification (bound derived in pccm_brute.hip) + the ring-1 stop rule of k_grid_query
            const double tq = sqrt((double)best) * (1.0 + 0x1.0p-20) + slack32;
            const double thr = tq * tq * (1.0 + 0x1.0p-30) + 1.0e-36;
            bool settled = false;
            double d64 = 0.0;
            int wrow = -1;
            if (bestpos != 0xffffffffu && (double)second > thr) {
                const double4 r = *reinterpret_cast<const double4 *>(&srecs[bestpos]);
                d64 = gdist64(qx, qy, qz, r.x, r.y, r.z);
                wrow = (int)(__double_as_longlong(r.w) & 0xffffffffll);
                double L = INFINITY;
                const double q[3] = {qx, qy, qz};
                const int c[3] = {cx, cy, cz};
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    if (c[a] - 1 > 0) L = fmin(L, (q[a] - (g.org[a] + (double)(c[a] - 1) * g.h[a])) - g.slack[a]);
                    if (c[a] + 1 < g.dim[a] - 1) L = fmin(L, ((g.org[a] + (double)(c[a] + 2) * g.h[a]) - q[a]) - g.slack[a]);
                }
                settled = (L == INFINITY) || (L > 0.0 && d64 < L * L * (1.0 - 0x1.0p-30));
            }
            if (settled) {
                idx_out[qrow - row_base] = wrow;
                d2_out[qrow - row_base] = d64;
            } else {
                const uint32_t pos = atomicAdd(tailcount, 1u);
                *reinterpret_cast<double4 *>(&tail[pos]) = qa;
            }
        }
        pending &= ~seg;
    }
}

// ---- tail: one wave per unsettled query ---------------------------------------------------------------
// The few queries ring 1 could not settle (typically ~2e-4 of uniform data) would cost a whole
// per-thread kernel lifetime (~0.15 ms, all of it dependent-load latency).  Here a wave takes ONE query
// and its lanes take one x-run of the cube [c-r, c+r]^3 each ((2r+1)^2 <= 49 runs), so a ring costs one
// bounds load plus a few batched record loads; the lexicographic (d2, row) minimum is reduced across the
// wave and the stop rule is evaluated wave-uniformly.  Rings already scanned are simply scanned again
// (the minimum is idempotent).  Used when the tail is short; long tails (lattice data, where exact ties
// defeat the fp32 certification) go through the per-thread kernel, which has the parallelism then.
template <bool SELF>
__global__ __launch_bounds__(256) void k_grid_tail_wave(const GridRec *__restrict__ tail, const uint32_t *__restrict__ tailcount,
                                                        GridGeom g, const uint32_t *__restrict__ cell_start,
                                                        const GridRec *__restrict__ srecs, int64_t row_base, double slack32,
                                                        int32_t *__restrict__ idx_out, double *__restrict__ d2_out,
                                                        int32_t *__restrict__ flagged, float *__restrict__ flag_thr,
                                                        uint32_t *__restrict__ nflag)
{
    const uint32_t count = *tailcount;
    if (count > kTailWaveMax) return;                   // long tail: k_grid_query handles it
    const int lane = threadIdx.x & 63;
    const uint32_t wave0 = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = gridDim.x * 4u;
    const int dimx = g.dim[0], dimy = g.dim[1], dimz = g.dim[2];
    for (uint32_t qi = wave0; qi < count; qi += nwaves) {
        const double4 qa = *reinterpret_cast<const double4 *>(&tail[qi]);   // wave-uniform
        const double qx = qa.x, qy = qa.y, qz = qa.z;
        const int qrow = (int)(__double_as_longlong(qa.w) & 0xffffffffll);
        const int cx = cell_coord(qx, g.org[0], g.inv_h[0], dimx);
        const int cy = cell_coord(qy, g.org[1], g.inv_h[1], dimy);
        const int cz = cell_coord(qz, g.org[2], g.inv_h[2], dimz);
        Best b;
        b.d = INFINITY;
        b.idx = 0x7fffffff;
        bool done = false;
        for (int r = 1; r <= kMaxRing && !done; ++r) {
            const int side = 2 * r + 1;
            if (lane < side * side) {
                const int z = cz + lane / side - r, y = cy + lane % side - r;
                if (z >= 0 && z < dimz && y >= 0 && y < dimy) {
                    const uint32_t row = ((uint32_t)z * dimy + y) * dimx;
                    const int x0 = max(cx - r, 0), x1 = min(cx + r, dimx - 1);
                    scan_range<SELF>(srecs, cell_start[row + x0], cell_start[row + x1 + 1], qx, qy, qz, qrow, b);
                }
            }
            const double m = b.d;
            double wm = m;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wm = fmin(wm, __shfl_xor(wm, off));
            int wi = (m == wm) ? b.idx : 0x7fffffff;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) wi = min(wi, __shfl_xor(wi, off));
            b.d = wm;
            b.idx = wi;
            double L = INFINITY;
            const double q[3] = {qx, qy, qz};
            const int c[3] = {cx, cy, cz};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                if (c[a] - r > 0) L = fmin(L, (q[a] - (g.org[a] + (double)(c[a] - r) * g.h[a])) - g.slack[a]);
                if (c[a] + r < g.dim[a] - 1) L = fmin(L, ((g.org[a] + (double)(c[a] + r + 1) * g.h[a]) - q[a]) - g.slack[a]);
            }
            if (L == INFINITY) done = true;
            else if (L > 0.0 && b.d < L * L * (1.0 - 0x1.0p-30)) done = true;
        }
        if (lane == 0) {
            if (done) {
                if (b.idx == 0x7fffffff) { b.idx = -1; b.d = 0.0; }
                idx_out[qrow - row_base] = b.idx;
                d2_out[qrow - row_base] = b.d;
            } else {
                double tq = (b.d == INFINITY) ? 1.0e18 : sqrt(b.d) * (1.0 + 0x1.0p-20) + slack32;
                double thr = tq * tq * (1.0 + 0x1.0p-30) + 1.0e-36;
                float tf = thr > 3.0e38 ? 3.0e38f : (float)thr;
                tf = __uint_as_float(__float_as_uint(tf) + 1u);
                const uint32_t pos = atomicAdd(nflag, 1u);
                flagged[pos] = (int32_t)(qrow - row_base);
                flag_thr[pos] = tf;
            }
        }
    }
}

// ---- host ---------------------------------------------------------------------------------------
static double points_per_cell()
{
    static double k = [] {
        const char *e = getenv("PCCM_GRID_PPC");
        double v = e ? atof(e) : 2.0;
        return (v > 0.05 && v < 64.0) ? v : 2.0;
    }();
    return k;
}

// One geometry for BOTH clouds (union bounding box, cell edge from the mean point count): a query's
// cell in the searched grid is then the cell it was sorted into in its own grid, which is what lets
// the cooperative kernel work on runs of consecutive cells.
static void choose_geometry(const pccm_ctx *ctx, GridGeom &g, int64_t &ncells)
{
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    double npts = 0.0;
    int nset = 0;
    for (int k = 0; k < 2; ++k) {
        const Cloud &c = ctx->cloud[k];
        if (c.n <= 0) continue;
        for (int a = 0; a < 3; ++a) {
            lo[a] = c.bb_min[a] < lo[a] ? c.bb_min[a] : lo[a];
            hi[a] = c.bb_max[a] > hi[a] ? c.bb_max[a] : hi[a];
        }
        npts += (double)c.n;
        ++nset;
    }
    npts /= (nset > 0 ? nset : 1);
    double ext[3];
    int nz = 0;
    double vol = 1.0;
    for (int a = 0; a < 3; ++a) {
        ext[a] = hi[a] - lo[a];
        if (ext[a] > 0.0) { vol *= ext[a]; ++nz; }
    }
    double h = 1.0;
    if (nz > 0) h = pow(vol * points_per_cell() / npts, 1.0 / nz);
    const int64_t cap = 1ll << 27;
    for (int pass = 0; pass < 64; ++pass) {
        ncells = 1;
        for (int a = 0; a < 3; ++a) {
            double d = ext[a] > 0.0 ? ceil(ext[a] / h) : 1.0;
            if (!(d >= 1.0)) d = 1.0;
            if (d > 4096.0) d = 4096.0;
            g.dim[a] = (int)d;
            ncells *= g.dim[a];
        }
        if (ncells <= cap) break;
        h *= 1.26;   // halve the cell count
    }
    for (int a = 0; a < 3; ++a) {
        g.org[a] = lo[a];
        g.h[a] = ext[a] > 0.0 ? ext[a] / g.dim[a] : 1.0;
        g.h[a] *= (1.0 + 0x1.0p-40);   // the top face must map into the last cell
        g.inv_h[a] = 1.0 / g.h[a];
        g.slack[a] = (fabs(g.org[a]) + (g.dim[a] + 2) * g.h[a]) * 0x1.0p-48;
    }
}

static GridGeom geom_of(const Grid &gr)
{
    GridGeom g;
    for (int a = 0; a < 3; ++a) {
        g.dim[a] = gr.dim[a];
        g.org[a] = gr.org[a];
        g.h[a] = gr.h[a];
        g.inv_h[a] = gr.inv_h[a];
        g.slack[a] = (fabs(gr.org[a]) + (gr.dim[a] + 2) * gr.h[a]) * 0x1.0p-48;
    }
    return g;
}

// (re)build the grids of both clouds when either cloud changed or the caches were dropped
static int ensure_grids(pccm_ctx *ctx)
{
    const uint64_t key = ctx->cloud[0].version * 1000003ull + ctx->cloud[1].version + 1;
    bool fresh = true;
    for (int k = 0; k < 2; ++k)
        if (ctx->cloud[k].n > 0 && (ctx->grid[k].version != key || ctx->grid[k].n != ctx->cloud[k].n || !ctx->grid[k].recs.p)) fresh = false;
    if (fresh) return PCCM_OK;
    ProfScope ps(ctx, PCCM_K_GRID_BUILD);
    GridGeom g;
    int64_t ncells;
    choose_geometry(ctx, g, ncells);
    for (int k = 0; k < 2; ++k) {
        const Cloud &c = ctx->cloud[k];
        if (c.n <= 0) continue;
        Grid &gr = ctx->grid[k];
        int rc;
        if ((rc = ensure(ctx, gr.cell_start, (size_t)(ncells + 1) * sizeof(uint32_t)))) return rc;
        if ((rc = ensure(ctx, gr.recs, (size_t)c.n * sizeof(GridRec)))) return rc;
        if ((rc = sort_by_cell(ctx, c.xyz64, 0, c.n, g, ncells, (uint32_t *)gr.cell_start.p, (GridRec *)gr.recs.p))) return rc;
        for (int a = 0; a < 3; ++a) {
            gr.dim[a] = g.dim[a];
            gr.org[a] = g.org[a];
            gr.h[a] = g.h[a];
            gr.inv_h[a] = g.inv_h[a];
        }
        gr.ncells = ncells;
        gr.n = c.n;
        gr.version = key;
    }
    return PCCM_OK;
}

static bool use_coop()
{
    static bool on = [] {
        const char *e = getenv("PCCM_GRID_COOP");
        return !(e && e[0] == '0');
    }();
    return on;
}

int nn_grid(pccm_ctx *ctx, int dir, const Cloud &it, const Cloud &se, bool self, NNResult &res)
{
    const int64_t nq = res.end - res.begin;
    if (nq <= 0) return PCCM_OK;
    const int si = (dir == PCCM_DIR_LEFT) ? 1 : 0;      // searched cloud
    const int ii = (dir == PCCM_DIR_RIGHT) ? 1 : 0;     // iterating cloud
    int rc;
    if ((rc = ensure_grids(ctx))) return rc;
    const Grid &sg = ctx->grid[si];
    const GridGeom g = geom_of(sg);

    const GridRec *qrecs;
    if (res.begin == 0 && res.end == it.n) {
        qrecs = (const GridRec *)ctx->grid[ii].recs.p;   // whole cloud: its own cell-sorted records
    } else {
        ProfScope ps(ctx, PCCM_K_GRID_BUILD);            // shard: sort its rows by the same cells
        if ((rc = ensure(ctx, ctx->g_qrecs, (size_t)nq * sizeof(GridRec)))) return rc;
        if ((rc = sort_by_cell(ctx, it.xyz64, res.begin, nq, g, sg.ncells, nullptr, (GridRec *)ctx->g_qrecs.p))) return rc;
        qrecs = (const GridRec *)ctx->g_qrecs.p;
    }
    if ((rc = ensure(ctx, ctx->flagged, (size_t)nq * sizeof(int32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->flag_thr, (size_t)nq * sizeof(float)))) return rc;
    if ((rc = ensure(ctx, ctx->g_tail, (size_t)nq * sizeof(GridRec)))) return rc;
    if ((rc = ensure(ctx, ctx->g_tailcount, 2 * sizeof(uint32_t)))) return rc;
    PCCM_HIP(hipMemsetAsync(res.nflag_dev, 0, sizeof(uint32_t), ctx->stream));
    const bool exact = it.exact32 && se.exact32;
    const double maxabs = it.maxabs > se.maxabs ? it.maxabs : se.maxabs;
    const double slack32 = exact ? 0.0 : maxabs * 0x1.0p-20;
    const uint32_t *cs = (const uint32_t *)sg.cell_start.p;
    const GridRec *srecs = (const GridRec *)sg.recs.p;
    int32_t *flg = (int32_t *)ctx->flagged.p;
    float *fthr = (float *)ctx->flag_thr.p;
    {
        ProfScope ps(ctx, PCCM_K_GRID_QUERY);
        if (use_coop()) {
            uint32_t *tailcount = (uint32_t *)ctx->g_tailcount.p;
            GridRec *tail = (GridRec *)ctx->g_tail.p;
            PCCM_HIP(hipMemsetAsync(tailcount, 0, sizeof(uint32_t), ctx->stream));
            const int64_t chunks = (nq + 63) / 64;
            dim3 grid((unsigned)((chunks + 3) / 4));
            if (self)
                hipLaunchKernelGGL((k_grid_query_coop<true>), grid, dim3(256), 0, ctx->stream, qrecs, nq, g, cs, srecs, res.begin,
                                   slack32, res.idx, res.d2, tail, tailcount);
            else
                hipLaunchKernelGGL((k_grid_query_coop<false>), grid, dim3(256), 0, ctx->stream, qrecs, nq, g, cs, srecs, res.begin,
                                   slack32, res.idx, res.d2, tail, tailcount);
            // queries ring 1 could not settle: a wave per query when they are few, else the per-thread kernel
            dim3 wgrid((unsigned)(nq < 4096 ? (nq + 3) / 4 : 1024));
            if (self)
                hipLaunchKernelGGL((k_grid_tail_wave<true>), wgrid, dim3(256), 0, ctx->stream, (const GridRec *)tail, tailcount, g, cs,
                                   srecs, res.begin, slack32, res.idx, res.d2, flg, fthr, res.nflag_dev);
            else
                hipLaunchKernelGGL((k_grid_tail_wave<false>), wgrid, dim3(256), 0, ctx->stream, (const GridRec *)tail, tailcount, g, cs,
                                   srecs, res.begin, slack32, res.idx, res.d2, flg, fthr, res.nflag_dev);
            dim3 tgrid((unsigned)(nq < 256 * 256 ? (nq + 255) / 256 : 256));
            if (self)
                hipLaunchKernelGGL((k_grid_query<true>), tgrid, dim3(256), 0, ctx->stream, (const GridRec *)tail, (int64_t)-1, tailcount, g,
                                   cs, srecs, res.begin, slack32, res.idx, res.d2, flg, fthr, res.nflag_dev);
            else
                hipLaunchKernelGGL((k_grid_query<false>), tgrid, dim3(256), 0, ctx->stream, (const GridRec *)tail, (int64_t)-1, tailcount, g,
                                   cs, srecs, res.begin, slack32, res.idx, res.d2, flg, fthr, res.nflag_dev);
        } else {
            dim3 grid((unsigned)((nq + 255) / 256));
            if (self)
                hipLaunchKernelGGL((k_grid_query<true>), grid, dim3(256), 0, ctx->stream, qrecs, nq, (const uint32_t *)nullptr, g, cs,
                                   srecs, res.begin, slack32, res.idx, res.d2, flg, fthr, res.nflag_dev);
            else
                hipLaunchKernelGGL((k_grid_query<false>), grid, dim3(256), 0, ctx->stream, qrecs, nq, (const uint32_t *)nullptr, g, cs,
                                   srecs, res.begin, slack32, res.idx, res.d2, flg, fthr, res.nflag_dev);
        }
        PCCM_HIP(hipGetLastError());
    }
    if ((rc = launch_fallback(ctx, it, se, self, res))) return rc;
    res.stats[1] = 0;
    res.stats[2] = 0;
    return PCCM_OK;
}

void grid_release(pccm_ctx *ctx)
{
    for (int k = 0; k < 2; ++k) {
        if (ctx->grid[k].cell_start.p) (void)hipFree(ctx->grid[k].cell_start.p);
        if (ctx->grid[k].recs.p) (void)hipFree(ctx->grid[k].recs.p);
        ctx->grid[k] = Grid();
    }
    DevBuf *bufs[] = {&ctx->g_cell_of, &ctx->g_hist, &ctx->g_blocksum, &ctx->g_qrecs, &ctx->g_tail, &ctx->g_tailcount};
    for (DevBuf *b : bufs) {
        if (b->p) (void)hipFree(b->p);
        b->p = nullptr;
        b->bytes = 0;
    }
}

void grid_invalidate(pccm_ctx *ctx)
{
    ctx->grid[0].version = 0;
    ctx->grid[1].version = 0;
}

}  // namespace pccm
