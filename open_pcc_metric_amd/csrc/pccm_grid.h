// Shared device/host declarations of the uniform-grid engine (pccm_grid.hip, pccm_gridbuild.hip, pccm_brick.hip).
//
// The grid stands under the two KDTreeFlann builds of open_pcc_metric/cloud_pair.py:65 and the per-point
// search_knn_vector_3d calls of cloud_pair.py:16-32 (see pccm_grid.hip for the search itself).
#pragma once
#include "pccm_internal.h"

namespace pccm {

struct GridGeom {
    int dim[3];
    int morton = 0;      // cells are numbered along a Z-order curve (dims are equal powers of two): the ingest-time spatial sort
    double org[3], h[3], inv_h[3], slack[3];
};

// x | y << 1 | z << 2 per bit: 10 bits per axis
__device__ __forceinline__ uint32_t spread3(uint32_t v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__device__ __forceinline__ int cell_coord(double v, double org, double inv_h, int dim)
{
    double t = floor(__dmul_rn(__dsub_rn(v, org), inv_h));
    t = t < 0.0 ? 0.0 : t;
    const double top = (double)(dim - 1);
    t = t > top ? top : t;
    return (int)t;
}

__device__ __forceinline__ uint32_t cell_linear(const GridGeom &g, double x, double y, double z)
{
    const int cx = cell_coord(x, g.org[0], g.inv_h[0], g.dim[0]);
    const int cy = cell_coord(y, g.org[1], g.inv_h[1], g.dim[1]);
    const int cz = cell_coord(z, g.org[2], g.inv_h[2], g.dim[2]);
    if (g.morton) return spread3((uint32_t)cx) | (spread3((uint32_t)cy) << 1) | (spread3((uint32_t)cz) << 2);
    return ((uint32_t)cz * g.dim[1] + cy) * g.dim[0] + cx;
}

__device__ __forceinline__ double gdist64(double qx, double qy, double qz, double rx, double ry, double rz)
{
    double dx = __dsub_rn(qx, rx), dy = __dsub_rn(qy, ry), dz = __dsub_rn(qz, rz);
    double d = __dmul_rn(dx, dx);
    d = __dadd_rn(d, __dmul_rn(dy, dy));
    d = __dadd_rn(d, __dmul_rn(dz, dz));
    return d;
}

// ---- cell-sorted records -------------------------------------------------------------------------------
// Two layouts.  GridRec (32 B, pccm_internal.h) carries the fp64 position; Rec32 (16 B) carries the fp32
// position and is used when BOTH clouds are fp32-exact (every PLY / voxelised input; the benchmark's clouds):
// (double)(float)x == x then, so every fp64 decision made from a Rec32 is the one made from the originals,
// at half the bytes per candidate.
struct Rec32 {
    float x, y, z;
    int32_t row;
};

struct P3 {            // a record as the kernels compute with it
    double x, y, z;
    int row;
};

__device__ __forceinline__ P3 load_rec(const GridRec *__restrict__ recs, uint32_t p)
{
    const double4 a = *reinterpret_cast<const double4 *>(&recs[p]);
    P3 r;
    r.x = a.x; r.y = a.y; r.z = a.z;
    r.row = (int)(__double_as_longlong(a.w) & 0xffffffffll);
    return r;
}

__device__ __forceinline__ P3 load_rec(const Rec32 *__restrict__ recs, uint32_t p)
{
    const float4 a = *reinterpret_cast<const float4 *>(&recs[p]);
    P3 r;
    r.x = (double)a.x; r.y = (double)a.y; r.z = (double)a.z;
    r.row = __float_as_int(a.w);
    return r;
}

__device__ __forceinline__ void store_rec(GridRec *__restrict__ recs, uint32_t p, const P3 &v)
{
    double4 r;
    r.x = v.x; r.y = v.y; r.z = v.z;
    r.w = __longlong_as_double((long long)(uint32_t)v.row);
    *reinterpret_cast<double4 *>(&recs[p]) = r;
}

__device__ __forceinline__ void store_rec(Rec32 *__restrict__ recs, uint32_t p, const P3 &v)
{
    float4 r;
    r.x = (float)v.x; r.y = (float)v.y; r.z = (float)v.z;      // exact: Rec32 is only used for fp32-exact clouds
    r.w = __int_as_float(v.row);
    *reinterpret_cast<float4 *>(&recs[p]) = r;
}

// ---- build (pccm_gridbuild.hip) --------------------------------------------------------------------------
struct BuildJob {
    const double *x64;   // [.][3]
    const float *x32;    // quad layout (Cloud::xyz32); read instead of x64 when the records are Rec32
    int64_t row0, n;     // rows [row0, row0 + n)
    uint32_t *cs;        // this job's cell starts, [ncells + 1]
    uint32_t *occ = nullptr;     // occupancy bitmap to write next to cs: one bit per cell, set when the cell holds a record
                                 // ([ncells / 32 + 2] words; voxelised pairs, pccm_lattice.hip), or null
    const Rec32 *sp = nullptr;   // the cloud in its ingest-time spatial order ({x, y, z, original row}; Cloud::sp), read instead
                                 // of x32 when the job covers the whole cloud: the sort's scatters then stay inside a few bins
};

struct BuildJobs {
    BuildJob j[2];
    int njobs;
    int64_t total;
};

// Counting sort of the jobs' rows by cell into `recs` (job 0's records, then job 1's); on return every job's
// cs[c] holds the position of the first record of cell c RELATIVE to the job's first record and cs[ncells] the
// number of records of the job.
int sort_by_cell(pccm_ctx *ctx, const BuildJobs &jobs, const GridGeom &g, int64_t ncells, void *recs, bool rec32,
                 uint32_t *zero = nullptr, int nzero = 0);   // zero: up to 256 words the last kernel clears for the caller

// scratch sizes (so that callers can allocate before a graph capture)
int64_t scan_tiles(int64_t m);

// ---- result of one query (all grid kernels and the exact rescan write through this) ---------------------------
// A settled query leaves ONE record in row order: squared distance, signed point-to-plane projection (0 when no
// normals were attached: pccm_nn_fuse) and -- when somebody will ask for it (pccm_nn_want_idx: getters, colour
// metrics) -- the matched row: 32 bytes = one full sector, written by two 16-byte stores; without the row 16 bytes, one
// store, and a column the reductions read densely.  The reductions read the records directly.
struct NNOut {
    double *rec;           // [rows of the shard][stride], indexed row - row_base
    int stride;            // doubles per record: 4 = {d2, projection, row bits, -}, 2 = {d2, projection}
    const double *nrm;     // normals of the searched cloud, [.][3], or null: projection not fused
    const float4 *nrm32;   // the same normals as aligned 16-byte words when they are fp32-exact (the brick kernel's gather), or null
    int64_t row_base;
    int normal_mode;       // PCCM_NORMAL_ROW / PCCM_NORMAL_NEIGHBOUR
    int layout;            // 0: {d2, projection[, row, -]}; 1: the MATCHED RECORD itself, {rx, ry, rz (fp32), row} = 16 bytes (stride 2
                           //    doubles; Rec32 grids: the fp32 coordinates are exact).  Squared distance and row-indexed
                           //    projection are formed by the reduction that reads the records, from the iterating cloud's rows
                           //    and the searched cloud's normals -- both read in row order there, i.e. coalesced -- with the
                           //    search's own expressions, bit for bit.  No search kernel gathers a normal (round 2's brick
                           //    kernel: 2 M random 16-byte reads per launch), a result is one 16-byte store, and the matched
                           //    row always comes along.
};

// (non-temporal stores here: the brick kernel 90 instead of 70 us, the step 0.194 instead of 0.181 ms -- round 3)
__device__ __forceinline__ void store_result_rec(const NNOut &o, int qrow, float rx, float ry, float rz, int wrow)
{
    reinterpret_cast<float4 *>(o.rec)[qrow - o.row_base] = make_float4(rx, ry, rz, __int_as_float(wrow));
}

__device__ __forceinline__ void store_result(const NNOut &o, int qrow, double d2, double p, int wrow)
{
    double *dst = o.rec + (int64_t)(qrow - o.row_base) * o.stride;
    *reinterpret_cast<double2 *>(dst) = make_double2(d2, p);
    if (o.stride == 4) *reinterpret_cast<double2 *>(dst + 2) = make_double2(__longlong_as_double((long long)(uint32_t)wrow), 0.0);
}

__device__ __forceinline__ void emit_result(const NNOut &o, int qrow, double qx, double qy, double qz, int wrow, double d2,
                                            double rx, double ry, double rz)
{
    if (o.layout == 1) {                        // (no neighbour -- a self search on a one-point cloud: the point itself, d2 = 0)
        const bool has = wrow >= 0;
        store_result_rec(o, qrow, (float)(has ? rx : qx), (float)(has ? ry : qy), (float)(has ? rz : qz), wrow);
        return;
    }
    double p = 0.0;
    if (o.nrm && wrow >= 0) {
        // metric.py:146-153: err . normal_other[row]; FMA chain as np.dot evaluates it (see pccm_point.hip, K3)
        const int64_t k = (o.normal_mode == PCCM_NORMAL_ROW) ? (int64_t)qrow : (int64_t)wrow;
        const double ex = __dsub_rn(qx, rx), ey = __dsub_rn(qy, ry), ez = __dsub_rn(qz, rz);
        p = __dmul_rn(ex, o.nrm[3 * k]);
        p = __fma_rn(ey, o.nrm[3 * k + 1], p);
        p = __fma_rn(ez, o.nrm[3 * k + 2], p);
    }
    store_result(o, qrow, d2, p, wrow);
}


// lookup form: the winner's coordinates come from the searched cloud's fp64 rows (rare paths: tails, rescans)
__device__ __forceinline__ void emit_result_lookup(const NNOut &o, const double *__restrict__ s64, int qrow, double qx, double qy,
                                                   double qz, int wrow, double d2)
{
    double rx = 0.0, ry = 0.0, rz = 0.0;
    if ((o.nrm || o.layout == 1) && wrow >= 0) {
        rx = s64[3 * (int64_t)wrow];
        ry = s64[3 * (int64_t)wrow + 1];
        rz = s64[3 * (int64_t)wrow + 2];
    }
    emit_result(o, qrow, qx, qy, qz, wrow, d2, rx, ry, rz);
}

// ---- query jobs (pccm_grid.hip launches; pccm_brick.hip holds the LDS-brick kernel) ----------------------
struct QueryJob {
    const void *qrecs;          // cell-sorted queries of this job, [nq] (GridRec or Rec32: Grid::rec32)
    const void *qbase;          // first record of the query cloud / shard: what its cell starts (qcs) index
    const uint32_t *qcs;        // query cloud's (or shard's) cell starts
    int64_t nq, nchunks;        // nchunks = ceil(nq / 64)
    const uint32_t *cs;         // searched cloud's cell starts (positions in srecs)
    const uint32_t *occ;        // ... and its occupancy bitmap (voxelised pairs), or null
    const uint32_t *vbricks = nullptr;   // voxel-brick grids (pccm_vox.hip): the searched cloud's bricks, 32 words at the index of
                                         // a cell's first record ...
    const uint32_t *vlist = nullptr;     // ... the ITERATING cloud's occupied cells, in cell order ...
    const uint32_t *vcount = nullptr;    // ... and how many they are
    const int32_t *vminrow = nullptr;    // voxel-brick grids built with rows: the searched cloud's smallest row per occupied voxel
    const void *srecs;          // searched cloud's records
    const double *s64;          // searched cloud's fp64 rows (emit_result_lookup)
    int64_t row_base;           // first row of the shard (outputs are indexed row - row_base)
    double slack32;             // fp32 rounding slack of inexact inputs (see pccm_brute.hip); 0 = both clouds fp32-exact
    NNOut out;
    void *tail;                 // queries ring 1 could not settle (same record type)
    uint32_t *counters;         // [0] = queries handed to the exact full rescan (k2b_fallback), [1] = tail length
    int32_t *flagged;           // ... their rows (relative to row_base) and fp32 filter thresholds
    float *flag_thr;
};

struct QueryJobs {
    QueryJob j[2];
    int njobs;
    uint32_t *err = nullptr;    // the context's device error word (pinned host memory; pccm_ctx::host_err), or null
};

// bits of the device error word: a kernel that meets a state it cannot be in raises one instead of answering wrongly in silence
constexpr uint32_t kErrTailWait = 1u;      // k_grid_tail: the rescan's wait for the tail workgroups ran out
constexpr uint32_t kErrVoxState = 2u;      // pccm_vox.hip: a record outside its tile / a query outside its cell

constexpr int kMaxRing = 3;

// distance from q to the nearest face of the cube [c-r, c+r]^3 that still has cells behind it
__device__ __forceinline__ double face_bound(const GridGeom &g, double qx, double qy, double qz, int cx, int cy, int cz, int r)
{
    double L = INFINITY;
    const double q[3] = {qx, qy, qz};
    const int c[3] = {cx, cy, cz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (c[a] - r > 0) L = fmin(L, (q[a] - (g.org[a] + (double)(c[a] - r) * g.h[a])) - g.slack[a]);
        if (c[a] + r < g.dim[a] - 1) L = fmin(L, ((g.org[a] + (double)(c[a] + r + 1) * g.h[a]) - q[a]) - g.slack[a]);
    }
    return L;
}

__device__ __forceinline__ bool settled_by(double L, double d)
{
    return (L == INFINITY) || (L > 0.0 && d < L * L * (1.0 - 0x1.0p-30));
}

// voxel-brick grids (pccm_vox.hip): cells of 8 x 8 x 8 voxels, a 512-bit occupancy brick + a brick of the voxels that hold several
// points per occupied cell
struct VoxBuildJob {
    const uint32_t *cs;     // the cloud's cell starts
    const void *recs;       // its cell-sorted Rec32 records
    uint32_t *bricks;       // [records][32]
    const uint32_t *occ;    // the cloud's occupancy bitmap (one bit per cell, written by the build: BuildJob::occ)
    uint32_t *list;         // occupied cells in cell order, [<= records]
    uint32_t *count;        // their number
    int32_t *minrow = nullptr;   // [records]: smallest row of every occupied voxel, at the cell's first record + the voxel's rank among
                                 // the brick's set bits (what the search with matched rows gathers), or null: distances-only grid
};
struct VoxBuild {
    VoxBuildJob j[2];
    int njobs;
    int64_t ncells;
    uint32_t *err = nullptr;    // device error word (QueryJobs::err)
};
int launch_vox_bricks(pccm_ctx *ctx, const VoxBuild &vb, const GridGeom &g);
int launch_vox_query(pccm_ctx *ctx, const QueryJobs &jobs, const GridGeom &g, bool self, bool rows);

// per-thread search for voxelised (integer-valued) pairs on Rec32 grids (pccm_lattice.hip)
int launch_lattice_query(pccm_ctx *ctx, const QueryJobs &jobs, const GridGeom &g, bool self);

// LDS-brick ring-1 kernel for Rec32 grids (pccm_brick.hip)
int launch_brick_query(pccm_ctx *ctx, const QueryJobs &jobs, const GridGeom &g, bool self);

}  // namespace pccm
