// Uniform-grid exact 1-NN engine for gfx950.
//
// Replaces the same reference interface as the brute-force engine -- get_neighbour_cloud(),
// open_pcc_metric/cloud_pair.py:10-42, and the KD-tree builds of cloud_pair.py:65 -- with O(N) work:
// both clouds are counting-sorted into ONE uniform grid geometry (the analogue of the reference's two
// KDTreeFlann builds), every query scans the 3x3x3 cells around its own cell and widens ring by ring
// until the best distance found is provably smaller than the distance to anything not scanned yet.
// Winners are decided in fp64 with the reference's arithmetic ((dx*dx)+(dy*dy))+(dz*dz) (no FMA), exact
// ties go to the smallest original row: results are bit-identical to the brute-force engine and to the
// oracle.
//
// Data layout in HBM: cell-sorted records of BOTH clouds in one array (cloud A's records, then cloud B's), cells in
// x-fastest order, so the 3 cells of one x-run are one contiguous range and a 3x3x3 neighbourhood is 9 ranges.
// Record = Rec32 {float x, y, z; int32 row} (16 B) when both clouds are fp32-exact, else GridRec
// {double x, y, z; int32 row; pad} (32 B) (pccm_grid.h); cell_start is uint32[2][ncells + 1] holding positions in
// the combined array.  Queries are processed in cell-sorted order (the iterating cloud's own records, or a
// shard's rows sorted by the same cells).  Results leave as 32-byte records in row order (NNOut, pccm_grid.h).
//
// Launch structure (everything that exists per direction or per cloud is fused into one launch over "jobs"):
//   build   pccm_gridbuild.hip: bin count -> scan -> bin scatter -> bin sort (LDS atomics only)
//   query   well-filled x-rows (volumetric float data):
//             fp32-exact clouds: k_brick_query (pccm_brick.hip; LDS brick, fp32 filter, fp64 certification,
//               fused D2 projection; both directions)
//             otherwise: k_grid_query_coop (per-wave staging, segment-local fp32)
//             -> k_grid_tail, first half: few unsettled queries one wave each (wave_tail), many one thread each
//                (thread_search), rings 2..kMaxRing; second half of the same launch: the exact rescan below
//           surfaces and integer lattices (decide_scale, use_coop):
//             k_grid_query = thread_search for every query, fp64 throughout
//           -> the exact rescan (pccm_rescan.h; behind the per-thread kernels as k2b_fallback, pccm_brute.hip): scan of the whole
//              searched cloud for the queries kMaxRing rings could not settle (flagged list; its length is read on the device)
//
// Exactness of the stop rule.  cell(x) = clamp(floor((x - org) * inv_h)) is monotonic in x, so a
// point in a cell left of cell c lies below org + c*h up to a few ulps of the grid's size; the
// kernels subtract that slack (g.slack[a]) from every face distance and compare with a strict
// "<" after shrinking the bound by 2^-30, so a stop is never taken on a rounding coincidence.
#include "pccm_grid.h"
#include "pccm_rescan.h"

namespace pccm {

// tails up to this many queries take the wave-per-query search, longer ones (lattice data, where exact ties defeat the fp32
// certification of the cooperative kernel) the thread-per-query one.  (16384 until round 4: uniform clouds of 32M points leave
// 25 000 queries per direction -- 8e-4 of them -- to the tail, and the thread-per-query search took 301 us for them where the
// wave-per-query one takes ~60)
constexpr uint32_t kTailWaveMax = 1u << 18;

// per-cell histogram of one cloud (measure_occupancy: cell-edge decision, once per pair of clouds)
__global__ __launch_bounds__(256) void k_cell_hist(const double *__restrict__ x64, int64_t n, GridGeom g, uint32_t *__restrict__ hist)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *p = x64 + 3 * i;
    atomicAdd(&hist[cell_linear(g, p[0], p[1], p[2])], 1u);
}

// ---- query --------------------------------------------------------------------------------------------
// QueryJob / QueryJobs / NNOut: pccm_grid.h.  Every kernel below is templated on the record type of the grid
// (GridRec: fp64 positions; Rec32: fp32-exact clouds, half the bytes per candidate).
struct Best {
    double d;
    int idx;
};

__device__ __forceinline__ void consider(const P3 &a, double qx, double qy, double qz, int qrow, bool self, Best &b)
{
    const double d = gdist64(qx, qy, qz, a.x, a.y, a.z);
    bool better = d < b.d || (d == b.d && a.row < b.idx);
    if (self) better = better && (a.row != qrow);
    b.d = better ? d : b.d;
    b.idx = better ? a.row : b.idx;
}

// Scan records [s, e) kBatch at a time: the loads of a batch are independent and issued together
// (one memory round trip per batch instead of one per record); indices past the end are clamped to
// e-1, and re-evaluating a record is harmless because the lexicographic min is idempotent.
template <typename REC, bool SELF, int kBatch = 4>
__device__ __forceinline__ void scan_range(const REC *__restrict__ recs, uint32_t s, uint32_t e, double qx, double qy,
                                           double qz, int qrow, Best &b)
{
    for (uint32_t p = s; p < e; p += kBatch) {
        P3 a[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) a[j] = load_rec(recs, (p + j < e) ? p + j : e - 1);
#pragma unroll
        for (int j = 0; j < kBatch; ++j) consider(a[j], qx, qy, qz, qrow, SELF, b);
    }
}

// ... the same with the winner's coordinates kept (wave_tail: what the result record or the fused projection needs is then in
// registers when the search ends)
struct BestAt {
    double d, x, y, z;
    int idx;
};

template <typename REC, bool SELF, int kBatch>
__device__ __forceinline__ void scan_range_at(const REC *__restrict__ recs, uint32_t s, uint32_t e, double qx, double qy, double qz, int qrow,
                                              BestAt &b)
{
    for (uint32_t p = s; p < e; p += kBatch) {
        P3 a[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) a[j] = load_rec(recs, (p + j < e) ? p + j : e - 1);
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const double d = gdist64(qx, qy, qz, a[j].x, a[j].y, a[j].z);
            bool better = d < b.d || (d == b.d && a[j].row < b.idx);
            if (SELF) better = better && (a[j].row != qrow);
            b.d = better ? d : b.d;
            b.idx = better ? a[j].row : b.idx;
            b.x = better ? a[j].x : b.x;
            b.y = better ? a[j].y : b.y;
            b.z = better ? a[j].z : b.z;
        }
    }
}

// A query kMaxRing rings could not settle (isolated outliers, clouds that overlap only in part) goes on the
// result's flagged list; the exact rescan (pccm_rescan.h: the second half of k_grid_tail's launch, or k2b_fallback right after a
// per-thread kernel) finds its exact answer by a scan of the whole searched cloud.  Same filter as there: every possible fp64 winner has
// d32 <= thr(best found so far).
__device__ __forceinline__ void defer_rescan(const QueryJob &J, int qrow, double best)
{
    const double tq = (best == INFINITY) ? 1.0e18 : sqrt(best) * (1.0 + 0x1.0p-20) + J.slack32;
    const double thr = tq * tq * (1.0 + 0x1.0p-30) + 1.0e-36;
    float tf = thr > 3.0e38 ? 3.0e38f : (float)thr;
    tf = __uint_as_float(__float_as_uint(tf) + 1u);
    const uint32_t pos = atomicAdd(&J.counters[0], 1u);
    // (write-through stores: the rescan may run in the same launch on another XCD -- k_grid_tail -- and reads the list with
    // agent-scope loads; a release fence per workgroup instead costs a write-back of the XCD's whole L2 each: 57 against 12 us)
    __hip_atomic_store(&J.flagged[pos], qrow - (int)J.row_base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&J.flag_thr[pos], tf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- per-thread ring search: the long-tail kernel, and the whole query for surfaces and lattices ----------
// One query per thread, rings 1..kMaxRing.  from_tail: the queries are the job's tail list and the kernel
// only runs when that list is too long for wave_tail.
template <typename REC, bool SELF>
__device__ __forceinline__ void thread_search(const QueryJobs &jobs, const GridGeom &g, int from_tail, uint32_t nblocks,
                                              uint32_t *retired = nullptr)
{
    const int dimx = g.dim[0], dimy = g.dim[1], dimz = g.dim[2];
    for (int jb = 0; jb < jobs.njobs; ++jb) {
        const QueryJob &J = jobs.j[jb];
        const REC *__restrict__ qrecs = (const REC *)(from_tail ? J.tail : J.qrecs);
        int64_t nq = J.nq;
        if (from_tail) {
            nq = (int64_t)J.counters[1];
            if (nq <= (int64_t)kTailWaveMax) continue;      // short tail: wave_tail handled it
        }
        const uint32_t *__restrict__ cell_start = J.cs;
        const REC *__restrict__ srecs = (const REC *)J.srecs;
        for (int64_t t0 = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63); t0 < nq; t0 += (int64_t)nblocks * 256) {
            const int64_t t = t0 + (threadIdx.x & 63);
            const bool live = t < nq;
            const P3 qa = load_rec(qrecs, (uint32_t)(live ? t : nq - 1));
            const double qx = qa.x, qy = qa.y, qz = qa.z;
            const int qrow = qa.row;
            const int cx = cell_coord(qx, g.org[0], g.inv_h[0], dimx);
            const int cy = cell_coord(qy, g.org[1], g.inv_h[1], dimy);
            const int cz = cell_coord(qz, g.org[2], g.inv_h[2], dimz);
            Best b;
            b.d = INFINITY;
            b.idx = 0x7fffffff;
            bool done = !live;
            if (live) {
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
                for (int k = 0; k < 9; ++k) scan_range<REC, SELF>(srecs, rs[k], re[k], qx, qy, qz, qrow, b);
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
                                scan_range<REC, SELF>(srecs, cell_start[row + x0], cell_start[row + x1 + 1], qx, qy, qz, qrow, b);
                            } else {                                    // interior of the shell: only the two end cells
                                if (cx - r >= 0)
                                    scan_range<REC, SELF>(srecs, cell_start[row + cx - r], cell_start[row + cx - r + 1], qx, qy, qz, qrow, b);
                                if (cx + r <= dimx - 1)
                                    scan_range<REC, SELF>(srecs, cell_start[row + cx + r], cell_start[row + cx + r + 1], qx, qy, qz, qrow, b);
                            }
                        }
                    }
                }
                done = settled_by(face_bound(g, qx, qy, qz, cx, cy, cz, r), b.d);
            }
            if (done && live) {
                if (b.idx == 0x7fffffff) { b.idx = -1; b.d = 0.0; }   // SELF on a one-point cloud (host handles it earlier)
                emit_result_lookup(J.out, J.s64, qrow, qx, qy, qz, b.idx, b.d);
            }
            if (!done) defer_rescan(J, qrow, b.d);           // still open after kMaxRing rings
            if (from_tail && retired) {                      // (k_grid_tail: the rescan half of the launch waits for this count)
                const unsigned long long m = __ballot(live);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if ((threadIdx.x & 63) == 0 && m)
                    __hip_atomic_fetch_add(&retired[(jb * 8 + (blockIdx.x & 7u)) * 32], (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---- cooperative ring-1 kernel for clouds that are NOT fp32-exact (GridRec grids) ----------------------------
// (fp32-exact pairs take the LDS-brick kernel of pccm_brick.hip instead.)
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
// fp32 works on coordinates relative to the segment's corner (subtracted in fp64 from queries and candidates
// alike, so it cancels in exact arithmetic): the rounding of the fp64 -> fp32 conversion scales with the
// local extent instead of the coordinate magnitude (geo-referenced clouds: |x| ~ 1e6 m at mm resolution).
// Queries that cannot be certified (near ties) or whose ring-1 result does not satisfy the stop rule go
// to `tail`.  Integer-valued clouds, where exact ties are the rule, never come here (use_coop).
constexpr int kCap = 384;         // fp32 records staged per wave (4.5 KB); more -> several windows
constexpr int kSegWidth = 61;     // + 3 bounds = 64 lanes
constexpr float kBigF = 3.0e38f;

template <bool SELF>
__global__ __launch_bounds__(256) void k_grid_query_coop(QueryJobs jobs, GridGeom g)
{
    __shared__ float lx[4][kCap + 1], ly[4][kCap + 1], lz[4][kCap + 1];
    __shared__ int lrow[SELF ? 4 : 1][SELF ? kCap + 1 : 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // XCD-aware order: workgroups b, b+8, b+16, ... share an XCD (and its 4 MB L2), so give every XCD one
    // contiguous eighth of the cell-sorted chunk list = one slab of the grid.  Speed only.
    const uint32_t nblk = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t bq = nblk >> 3, br = nblk & 7u;
    const uint32_t vb = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + slot;
    int64_t chunk = (int64_t)vb * 4 + w;
    const int jb = (jobs.njobs > 1 && chunk >= jobs.j[0].nchunks) ? 1 : 0;     // wave-uniform
    if (jb) chunk -= jobs.j[0].nchunks;
    const QueryJob &J = jobs.j[jb];
    const int64_t nq = J.nq;
    if (chunk * 64 >= nq) return;                       // wave-uniform
    const uint32_t *__restrict__ cell_start = J.cs;
    const GridRec *__restrict__ srecs = (const GridRec *)J.srecs;
    const int64_t t = chunk * 64 + lane;
    const bool valid = t < nq;
    const double4 qa = *reinterpret_cast<const double4 *>(&((const GridRec *)J.qrecs)[valid ? t : nq - 1]);
    const double qx = qa.x, qy = qa.y, qz = qa.z;
    float fx, fy, fz;
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
        const double ox = g.org[0] + (double)x_lo * g.h[0];
        const double oy = g.org[1] + (double)cyl * g.h[1];
        const double oz = g.org[2] + (double)czl * g.h[2];
        fx = (float)(qx - ox);
        fy = (float)(qy - oy);
        fz = (float)(qz - oz);

        // per run k: wave-uniform S (first record), off (start in the flattened candidate list) in
        // SGPRs; per lane only the flat start and the length of its own three-cell range
        uint32_t S[9], off[10], rs[9], rl[9];
        off[0] = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int z = czl + k / 3 - 1, y = cyl + k % 3 - 1;
            const bool in = z >= 0 && z < dimz && y >= 0 && y < dimy;          // wave-uniform
            const uint32_t rowbase = in ? ((uint32_t)z * dimy + y) * dimx : 0u;
            const uint32_t csv = in ? cell_start[rowbase + min(x_lo + lane, x_hi)] : 0u;
            const uint32_t s = __shfl(csv, my_s), e = __shfl(csv, my_e);
            S[k] = __builtin_amdgcn_readfirstlane(csv);                          // lane 0 holds the bound of x_lo
            const uint32_t E = __builtin_amdgcn_readfirstlane(__shfl(csv, x_hi - x_lo));
            off[k + 1] = off[k] + (E - S[k]);                                  // 0 for rows outside the grid
            rs[k] = off[k] + (s - S[k]);
            rl[k] = e - s;
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
                    lx[w][f - W0] = (float)(r.x - ox);
                    ly[w][f - W0] = (float)(r.y - oy);
                    lz[w][f - W0] = (float)(r.z - oz);
                    if (SELF) lrow[w][f - W0] = (int)(__double_as_longlong(r.w) & 0xffffffffll);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (inseg) {
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const uint32_t fa = rs[k], fb = rs[k] + rl[k];
                    const uint32_t lo = fa > W0 ? fa : W0, hi = fb < W1 ? fb : W1;
                    const uint32_t delta = S[k] - off[k];                         // flat position -> record in srecs
                    // two candidates per trip: their LDS reads are issued together, so the LDS latency is
                    // paid once per pair (slot o+1 always exists: the arrays carry one spare element)
                    for (uint32_t f = lo; f < hi; f += 2) {
                        const uint32_t o = f - W0;
                        const bool two = f + 1 < hi;
                        const float x0 = lx[w][o], x1 = lx[w][o + 1];
                        const float y0 = ly[w][o], y1 = ly[w][o + 1];
                        const float z0 = lz[w][o], z1 = lz[w][o + 1];
                        const float ax = fx - x0, ay = fy - y0, az = fz - z0;
                        const float bx = fx - x1, by = fy - y1, bz = fz - z1;
                        float d0 = ax * ax, d1 = bx * bx;
                        d0 = __builtin_fmaf(ay, ay, d0);
                        d1 = __builtin_fmaf(by, by, d1);
                        d0 = __builtin_fmaf(az, az, d0);
                        d1 = __builtin_fmaf(bz, bz, d1);
                        if (SELF) {
                            d0 = (lrow[w][o] == qrow) ? kBigF : d0;
                            d1 = (lrow[w][o + 1] == qrow) ? kBigF : d1;
                        }
                        d1 = two ? d1 : kBigF;
                        second = __builtin_amdgcn_fmed3f(best, second, d0);
                        const bool u0 = d0 < best;
                        best = u0 ? d0 : best;
                        bestpos = u0 ? f + delta : bestpos;
                        second = __builtin_amdgcn_fmed3f(best, second, d1);
                        const bool u1 = d1 < best;
                        best = u1 ? d1 : best;
                        bestpos = u1 ? f + 1 + delta : bestpos;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (inseg) {
            // certification (bound derived in pccm_brute.hip) + the ring-1 stop rule
            // input-rounding slack: a candidate that can win or tie in fp64 lies within d(best) of the query, so its
            // local coordinates are bounded by the query's plus that distance; each conversion is off by at most
            // 2^-24 of the coordinate, 2^-20 * (|q|_inf + d) covers the two points involved with room to spare
            const double slack = ((double)fmaxf(fmaxf(fabsf(fx), fabsf(fy)), fabsf(fz)) + sqrt((double)best)) * 0x1.0p-20;
            const double tq = sqrt((double)best) * (1.0 + 0x1.0p-20) + slack;
            const double thr = tq * tq * (1.0 + 0x1.0p-30) + 1.0e-36;
            bool settled = false;
            if (bestpos != 0xffffffffu && (double)second > thr) {
                const P3 r = load_rec(srecs, bestpos);
                const double d64 = gdist64(qx, qy, qz, r.x, r.y, r.z);
                settled = settled_by(face_bound(g, qx, qy, qz, cx, cy, cz, 1), d64);
                if (settled) emit_result(J.out, qrow, qx, qy, qz, r.row, d64, r.x, r.y, r.z);
            }
            if (!settled) {
                const uint32_t pos = atomicAdd(&J.counters[1], 1u);
                *reinterpret_cast<double4 *>(&((GridRec *)J.tail)[pos]) = qa;
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
// (the minimum is idempotent).  Long tails (lattice data, where exact ties defeat the fp32
// certification) go through thread_search instead, which has the parallelism then.
template <typename REC, bool SELF>
__device__ __forceinline__ void wave_tail(const QueryJobs &jobs, const GridGeom &g, uint32_t nblocks, uint32_t *retired)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave0 = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = nblocks * 4u;      // (the launch's tail workgroups, not gridDim.x)
    const int dimx = g.dim[0], dimy = g.dim[1], dimz = g.dim[2];
    // The first job's tail is handed out to the waves from the front, the second job's from the back (a wave that took an
    // entry of each job in turn walked the chain of dependent round trips -- tail record, cell starts, records, winner --
    // twice), and the list lengths travel together with the wave's first candidate entry of every job (an entry beyond
    // its list's length is loaded and dropped: the buffers hold nq rows).
    uint32_t cnt[2] = {0u, 0u}, start[2] = {wave0, nwaves - 1u - wave0};
    P3 first[2];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        first[jb].x = first[jb].y = first[jb].z = 0.0;
        first[jb].row = 0;
        if (jb < jobs.njobs) {
            const QueryJob &J = jobs.j[jb];
            cnt[jb] = J.counters[1];
            if ((int64_t)start[jb] < J.nq) first[jb] = load_rec((const REC *)J.tail, start[jb]);
        }
    }
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        if (jb >= jobs.njobs) continue;
        const QueryJob &J = jobs.j[jb];
        const uint32_t count = cnt[jb];
        if (count > kTailWaveMax) continue;                 // long tail: thread_search handles it
        const uint32_t *__restrict__ cell_start = J.cs;
        const REC *__restrict__ srecs = (const REC *)J.srecs;
        for (uint32_t qi = start[jb]; qi < count; qi += nwaves) {
            const P3 qa = (qi == start[jb]) ? first[jb] : load_rec((const REC *)J.tail, qi);   // wave-uniform
            const double qx = qa.x, qy = qa.y, qz = qa.z;
            const int qrow = qa.row;
            const int cx = cell_coord(qx, g.org[0], g.inv_h[0], dimx);
            const int cy = cell_coord(qy, g.org[1], g.inv_h[1], dimy);
            const int cz = cell_coord(qz, g.org[2], g.inv_h[2], dimz);
            BestAt b;                                          // (the winner's coordinates travel with it: no look-up behind the search)
            b.d = INFINITY;
            b.idx = 0x7fffffff;
            b.x = b.y = b.z = 0.0;
            bool done = false;
            // every query on the tail list has been through a ring-1 kernel that could not settle it -- mostly because the
            // stop rule failed, which a second look at ring 1 cannot change: start with the 5 x 5 x 5 cube (it contains
            // ring 1, scanned here in exact fp64, so uncertified near ties are resolved too)
            for (int r = 2; r <= kMaxRing && !done; ++r) {
                const int side = 2 * r + 1;
                if (lane < side * side) {
                    const int z = cz + lane / side - r, y = cy + lane % side - r;
                    if (z >= 0 && z < dimz && y >= 0 && y < dimy) {
                        const uint32_t row = ((uint32_t)z * dimy + y) * dimx;
                        const int x0 = max(cx - r, 0), x1 = min(cx + r, dimx - 1);
                        scan_range_at<REC, SELF, 8>(srecs, cell_start[row + x0], cell_start[row + x1 + 1], qx, qy, qz, qrow, b);
                    }
                }
                double wm = b.d;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) wm = fmin(wm, __shfl_xor(wm, off));
                int wi = (b.d == wm) ? b.idx : 0x7fffffff;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) wi = min(wi, __shfl_xor(wi, off));
                // a lane that holds the winner hands its coordinates to everybody (rows are unique per record: one such lane)
                const unsigned long long holds = __ballot(b.d == wm && b.idx == wi);
                const int src = holds ? __ffsll((long long)holds) - 1 : 0;
                b.x = __shfl(b.x, src);
                b.y = __shfl(b.y, src);
                b.z = __shfl(b.z, src);
                b.d = wm;
                b.idx = wi;
                done = settled_by(face_bound(g, qx, qy, qz, cx, cy, cz, r), b.d);
            }
            if (lane == 0) {
                if (!done) {                                 // (wave-uniform)
                    defer_rescan(J, qrow, b.d);
                } else {
                    if (b.idx == 0x7fffffff) { b.idx = -1; b.d = 0.0; }
                    emit_result(J.out, qrow, qx, qy, qz, b.idx, b.d, b.x, b.y, b.z);
                }
                // retired: the rescan half of the launch may count on this entry's stores
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(&retired[(jb * 8 + (blockIdx.x & 7u)) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ONE launch finishes whatever the ring-1 kernel left open (round 4; rounds 1-3: two launches, 11.8 + 6.3 us at 1M + 1M points of
// which the second found its list empty):
//   workgroups [0, n_tail): short tails wave-per-query, long tails thread-per-query (each checks the tail length on the device
//     and does nothing when it is not its case); what three rings cannot settle goes on the flagged list (write-through stores).
//     A wave that has retired tail entries -- result or flagged entry stored and drained -- adds their number to its XCD's shard
//     of the job's count (one add per retired query: ~600 per launch at 1M + 1M points, spread over eight lines; a count of
//     finished WORKGROUPS on one word cost 6-12 us of serialised atomics for 512-1024 workgroups);
//   workgroups [n_tail, n_tail + njobs * n_rescan): the exact rescan of the flagged lists (pccm_rescan.h) -- they wait until
//     their job's tail list (complete since the previous launch) is retired entirely (one lane polls, with s_sleep) and run
//     k2b_fallback's body on a flagged list that is complete by then, reading it with agent-scope loads.
// The wait always ends: the tail workgroups never wait for anybody, and the waiting ones are too few (<= 256 workgroups of 256
// threads) to keep them off the GPU whatever the dispatch order; should the poll still run out (seconds), the workgroup raises
// the context's device error word (host memory: the next call that hands results out fails with PCCM_E_STATE) instead of spinning on.
struct TailSync {
    uint32_t *retired;       // [2 jobs][8 shards] tail entries retired (settled or flagged), one 128-byte line per shard
    uint32_t *ticket;        // rescan workgroups that are through (the last one rearms everything)
    uint32_t *host_err;      // the context's device error word (pinned host memory), or null
    uint32_t n_tail, n_rescan;
    uint32_t delay, nap;     // the poll: s_sleep(127) units before the first look, s_sleep(16) units between looks
};
constexpr int kTailShardWords = 32;     // words between two shards of TailSync::retired

template <typename REC, bool SELF>
__global__ __launch_bounds__(256) void k_grid_tail(QueryJobs jobs, GridGeom g, RescanJobs rjobs, TailSync ts)
{
    if (blockIdx.x < ts.n_tail) {
        wave_tail<REC, SELF>(jobs, g, ts.n_tail, ts.retired);
        thread_search<REC, SELF>(jobs, g, 1, ts.n_tail, ts.retired);
        return;
    }
    const uint32_t rb = blockIdx.x - ts.n_tail, jb = rb / ts.n_rescan;
    __shared__ uint32_t s_ok;
    if (threadIdx.x == 0) {
        // this job's tail list is complete (the ring-1 kernel of the previous launch wrote it): wait until all of it is retired
        const uint32_t want = __hip_atomic_load(&jobs.j[jb].counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t ok = want == 0u ? 1u : 0u;          // (an empty tail list -- the rule on voxelised content -- has nothing to wait for)
        for (uint32_t k = 0; k < ts.delay && !ok; ++k) __builtin_amdgcn_s_sleep(127);
        for (uint32_t spin = 0; spin < (1u << 21) && !ok; ++spin) {
            uint32_t got = 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                got += __hip_atomic_load(&ts.retired[(jb * 8 + k) * kTailShardWords], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got >= want) {
                ok = 1u;
                break;
            }
            for (uint32_t k = 0; k < ts.nap; ++k) __builtin_amdgcn_s_sleep(16);
        }
        if (!ok && ts.host_err) atomicOr(ts.host_err, kErrTailWait);
        s_ok = ok;
    }
    __syncthreads();
    if (s_ok) rescan_body<SELF>(rjobs.j[jb], rb % ts.n_rescan, ts.n_rescan);
    __syncthreads();
    if (threadIdx.x == 0) {
        // the last rescan workgroup through rearms the counts for the next launch (every tail entry has been retired by then)
        const uint32_t t = __hip_atomic_fetch_add(ts.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - ts.n_tail - 1u) {
            for (int k = 0; k < 16; ++k) __hip_atomic_store(&ts.retired[k * kTailShardWords], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ts.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// the whole query by the per-thread search (surfaces, lattices; PCCM_GRID_COOP=0)
template <typename REC, bool SELF>
__global__ __launch_bounds__(256) void k_grid_query(QueryJobs jobs, GridGeom g)
{
    thread_search<REC, SELF>(jobs, g, 0, gridDim.x);
}

// ---- host ---------------------------------------------------------------------------------------
static double points_per_cell()
{
    static double k = [] {
        const char *e = PCCM_DIAG_ENV("PCCM_GRID_PPC");
        // 1.4 points per cell: per step at 1M / 4M / 8M points 0.250 / 0.763 / 1.627 ms (1.5: 0.252 / 0.768 / 1.634; 1.3: 0.251 / 0.822 /
        // 1.639; 1.2: 0.253 / 0.803 / 1.825 -- fewer candidates per query, but more tails and more cell starts)
        double v = e ? atof(e) : 1.4;
        return (v > 0.05 && v < 64.0) ? v : 1.4;
    }();
    return k;
}

// Which query kernel?  (PCCM_GRID_COOP=0/1 forces one; default: decided per pair of clouds in decide_scale)
// The cooperative kernel amortises its per-segment work (bounds, staging) over the queries of a wave that share an
// x-row of cells; that pays when rows are well filled -- volumetric float data (1.5x over the per-thread kernel) --
// and backfires when they are not: on surfaces a row holds a handful of points, a wave
// walks a dozen segments with a few lanes active in each (sphere surface, 1M points: 0.83 ms vs 0.28 ms per-thread;
// voxelised: 1.55 vs 0.33 ms), and on integer lattices its in-place tie resolution is no match for plain fp64
// (128^3 lattice volume: 4.4 vs 0.22 ms).  Measured crossover for float data: ~40 points per x-row of the grid.
constexpr double kCoopMinPointsPerRow = 40.0;

static int coop_override()
{
    static int v = [] {
        const char *e = getenv("PCCM_GRID_COOP");
        return e ? (e[0] == '0' ? 0 : 1) : -1;
    }();
    return v;
}

static bool use_coop(const pccm_ctx *ctx)
{
    const int o = coop_override();
    return o >= 0 ? o == 1 : ctx->grid.coop;
}

// One geometry for BOTH clouds (union bounding box, cell edge from the mean point count): a query's
// cell in the searched cloud's grid is then the cell it was sorted into in its own cloud's grid, which is
// what lets the cooperative kernel work on runs of consecutive cells.
static void choose_geometry(const pccm_ctx *ctx, GridGeom &g, int64_t &ncells, double h_scale, bool vox = false, int solo = -1)
{   // solo >= 0: a grid over that cloud alone (its own box and count: estimate_normals), never trimmed
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    double npts = 0.0;
    int nset = 0;
    for (int k = 0; k < 2; ++k) {
        const Cloud &c = ctx->cloud[k];
        if (c.n <= 0 || (solo >= 0 && k != solo)) continue;
        for (int a = 0; a < 3; ++a) {
            lo[a] = c.bb_min[a] < lo[a] ? c.bb_min[a] : lo[a];
            hi[a] = c.bb_max[a] > hi[a] ? c.bb_max[a] : hi[a];
        }
        npts += (double)c.n;
        ++nset;
    }
    if (ctx->grid.boxed && solo < 0)     // outliers trimmed away (decide_geometry): they clamp into the boundary cells
        for (int a = 0; a < 3; ++a) {
            lo[a] = ctx->grid.box_lo[a];
            hi[a] = ctx->grid.box_hi[a];
        }
    npts /= (nset > 0 ? nset : 1);
    if (vox) {
        // voxel-brick flavour (pccm_vox.hip; vox_feasible() has checked the box): cells of exactly 8 x 8 x 8 voxels from the
        // integer lower corner of the bounding box -- cell_coord is exact integer arithmetic then
        ncells = 1;
        for (int a = 0; a < 3; ++a) {
            g.org[a] = lo[a];
            g.h[a] = 8.0;
            g.inv_h[a] = 0.125;
            g.dim[a] = (int)floor((hi[a] - lo[a]) * 0.125) + 1;
            g.slack[a] = (fabs(g.org[a]) + (g.dim[a] + 2) * g.h[a]) * 0x1.0p-48;
            ncells *= g.dim[a];
        }
        return;
    }
    double ext[3];
    int nz = 0;
    double vol = 1.0;
    for (int a = 0; a < 3; ++a) {
        ext[a] = hi[a] - lo[a];
        if (ext[a] > 0.0) { vol *= ext[a]; ++nz; }
    }
    double h = 1.0;
    if (nz > 0) h = pow(vol * points_per_cell() / npts, 1.0 / nz) * h_scale;
    const int64_t cap = 1ll << 26;
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

// occupied cells of a histogram (out[0]) and the sum of the squared counts (out[1]: size-biased occupancy)
__global__ __launch_bounds__(256) void k_count_occupied(const uint32_t *__restrict__ hist, int64_t m, unsigned long long *__restrict__ out)
{
    unsigned int c = 0;
    unsigned long long sq = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (int64_t)gridDim.x * 256) {
        const unsigned long long h = hist[i];
        c += h ? 1u : 0u;
        sq += h * h;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c += __shfl_xor(c, off);
        sq += __shfl_xor(sq, off);
    }
    if ((threadIdx.x & 63) == 0 && c) {
        atomicAdd(&out[0], (unsigned long long)c);
        atomicAdd(&out[1], sq);
    }
}

// per-axis histograms (kTrimBins bins over [lo, lo + kTrimBins / inv_w)) of a cloud's coordinates
constexpr int kTrimBins = 1024;
__global__ __launch_bounds__(256) void k_axis_hist(const double *__restrict__ x64, int64_t n, double lo0, double lo1, double lo2,
                                                   double iw0, double iw1, double iw2, unsigned int *__restrict__ hist)
{
    __shared__ unsigned int s_h[3 * kTrimBins];
    for (int k = threadIdx.x; k < 3 * kTrimBins; k += 256) s_h[k] = 0u;
    __syncthreads();
    const double lo[3] = {lo0, lo1, lo2}, iw[3] = {iw0, iw1, iw2};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double t = (x64[3 * i + a] - lo[a]) * iw[a];
            t = t < 0.0 ? 0.0 : (t > (double)(kTrimBins - 1) ? (double)(kTrimBins - 1) : t);   // outside: first / last bin
            atomicAdd(&s_h[a * kTrimBins + (int)t], 1u);
        }
    __syncthreads();
    for (int k = threadIdx.x; k < 3 * kTrimBins; k += 256)
        if (s_h[k]) atomicAdd(&hist[k], s_h[k]);
}

// Cell edge for this pair of clouds.  The volume rule (ppc points per cell of the bounding box) is
// right for space-filling data; point clouds of surfaces (all real PCC content) leave most cells empty
// and pile dozens of points into the occupied ones, which multiplies the candidates per query.  So the
// edge is shrunk until the OCCUPIED cells of the larger cloud hold ~2 ppc points on average (or the cell
// budget is reached).  Costs a few histogram passes and host round trips, once per pair of clouds:
// the result is cached and survives pccm_drop_caches().
//
// Hostile distributions.  A stray point far away blows the bounding box up until the cell budget leaves
// the real data in a handful of cells, and the ring search degenerates into all-pairs inside them.  When
// the size-biased occupancy  sb = sum(h^2) / sum(h)  (the cell population a random point sees; 2.5 for
// uniform data, ~5 for surfaces) explodes.  So the grid covers the per-axis [0.1 %, 99.9 %] quantile box of
// both clouds whenever that is less than half of the bounding box on some axis (coarse LDS histograms,
// repeated while the extent keeps collapsing): cell_coord clamps, so the trimmed points live in boundary
// cells, and face_bound already treats the grid's outer faces as infinitely far, so the search stays
// exact.  If the size-biased occupancy is above kHeavyCell even so, and so high that
// the ring search would evaluate more pairs per query than 1/800 of the other cloud (clumps, duplicates:
// nothing a uniform grid can separate), the pair is marked hostile and PCCM_ENGINE_AUTO takes the
// brute-force engine, whose scan cost does not depend on the distribution.
constexpr double kHeavyCell = 32.0;

struct Occupancy {
    double mean = 0.0, sb = 0.0;   // points per occupied cell: plain and size-biased mean
    int64_t ncells = 0;
};

static int measure_occupancy(pccm_ctx *ctx, const Cloud &c, double scale, Occupancy &o, int solo = -1)
{
    GridGeom g;
    int64_t ncells;
    choose_geometry(ctx, g, ncells, scale, false, solo);
    int rc;
    if ((rc = ensure(ctx, ctx->g_hist, (size_t)(ncells + 1) * sizeof(uint32_t)))) return rc;
    uint32_t *hist = (uint32_t *)ctx->g_hist.p;
    unsigned long long *counter = (unsigned long long *)ctx->stats.p;
    PCCM_HIP(hipMemsetAsync(hist, 0, (size_t)(ncells + 1) * sizeof(uint32_t), ctx->stream));
    PCCM_HIP(hipMemsetAsync(counter, 0, 2 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(k_cell_hist, dim3((unsigned)((c.n + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)c.xyz64, c.n, g, hist);
    hipLaunchKernelGGL(k_count_occupied, dim3(1024), dim3(256), 0, ctx->stream, (const uint32_t *)hist, ncells, counter);
    unsigned long long h[2] = {0, 0};
    PCCM_HIP(hipMemcpyAsync(h, counter, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    o.mean = (double)c.n / (double)(h[0] ? h[0] : 1);
    o.sb = (double)h[1] / (double)c.n;
    o.ncells = ncells;
    return PCCM_OK;
}

// Shrink the cell edge until the cells are no longer crowded (or the cell budget binds).  Two measures, both taken
// over the larger cloud: points per OCCUPIED cell (surfaces leave most of a volume-rule grid empty and pile their
// points into the rest) and the size-biased mean sum(h^2)/sum(h) -- the population of the cell a random POINT sits in,
// which is what a query pays for: scanner data is sparse far out and crowded near the sensor, so its plain mean
// looks harmless (4) while the typical point shares its cell with 50 others.  A step that does not relieve the
// crowding (duplicates: identical points cannot be separated) is taken back.
static int fit_scale(pccm_ctx *ctx, const Cloud &c, double &scale, Occupancy &o, int solo = -1)
{
    const double target = 2.0 * points_per_cell(), target_sb = 4.0 * points_per_cell();
    auto crowding = [&](const Occupancy &q) { return fmax(q.mean / target, q.sb / target_sb); };
    scale = 1.0;
    Occupancy prev;
    double prev_scale = 1.0;
    for (int it = 0; it < 4; ++it) {
        int rc = measure_occupancy(ctx, c, scale, o, solo);
        if (rc) return rc;
        if (it > 0 && crowding(o) > 0.8 * crowding(prev)) {      // no relief: keep the coarser (cheaper) grid
            scale = prev_scale;
            o = prev;
            break;
        }
        if (crowding(o) <= 1.5) break;
        GridGeom g2;
        int64_t nc2;
        const double next = scale * fmax(0.35, pow(1.0 / crowding(o), 1.0 / 2.4));
        choose_geometry(ctx, g2, nc2, next, false, solo);
        if (nc2 == o.ncells) break;                // cell budget or per-axis limit reached
        prev = o;
        prev_scale = scale;
        scale = next;
    }
    return PCCM_OK;
}

// per-axis [eps, 1 - eps] quantile box of both clouds; returns whether it is materially smaller
static int trim_box(pccm_ctx *ctx, bool &changed)
{
    Grid &gr = ctx->grid;
    double lo[3], hi[3];
    int64_t total = 0;
    for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    for (int k = 0; k < 2; ++k) {
        const Cloud &c = ctx->cloud[k];
        if (c.n <= 0) continue;
        total += c.n;
        for (int a = 0; a < 3; ++a) {
            lo[a] = fmin(lo[a], c.bb_min[a]);
            hi[a] = fmax(hi[a], c.bb_max[a]);
        }
    }
    const double lo0[3] = {lo[0], lo[1], lo[2]}, hi0[3] = {hi[0], hi[1], hi[2]};
    int rc = ensure(ctx, ctx->g_blocksum, 3 * kTrimBins * sizeof(unsigned int));
    if (rc) return rc;
    unsigned int *dh = (unsigned int *)ctx->g_blocksum.p;
    std::vector<unsigned int> h(3 * kTrimBins);
    const unsigned long long allow = (unsigned long long)((double)total * 1.0e-3);    // points given up per side and axis
    for (int it = 0; it < 8; ++it) {
        double iw[3];
        for (int a = 0; a < 3; ++a) iw[a] = hi[a] > lo[a] ? (double)kTrimBins / (hi[a] - lo[a]) : 0.0;
        PCCM_HIP(hipMemsetAsync(dh, 0, 3 * kTrimBins * sizeof(unsigned int), ctx->stream));
        for (int k = 0; k < 2; ++k) {
            const Cloud &c = ctx->cloud[k];
            if (c.n <= 0) continue;
            hipLaunchKernelGGL(k_axis_hist, dim3(512), dim3(256), 0, ctx->stream, (const double *)c.xyz64, c.n, lo[0], lo[1],
                               lo[2], iw[0], iw[1], iw[2], dh);
        }
        PCCM_HIP(hipMemcpyAsync(h.data(), dh, 3 * kTrimBins * sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
        PCCM_HIP(hipStreamSynchronize(ctx->stream));
        bool shrunk = false;
        for (int a = 0; a < 3; ++a) {
            if (!(hi[a] > lo[a])) continue;
            const unsigned int *ha = h.data() + a * kTrimBins;
            int b0 = 0, b1 = kTrimBins - 1;
            unsigned long long acc = 0;
            while (b0 < b1 && acc + ha[b0] <= allow) acc += ha[b0++];
            acc = 0;
            while (b1 > b0 && acc + ha[b1] <= allow) acc += ha[b1--];
            const double w = (hi[a] - lo[a]) / kTrimBins;
            const double nlo = lo[a] + b0 * w, nhi = lo[a] + (b1 + 1) * w;
            if ((nhi - nlo) < 0.5 * (hi[a] - lo[a])) shrunk = true;
            lo[a] = nlo;
            hi[a] = nhi < hi[a] ? nhi : hi[a];
        }
        if (!shrunk) break;
    }
    changed = false;
    for (int a = 0; a < 3; ++a)
        if ((hi[a] - lo[a]) < 0.5 * (hi0[a] - lo0[a])) changed = true;
    if (changed) {
        for (int a = 0; a < 3; ++a) {
            gr.box_lo[a] = lo[a];
            gr.box_hi[a] = hi[a];
        }
        gr.boxed = true;
    }
    return PCCM_OK;
}

static int decide_scale(pccm_ctx *ctx, uint64_t key)
{
    Grid &gr = ctx->grid;
    if (gr.scale_key == key) return PCCM_OK;
    if (ctx->capturing) {
        ctx->capture_failed = true;
        return fail(PCCM_E_STATE, "the grid geometry must be decided before graph capture: run pccm_nn once first");
    }
    const int big = ctx->cloud[1].n > ctx->cloud[0].n ? 1 : 0;
    const Cloud &c = ctx->cloud[big];
    // The frames of a sequence -- or the decoded versions of one frame -- look alike: same point counts, same
    // bounding box, same kind of coordinates.  The decisions below only steer speed, never results, so a pair that
    // matches the previous one in all of that simply inherits them (0.3-0.7 ms of histogram passes and host round
    // trips saved per fresh pair); anything unusual about the previous pair (trimmed box, brute engine) or any
    // visible change makes the pair decide for itself.
    PairSignature sig;
    for (int k = 0; k < 2; ++k) {
        const Cloud &q = ctx->cloud[k];
        sig.n[k] = q.n;
        sig.flags[k] = (q.exact32 ? 1 : 0) | (q.all_int ? 2 : 0);
        for (int a = 0; a < 3; ++a) { sig.lo[k][a] = q.bb_min[a]; sig.hi[k][a] = q.bb_max[a]; }
    }
    if (gr.scale_key != 0 && !gr.boxed && !gr.hostile && sig.resembles(gr.sig) && !getenv("PCCM_GRID_NO_REUSE")) {
        gr.scale_key = key;
        gr.iso_key = key;                              // the isolation verdict (not hostile) is inherited too
        return PCCM_OK;                                // gr.sig stays that of the pair that decided: no drift
    }
    gr.sig = sig;
    gr.boxed = false;
    gr.hostile = false;
    double scale = 1.0;
    if (c.n > 0) {
        Occupancy o;
        int rc;
        if (!PCCM_DIAG_ENV("PCCM_GRID_NO_TRIM")) {        // first, so that no histogram is ever built over a blown-up box
            bool changed = false;
            if ((rc = trim_box(ctx, changed))) return rc;
        }
        if ((rc = fit_scale(ctx, c, scale, o))) return rc;
        const double n_other = (double)ctx->cloud[1 - big].n > 0 ? (double)ctx->cloud[1 - big].n : (double)c.n;
        // integer-valued clouds (voxelised content, duplicates) are excluded: exact ties are the rule there, the grid's
        // per-thread fp64 search handles them directly while the brute engine would send every query to its exact rescan
        const bool lattice = ctx->cloud[0].all_int && ctx->cloud[1].all_int;
        gr.hostile = !lattice && o.sb > kHeavyCell && o.sb > n_other / 800.0;
        gr.sb = o.sb;
        GridGeom gg;
        int64_t nc;
        choose_geometry(ctx, gg, nc, scale);
        gr.coop = !lattice && (double)c.n / ((double)gg.dim[1] * (double)gg.dim[2]) >= kCoopMinPointsPerRow;
    }
    gr.scale = scale;
    gr.scale_key = key;
    if (getenv("PCCM_DEBUG")) {
        GridGeom g;
        int64_t nc;
        choose_geometry(ctx, g, nc, scale);
        fprintf(stderr, "[pccm] grid %d x %d x %d, scale %.3f, sb %.1f, boxed %d [%g %g %g .. %g %g %g], hostile %d, kernel %s\n",
                g.dim[0], g.dim[1], g.dim[2], scale, gr.sb, (int)gr.boxed, gr.box_lo[0], gr.box_lo[1], gr.box_lo[2], gr.box_hi[0],
                gr.box_hi[1], gr.box_hi[2], (int)gr.hostile, gr.coop ? "cooperative" : "per-thread");
    }
    return PCCM_OK;
}

int grid_decide(pccm_ctx *ctx, bool *hostile)
{
    const uint64_t key = ctx->cloud[0].version * 1000003ull + ctx->cloud[1].version + 1;
    int rc = decide_scale(ctx, key);
    if (rc) return rc;
    *hostile = ctx->grid.hostile;
    return PCCM_OK;
}

// ---- spatial order of a resident cloud (Cloud::sp) ---------------------------------------------------------------
// One counting sort of the cloud along a Z-order curve over its own bounding box (2^b cells per axis, ~4 points per cell):
// the same three kernels as the grid build, with Morton-numbered cells.  Per cloud, once, by the first pccm_drop_caches behind
// a search (the caller is about to rebuild the search structures of resident clouds) -- never part of a step, whose build
// re-sorts this array into the pair's grid every time, and never paid by a pair that is searched once.
// PCCM_SPATIAL=0 switches it off (A/B runs: the build then reads the rows in the caller's order, as in round 2).
int spatial_order(pccm_ctx *ctx, Cloud &c)
{
    c.sp_valid = false;
    static const bool off = [] { const char *e = getenv("PCCM_SPATIAL"); return e && e[0] == '0'; }();
    if (off || !c.exact32 || c.n < 4096) return PCCM_OK;            // small clouds: nothing to gain
    int bits = 4;
    while (bits < 8 && (double)(1ll << (3 * (bits + 1))) * 4.0 <= (double)c.n * 2.0) ++bits;   // ~4 points per cell
    GridGeom g;
    g.morton = 1;
    for (int a = 0; a < 3; ++a) {
        const double ext = c.bb_max[a] - c.bb_min[a];
        g.dim[a] = 1 << bits;
        g.org[a] = c.bb_min[a];
        g.h[a] = ext > 0.0 ? ext / g.dim[a] * (1.0 + 0x1.0p-40) : 1.0;
        g.inv_h[a] = 1.0 / g.h[a];
        g.slack[a] = 0.0;
    }
    const int64_t ncells = 1ll << (3 * bits);
    int rc;
    if ((rc = grow(&c.sp, c.cap_sp, (size_t)c.n * sizeof(Rec32)))) return rc;
    if ((rc = ensure(ctx, ctx->g_hist, (size_t)3 * (ncells + 1) * sizeof(uint32_t)))) return rc;
    ProfScope ps(ctx, PCCM_K_INGEST);
    BuildJobs jobs;
    jobs.njobs = 1;
    jobs.total = c.n;
    jobs.j[0] = {c.xyz64, (const float *)c.xyz32, 0, c.n, (uint32_t *)ctx->g_hist.p};
    jobs.j[1] = jobs.j[0];
    if ((rc = sort_by_cell(ctx, jobs, g, ncells, c.sp, true))) return rc;
    c.sp_valid = true;
    return PCCM_OK;
}

// record layout for the current pair: Rec32 when both clouds are fp32-exact (PCCM_GRID_REC64=1 forces GridRec, for A/B runs)
static bool pair_rec32(const pccm_ctx *ctx)
{
    static const bool force64 = [] { const char *e = getenv("PCCM_GRID_REC64"); return e && e[0] == '1'; }();
    return !force64 && ctx->cloud[0].exact32 && ctx->cloud[1].exact32;
}

// Voxelised pairs on the per-thread path: fp32 arithmetic is exact for every candidate the rings can reach (integers below
// 2^22, cells of at most 256 units: differences below 2^11, squared distances below 2^24) -- pccm_lattice.hip, which asks the
// searched cloud's occupancy bitmap before it touches cell starts.  PCCM_LATTICE=0: the general per-thread kernel.
static bool lattice_pair(const pccm_ctx *ctx, const GridGeom &g, bool rec32)
{
    static const bool off = [] { const char *e = getenv("PCCM_LATTICE"); return e && e[0] == '0'; }();
    const Cloud &c0 = ctx->cloud[0], &c1 = ctx->cloud[1];
    return rec32 && !off && !use_coop(ctx) && c0.all_int && c1.all_int && c0.maxabs < 4194304.0 && c1.maxabs < 4194304.0 &&
           g.h[0] <= 256.0 && g.h[1] <= 256.0 && g.h[2] <= 256.0;
}

// Voxel-brick flavour of the pair's grid (pccm_vox.hip): both clouds voxelised, whole, and a bounding box that 8-voxel cells
// cover within the budgets below.  PCCM_VOX=0 switches it off (A/B runs: the per-thread lattice search then).
static bool vox_feasible(const pccm_ctx *ctx)
{
    static const bool off = [] { const char *e = getenv("PCCM_VOX"); return e && e[0] == '0'; }();
    const Cloud &c0 = ctx->cloud[0], &c1 = ctx->cloud[1];
    static const bool lattice_off = [] { const char *e = getenv("PCCM_LATTICE"); return e && e[0] == '0'; }();
    if (off || lattice_off || use_coop(ctx) || ctx->grid.boxed || c0.n <= 0 || c1.n <= 0 || !c0.exact32 || !c1.exact32 || !c0.all_int ||
        !c1.all_int) return false;
    if (!(c0.maxabs < 4194304.0 && c1.maxabs < 4194304.0)) return false;
    if (c0.n + c1.n > (16ll << 20)) return false;          // brick slots are addressed by record index: 128 B per point reserved (2 GB here)
    double cells = 1.0;
    for (int a = 0; a < 3; ++a) {
        const double lo = fmin(c0.bb_min[a], c1.bb_min[a]), hi = fmax(c0.bb_max[a], c1.bb_max[a]);
        const double d = floor((hi - lo) * 0.125) + 1.0;
        if (!(d <= 2048.0)) return false;
        cells *= d;
    }
    return cells <= (double)(1ll << 24);
}

// (re)build the combined grid when either cloud changed, the caches were dropped or the record layout asked for
// differs from the built one (need64: a caller that reads GridRec records, pccm_normals.hip)
// A shard's rows of the iterating cloud that want the same cell sort as the grid being built: when exactly one cloud is
// (re)built, they ride along as the second job of the same three launches instead of paying three more
struct ShardSort {
    const Cloud *it = nullptr;
    int64_t begin = 0, n = 0;
    bool done = false;
    const char *recs = nullptr;        // where the shard's cell-sorted records went
    const uint32_t *cs = nullptr;      // ... and their cell starts
};

static int ensure_grid(pccm_ctx *ctx, bool need64 = false, int need_mask = 3, uint32_t *zero = nullptr, int nzero = 0, bool *rebuilt = nullptr,
                       ShardSort *shard = nullptr, int want_vox = 0)     // want_vox: 1 voxel bricks, 2 ... with the rows' table
{
    if (rebuilt) *rebuilt = false;
    Grid &gr = ctx->grid;
    const uint64_t key = ctx->cloud[0].version * 1000003ull + ctx->cloud[1].version + 1;
    const bool rec32 = !need64 && pair_rec32(ctx);
    // (the geometry decisions first: which kernel family serves the pair -- Grid::coop, Grid::boxed -- is part of what makes the
    // voxel bricks feasible; cached per pair)
    if (ctx->cloud[0].n > 0 || ctx->cloud[1].n > 0) {
        const int rcd = decide_scale(ctx, key);
        if (rcd) return rcd;
    }
    const bool vox = want_vox && rec32 && !shard && vox_feasible(ctx);
    // a pair that has been asked for matched rows once keeps the rows' table in every later build (a report that alternates
    // distance-only and row requests would otherwise rebuild the grid at every switch)
    if (vox && want_vox >= 2) gr.vox_rows_pair = key;
    const bool vox_rows = vox && (want_vox >= 2 || gr.vox_rows_pair == key);
    const bool same = gr.key == key && gr.n[0] == ctx->cloud[0].n && gr.n[1] == ctx->cloud[1].n && gr.recs.p && gr.rec32 == rec32 &&
                      gr.vox == vox && (!vox_rows || gr.vox_rows) && (!rec32 || gr.lattice == lattice_pair(ctx, geom_of(gr), rec32));
    if (same && (gr.built & need_mask) == need_mask) return PCCM_OK;
    if (same) need_mask |= gr.built;                   // keep what is there, add what is missing
    ProfScope ps(ctx, PCCM_K_GRID_BUILD);
    GridGeom g;
    int64_t ncells;
    choose_geometry(ctx, g, ncells, gr.scale, vox);
    const int64_t n0 = ctx->cloud[0].n, n1 = ctx->cloud[1].n;
    int rc;
    if (vox) {
        if ((rc = ensure(ctx, gr.vbricks, (size_t)(n0 + n1) * 32 * sizeof(uint32_t)))) return rc;
        if ((rc = ensure(ctx, gr.vlist, (size_t)(n0 + n1) * sizeof(uint32_t)))) return rc;
        if ((rc = ensure(ctx, gr.vcount, 2 * sizeof(uint32_t)))) return rc;
        if (vox_rows && (rc = ensure(ctx, gr.vminrow, (size_t)(n0 + n1) * sizeof(int32_t)))) return rc;
    }
    if ((rc = ensure(ctx, gr.cell_start, (size_t)2 * (ncells + 1) * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, gr.recs, (size_t)(n0 + n1 > 0 ? n0 + n1 : 1) * sizeof(GridRec)))) return rc;   // either layout fits
    uint32_t *cs = (uint32_t *)gr.cell_start.p;
    const bool lattice = lattice_pair(ctx, g, rec32);
    const int64_t occ_words = ncells / 32 + 2;
    if (lattice && (rc = ensure(ctx, gr.occ, (size_t)2 * occ_words * sizeof(uint32_t)))) return rc;
    const size_t rsz = rec32 ? sizeof(Rec32) : sizeof(GridRec);
    // cloud k's records live at recs + (k ? n0 : 0) whichever clouds are built; cell starts are relative to that
    BuildJobs jobs;
    jobs.njobs = 0;
    jobs.total = 0;
    char *first = nullptr;
    for (int k = 0; k < 2; ++k) {
        if (!(need_mask & (1 << k))) continue;
        const Cloud &c = ctx->cloud[k];
        if (jobs.njobs == 0) first = (char *)gr.recs.p + (size_t)(k ? n0 : 0) * rsz;
        jobs.j[jobs.njobs] = {c.xyz64, (const float *)c.xyz32, 0, c.n, cs + (size_t)k * (ncells + 1)};
        if (rec32 && c.sp_valid) jobs.j[jobs.njobs].sp = (const Rec32 *)c.sp;
        if (lattice) jobs.j[jobs.njobs].occ = (uint32_t *)gr.occ.p + (size_t)k * occ_words;
        ++jobs.njobs;
        jobs.total += c.n;
    }
    if (jobs.njobs == 1) jobs.j[1] = jobs.j[0];
    if (shard && shard->n > 0 && jobs.njobs == 1 && jobs.total > 0 && rec32 &&
        (size_t)(first - (char *)gr.recs.p) + (size_t)(jobs.total + shard->n) * rsz <= gr.recs.bytes) {
        // (16-byte records fill half of the buffer that is sized for either layout: the shard's records fit behind the cloud's)
        if ((rc = ensure(ctx, ctx->g_hist, (size_t)3 * (ncells + 1) * sizeof(uint32_t)))) return rc;
        jobs.j[1] = {shard->it->xyz64, (const float *)shard->it->xyz32, shard->begin, shard->n, (uint32_t *)ctx->g_hist.p};
        jobs.njobs = 2;
        shard->recs = first + (size_t)jobs.total * rsz;
        shard->cs = (const uint32_t *)ctx->g_hist.p;
        jobs.total += shard->n;
        shard->done = true;
    }
    if (jobs.total > 0 && (rc = sort_by_cell(ctx, jobs, g, ncells, first, rec32, zero, nzero))) return rc;
#ifdef PCCM_DIAG
    // diagnostic builds only (tests/test_gpu_device_errors.py): a cell start that contradicts the records, so that the brick
    // build meets records outside its tile -- the state pccm_vox.hip reports through the device error word
    if (vox && jobs.total > 0 && getenv("PCCM_DIAG_CORRUPT_CS") && ncells > 200) {
        const uint32_t bad = (uint32_t)ctx->cloud[0].n;
        PCCM_HIP(hipMemcpyAsync(cs + 128, &bad, sizeof(bad), hipMemcpyHostToDevice, ctx->stream));
        PCCM_HIP(hipStreamSynchronize(ctx->stream));
    }
#endif
    if (vox && jobs.total > 0) {                           // bricks of the clouds just built
        VoxBuild vb;
        vb.njobs = 0;
        vb.ncells = ncells;
        vb.err = ctx->host_err;
        for (int k = 0; k < 2; ++k) {
            if (!(need_mask & (1 << k))) continue;
            const size_t r0 = (size_t)(k ? n0 : 0);
            vb.j[vb.njobs++] = {cs + (size_t)k * (ncells + 1), (const char *)gr.recs.p + r0 * rsz, (uint32_t *)gr.vbricks.p + r0 * 32,
                                (const uint32_t *)gr.occ.p + (size_t)k * occ_words, (uint32_t *)gr.vlist.p + r0, (uint32_t *)gr.vcount.p + k,
                                vox_rows ? (int32_t *)gr.vminrow.p + r0 : nullptr};
        }
        if (vb.njobs == 1) vb.j[1] = vb.j[0];
        if (vb.njobs > 0 && (rc = launch_vox_bricks(ctx, vb, g))) return rc;
    }
    if (rebuilt) *rebuilt = jobs.total > 0;
    for (int a = 0; a < 3; ++a) {
        gr.dim[a] = g.dim[a];
        gr.org[a] = g.org[a];
        gr.h[a] = g.h[a];
        gr.inv_h[a] = g.inv_h[a];
    }
    gr.ncells = ncells;
    gr.n[0] = n0;
    gr.n[1] = n1;
    gr.key = key;
    gr.rec32 = rec32;
    gr.lattice = lattice;
    gr.vox = vox;
    gr.vox_rows = vox_rows;
    gr.built = need_mask;
    return PCCM_OK;
}

// Queries with no point of the searched cloud within kMaxRing cells end in an exact full rescan each
// (k2b_fallback): fine for stray points, ruinous when a whole region of one cloud has no counterpart
// (clouds that overlap only in part, or not at all).  Counted once per pair of clouds on the freshly
// built grid; beyond ~3 % of the queries the brute-force engine is the cheaper way to be exact.
template <typename REC>
__global__ __launch_bounds__(256) void k_count_isolated(const REC *__restrict__ qrecs, int64_t nq,
                                                        const uint32_t *__restrict__ cs, GridGeom g,
                                                        unsigned long long *__restrict__ out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool isolated = false;
    if (t < nq) {
        const P3 qa = load_rec(qrecs, (uint32_t)t);
        const int dimx = g.dim[0], dimy = g.dim[1], dimz = g.dim[2];
        const int cx = cell_coord(qa.x, g.org[0], g.inv_h[0], dimx);
        const int cy = cell_coord(qa.y, g.org[1], g.inv_h[1], dimy);
        const int cz = cell_coord(qa.z, g.org[2], g.inv_h[2], dimz);
        const int x0 = max(cx - kMaxRing, 0), x1 = min(cx + kMaxRing, dimx - 1);
        isolated = true;
        for (int dz = 0; dz <= 2 * kMaxRing && isolated; ++dz) {
            const int z = cz + ((dz & 1) ? (dz + 1) / 2 : -(dz / 2));          // centre row first
            if (z < 0 || z >= dimz) continue;
            for (int dy = 0; dy <= 2 * kMaxRing; ++dy) {
                const int y = cy + ((dy & 1) ? (dy + 1) / 2 : -(dy / 2));
                if (y < 0 || y >= dimy) continue;
                const uint32_t row = ((uint32_t)z * dimy + y) * dimx;
                if (cs[row + x1 + 1] != cs[row + x0]) {
                    isolated = false;
                    break;
                }
            }
        }
    }
    const unsigned long long m = __ballot(isolated);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, (unsigned long long)__popcll(m));
}

static int check_isolation(pccm_ctx *ctx)
{
    Grid &gr = ctx->grid;
    if (gr.iso_key == gr.scale_key) return PCCM_OK;
    if (ctx->capturing) {
        ctx->capture_failed = true;
        return fail(PCCM_E_STATE, "run pccm_nn once before graph capture");
    }
    const GridGeom g = geom_of(gr);
    unsigned long long *counter = (unsigned long long *)ctx->stats.p;
    PCCM_HIP(hipMemsetAsync(counter, 0, 2 * sizeof(unsigned long long), ctx->stream));
    const uint32_t *cs = (const uint32_t *)gr.cell_start.p;
    for (int ii = 0; ii < 2; ++ii) {                       // queries of cloud ii against the cells of the other cloud
        const int64_t nq = gr.n[ii];
        if (nq <= 0 || gr.n[1 - ii] <= 0) continue;
        dim3 grid((unsigned)((nq + 255) / 256));
        if (gr.rec32)
            hipLaunchKernelGGL((k_count_isolated<Rec32>), grid, dim3(256), 0, ctx->stream,
                               (const Rec32 *)gr.recs.p + (ii ? gr.n[0] : 0), nq, cs + (ii ? 0 : gr.ncells + 1), g, counter + ii);
        else
            hipLaunchKernelGGL((k_count_isolated<GridRec>), grid, dim3(256), 0, ctx->stream,
                               (const GridRec *)gr.recs.p + (ii ? gr.n[0] : 0), nq, cs + (ii ? 0 : gr.ncells + 1), g, counter + ii);
    }
    unsigned long long h[2] = {0, 0};
    PCCM_HIP(hipMemcpyAsync(h, counter, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    for (int ii = 0; ii < 2; ++ii) gr.isolated[ii] = (int64_t)h[ii];
    gr.iso_key = gr.scale_key;
    if (getenv("PCCM_DEBUG")) fprintf(stderr, "[pccm] isolated queries: %lld of %lld, %lld of %lld\n", (long long)h[0],
                                      (long long)gr.n[0], (long long)h[1], (long long)gr.n[1]);
    return PCCM_OK;
}

// Does the built grid call for the other engine?  (PCCM_ENGINE_AUTO only; decided once per pair of clouds)
int grid_prefers_brute(pccm_ctx *ctx, bool *yes)
{
    int rc;
    *yes = false;
    if (ctx->grid.iso_key == ctx->grid.scale_key && ctx->grid.scale_key != 0) {     // verdict already known for this pair
        Grid &g0 = ctx->grid;
        for (int ii = 0; ii < 2; ++ii)
            if (g0.n[ii] > 0 && (double)g0.isolated[ii] > 0.03 * (double)g0.n[ii] && g0.isolated[ii] > 64) *yes = true;
        return PCCM_OK;
    }
    if ((rc = ensure_grid(ctx))) return rc;             // both clouds: the count looks at both directions
    if ((rc = check_isolation(ctx))) return rc;
    Grid &gr = ctx->grid;
    for (int ii = 0; ii < 2; ++ii)
        if (gr.n[ii] > 0 && (double)gr.isolated[ii] > 0.03 * (double)gr.n[ii] && gr.isolated[ii] > 64) *yes = true;
    if (*yes) gr.hostile = true;                           // remembered with the pair: later calls skip the grid build
    return PCCM_OK;
}

// Is the D2 projection of direction `dir` fused into the search?  (pccm_nn_fuse; decided per call, never an error:
// a request that cannot be honoured -- no normals, row-indexed normals out of range -- simply stays unfused and the
// reduction falls back to the separate point kernel, which reports the reference's IndexError)
static int fused_mode(const pccm_ctx *ctx, int dir, const Cloud &it, const Cloud &se)
{
    if (dir == PCCM_DIR_SELF) return -1;
    static const bool off = [] { const char *e = getenv("PCCM_NO_FUSE"); return e && e[0] == '1'; }();   // A/B runs
    if (off) return -1;
    const int mode = ctx->fuse_mode[dir];
    if (mode != PCCM_NORMAL_ROW && mode != PCCM_NORMAL_NEIGHBOUR) return -1;
    if (se.n_nrm <= 0) return -1;
    if (mode == PCCM_NORMAL_ROW && it.n > se.n_nrm) return -1;      // the whole cloud, not the shard: ranks agree
    if (mode == PCCM_NORMAL_NEIGHBOUR && se.n_nrm != se.n) return -1;
    return mode;
}

template <typename REC>
static void launch_queries(pccm_ctx *ctx, const QueryJobs &jobs, const GridGeom &g, bool self, dim3 grid)
{
    if (self) hipLaunchKernelGGL((k_grid_query<REC, true>), grid, dim3(256), 0, ctx->stream, jobs, g);
    else hipLaunchKernelGGL((k_grid_query<REC, false>), grid, dim3(256), 0, ctx->stream, jobs, g);
}

// the one tail launch behind a ring-1 kernel: rings 2..3 for its tail lists, then the exact rescan of what is left (k_grid_tail)
static int launch_tail(pccm_ctx *ctx, const QueryJobs &jobs, const GridGeom &g, bool self, bool rec32, int64_t nqmax, const Cloud *const *its,
                       const Cloud *const *ses, NNResult *const *ress)
{
    ProfScope pf(ctx, PCCM_K_GRID_FINISH);
    RescanJobs rj;
    int rc = rescan_jobs(ctx, jobs.njobs, its, ses, ress, self, &rj);
    if (rc) return rc;
    // the rescan's workgroups are few enough never to keep the tail's off the GPU (see k_grid_tail) and enough to split a cloud finely;
    // 1024 tail workgroups whatever the size (the grid-stride loops inside take any list): measured at 1M / 8M / 32M points per
    // cloud -- 2 110 / 13 565 / 51 516 tail queries, ~1e-3 of uniform data -- 256 / 512 / 1024 / 2048 / 4096 workgroups take
    // 23 / 20 / 15.4 / 18 / 22 us, 119 / 90 / 41 / 43 / 49 us and 421 / 323 / 140 / 142 / 139 us (round 4)
    const int64_t qblocks = (nqmax + 255) / 256;
    TailSync ts;
    ts.n_tail = (uint32_t)(qblocks < 1024 ? qblocks : 1024);
    ts.n_rescan = (uint32_t)(nqmax < 64 ? nqmax : 64);
    if ((rc = ensure(ctx, ctx->tail_sync, (size_t)2 * (16 * kTailShardWords + 32) * sizeof(uint32_t)))) return rc;
    if (!ctx->tail_sync_clean) {                           // (first use: the kernels rearm the counts themselves afterwards)
        if (ctx->capturing) {
            ctx->capture_failed = true;
            return fail(PCCM_E_STATE, "run pccm_nn once before graph capture");
        }
        PCCM_HIP(hipMemsetAsync(ctx->tail_sync.p, 0, ctx->tail_sync.bytes, ctx->stream));
        ctx->tail_sync_clean = true;
    }
    ts.retired = (uint32_t *)ctx->tail_sync.p + (self ? 16 * kTailShardWords + 32 : 0);
    ts.ticket = ts.retired + 16 * kTailShardWords;
    ts.host_err = ctx->host_err;
    // the pollers keep off the lines the tail waves add to for the first ~7 us (no tail is retired earlier) and look every ~0.9 us
    // afterwards: 64 + 64 pollers that start at once and look every 0.4 us cost the tail waves 6 us (21.8 against 15.5 us)
    ts.delay = 2;
    ts.nap = 2;
    dim3 grid(ts.n_tail + ts.n_rescan * (uint32_t)jobs.njobs);
    if (rec32) {
        if (self) hipLaunchKernelGGL((k_grid_tail<Rec32, true>), grid, dim3(256), 0, ctx->stream, jobs, g, rj, ts);
        else hipLaunchKernelGGL((k_grid_tail<Rec32, false>), grid, dim3(256), 0, ctx->stream, jobs, g, rj, ts);
    } else {
        if (self) hipLaunchKernelGGL((k_grid_tail<GridRec, true>), grid, dim3(256), 0, ctx->stream, jobs, g, rj, ts);
        else hipLaunchKernelGGL((k_grid_tail<GridRec, false>), grid, dim3(256), 0, ctx->stream, jobs, g, rj, ts);
    }
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

// Exact 1-NN for `ndirs` directions (LEFT and RIGHT fused into the same launches when both are asked for).
int nn_grid(pccm_ctx *ctx, int ndirs, const int *dirs, int force_idx)
{
    int rc;
    // which clouds' search structures this call needs: the searched cloud of every direction with rows here, and the
    // iterating cloud when all of its rows are here (its cell-sorted records are the query list then)
    int need = 0;
    for (int d = 0; d < ndirs; ++d) {
        const NNResult &res = ctx->nn[dirs[d]];
        if (res.end <= res.begin) continue;
        const int si = (dirs[d] == PCCM_DIR_LEFT) ? 1 : 0, ii = (dirs[d] == PCCM_DIR_RIGHT) ? 1 : 0;
        need |= 1 << si;
        if (res.begin == 0 && res.end == ctx->cloud[ii].n) need |= 1 << ii;
    }
    if (need == 0) {                                       // no rows of these directions on this rank
        for (int d = 0; d < ndirs; ++d) PCCM_HIP(hipMemsetAsync(ctx->nn[dirs[d]].nflag_dev, 0, 2 * sizeof(uint32_t), ctx->stream));
        return PCCM_OK;
    }
    // the directions' counter pairs {rescans, tail length} live side by side: cleared by the last kernel of the grid build
    // when this call builds the grid (the usual case: the search structure is rebuilt every step), else by one memset
    int dlo = 2, dhi = 0;
    for (int d = 0; d < ndirs; ++d) {
        dlo = dirs[d] < dlo ? dirs[d] : dlo;
        dhi = dirs[d] > dhi ? dirs[d] : dhi;
    }
    const bool side_by_side = ndirs > 0 && dhi - dlo + 1 == ndirs;
    bool rebuilt = false;
    // a single sharded direction of this call (the direction-first split): its rows are cell-sorted by the grid build itself
    ShardSort ride;
    int ride_dir = -1, nsh = 0;
    for (int d = 0; d < ndirs; ++d) {
        const NNResult &res = ctx->nn[dirs[d]];
        const Cloud &it = ctx->cloud[dirs[d] == PCCM_DIR_RIGHT ? 1 : 0];
        if (res.end > res.begin && !(res.begin == 0 && res.end == it.n)) {
            ++nsh;
            ride_dir = dirs[d];
            ride.it = &it;
            ride.begin = res.begin;
            ride.n = res.end - res.begin;
        }
    }
    if (nsh != 1) ride_dir = -1;
    // Distances only?  Then a voxelised pair is searched through its voxel bricks (pccm_vox.hip): nobody has asked for the matched
    // rows (pccm_nn_want_idx off, no repeat of a search for them) and no projection is to be fused (an exact tie decides whose
    // error vector is projected).  Whole clouds only.
    // matched rows wanted (pccm_nn_want_idx, a repeat of a search for them, a projection to fuse: an exact tie decides whose error
    // vector is projected)?  Then the bricks come with the rows' table and the search enumerates the equidistant voxels; the self
    // search with rows (nobody in the metric DAG asks for it) takes the lattice kernel
    int want_vox = nsh == 0 ? 1 : 0;
    for (int d = 0; d < ndirs && want_vox; ++d) {
        const int dir = dirs[d];
        const NNResult &res = ctx->nn[dir];
        if (res.end <= res.begin) continue;
        const Cloud &it = ctx->cloud[dir == PCCM_DIR_RIGHT ? 1 : 0], &se = ctx->cloud[dir == PCCM_DIR_LEFT ? 1 : 0];
        // (pccm_nn_want_idx speaks for the two directions colour metrics and getters read; the self search is read for its
        // distances -- cloud_pair.py:108-109 -- and returns rows only to a caller who asks for them explicitly)
        const bool rows = dir == PCCM_DIR_SELF ? force_idx != 0 : (ctx->want_idx || force_idx || fused_mode(ctx, dir, it, se) >= 0);
        if (rows && dir == PCCM_DIR_SELF) want_vox = 0;
        else if (rows) want_vox = 2;
    }
    if ((rc = ensure_grid(ctx, false, need, side_by_side ? (uint32_t *)ctx->counters.p + 2 * dlo : nullptr, 2 * ndirs, &rebuilt,
                          ride_dir >= 0 ? &ride : nullptr, want_vox))) return rc;
    const Grid &gr = ctx->grid;
    const GridGeom g = geom_of(gr);
    const uint32_t *cs_all = (const uint32_t *)gr.cell_start.p;
    const char *recs_all = (const char *)gr.recs.p;
    const size_t rsz = gr.rec32 ? sizeof(Rec32) : sizeof(GridRec);

    QueryJobs normal, selfj;
    normal.njobs = 0;
    selfj.njobs = 0;
    normal.err = selfj.err = ctx->host_err;
    int shard_dirs[3], nshard = 0, job_of_dir[3] = {-1, -1, -1};
    int normal_dirs[2] = {0, 0}, self_dirs[2] = {0, 0};
    int64_t shard_off[3], shard_total = 0;
    if (side_by_side && !rebuilt)
        PCCM_HIP(hipMemsetAsync((uint32_t *)ctx->counters.p + 2 * dlo, 0, (size_t)2 * ndirs * sizeof(uint32_t), ctx->stream));
    else if (!side_by_side)
        for (int d = 0; d < ndirs; ++d)
            PCCM_HIP(hipMemsetAsync(ctx->nn[dirs[d]].nflag_dev, 0, 2 * sizeof(uint32_t), ctx->stream));
    for (int d = 0; d < ndirs; ++d) {
        const int dir = dirs[d];
        NNResult &res = ctx->nn[dir];
        const int64_t nq = res.end - res.begin;
        if (nq <= 0) continue;
        const int si = (dir == PCCM_DIR_LEFT) ? 1 : 0;      // searched cloud
        const int ii = (dir == PCCM_DIR_RIGHT) ? 1 : 0;     // iterating cloud
        const Cloud &it = ctx->cloud[ii], &se = ctx->cloud[si];
        if ((rc = ensure(ctx, res.tail, (size_t)nq * sizeof(GridRec)))) return rc;
        QueryJob J;
        if (res.begin == 0 && res.end == it.n) {
            J.qrecs = recs_all + (size_t)(ii ? gr.n[0] : 0) * rsz;   // whole cloud: its own cell-sorted records
            J.qbase = J.qrecs;
            J.qcs = cs_all + (ii ? gr.ncells + 1 : 0);
        } else {
            // shard: its rows are sorted by the same cells into the shared shard-record buffer (below, one
            // counting sort for up to two directions of this call); remember where this direction's slice starts
            J.qrecs = nullptr;
            J.qbase = nullptr;
            J.qcs = nullptr;
            shard_dirs[nshard] = dir;
            shard_off[nshard] = shard_total;
            shard_total += nq;
            ++nshard;
        }
        const bool exact = it.exact32 && se.exact32;
        const double maxabs = it.maxabs > se.maxabs ? it.maxabs : se.maxabs;
        J.nq = nq;
        J.nchunks = (nq + 63) / 64;
        J.cs = cs_all + (si ? gr.ncells + 1 : 0);
        J.occ = gr.lattice ? (const uint32_t *)gr.occ.p + (size_t)si * (gr.ncells / 32 + 2) : nullptr;
        if (gr.vox) {
            J.vbricks = (const uint32_t *)gr.vbricks.p + (size_t)(si ? gr.n[0] : 0) * 32;
            J.vlist = (const uint32_t *)gr.vlist.p + (size_t)(ii ? gr.n[0] : 0);
            J.vcount = (const uint32_t *)gr.vcount.p + ii;
            J.vminrow = gr.vox_rows ? (const int32_t *)gr.vminrow.p + (size_t)(si ? gr.n[0] : 0) : nullptr;
        }
        J.srecs = recs_all + (size_t)(si ? gr.n[0] : 0) * rsz;
        J.s64 = se.xyz64;
        J.row_base = res.begin;
        J.slack32 = exact ? 0.0 : maxabs * 0x1.0p-20;
        int fm = fused_mode(ctx, dir, it, se);
        if (gr.vox && fm == PCCM_NORMAL_NEIGHBOUR) fm = -1;    // matched records carry no projection: the point kernel forms it from the rows
        res.rec_stride = (ctx->want_idx || force_idx) ? 4 : 2;
        // fp32-exact pairs, and no neighbour-indexed projection to fuse: the searches leave the matched record itself (16 bytes,
        // one store) and the reductions form distance and row-indexed projection from rows and normals they read in row order
        // (NNOut::layout).  PCCM_DEFER=0: the searches form both themselves, as in round 2 (A/B runs)
        static const bool defer_off = [] { const char *e = getenv("PCCM_DEFER"); return e && e[0] == '0'; }();
        const bool defer = gr.vox || (!defer_off && gr.rec32 && fm != PCCM_NORMAL_NEIGHBOUR);
        if (defer) res.rec_stride = 2;
        else if (fm >= 0 && (rc = normals_ready(ctx, ctx->cloud[si]))) return rc;     // the search itself projects: announced normals cross PCIe now
        res.rec_layout = defer ? 1 : 0;
        res.no_rows = gr.vox && !(gr.vox_rows && want_vox == 2 && dir != PCCM_DIR_SELF);
        J.out.rec = (double *)res.rec.p;
        J.out.stride = res.rec_stride;
        J.out.nrm = fm >= 0 ? se.nrm64 : nullptr;
        static const bool nrm32_off = [] { const char *e = getenv("PCCM_NRM32"); return e && e[0] == '0'; }();
        J.out.nrm32 = (fm >= 0 && se.nrm_exact32 && !nrm32_off) ? se.nrm32 : nullptr;
        J.out.row_base = res.begin;
        J.out.normal_mode = fm >= 0 ? fm : PCCM_NORMAL_ROW;
        J.out.layout = res.rec_layout;
        res.fused_mode = fm;
        res.rec_valid = true;
        res.plain_valid = res.plain_d2_valid = false;
        J.tail = res.tail.p;
        J.counters = res.nflag_dev;                         // [0] full rescans, [1] tail length
        if ((rc = ensure(ctx, res.flagged, (size_t)nq * sizeof(int32_t)))) return rc;
        if ((rc = ensure(ctx, res.flag_thr, (size_t)nq * sizeof(float)))) return rc;
        J.flagged = (int32_t *)res.flagged.p;
        J.flag_thr = (float *)res.flag_thr.p;
        QueryJobs &dst = (dir == PCCM_DIR_SELF) ? selfj : normal;
        job_of_dir[dir] = dst.njobs;
        (dir == PCCM_DIR_SELF ? self_dirs : normal_dirs)[dst.njobs] = dir;
        dst.j[dst.njobs++] = J;
        res.stats[1] = gr.ncells;       // pccm_nn_stats: the grid this search ran on (the bench's byte count needs it)
        res.stats[2] = 0;
    }
    if (nshard == 1 && ride_dir == shard_dirs[0] && ride.done) {
        QueryJobs &dst = (ride_dir == PCCM_DIR_SELF) ? selfj : normal;
        QueryJob &J = dst.j[job_of_dir[ride_dir]];
        J.qrecs = ride.recs;
        J.qbase = J.qrecs;                                           // cell starts are relative to the shard's first record
        J.qcs = ride.cs;
    } else if (nshard > 0) {
        // cell-sort the shards' rows: up to two directions per counting sort (same launches as a grid build); every
        // shard keeps its own cell starts (the brick kernel finds a brick's queries through them)
        ProfScope ps(ctx, PCCM_K_GRID_BUILD);
        if ((rc = ensure(ctx, ctx->g_qrecs, (size_t)shard_total * rsz))) return rc;
        if ((rc = ensure(ctx, ctx->g_hist, (size_t)3 * (gr.ncells + 1) * sizeof(uint32_t)))) return rc;
        char *qbuf = (char *)ctx->g_qrecs.p;
        for (int s0 = 0; s0 < nshard; s0 += 2) {
            const int cnt = nshard - s0 >= 2 ? 2 : 1;
            BuildJobs bj;
            bj.njobs = cnt;
            bj.total = 0;
            for (int k = 0; k < cnt; ++k) {
                const int dir = shard_dirs[s0 + k];
                const NNResult &res = ctx->nn[dir];
                const Cloud &it = ctx->cloud[dir == PCCM_DIR_RIGHT ? 1 : 0];
                bj.j[k] = {it.xyz64, (const float *)it.xyz32, res.begin, res.end - res.begin,
                           (uint32_t *)ctx->g_hist.p + (size_t)(s0 + k) * (gr.ncells + 1)};
                bj.total += res.end - res.begin;
            }
            if (cnt == 1) bj.j[1] = bj.j[0];
            if ((rc = sort_by_cell(ctx, bj, g, gr.ncells, qbuf + (size_t)shard_off[s0] * rsz, gr.rec32))) return rc;
        }
        for (int s = 0; s < nshard; ++s) {
            const int dir = shard_dirs[s];
            QueryJobs &dst = (dir == PCCM_DIR_SELF) ? selfj : normal;
            QueryJob &J = dst.j[job_of_dir[dir]];
            J.qrecs = qbuf + (size_t)shard_off[s] * rsz;
            J.qbase = J.qrecs;                                       // cell starts are relative to the shard's first record
            J.qcs = (const uint32_t *)ctx->g_hist.p + (size_t)s * (gr.ncells + 1);
        }
    }
    for (int pass = 0; pass < 2; ++pass) {
        QueryJobs &jobs = pass ? selfj : normal;
        if (jobs.njobs == 0) continue;
        if (jobs.njobs == 1) jobs.j[1] = jobs.j[0];
        const bool self = pass == 1;
        int64_t chunks = 0, nqmax = 0;
        for (int k = 0; k < jobs.njobs; ++k) {
            chunks += jobs.j[k].nchunks;
            nqmax = jobs.j[k].nq > nqmax ? jobs.j[k].nq : nqmax;
        }
        const int64_t qblocks = (nqmax + 255) / 256;
        const Cloud *its[2], *ses[2];
        NNResult *ress[2];
        const int *pd = self ? self_dirs : normal_dirs;
        for (int k = 0; k < jobs.njobs; ++k) {
            its[k] = &ctx->cloud[pd[k] == PCCM_DIR_RIGHT ? 1 : 0];
            ses[k] = &ctx->cloud[pd[k] == PCCM_DIR_LEFT ? 1 : 0];
            ress[k] = &ctx->nn[pd[k]];
        }
        bool tail_launch = true;                 // a ring-1 kernel ran: its tails and the exact rescan share one launch
        if (gr.vox) {
            ProfScope ps(ctx, PCCM_K_GRID_QUERY);
            if ((rc = launch_vox_query(ctx, jobs, g, self, gr.vox_rows && want_vox == 2 && !self))) return rc;      // (queries with nothing within 8 voxels: the general search)
        } else if (use_coop(ctx)) {
            ProfScope ps(ctx, PCCM_K_GRID_QUERY);
            if (gr.rec32) {
                if ((rc = launch_brick_query(ctx, jobs, g, self))) return rc;
            } else {
                dim3 grid((unsigned)((chunks + 3) / 4));
                if (self) hipLaunchKernelGGL((k_grid_query_coop<true>), grid, dim3(256), 0, ctx->stream, jobs, g);
                else hipLaunchKernelGGL((k_grid_query_coop<false>), grid, dim3(256), 0, ctx->stream, jobs, g);
            }
        } else {
            dim3 grid((unsigned)qblocks);
            ProfScope ps(ctx, PCCM_K_GRID_QUERY);
            if (gr.lattice) {                                // voxelised pair: pccm_lattice.hip
                if ((rc = launch_lattice_query(ctx, jobs, g, self))) return rc;
            } else if (gr.rec32) launch_queries<Rec32>(ctx, jobs, g, self, grid);
            else launch_queries<GridRec>(ctx, jobs, g, self, grid);
            tail_launch = false;                 // the per-thread kernels walk all three rings themselves
        }
        PCCM_HIP(hipGetLastError());
        if (tail_launch) {
            if ((rc = launch_tail(ctx, jobs, g, self, gr.rec32, nqmax, its, ses, ress))) return rc;
        } else {
            // whatever the rings could not settle: exact rescan, list lengths read on the device (an empty list is an early exit)
            if ((rc = launch_fallback(ctx, jobs.njobs, its, ses, ress, self))) return rc;
        }
    }
    return PCCM_OK;
}

// ---- tie exposure (diagnostic) ------------------------------------------------------------------------------------
// The reference keeps whichever of several equidistant nearest neighbours nanoflann's traversal meets last
// (cloud_pair.py:22-23: idx[-1] of a one-element search); the library keeps the smallest row (include/pccm.h).  D1 does not
// care, the point-to-plane projection (metric.py:146-153: err = a_i - b_nn(i)) does.  For every query of the shard this kernel
// finds ALL points of the searched cloud at exactly the nearest distance (the grid's cells inside the ball) and evaluates the
// squared projection for each: the smallest, the largest and the library's own.  Summed over the queries they bound what ANY
// tie rule can report as D2 MSE -- the reference's included.
struct TieJob {
    const void *srecs;          // cell-sorted records of the searched cloud
    const uint32_t *cs;         // ... its cell starts
    const double *q64;          // iterating cloud, fp64 rows
    const double *s64;          // searched cloud, fp64 rows (the pick's coordinates when its ball is not enumerated)
    const double *nrm;          // searched cloud's normals, or null (counts only)
    const int32_t *idx;         // shard's matched rows
    const double *d2;           // ... and squared distances
    int64_t q_begin, ns;
    int normal_mode;
    bool self;
};

template <typename REC>
__global__ __launch_bounds__(256) void k_tie_exposure(TieJob J, GridGeom g, double *__restrict__ sums, unsigned long long *__restrict__ counts)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double vmin = 0.0, vmax = 0.0, vpick = 0.0;
    unsigned int tied = 0, skipped = 0, mult = 0, live = 0;
    if (i < J.ns && J.idx[i] >= 0) {
        live = 1;
        const int64_t row = J.q_begin + i;
        const double qx = J.q64[3 * row], qy = J.q64[3 * row + 1], qz = J.q64[3 * row + 2];
        const double d = J.d2[i];
        const int pick = J.idx[i];
        const double rad = sqrt(d) * (1.0 + 0x1.0p-40) + fmax(g.slack[0], fmax(g.slack[1], g.slack[2]));
        const int xa = cell_coord(qx - rad, g.org[0], g.inv_h[0], g.dim[0]), xb = cell_coord(qx + rad, g.org[0], g.inv_h[0], g.dim[0]);
        const int ya = cell_coord(qy - rad, g.org[1], g.inv_h[1], g.dim[1]), yb = cell_coord(qy + rad, g.org[1], g.inv_h[1], g.dim[1]);
        const int za = cell_coord(qz - rad, g.org[2], g.inv_h[2], g.dim[2]), zb = cell_coord(qz + rad, g.org[2], g.inv_h[2], g.dim[2]);
        auto value = [&](double rx, double ry, double rz, int rrow) {
            if (!J.nrm) return 0.0;
            const int64_t k = (J.normal_mode == PCCM_NORMAL_ROW) ? row : (int64_t)rrow;
            const double ex = __dsub_rn(qx, rx), ey = __dsub_rn(qy, ry), ez = __dsub_rn(qz, rz);
            double p = __dmul_rn(ex, J.nrm[3 * k]);
            p = __fma_rn(ey, J.nrm[3 * k + 1], p);
            p = __fma_rn(ez, J.nrm[3 * k + 2], p);
            return __dmul_rn(p, p);
        };
        vmin = INFINITY;
        vmax = -INFINITY;
        if ((int64_t)(xb - xa + 1) * (yb - ya + 1) * (zb - za + 1) > 4096) {
            skipped = 1;                      // an outlier's ball covers too many cells: only the library's pick is looked at
        } else {
            const REC *__restrict__ srecs = (const REC *)J.srecs;
            for (int z = za; z <= zb; ++z)
                for (int y = ya; y <= yb; ++y) {
                    const uint32_t rowc = ((uint32_t)z * g.dim[1] + y) * g.dim[0];
                    for (uint32_t p = J.cs[rowc + xa], e = J.cs[rowc + xb + 1]; p < e; ++p) {
                        const P3 a = load_rec(srecs, p);
                        if (J.self && a.row == (int)row) continue;
                        if (gdist64(qx, qy, qz, a.x, a.y, a.z) != d) continue;
                        const double v = value(a.x, a.y, a.z, a.row);
                        vmin = fmin(vmin, v);
                        vmax = fmax(vmax, v);
                        vpick = a.row == pick ? v : vpick;
                        ++mult;
                    }
                }
        }
        if (mult == 0) {                       // skipped, or the winner came from beyond the grid (rescan): the pick alone counts,
            const double *r = J.s64 + 3 * (int64_t)pick;      // with its own squared projection (the interval stays an interval)
            vmin = vmax = vpick = value(r[0], r[1], r[2], pick);
            mult = 1;
            skipped = 1;
        }
        tied = mult > 1 ? 1u : 0u;
    }
    // block totals
    __shared__ double s_v[3][4];
    __shared__ unsigned int s_c[4][4];
    double a0 = live ? vmin : 0.0, a1 = live ? vmax : 0.0, a2 = vpick;
    unsigned int c0 = live, c1 = tied, c2 = skipped, c3 = mult;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a0 += __shfl_xor(a0, off);
        a1 += __shfl_xor(a1, off);
        a2 += __shfl_xor(a2, off);
        c0 += __shfl_xor(c0, off);
        c1 += __shfl_xor(c1, off);
        c2 += __shfl_xor(c2, off);
        c3 = max(c3, (unsigned int)__shfl_xor((int)c3, off));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_v[0][w] = a0; s_v[1][w] = a1; s_v[2][w] = a2;
        s_c[0][w] = c0; s_c[1][w] = c1; s_c[2][w] = c2; s_c[3][w] = c3;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // one partial per workgroup, added up on the host in workgroup order: the same on every run, and the three sums of
        // tie-free data come out identical
        sums[8 + 3 * (size_t)blockIdx.x + 0] = s_v[0][0] + s_v[0][1] + s_v[0][2] + s_v[0][3];
        sums[8 + 3 * (size_t)blockIdx.x + 1] = s_v[1][0] + s_v[1][1] + s_v[1][2] + s_v[1][3];
        sums[8 + 3 * (size_t)blockIdx.x + 2] = s_v[2][0] + s_v[2][1] + s_v[2][2] + s_v[2][3];
        atomicAdd(&counts[0], (unsigned long long)(s_c[0][0] + s_c[0][1] + s_c[0][2] + s_c[0][3]));
        atomicAdd(&counts[1], (unsigned long long)(s_c[1][0] + s_c[1][1] + s_c[1][2] + s_c[1][3]));
        atomicAdd(&counts[2], (unsigned long long)(s_c[2][0] + s_c[2][1] + s_c[2][2] + s_c[2][3]));
        atomicMax(&counts[3], (unsigned long long)max(max(s_c[3][0], s_c[3][1]), max(s_c[3][2], s_c[3][3])));
    }
}

// out[0] queries looked at, [1] queries with two or more equidistant nearest neighbours, [2] / [3] / [4] sum over the queries
// of the smallest / largest / the library's own squared projection, [5] queries whose ball was not enumerated, [6] the largest
// number of equidistant nearest neighbours of one query.  res: plain columns valid (ensure_plain).
int tie_exposure(pccm_ctx *ctx, int dir, const Cloud &it, const Cloud &se, const NNResult &res, int normal_mode, double out[8])
{
    for (int k = 0; k < 8; ++k) out[k] = 0.0;
    const int64_t ns = res.end - res.begin;
    if (ns <= 0) return PCCM_OK;
    const int si = (dir == PCCM_DIR_LEFT) ? 1 : 0;
    int rc = ensure_grid(ctx, false, 1 << si);
    if (rc) return rc;
    const Grid &gr = ctx->grid;
    const GridGeom g = geom_of(gr);
    const size_t nblk = (size_t)((ns + 255) / 256);
    if ((rc = ensure(ctx, ctx->g_blocksum, (8 + 3 * nblk) * sizeof(double)))) return rc;
    double *sums = (double *)ctx->g_blocksum.p;
    unsigned long long *counts = (unsigned long long *)(sums + 4);
    PCCM_HIP(hipMemsetAsync(sums, 0, 64, ctx->stream));
    TieJob J;
    J.srecs = (const char *)gr.recs.p + (size_t)(si ? gr.n[0] : 0) * (gr.rec32 ? sizeof(Rec32) : sizeof(GridRec));
    J.cs = (const uint32_t *)gr.cell_start.p + (si ? gr.ncells + 1 : 0);
    J.q64 = it.xyz64;
    J.s64 = se.xyz64;
    J.nrm = normal_mode >= 0 ? se.nrm64 : nullptr;
    J.idx = res.idx;
    J.d2 = res.d2;
    J.q_begin = res.begin;
    J.ns = ns;
    J.normal_mode = normal_mode >= 0 ? normal_mode : PCCM_NORMAL_ROW;
    J.self = dir == PCCM_DIR_SELF;
    dim3 grid((unsigned)((ns + 255) / 256));
    if (gr.rec32) hipLaunchKernelGGL((k_tie_exposure<Rec32>), grid, dim3(256), 0, ctx->stream, J, g, sums, counts);
    else hipLaunchKernelGGL((k_tie_exposure<GridRec>), grid, dim3(256), 0, ctx->stream, J, g, sums, counts);
    PCCM_HIP(hipGetLastError());
    std::vector<double> hv(8 + 3 * nblk);
    double *h = hv.data();
    PCCM_HIP(hipMemcpyAsync(h, sums, hv.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    h[0] = h[1] = h[2] = 0.0;
    for (size_t b = 0; b < nblk; ++b)
        for (int k = 0; k < 3; ++k) h[k] += h[8 + 3 * b + k];
    unsigned long long c[4];
    for (int k = 0; k < 4; ++k) c[k] = *reinterpret_cast<const unsigned long long *>(&h[4 + k]);
    out[0] = (double)c[0];
    out[1] = (double)c[1];
    out[2] = h[0];
    out[3] = h[1];
    out[4] = h[2];
    out[5] = (double)c[2];
    out[6] = (double)c[3];
    return PCCM_OK;
}

// A grid over ONE cloud with cells sized for that cloud (GridRec records: pccm_normals.hip): the pair's geometry follows its
// larger cloud, and a decoded cloud of a fifteenth of the reference's points -- a low rate of BASELINE configs[4] -- had its
// k = 30 neighbourhoods spread over six rings of such cells (43 ms of normal estimation for 59 000 points, round 4).
// The cell-edge factor is cached with the cloud; the pair's grid is gone afterwards (the next search builds it again).
int grid_ensure_solo(pccm_ctx *ctx, int which)
{
    Grid &gr = ctx->grid;
    Cloud &c = ctx->cloud[which];
    if (ctx->capturing) {
        ctx->capture_failed = true;
        return fail(PCCM_E_STATE, "normal estimation is not allowed during graph capture");
    }
    int rc;
    if (c.solo_scale_version != c.version) {
        Occupancy o;
        double scale = 1.0;
        if ((rc = fit_scale(ctx, c, scale, o, which))) return rc;
        c.solo_scale = scale;
        c.solo_scale_version = c.version;
    }
    ProfScope ps(ctx, PCCM_K_GRID_BUILD);
    GridGeom g;
    int64_t ncells;
    choose_geometry(ctx, g, ncells, c.solo_scale, false, which);
    const int64_t n0 = ctx->cloud[0].n, n1 = ctx->cloud[1].n;
    if ((rc = ensure(ctx, gr.cell_start, (size_t)2 * (ncells + 1) * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, gr.recs, (size_t)(n0 + n1 > 0 ? n0 + n1 : 1) * sizeof(GridRec)))) return rc;
    BuildJobs jobs;
    jobs.njobs = 1;
    jobs.total = c.n;
    jobs.j[0] = {c.xyz64, (const float *)c.xyz32, 0, c.n, (uint32_t *)gr.cell_start.p + (size_t)which * (ncells + 1)};
    jobs.j[1] = jobs.j[0];
    if ((rc = sort_by_cell(ctx, jobs, g, ncells, (char *)gr.recs.p + (size_t)(which ? n0 : 0) * sizeof(GridRec), false))) return rc;
    for (int a = 0; a < 3; ++a) {
        gr.dim[a] = g.dim[a];
        gr.org[a] = g.org[a];
        gr.h[a] = g.h[a];
        gr.inv_h[a] = g.inv_h[a];
    }
    gr.ncells = ncells;
    gr.n[0] = n0;
    gr.n[1] = n1;
    gr.key = 0;                                            // not the pair's grid: whoever searches next builds that
    gr.rec32 = false;
    gr.lattice = gr.vox = gr.vox_rows = false;
    gr.built = 0;
    return PCCM_OK;
}

void grid_release(pccm_ctx *ctx)
{
    DevBuf *bufs[] = {&ctx->grid.cell_start, &ctx->grid.occ, &ctx->grid.recs, &ctx->grid.vbricks, &ctx->grid.vlist, &ctx->grid.vcount, &ctx->grid.vminrow,
                      &ctx->g_cell_of, &ctx->g_rank, &ctx->g_hist, &ctx->g_blocksum, &ctx->g_qrecs, &ctx->g_bins, &ctx->g_tmp};
    for (DevBuf *b : bufs) {
        if (b->p) (void)hipFree(b->p);
        b->p = nullptr;
        b->bytes = 0;
    }
    ctx->grid.key = 0;
}

void grid_invalidate(pccm_ctx *ctx) { ctx->grid.key = 0; }

int grid_ensure(pccm_ctx *ctx, bool need64, int need_mask) { return ensure_grid(ctx, need64, need_mask); }

}  // namespace pccm
