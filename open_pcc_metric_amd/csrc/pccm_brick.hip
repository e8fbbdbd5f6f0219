// LDS-brick ring-1 search for gfx950: the dominant kernel of the default (grid) engine on fp32-exact clouds.
//
// Stands under get_neighbour_cloud(), open_pcc_metric/cloud_pair.py:10-42 (one search_knn_vector_3d call per
// point there), with the D2 projection of metric.py:146-153 fused into the same pass.
//
// One workgroup owns a *brick* of the grid: BX x BY x BZ cells (BX <= 64 cells along the x-fastest axis,
// BY = 4, BZ = 2).  Its queries are the iterating cloud's records in those cells -- eight contiguous runs of
// the cell-sorted array -- and every candidate any of them can have in ring 1 lies in the (BX+2) x 6 x 4 cells
// around the brick: 24 contiguous x-runs of the searched cloud's cell-sorted array.
//   1. 24 + 8 lanes fetch the run bounds (cell_start) of the searched and the iterating cloud; two wave scans
//      turn them into LDS offsets,
//   2. the 24 runs are copied into LDS as 16-byte Rec32 records (x, y, z, row: one coalesced global_load_dwordx4
//      and one ds_write_b128 per record, ~2.4 staged records per query instead of 9.5 per query for the
//      per-wave staging of round 1) together with their cell starts, rebased to LDS positions,
//   3. every lane takes queries of the brick in cell-sorted order, so the lanes of a wave sit in neighbouring
//      cells of one x-row and their LDS reads broadcast or fall on consecutive banks; per query nine
//      x-runs of three cells each are scanned with one ds_read_b128 per candidate, in fp32, tracking
//      (best, second best, position of the best),
//   4. certification as in pccm_brute.hip K2 -- the fp32 winner is the unique fp64 winner when the second best
//      d32 lies above thr(best) -- then the exact fp64 d2 from the winner's LDS record (fp32-exact inputs:
//      (double)(float)x == x), the ring-1 stop rule, and the fused epilogue: error vector, projection on the
//      searched cloud's normal (row i of it: reference quirk Q1, or row nn(i)), ONE 32-byte result record.
// Queries that are not certified (near ties) or not settled by ring 1 go to the tail list, as before
// (k_grid_finish -> k2b_fallback).  A brick whose 24 runs do not fit the LDS budget (clumped data) sends all
// its queries there.
//
// Bound: LDS/VALU issue on ~40 candidates per query; HBM sees every record of both clouds about once
// (neighbouring bricks share runs through the XCD's L2: XCD-aware brick order, as in round 1).
#include "pccm_grid.h"

namespace pccm {

constexpr int kBXMax = 64;
constexpr int kLcsPitch = kBXMax + 3;                // cell starts per staged run (BX + 2 cells + 1), odd pitch
constexpr float kBigF = 3.0e38f;

struct BrickParams {
    int bx;                 // cells per brick along x
    int nbx, nby, nbz;      // bricks per axis
    int64_t per_job;        // nbx * nby * nbz
    int cap;                // staged records that fit (< 65536: LDS positions are kept as uint16)
};

// NT threads per workgroup, brick of BX x BY x BZ cells.  Occupancy is set by LDS (16 B per staged record), so the
// kernel spends registers freely: the next query of a lane and its row-indexed normal are in flight while the
// current one is scanned.
template <bool SELF, int NT, int BY, int BZ>
__global__ __launch_bounds__(NT) void k_brick_query(QueryJobs jobs, GridGeom g, BrickParams bp)
{
    constexpr int kNRow = BY * BZ;                   // query rows of a brick
    constexpr int kNRun = (BY + 2) * (BZ + 2);       // staged x-runs
    static_assert(kNRun <= 64 && kNRow <= 64, "one wave scans the run / row lengths");
    extern __shared__ float4 s_rec[];                // [cap + 1]
    __shared__ uint16_t s_lcs[kNRun * kLcsPitch];
    __shared__ uint32_t s_g0[kNRun], s_base[kNRun + 1], s_qg0[kNRow], s_qoff[kNRow + 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // XCD-aware order: workgroups b, b+8, ... share an XCD; give every XCD one contiguous eighth of the brick list
    const uint32_t nblk = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t bq = nblk >> 3, br = nblk & 7u;
    int64_t vb = (int64_t)(xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + slot;
    const int jb = (jobs.njobs > 1 && vb >= bp.per_job) ? 1 : 0;
    if (jb) vb -= bp.per_job;
    if (vb >= bp.per_job) return;
    const QueryJob &J = jobs.j[jb];
    const int dimx = g.dim[0], dimy = g.dim[1], dimz = g.dim[2];
    const int ibx = (int)(vb % bp.nbx), iby = (int)((vb / bp.nbx) % bp.nby), ibz = (int)(vb / ((int64_t)bp.nbx * bp.nby));
    const int bx0 = ibx * bp.bx, bx1 = min(bx0 + bp.bx, dimx);
    const int by0 = iby * BY, bz0 = ibz * BZ;
    const int sx0 = max(bx0 - 1, 0), sx1 = min(bx1 + 1, dimx);
    const int ncs = sx1 - sx0;                                        // staged cells per run, <= bx + 2
    const uint32_t *__restrict__ cs = J.cs;
    const uint32_t *__restrict__ qcs = J.qcs;
    const Rec32 *__restrict__ srecs = (const Rec32 *)J.srecs;
    const Rec32 *__restrict__ qbase = (const Rec32 *)J.qbase;

    // ---- 1. run bounds -> LDS offsets (wave 0: searched runs, wave 1: query rows) ---------------------------
    if (w == 0) {
        uint32_t len = 0, g0 = 0;
        if (lane < kNRun) {
            const int y = by0 - 1 + lane % (BY + 2), z = bz0 - 1 + lane / (BY + 2);
            if (y >= 0 && y < dimy && z >= 0 && z < dimz) {
                const uint32_t rowbase = ((uint32_t)z * dimy + y) * dimx;
                g0 = cs[rowbase + sx0];
                len = cs[rowbase + sx1] - g0;
            }
            s_g0[lane] = g0;
        }
        uint32_t inc = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane < kNRun) s_base[lane] = inc - len;
        if (lane == kNRun - 1) s_base[kNRun] = inc;
    } else if (w == 1) {
        uint32_t len = 0, q0 = 0;
        if (lane < kNRow) {
            const int y = by0 + lane % BY, z = bz0 + lane / BY;
            if (y < dimy && z < dimz) {
                const uint32_t rowbase = ((uint32_t)z * dimy + y) * dimx;
                q0 = qcs[rowbase + bx0];
                len = qcs[rowbase + bx1] - q0;
            }
            s_qg0[lane] = q0;
        }
        uint32_t inc = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane < kNRow) s_qoff[lane] = inc - len;
        if (lane == kNRow - 1) s_qoff[kNRow] = inc;
    }
    __syncthreads();
    const uint32_t T = s_base[kNRun], NQ = s_qoff[kNRow];
    if (NQ == 0) return;                                               // block-uniform

    // query qi of the brick -> its row of the brick and its record
    auto query_row = [&](uint32_t qi) {
        int r = 0;
#pragma unroll
        for (int k = 1; k < kNRow; ++k) r += (qi >= s_qoff[k]) ? 1 : 0;
        return r;
    };
    auto load_query = [&](uint32_t qi, int r) {
        return *reinterpret_cast<const float4 *>(&qbase[s_qg0[r] + (qi - s_qoff[r])]);
    };

    if (T > (uint32_t)bp.cap) {
        // clumped data: more candidates than the LDS budget holds -- the general kernels take this brick's queries
        for (uint32_t qi = tid; qi < NQ; qi += NT) {
            const float4 q = load_query(qi, query_row(qi));
            const uint32_t pos = atomicAdd(&J.counters[1], 1u);
            reinterpret_cast<float4 *>(J.tail)[pos] = q;
        }
        return;
    }

    // the lane's first query is fetched while the brick is staged
    uint32_t qi = tid;
    bool have = qi < NQ;
    int rn = 0;
    float4 qn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have) {
        rn = query_row(qi);
        qn = load_query(qi, rn);
    }

    // ---- 2. stage cell starts (wave w: runs w, w + NT/64, ...) and records (interleaved 64-record pieces) -------
    for (int r = w; r < kNRun; r += NT / 64) {
        const int y = by0 - 1 + r % (BY + 2), z = bz0 - 1 + r / (BY + 2);
        const bool in = y >= 0 && y < dimy && z >= 0 && z < dimz;      // wave-uniform
        const uint32_t rowbase = in ? ((uint32_t)z * dimy + y) * dimx : 0u;
        const uint32_t rebase = s_base[r] - s_g0[r];
        for (int j = lane; j <= ncs; j += 64)
            s_lcs[r * kLcsPitch + j] = (uint16_t)(in ? cs[rowbase + sx0 + j] + rebase : s_base[r]);
    }
    {
        int run = 0;                                                   // wave-uniform, monotone
        for (uint32_t f0 = (uint32_t)w * 64u; f0 < T; f0 += (uint32_t)NT) {
            while (run + 1 < kNRun && f0 >= s_base[run + 1]) ++run;
            const uint32_t f = f0 + lane;
            int myrun = run;
            while (myrun + 1 < kNRun && f >= s_base[myrun + 1]) ++myrun;
            if (f < T) s_rec[f] = *reinterpret_cast<const float4 *>(&srecs[s_g0[myrun] + (f - s_base[myrun])]);
        }
    }
    const NNOut &out = J.out;
    const bool fuse = out.nrm != nullptr;
    const bool fuse_row = fuse && out.normal_mode == PCCM_NORMAL_ROW;
    // row-indexed normal (quirk Q1) of the first query: its address is known as soon as the query is here
    double n0 = 0.0, n1 = 0.0, n2 = 0.0;
    if (have && fuse_row) {
        const double *np = out.nrm + 3 * (int64_t)__float_as_int(qn.w);
        n0 = np[0]; n1 = np[1]; n2 = np[2];
    }
    __syncthreads();

    // ---- 3. queries ------------------------------------------------------------------------------------------
    while (have) {                                                     // no barrier below: lanes may leave
        const float4 q = qn;
        const int r = rn;
        const double m0 = n0, m1 = n1, m2 = n2;
        const int qrow = __float_as_int(q.w);
        // next query of this lane, and its normal: in flight during the scan
        const uint32_t qnext = qi + NT;
        const bool hn = qnext < NQ;
        if (hn) {
            rn = query_row(qnext);
            qn = load_query(qnext, rn);
            if (fuse_row) {
                const double *np = out.nrm + 3 * (int64_t)__float_as_int(qn.w);
                n0 = np[0]; n1 = np[1]; n2 = np[2];
            }
        }
        const double qx = (double)q.x, qy = (double)q.y, qz = (double)q.z;
        const int ly = r % BY, lz = r / BY;
        const int cx = cell_coord(qx, g.org[0], g.inv_h[0], dimx);
        const int cy = by0 + ly, cz = bz0 + lz;
        const int ja = max(cx - 1, 0) - sx0, jb2 = min(cx + 2, dimx) - sx0;
        float best = kBigF, second = kBigF;
        uint32_t bestpos = 0xffffffffu;
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int run = (lz + dz) * (BY + 2) + (ly + dy);
                const uint32_t fs = s_lcs[run * kLcsPitch + ja], fe = s_lcs[run * kLcsPitch + jb2];
                for (uint32_t f = fs; f < fe; f += 2) {
                    // two records = 32 contiguous bytes: two ds_read_b128 (4 LDS cycles each).  Left to itself hipcc
                    // drops the unused row word and issues ds_read_b96, which the LDS serves at 8 cycles per wave
                    // (MI355X_MICROARCH.md, LDS table): the reads, not the arithmetic, then pace the loop.
                    float4 c0, c1;                                     // slot f + 1 always exists (cap + 1 slots)
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(c0), "=&v"(c1)
                                 : "v"((uint32_t)(uintptr_t)(&s_rec[f]))
                                 : "memory");
                    const bool two = f + 1 < fe;
                    const float ax = q.x - c0.x, ay = q.y - c0.y, az = q.z - c0.z;
                    const float bx = q.x - c1.x, by = q.y - c1.y, bz = q.z - c1.z;
                    float d0 = ax * ax, d1 = bx * bx;
                    d0 = __builtin_fmaf(ay, ay, d0);
                    d1 = __builtin_fmaf(by, by, d1);
                    d0 = __builtin_fmaf(az, az, d0);
                    d1 = __builtin_fmaf(bz, bz, d1);
                    if (SELF) {
                        d0 = (__float_as_int(c0.w) == qrow) ? kBigF : d0;
                        d1 = (__float_as_int(c1.w) == qrow) ? kBigF : d1;
                    }
                    d1 = two ? d1 : kBigF;
                    second = __builtin_amdgcn_fmed3f(best, second, d0);
                    const bool u0 = d0 < best;
                    best = u0 ? d0 : best;
                    bestpos = u0 ? f : bestpos;
                    second = __builtin_amdgcn_fmed3f(best, second, d1);
                    const bool u1 = d1 < best;
                    best = u1 ? d1 : best;
                    bestpos = u1 ? f + 1 : bestpos;
                }
            }
        }
        // certification (bound derived in pccm_brute.hip) + the ring-1 stop rule
        const double tq = sqrt((double)best) * (1.0 + 0x1.0p-20);
        const double thr = tq * tq * (1.0 + 0x1.0p-30) + 1.0e-36;
        bool settled = false;
        if (bestpos != 0xffffffffu && (double)second > thr) {
            const float4 c = s_rec[bestpos];
            const double rx = (double)c.x, ry = (double)c.y, rz = (double)c.z;
            const int wrow = __float_as_int(c.w);
            const double d64 = gdist64(qx, qy, qz, rx, ry, rz);
            settled = settled_by(face_bound(g, qx, qy, qz, cx, cy, cz, 1), d64);
            if (settled) {
                double p = 0.0;
                if (fuse) {
                    double e0 = m0, e1 = m1, e2 = m2;
                    if (!fuse_row) {
                        const double *np = out.nrm + 3 * (int64_t)wrow;
                        e0 = np[0]; e1 = np[1]; e2 = np[2];
                    }
                    const double ex = __dsub_rn(qx, rx), ey = __dsub_rn(qy, ry), ez = __dsub_rn(qz, rz);
                    p = __dmul_rn(ex, e0);
                    p = __fma_rn(ey, e1, p);
                    p = __fma_rn(ez, e2, p);
                }
                double4 o;
                o.x = d64;
                o.y = p;
                o.z = __longlong_as_double((long long)(uint32_t)wrow);
                o.w = 0.0;
                out.rec[qrow - out.row_base] = o;
            }
        }
        if (!settled) {
            const uint32_t pos = atomicAdd(&J.counters[1], 1u);
            reinterpret_cast<float4 *>(J.tail)[pos] = q;
        }
        qi = qnext;
        have = hn;
    }
}

// brick shape / workgroup size: PCCM_BRICK="NT,BY,BZ" picks one of the compiled variants (A/B runs)
struct BrickShape {
    int nt, by, bz;
};

static BrickShape brick_shape()
{
    static const BrickShape s = [] {
        BrickShape v = {512, 4, 2};
        const char *e = getenv("PCCM_BRICK");
        if (e) {
            int a = 0, b = 0, c = 0;
            if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3) v = {a, b, c};
        }
        return v;
    }();
    return s;
}

template <int NT, int BY, int BZ>
static void launch_shape(pccm_ctx *ctx, const QueryJobs &jobs, const GridGeom &g, bool self, BrickParams bp, double density)
{
    bp.nby = (g.dim[1] + BY - 1) / BY;
    bp.nbz = (g.dim[2] + BZ - 1) / BZ;
    bp.per_job = (int64_t)bp.nbx * bp.nby * bp.nbz;
    // LDS budget: the expected number of staged records (runs x cells x points per cell) plus a quarter -- occupancy
    // is set by it.  Bricks that hold more (clumped data) hand their queries to the general kernels.
    const double expect = (double)((BY + 2) * (BZ + 2)) * (bp.bx + 2) * density;
    int cap = (int)(1.25 * expect) + 64;
    static const int cap_env = [] { const char *e = getenv("PCCM_BRICK_CAP"); return e ? atoi(e) : 0; }();
    if (cap_env > 0) cap = cap_env;
    if (cap < 256) cap = 256;
    if (cap > 3800) cap = 3800;                         // 60 KB of records: static + dynamic LDS stay under 64 KB
    bp.cap = cap;
    const size_t lds = (size_t)(bp.cap + 1) * sizeof(float4);
    dim3 grid((unsigned)(bp.per_job * jobs.njobs));
    if (self) hipLaunchKernelGGL((k_brick_query<true, NT, BY, BZ>), grid, dim3(NT), lds, ctx->stream, jobs, g, bp);
    else hipLaunchKernelGGL((k_brick_query<false, NT, BY, BZ>), grid, dim3(NT), lds, ctx->stream, jobs, g, bp);
}

int launch_brick_query(pccm_ctx *ctx, const QueryJobs &jobs, const GridGeom &g, bool self)
{
    BrickParams bp;
    const int nbx = (g.dim[0] + 47) / 48;                 // bricks of <= 48 cells: ~70 records per staged run at 1.5 points per cell
    bp.bx = (g.dim[0] + nbx - 1) / nbx;
    if (bp.bx > kBXMax) bp.bx = kBXMax;
    bp.nbx = (g.dim[0] + bp.bx - 1) / bp.bx;
    bp.nby = bp.nbz = 0;
    bp.per_job = 0;
    bp.cap = 0;
    const Grid &gr = ctx->grid;
    const int64_t nmax = gr.n[0] > gr.n[1] ? gr.n[0] : gr.n[1];
    const double density = gr.ncells > 0 ? (double)nmax / (double)gr.ncells : 1.5;      // points per cell of the denser cloud
    const BrickShape sh = brick_shape();
    if (sh.nt == 256 && sh.by == 4 && sh.bz == 2) launch_shape<256, 4, 2>(ctx, jobs, g, self, bp, density);
    else if (sh.nt == 256 && sh.by == 2 && sh.bz == 2) launch_shape<256, 2, 2>(ctx, jobs, g, self, bp, density);
    else if (sh.nt == 512 && sh.by == 4 && sh.bz == 4) launch_shape<512, 4, 4>(ctx, jobs, g, self, bp, density);
    else if (sh.nt == 1024 && sh.by == 4 && sh.bz == 4) launch_shape<1024, 4, 4>(ctx, jobs, g, self, bp, density);
    else launch_shape<512, 4, 2>(ctx, jobs, g, self, bp, density);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

}  // namespace pccm
