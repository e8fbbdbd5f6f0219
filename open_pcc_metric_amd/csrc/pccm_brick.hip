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

constexpr int kBY = 4, kBZ = 2;
constexpr int kNRow = kBY * kBZ;                     // query rows of a brick
constexpr int kNRun = (kBY + 2) * (kBZ + 2);         // staged x-runs
constexpr int kBXMax = 64;
constexpr int kLcsPitch = kBXMax + 3;                // cell starts per staged run (BX + 2 cells + 1), odd pitch
constexpr int kBrickCap = 2560;                      // staged records per brick: 40 KB of LDS (+ 6.5 KB of cell starts)
constexpr float kBigF = 3.0e38f;

struct BrickParams {
    int bx;                 // cells per brick along x
    int nbx, nby, nbz;      // bricks per axis
    int64_t per_job;        // nbx * nby * nbz
    int cap;                // staged records that fit
};

template <bool SELF>
__global__ __launch_bounds__(256) void k_brick_query(QueryJobs jobs, GridGeom g, BrickParams bp)
{
    extern __shared__ float4 s_rec[];                // [cap + 1]
    __shared__ uint32_t s_lcs[kNRun * kLcsPitch];
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
    const int by0 = iby * kBY, bz0 = ibz * kBZ;
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
            const int y = by0 - 1 + lane % (kBY + 2), z = bz0 - 1 + lane / (kBY + 2);
            if (y >= 0 && y < dimy && z >= 0 && z < dimz) {
                const uint32_t rowbase = ((uint32_t)z * dimy + y) * dimx;
                g0 = cs[rowbase + sx0];
                len = cs[rowbase + sx1] - g0;
            }
            s_g0[lane] = g0;
        }
        uint32_t inc = len;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane < kNRun) s_base[lane] = inc - len;
        if (lane == kNRun - 1) s_base[kNRun] = inc;
    } else if (w == 1) {
        uint32_t len = 0, q0 = 0;
        if (lane < kNRow) {
            const int y = by0 + lane % kBY, z = bz0 + lane / kBY;
            if (y < dimy && z < dimz) {
                const uint32_t rowbase = ((uint32_t)z * dimy + y) * dimx;
                q0 = qcs[rowbase + bx0];
                len = qcs[rowbase + bx1] - q0;
            }
            s_qg0[lane] = q0;
        }
        uint32_t inc = len;
#pragma unroll
        for (int off = 1; off < kNRow; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane < kNRow) s_qoff[lane] = inc - len;
        if (lane == kNRow - 1) s_qoff[kNRow] = inc;
    }
    __syncthreads();
    const uint32_t T = s_base[kNRun], NQ = s_qoff[kNRow];
    if (NQ == 0) return;                                               // block-uniform
    if (T > (uint32_t)bp.cap) {
        // clumped data: more candidates than the LDS budget holds -- the general kernels take this brick's queries
        for (uint32_t qi = tid; qi < NQ; qi += 256) {
            int r = 0;
#pragma unroll
            for (int k = 1; k < kNRow; ++k) r += (qi >= s_qoff[k]) ? 1 : 0;
            const float4 q = *reinterpret_cast<const float4 *>(&qbase[s_qg0[r] + (qi - s_qoff[r])]);
            const uint32_t pos = atomicAdd(&J.counters[1], 1u);
            reinterpret_cast<float4 *>(J.tail)[pos] = q;
        }
        return;
    }

    // ---- 2. stage cell starts (wave w: runs w, w+4, ...) and records (interleaved 64-record pieces) ---------
    for (int r = w; r < kNRun; r += 4) {
        const int y = by0 - 1 + r % (kBY + 2), z = bz0 - 1 + r / (kBY + 2);
        const bool in = y >= 0 && y < dimy && z >= 0 && z < dimz;      // wave-uniform
        const uint32_t rowbase = in ? ((uint32_t)z * dimy + y) * dimx : 0u;
        const uint32_t rebase = s_base[r] - s_g0[r];
        for (int j = lane; j <= ncs; j += 64)
            s_lcs[r * kLcsPitch + j] = in ? cs[rowbase + sx0 + j] + rebase : s_base[r];
    }
    {
        int run = 0;                                                   // wave-uniform, monotone
        for (uint32_t f0 = (uint32_t)w * 64u; f0 < T; f0 += 256u) {
            while (run + 1 < kNRun && f0 >= s_base[run + 1]) ++run;
            const uint32_t f = f0 + lane;
            int myrun = run;
            while (myrun + 1 < kNRun && f >= s_base[myrun + 1]) ++myrun;
            if (f < T) s_rec[f] = *reinterpret_cast<const float4 *>(&srecs[s_g0[myrun] + (f - s_base[myrun])]);
        }
    }
    __syncthreads();

    // ---- 3. queries ------------------------------------------------------------------------------------------
    const NNOut &out = J.out;
    for (uint32_t q0 = 0; q0 < NQ; q0 += 256u) {
        const uint32_t qi = q0 + tid;
        if (qi >= NQ) break;                                           // no barrier below: lanes may leave
        int r = 0;
#pragma unroll
        for (int k = 1; k < kNRow; ++k) r += (qi >= s_qoff[k]) ? 1 : 0;
        const float4 q = *reinterpret_cast<const float4 *>(&qbase[s_qg0[r] + (qi - s_qoff[r])]);
        const int qrow = __float_as_int(q.w);
        const double qx = (double)q.x, qy = (double)q.y, qz = (double)q.z;
        const int ly = r % kBY, lz = r / kBY;
        const int cx = cell_coord(qx, g.org[0], g.inv_h[0], dimx);
        const int cy = by0 + ly, cz = bz0 + lz;
        // row-indexed normal (quirk Q1): its address is known now; issue the gather before the scan
        double n0 = 0.0, n1 = 0.0, n2 = 0.0;
        const bool fuse = out.nrm != nullptr;
        if (fuse && out.normal_mode == PCCM_NORMAL_ROW) {
            const double *np = out.nrm + 3 * (int64_t)qrow;
            n0 = np[0]; n1 = np[1]; n2 = np[2];
        }
        const int ja = max(cx - 1, 0) - sx0, jb2 = min(cx + 2, dimx) - sx0;
        float best = kBigF, second = kBigF;
        uint32_t bestpos = 0xffffffffu;
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int run = (lz + dz) * (kBY + 2) + (ly + dy);
                const uint32_t fs = s_lcs[run * kLcsPitch + ja], fe = s_lcs[run * kLcsPitch + jb2];
                for (uint32_t f = fs; f < fe; f += 2) {
                    const float4 c0 = s_rec[f], c1 = s_rec[f + 1];     // slot f + 1 always exists (cap + 1 slots)
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
                    if (out.normal_mode != PCCM_NORMAL_ROW) {
                        const double *np = out.nrm + 3 * (int64_t)wrow;
                        n0 = np[0]; n1 = np[1]; n2 = np[2];
                    }
                    const double ex = __dsub_rn(qx, rx), ey = __dsub_rn(qy, ry), ez = __dsub_rn(qz, rz);
                    p = __dmul_rn(ex, n0);
                    p = __fma_rn(ey, n1, p);
                    p = __fma_rn(ez, n2, p);
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
    }
}

bool brick_applicable(const GridGeom &g)
{
    (void)g;
    return true;
}

int launch_brick_query(pccm_ctx *ctx, const QueryJobs &jobs, const GridGeom &g, bool self)
{
    BrickParams bp;
    const int nbx = (g.dim[0] + 47) / 48;                 // bricks of <= 48 cells: ~70 records per staged run at 1.5 points per cell
    bp.bx = (g.dim[0] + nbx - 1) / nbx;
    if (bp.bx > kBXMax) bp.bx = kBXMax;
    bp.nbx = (g.dim[0] + bp.bx - 1) / bp.bx;
    bp.nby = (g.dim[1] + kBY - 1) / kBY;
    bp.nbz = (g.dim[2] + kBZ - 1) / kBZ;
    bp.per_job = (int64_t)bp.nbx * bp.nby * bp.nbz;
    bp.cap = kBrickCap;
    const int64_t nblk = bp.per_job * jobs.njobs;
    if (nblk > 0x7fffffffLL) return fail(PCCM_E_ARG, "grid too large for the brick kernel");
    const size_t lds = (size_t)(bp.cap + 1) * sizeof(float4);
    dim3 grid((unsigned)nblk);
    if (self) hipLaunchKernelGGL((k_brick_query<true>), grid, dim3(256), lds, ctx->stream, jobs, g, bp);
    else hipLaunchKernelGGL((k_brick_query<false>), grid, dim3(256), lds, ctx->stream, jobs, g, bp);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

}  // namespace pccm
