// Brute-force exact 1-NN engine for gfx950 (MI355X): K1 fp32 scan, K2 fp64 certify/refine,
// K2b exact rescan of the queries K2 could not certify.
//
// Replaces get_neighbour_cloud(), open_pcc_metric/cloud_pair.py:10-42 of the reference (one
// KD-tree query per point from a Python loop) and, with SELF, the Open3D self search behind
// cloud_pair.py:108-109.
//
// K1 (k1_scan) is the dominant kernel.  It is fp32-VALU bound, not HBM bound: every (query, ref)
// pair costs 3 v_sub + 1 v_mul + 2 v_fmac + 1/2 v_min3, while a 1M-point cloud is 16 MB and is
// streamed from L2/MALL.  Layout: a workgroup of 256 threads keeps QT queries per thread in
// VGPRs (256*QT queries per workgroup), stages the search cloud through LDS in double-buffered
// tiles of 1024 points (12 KB) copied with coalesced 16-byte loads, and reads four search points
// per iteration with three wave-uniform (broadcast, conflict-free) ds_read_b128.  The fp32 copy
// of a cloud is packed as "quads": [x0 x1 x2 x3][y0 y1 y2 y3][z0 z1 z2 z3] per four points
// (12 B/point, no padding lane), so that every byte an LDS read returns is used.
// Winners are tracked per 64-point granule, not per point: the inner loop only keeps a running
// minimum (v_min3 over two points), and once per granule a v_med3 / v_cmp / v_cndmask group
// updates (best granule-min, its granule id, second-best granule-min).  The difference form
// (q - r)^2 is used, never |q|^2 + |r|^2 - 2 q.r: the expanded form loses all fp32
// significance for voxelised content (coordinates ~1e3, distances ~1).
//
// K2 (k2_refine) turns that into the exact fp64 answer.  With u = 2^-24, fp32 d2 of exactly
// representable inputs is within (1 +- 6u) of the real value, and inputs that had to be rounded
// to fp32 add at most 2*sqrt(3)*u*maxabs to the distance.  So every point that can be the fp64
// winner, or tie with it, has d32 <= thr(b1) = (sqrt(b1)*(1+2^-20) + slack)^2.  If the
// second-best granule-min is above thr, the winner lies in the best granule: one wave rescans
// those 64 points in fp64 with the reference's arithmetic ((dx*dx)+(dy*dy))+(dz*dz), no FMA,
// smallest index on exact ties.  Otherwise the query goes to K2b, which rescans the whole
// search cloud, evaluating fp64 only for points with d32 <= thr.
#include "pccm_rescan.h"

namespace pccm {

__device__ __forceinline__ float min3f(float a, float b, float c)
{
    float o;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}

// ------------------------------------------------------------------------------------------
// K1: fp32 scan.  grid = (ceil(nq / (256*QT)), splits).  Each workgroup scans the tiles
// [split*tiles_per_split, ...) of the padded search cloud for its 256*QT queries and writes
// (best granule-min, granule id, second-best granule-min) per query into row `split` of the
// partial arrays [splits][nq].
// ------------------------------------------------------------------------------------------
template <int QT, bool SELF>
__global__ __launch_bounds__(kScanThreads) void k1_scan(const float *__restrict__ q32, int64_t q_begin,
                                                         int64_t nq, const float4 *__restrict__ r32,
                                                         int64_t ntiles, int tiles_per_split,
                                                         float *__restrict__ pb1, int32_t *__restrict__ pg,
                                                         float *__restrict__ pb2)
{
    __shared__ float4 tile[2][kTileVec];
    const int tid = threadIdx.x;
    const int64_t qbase = (int64_t)blockIdx.x * (kScanThreads * QT);

    float qx[QT], qy[QT], qz[QT], b1[QT], b2[QT];
    int bg[QT];
    int64_t qi[QT];   // global row of the query (SELF exclusion only)
#pragma unroll
    for (int k = 0; k < QT; ++k) {
        int64_t i = qbase + (int64_t)k * kScanThreads + tid;
        int64_t ii = q_begin + (i < nq ? i : nq - 1);
        load_pt32(q32, ii, qx[k], qy[k], qz[k]);
        b1[k] = kBig32; b2[k] = kBig32; bg[k] = 0;
        qi[k] = ii;
    }
    const int64_t qlo = q_begin + qbase;
    const int64_t qhi = qlo + (int64_t)kScanThreads * QT;

    const int64_t t0 = (int64_t)blockIdx.y * tiles_per_split;   // host guarantees t0 < ntiles
    const int64_t t1 = (t0 + tiles_per_split < ntiles) ? t0 + tiles_per_split : ntiles;
    constexpr int kStage = kTileVec / kScanThreads;   // 3 x 16 B per thread per tile

    float4 st[kStage];
#pragma unroll
    for (int s = 0; s < kStage; ++s) st[s] = r32[t0 * kTileVec + s * kScanThreads + tid];
#pragma unroll
    for (int s = 0; s < kStage; ++s) tile[0][s * kScanThreads + tid] = st[s];
    // the queries are in registers before the loop: no load wait is left inside it
#pragma unroll
    for (int k = 0; k < QT; ++k) asm volatile("" : "+v"(qx[k]), "+v"(qy[k]), "+v"(qz[k]));
    __syncthreads();

    for (int64_t t = t0; t < t1; ++t) {
        const int cur = (int)((t - t0) & 1);
        const int64_t tn = (t + 1 < t1) ? t + 1 : t;   // last pass re-stages its own tile (unused)
#pragma unroll
        for (int s = 0; s < kStage; ++s) st[s] = r32[tn * kTileVec + s * kScanThreads + tid];
        const float4 *tl = tile[cur];
        for (int g = 0; g < kScanTile / kGranule; ++g) {
            const float4 *gp = tl + g * (kGranule / 4 * 3);
            const int64_t gref0 = t * kScanTile + (int64_t)g * kGranule;
            float tmin[QT];
#pragma unroll
            for (int k = 0; k < QT; ++k) tmin[k] = kBig32;
            if (SELF && gref0 < qhi && gref0 + kGranule > qlo) {
                // granule contains rows of this workgroup's own queries: mask j == i
                for (int c = 0; c < kGranule / 4; ++c) {
                    const float4 X = gp[3 * c], Y = gp[3 * c + 1], Z = gp[3 * c + 2];
                    const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int64_t rj = gref0 + 4 * c + e;
#pragma unroll
                        for (int k = 0; k < QT; ++k) {
                            float d = dist32(qx[k], qy[k], qz[k], xs[e], ys[e], zs[e]);
                            d = (rj == qi[k]) ? kBig32 : d;
                            tmin[k] = d < tmin[k] ? d : tmin[k];
                        }
                    }
                }
            } else {
#pragma unroll 2
                for (int c = 0; c < kGranule / 4; ++c) {
                    const float4 X = gp[3 * c], Y = gp[3 * c + 1], Z = gp[3 * c + 2];
#pragma unroll
                    for (int k = 0; k < QT; ++k) {
                        float d0 = dist32(qx[k], qy[k], qz[k], X.x, Y.x, Z.x);
                        float d1 = dist32(qx[k], qy[k], qz[k], X.y, Y.y, Z.y);
                        float d2 = dist32(qx[k], qy[k], qz[k], X.z, Y.z, Z.z);
                        float d3 = dist32(qx[k], qy[k], qz[k], X.w, Y.w, Z.w);
                        tmin[k] = min3f(min3f(tmin[k], d0, d1), d2, d3);
                    }
                }
            }
            const int gid = (int)(gref0 / kGranule);
#pragma unroll
            for (int k = 0; k < QT; ++k) {
                float tm = tmin[k];
                b2[k] = __builtin_amdgcn_fmed3f(b1[k], b2[k], tm);   // second smallest of {b1<=b2, tm}
                bool imp = tm < b1[k];                                // strict: earliest granule keeps ties
                b1[k] = imp ? tm : b1[k];
                bg[k] = imp ? gid : bg[k];
            }
        }
#pragma unroll
        for (int s = 0; s < kStage; ++s) tile[cur ^ 1][s * kScanThreads + tid] = st[s];
        __syncthreads();
    }

    const int64_t row = (int64_t)blockIdx.y * nq;
#pragma unroll
    for (int k = 0; k < QT; ++k) {
        int64_t i = qbase + (int64_t)k * kScanThreads + tid;
        if (i < nq) {
            pb1[row + i] = b1[k];
            pg[row + i] = bg[k];
            pb2[row + i] = b2[k];
        }
    }
}

// ------------------------------------------------------------------------------------------
// K2: merge the splits, certify, refine the winner granule in fp64.  One wave = 64 queries.
// ------------------------------------------------------------------------------------------
template <bool SELF>
__global__ __launch_bounds__(256) void k2_refine(const double *__restrict__ q64, int64_t q_begin, int64_t nq,
                                                 const double *__restrict__ r64, int64_t nr,
                                                 const float *__restrict__ pb1, const int32_t *__restrict__ pg,
                                                 const float *__restrict__ pb2, int splits, double slack_scale,
                                                 int32_t *__restrict__ idx_out, double *__restrict__ d2_out,
                                                 int32_t *__restrict__ flagged, float *__restrict__ flag_thr,
                                                 uint32_t *__restrict__ nflag)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t i = wave * 64 + lane;
    const bool valid = i < nq;
    const int64_t ii = valid ? i : nq - 1;

    float b1 = kBig32, b2 = kBig32;
    int g = 0;
    for (int s = 0; s < splits; ++s) {
        float p1 = pb1[(int64_t)s * nq + ii], p2 = pb2[(int64_t)s * nq + ii];
        int pgr = pg[(int64_t)s * nq + ii];
        float nb2 = fminf(fmaxf(b1, p1), fminf(b2, p2));   // second smallest of the two sorted pairs
        if (p1 < b1) g = pgr;                               // strict: earliest split keeps ties
        b1 = fminf(b1, p1);
        b2 = nb2;
    }
    // every point that can win or tie in fp64 has d32 <= thr (derivation: file header).  Inexact inputs: such a
    // point lies within sqrt(b1) of the query, so |coordinate| <= |q|_inf + sqrt(b1) for both points involved and
    // the fp32 conversions move the distance by less than slack_scale (2^-20) times that -- per query, so that a
    // stray far-away point does not loosen the test for everybody
    const double qx = q64[3 * (q_begin + ii)], qy = q64[3 * (q_begin + ii) + 1], qz = q64[3 * (q_begin + ii) + 2];
    const double slack = slack_scale * (fmax(fmax(fabs(qx), fabs(qy)), fabs(qz)) + sqrt((double)b1));
    double tq = sqrt((double)b1) * (1.0 + 0x1.0p-20) + slack;
    double thr = tq * tq * (1.0 + 0x1.0p-30) + 1.0e-36;
    const bool amb = valid && !((double)b2 > thr);
    if (amb) {
        uint32_t pos = atomicAdd(nflag, 1u);
        float tf = (float)thr;
        tf = __uint_as_float(__float_as_uint(tf) + 1u);     // round up: never exclude a candidate
        flagged[pos] = (int32_t)i;
        flag_thr[pos] = tf;
    }

    double best = 0.0;
    int bidx = -1;
    unsigned long long todo = __ballot(valid && !amb);
    while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int gl = __shfl(g, l);
        const double x = __shfl(qx, l), y = __shfl(qy, l), z = __shfl(qz, l);
        const int64_t j = (int64_t)gl * kGranule + lane;
        double d = INFINITY;
        bool ok = j < nr;
        if (SELF) ok = ok && (j != q_begin + wave * 64 + l);
        if (ok) d = dist64(x, y, z, r64[3 * j], r64[3 * j + 1], r64[3 * j + 2]);
        const double m = wave_min_f64(d);
        const unsigned long long eq = __ballot(d == m);
        const int win = __ffsll((long long)eq) - 1;         // smallest lane == smallest row index
        if (lane == l) {
            best = m;
            bidx = gl * kGranule + win;
        }
    }
    if (valid && !amb) {
        idx_out[i] = bidx;
        d2_out[i] = best;
    }
}

template <bool SELF>
__global__ __launch_bounds__(256) void k2b_fallback(RescanJobs jobs)
{
    rescan_body<SELF>(jobs.j[blockIdx.y], blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------
static int scan_qt()
{
    static int qt = [] {
        const char *e = PCCM_DIAG_ENV("PCCM_SCAN_QT");
        int v = e ? atoi(e) : 8;
        return (v == 4 || v == 8) ? v : 8;
    }();
    return qt;
}

template <int QT>
static void launch_k1(bool self, dim3 grid, hipStream_t st, const float *q32, int64_t qb, int64_t nq,
                      const float4 *r32, int64_t ntiles, int tps, float *pb1, int32_t *pg, float *pb2)
{
    if (self)
        hipLaunchKernelGGL((k1_scan<QT, true>), grid, dim3(kScanThreads), 0, st, q32, qb, nq, r32, ntiles, tps, pb1, pg, pb2);
    else
        hipLaunchKernelGGL((k1_scan<QT, false>), grid, dim3(kScanThreads), 0, st, q32, qb, nq, r32, ntiles, tps, pb1, pg, pb2);
}

// Exact rescan of the flagged queries of up to two results (counts on the device in res.nflag_dev[0]).
int rescan_jobs(pccm_ctx *ctx, int njobs, const Cloud *const *its, const Cloud *const *ses, NNResult *const *ress, bool self, RescanJobs *out)
{
    RescanJobs &jobs = *out;
    jobs.njobs = njobs;
    const unsigned cap = kRescanCap;
    int rc = ensure(ctx, ctx->rescan_part, (size_t)2 * kSplitMax * cap * (sizeof(double) + sizeof(int32_t)));
    if (rc) return rc;
    for (int k = 0; k < njobs; ++k) {
        const Cloud &it = *its[k], &se = *ses[k];
        NNResult &res = *ress[k];
        RescanJob &J = jobs.j[k];
        J.q32 = (const float *)it.xyz32;
        J.q64 = it.xyz64;
        J.q_begin = res.begin;
        J.r32 = (const float *)se.xyz32;
        J.r64 = se.xyz64;
        J.nr = se.n;
        J.flagged = (const int32_t *)res.flagged.p;
        J.flag_thr = (const float *)res.flag_thr.p;
        J.nflag = res.nflag_dev;
        J.idx_out = res.idx;
        J.d2_out = res.d2;
        // the grid engine's results are 32-byte records (nn_grid set rec_valid for this run); the brute engine writes columns
        J.rec_out = res.rec_valid ? (double *)res.rec.p : nullptr;
        J.rec_stride = res.rec_stride;
        J.rec_layout = res.rec_valid ? res.rec_layout : 0;
        J.nrm = (res.rec_valid && res.fused_mode >= 0) ? se.nrm64 : nullptr;
        J.normal_mode = res.fused_mode >= 0 ? res.fused_mode : PCCM_NORMAL_ROW;
        J.part_d = (double *)ctx->rescan_part.p + (size_t)k * kSplitMax * cap;
        J.part_j = (int32_t *)((double *)ctx->rescan_part.p + (size_t)2 * kSplitMax * cap) + (size_t)k * kSplitMax * cap;
        J.ticket = (uint32_t *)ctx->counters.p + 8 + k + (self ? 2 : 0);
    }
    if (njobs == 1) jobs.j[1] = jobs.j[0];
    return PCCM_OK;
}

int launch_fallback(pccm_ctx *ctx, int njobs, const Cloud *const *its, const Cloud *const *ses, NNResult *const *ress, bool self)
{
    ProfScope ps(ctx, PCCM_K_FALLBACK);
    RescanJobs jobs;
    int rc = rescan_jobs(ctx, njobs, its, ses, ress, self, &jobs);
    if (rc) return rc;
    const unsigned cap = kRescanCap;     // workgroups: enough to split a cloud finely, few enough that an empty list costs a short launch
    int64_t nqmax = 1;
    for (int k = 0; k < njobs; ++k) {
        const int64_t nq = ress[k]->end - ress[k]->begin;
        nqmax = nq > nqmax ? nq : nqmax;
    }
    if (njobs == 1) jobs.j[1] = jobs.j[0];
    // enough workgroups to split a cloud finely; never more than there are queries to hand out one each
    dim3 grid((unsigned)(nqmax < cap ? nqmax : cap), (unsigned)njobs);
    if (self) hipLaunchKernelGGL((k2b_fallback<true>), grid, dim3(256), 0, ctx->stream, jobs);
    else hipLaunchKernelGGL((k2b_fallback<false>), grid, dim3(256), 0, ctx->stream, jobs);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

int nn_brute(pccm_ctx *ctx, const Cloud &it, const Cloud &se, bool self, NNResult &res)
{
    const int64_t nq = res.end - res.begin;
    if (nq <= 0) return PCCM_OK;
    const int qt = scan_qt();
    const int64_t qblocks = (nq + (int64_t)kScanThreads * qt - 1) / ((int64_t)kScanThreads * qt);
    const int64_t ntiles = se.n_pad / kScanTile;
    // enough workgroups for ~6 rounds over 256 CUs x 4 resident workgroups, so the last round is short
    const int64_t target = 6144;
    int64_t splits = (target + qblocks - 1) / qblocks;
    if (splits > ntiles) splits = ntiles;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
    const int tps = (int)((ntiles + splits - 1) / splits);
    splits = (ntiles + tps - 1) / tps;

    int rc;
    if ((rc = ensure(ctx, ctx->part_b1, (size_t)splits * nq * sizeof(float)))) return rc;
    if ((rc = ensure(ctx, ctx->part_g, (size_t)splits * nq * sizeof(int32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->part_b2, (size_t)splits * nq * sizeof(float)))) return rc;
    if ((rc = ensure(ctx, res.flagged, (size_t)nq * sizeof(int32_t)))) return rc;
    if ((rc = ensure(ctx, res.flag_thr, (size_t)nq * sizeof(float)))) return rc;
    PCCM_HIP(hipMemsetAsync(res.nflag_dev, 0, 2 * sizeof(uint32_t), ctx->stream));

    float *pb1 = (float *)ctx->part_b1.p;
    int32_t *pg = (int32_t *)ctx->part_g.p;
    float *pb2 = (float *)ctx->part_b2.p;
    {
        ProfScope ps(ctx, PCCM_K_SCAN);
        dim3 grid((unsigned)qblocks, (unsigned)splits);
        if (qt == 4) launch_k1<4>(self, grid, ctx->stream, (const float *)it.xyz32, res.begin, nq, se.xyz32, ntiles, tps, pb1, pg, pb2);
        else launch_k1<8>(self, grid, ctx->stream, (const float *)it.xyz32, res.begin, nq, se.xyz32, ntiles, tps, pb1, pg, pb2);
    }
    PCCM_HIP(hipGetLastError());

    // fp32 rounding of inexact inputs moves a distance by at most 2*sqrt(3)*2^-24*|coordinate|; 2^-20 times the
    // magnitude covers it (and the fp32 arithmetic of the scan) with a wide margin (applied per query in K2)
    const bool exact = it.exact32 && se.exact32;
    const double slack = exact ? 0.0 : 0x1.0p-20;
    {
        ProfScope ps(ctx, PCCM_K_REFINE);
        const int64_t waves = (nq + 63) / 64;
        dim3 grid((unsigned)((waves + 3) / 4));
        if (self)
            hipLaunchKernelGGL((k2_refine<true>), grid, dim3(256), 0, ctx->stream, it.xyz64, res.begin, nq, se.xyz64, se.n,
                               pb1, pg, pb2, (int)splits, slack, res.idx, res.d2, (int32_t *)res.flagged.p,
                               (float *)res.flag_thr.p, res.nflag_dev);
        else
            hipLaunchKernelGGL((k2_refine<false>), grid, dim3(256), 0, ctx->stream, it.xyz64, res.begin, nq, se.xyz64, se.n,
                               pb1, pg, pb2, (int)splits, slack, res.idx, res.d2, (int32_t *)res.flagged.p,
                               (float *)res.flag_thr.p, res.nflag_dev);
    }
    PCCM_HIP(hipGetLastError());
    {
        const Cloud *its[1] = {&it}, *ses[1] = {&se};
        NNResult *ress[1] = {&res};
        if ((rc = launch_fallback(ctx, 1, its, ses, ress, self))) return rc;
    }
    res.stats[1] = splits;
    res.stats[2] = nq * se.n;
    return PCCM_OK;
}

}  // namespace pccm
