// Grid build for gfx950: counting sort of both clouds by cell without a single global atomic per point.
//
// Stands under the two KDTreeFlann builds of open_pcc_metric/cloud_pair.py:65 (the search structure is
// rebuilt every step, like the trees).
//
// Round 1 ranked every point with one device-scope atomicAdd on its cell counter (78 us for 2M points:
// scattered agent-scope atomics run ~17x slower than contiguous ones, MI355X_MICROARCH.md "Global float
// atomics") and scattered 32-byte records from row order.  Here the sort is two-level, three kernels, and the
// per-point atomics are LDS atomics:
//   K_A  k_bin_count    one workgroup per tile of rows: LDS histogram over coarse bins (a bin = a run of
//                       2^lg consecutive cells); the tile's share of a bin starts where the bin's running
//                       total stood -- one returning global atomic per non-empty (tile, bin), 64 consecutive
//                       words per wave instruction -> toff[tile][bin]
//   K_B  k_bin_scatter  same tiles: every tile scans the job's bin totals in LDS itself (no scan kernel), adds
//                       its toff row, and sends the records to their bin (tmp array) through LDS cursors
//   K_C  k_bin_sort     one workgroup per bin: cell histogram + ranks by LDS atomics, block scan, cell_start
//                       written coalesced, records written into their cell (registers hold a bin of up to
//                       kRegRecs records between the two phases; larger bins are streamed twice); clears the
//                       bin's running total for the next build and the searches' counters for the caller
// (PCCM_BUILD_SCAN=1 keeps round 2's first form for A/B runs: hist[job][bin][tile] + a decoupled look-back
// scan kernel between K_A and K_B; 3 us slower per build, one more graph node.)
// Algorithmic bytes per point (Rec32 records): 12 (K_A) + 12 + 16 (K_B) + 16 + 16 (K_C) = 72, plus
// 4 B per cell for cell_start.  The order of the records inside a cell depends on the arrival order of
// LDS atomics, the order of the tiles inside a bin on that of the global ones; no result depends on either
// (ties are decided by (d2, row) explicitly).
#include "pccm_grid.h"

namespace pccm {

// ---- single-pass exclusive scan (decoupled look-back) --------------------------------------------------
// A tile of 16384 counters per workgroup; tiles are handed out by an atomic ticket, so every predecessor of a
// running tile is itself running or done and the look-back cannot starve.  Every tile publishes
// (status, value) in one 64-bit word -- first its own total (status 1), then, once the totals of all earlier
// tiles are known, its inclusive prefix (status 2); wave 0 of the tile looks back 64 predecessors at a time.
// `state` = [ntiles] words followed by the ticket counter, all zero on entry.
constexpr int kLbItems = 64;
constexpr int kLbTile = 256 * kLbItems;

int64_t scan_tiles(int64_t m) { return (m + kLbTile - 1) / kLbTile; }

__global__ __launch_bounds__(256) void k_scan_lookback(uint32_t *__restrict__ data, int64_t m, unsigned long long *__restrict__ state,
                                                       int64_t ntiles)
{
    __shared__ uint32_t s_tile, s_prefix, wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_tile = (uint32_t)atomicAdd(&state[ntiles], 1ull);
    __syncthreads();
    const int64_t tile = s_tile;
    const int64_t base = tile * kLbTile + (int64_t)tid * kLbItems;
    uint32_t v[kLbItems], tot = 0;
    if (base + kLbItems <= m) {
#pragma unroll
        for (int k = 0; k < kLbItems; k += 4) {
            const uint4 q = *reinterpret_cast<const uint4 *>(&data[base + k]);
            v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kLbItems; ++k) v[k] = (base + k < m) ? data[base + k] : 0u;
    }
#pragma unroll
    for (int k = 0; k < kLbItems; ++k) tot += v[k];
    uint32_t inc = tot;                                   // inclusive scan of thread totals in the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum[k];
    const uint32_t agg = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (w == 0) {
        if (lane == 0)
            __hip_atomic_store(&state[tile], ((tile == 0 ? 2ull : 1ull) << 32) | agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        int64_t p = tile - 1;
        while (p >= 0) {
            const int64_t idx = p - lane;
            const unsigned long long st = idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                   : (2ull << 32);          // before the first tile: prefix 0
            const uint32_t status = (uint32_t)(st >> 32), val = (uint32_t)st;
            const unsigned long long incl = __ballot(status == 2u), empty = __ballot(status == 0u);
            const int first = incl ? __ffsll((long long)incl) - 1 : 64;     // nearest predecessor with a full prefix
            const unsigned long long nearer = first >= 64 ? ~0ull : ((1ull << first) - 1ull);
            if (empty & nearer) {                                            // someone nearer has not published yet
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            uint32_t part = (lane <= first) ? val : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
            excl += part;
            if (first < 64) break;
            p -= 64;
        }
        if (lane == 0) {
            if (tile > 0) __hip_atomic_store(&state[tile], (2ull << 32) | (unsigned long long)(excl + agg), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
            s_prefix = excl;
        }
    }
    __syncthreads();
    uint32_t run = s_prefix + woff + inc - tot;
    if (base + kLbItems <= m) {
#pragma unroll
        for (int k = 0; k < kLbItems; k += 4) {
            uint4 q;
            q.x = run; run += v[k];
            q.y = run; run += v[k + 1];
            q.z = run; run += v[k + 2];
            q.w = run; run += v[k + 3];
            *reinterpret_cast<uint4 *>(&data[base + k]) = q;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kLbItems; ++k) {
            if (base + k < m) data[base + k] = run;
            run += v[k];
        }
    }
}

// ---- the two-level counting sort ---------------------------------------------------------------------------
constexpr int kRegK = 8;                    // records a thread of k_bin_sort keeps in registers
constexpr int kRegRecs = 256 * kRegK;
constexpr int kMaxLg = 13;                  // cells per bin <= 8192 (32 KB of LDS counters)

struct BinJob {
    const double *x64;
    const float *x32;
    int64_t row0, n;
    uint32_t *cs;          // [ncells + 1]
    int64_t tl;            // rows per tile
    int64_t nt;            // tiles of this job
    int64_t hoff;          // first hist entry of this job (= nbin * tiles of the jobs before it)
    int64_t base;          // records of the jobs before this one
    int64_t tile0;         // tiles of the jobs before this one
    const float4 *sp;      // the cloud in its spatial order, {x, y, z, original row} (BuildJob::sp), or null
    uint32_t *occ;         // occupancy bitmap to write (BuildJob::occ), or null
};

// cell starts of a bin out of its scanned counters (s_cnt[c] = records of the bin before cell c; m = records of the bin;
// first = the bin's first record, relative to the job's), and -- when the job keeps one -- the occupancy bitmap: one bit per
// cell, set when the cell holds a record (every wave covers 64 consecutive cells, i.e. two whole words: bins start on
// multiples of 1024 cells)
__device__ __forceinline__ void write_cell_starts(const BinJob &J, const uint32_t *s_cnt, int64_t c0, int nc, uint32_t first, uint32_t m)
{
    const int tid = threadIdx.x, nth = (int)blockDim.x;
    if (!J.occ) {
        for (int c = tid; c < nc; c += nth) J.cs[c0 + c] = first + s_cnt[c];
        return;
    }
    for (int c = tid; c < ((nc + 63) & ~63); c += nth) {
        const uint32_t here = c < nc ? s_cnt[c] : m, next = c + 1 < nc ? s_cnt[c + 1] : m;
        if (c < nc) J.cs[c0 + c] = first + here;
        const unsigned long long bits = __ballot(next != here);
        if ((tid & 63) == 0) {
            J.occ[(c0 + c) >> 5] = (uint32_t)bits;
            J.occ[((c0 + c) >> 5) + 1] = (uint32_t)(bits >> 32);
        }
    }
}

struct BinPlan {
    BinJob j[2];
    int njobs;
    int lg;                // log2(cells per bin)
    int nbin;              // bins per job
    int64_t ncells;
    int64_t ntiles;        // all jobs
    int64_t hlen;          // nbin * ntiles; hist[hlen] is the scan's sentinel (= total after the scan)
    int64_t state_off;     // uint32 offset of the scan's state words behind the histogram (8-byte aligned)
    int64_t nstate;        // scan tiles + 1 (the ticket)
    uint32_t *zero;        // words the last build kernel clears for the caller (the searches' counters), or null
    int nzero;
    // cursor variant (default): no scan kernel.  cursor[job][bin] = running total of the bin (zero on entry; k_bin_sort
    // leaves it zero again), toff[tile][bin] = where the tile's share starts inside the bin, bstart[job][bin] (+ sentinel)
    // = first record of the bin, written by the first tile of each job in k_bin_scatter
    uint32_t *cursor, *toff, *bstart;
};

template <bool X32>
__device__ __forceinline__ void load_point(const BinJob &J, int64_t i, double &x, double &y, double &z)
{
    const int64_t r = J.row0 + i;
    if (X32) {
        const float *q = J.x32 + (r >> 2) * 12 + (r & 3);
        x = (double)q[0];
        y = (double)q[4];
        z = (double)q[8];
    } else {
        const double *p = J.x64 + 3 * r;
        x = p[0];
        y = p[1];
        z = p[2];
    }
}

// four consecutive rows (row0 + i a multiple of 4) of the fp32 "quads" layout: three 16-byte loads
__device__ __forceinline__ void load_quad(const BinJob &J, int64_t i, float4 &x, float4 &y, float4 &z)
{
    const float4 *q = reinterpret_cast<const float4 *>(J.x32 + ((J.row0 + i) >> 2) * 12);
    x = q[0];
    y = q[1];
    z = q[2];
}

// tile -> (job, tile of the job); wave-uniform
__device__ __forceinline__ int tile_job(const BinPlan &P, int64_t &tile)
{
    if (P.njobs > 1 && tile >= P.j[0].nt) {
        tile -= P.j[0].nt;
        return 1;
    }
    return 0;
}

template <bool X32, bool CUR>
__global__ __launch_bounds__(1024) void k_bin_count(BinPlan P, GridGeom g, uint32_t *__restrict__ hist)
{
    extern __shared__ uint32_t s_hist[];                  // [nbin]
    const int tid = threadIdx.x, nth = (int)blockDim.x;     // 1024 threads: a tile is two trips of four rows per thread, all loads in flight
    if (!CUR && blockIdx.x == 0) {                                // the scan's sentinel and state: zero before the scan starts
        unsigned long long *state = reinterpret_cast<unsigned long long *>(hist + P.state_off);
        if (tid == 0) hist[P.hlen] = 0u;
        for (int64_t k = tid; k < P.nstate; k += nth) state[k] = 0ull;
    }
    for (int b = tid; b < P.nbin; b += nth) s_hist[b] = 0u;
    __syncthreads();
    int64_t tile = blockIdx.x;
    const int jb = tile_job(P, tile);
    const BinJob &J = P.j[jb];
    const int64_t i0 = tile * J.tl, i1 = (i0 + J.tl < J.n) ? i0 + J.tl : J.n;
    if (X32 && J.sp) {
        // the cloud in its spatial order: one 16-byte record per row, eight independent rows per trip
        for (int64_t i = i0 + tid; i < i1; i += 8 * nth) {
            float4 r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t ii = i + (int64_t)nth * k;
                r[k] = J.sp[J.row0 + (ii < i1 ? ii : i1 - 1)];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (i + (int64_t)nth * k < i1) atomicAdd(&s_hist[cell_linear(g, (double)r[k].x, (double)r[k].y, (double)r[k].z) >> P.lg], 1u);
        }
    } else if (X32) {
        // a thread takes whole quads of rows (three 16-byte loads each, every byte of a line used), two quads per trip
        const int64_t nquad = (i1 - i0 + 3) >> 2;
        for (int64_t qd = tid; qd < nquad; qd += 2 * nth) {
            float4 x[2], y[2], z[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t q = qd + (int64_t)u * nth;
                load_quad(J, i0 + 4 * (q < nquad ? q : nquad - 1), x[u], y[u], z[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t i = i0 + 4 * (qd + (int64_t)u * nth);
                const float xs[4] = {x[u].x, x[u].y, x[u].z, x[u].w}, ys[4] = {y[u].x, y[u].y, y[u].z, y[u].w},
                            zs[4] = {z[u].x, z[u].y, z[u].z, z[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (i + k < i1) atomicAdd(&s_hist[cell_linear(g, (double)xs[k], (double)ys[k], (double)zs[k]) >> P.lg], 1u);
            }
        }
    } else {
        for (int64_t i = i0 + tid; i < i1; i += 4 * nth) {       // four independent rows per trip
            uint32_t bin[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t ii = i + nth * k;
                double x, y, z;
                load_point<X32>(J, ii < i1 ? ii : i1 - 1, x, y, z);
                bin[k] = cell_linear(g, x, y, z) >> P.lg;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i + nth * k < i1) atomicAdd(&s_hist[bin[k]], 1u);
        }
    }
    __syncthreads();
    if (CUR) {
        // the tile's share of every bin starts where the bin's running total stood: one returning atomic per non-empty
        // (tile, bin), 64 consecutive words per wave instruction; which tile comes first inside a bin is left to chance
        // (like the order inside a cell: no result depends on it)
        uint32_t *cur = P.cursor + (int64_t)jb * P.nbin, *dst = P.toff + (J.tile0 + tile) * P.nbin;
        for (int b = tid; b < P.nbin; b += nth) {
            const uint32_t c = s_hist[b];
            dst[b] = c ? atomicAdd(&cur[b], c) : 0u;
        }
        return;
    }
    uint32_t *dst = hist + J.hoff + tile;
    for (int b = tid; b < P.nbin; b += nth) dst[(int64_t)b * J.nt] = s_hist[b];
}

__device__ __forceinline__ void block_scan_inplace(uint32_t *s_cnt, int nc, uint32_t *s_wsum, uint32_t *s_carry);

template <typename REC, bool X32, bool CUR>
__global__ __launch_bounds__(1024) void k_bin_scatter(BinPlan P, GridGeom g, const uint32_t *__restrict__ hist, REC *__restrict__ tmp)
{
    extern __shared__ uint32_t s_cur[];                   // [nbin]: next free position of each bin's share of this tile
    __shared__ uint32_t s_wsum[16], s_carry;
    const int tid = threadIdx.x, nth = (int)blockDim.x;
    int64_t tile = blockIdx.x;
    const int jb = tile_job(P, tile);
    const BinJob &J = P.j[jb];
    if (CUR) {
        // every tile scans the job's bin totals itself (nbin <= 8192 words of LDS, a few barriers): no scan kernel
        const uint32_t *tot = P.cursor + (int64_t)jb * P.nbin, *off = P.toff + (J.tile0 + tile) * P.nbin;
        uint32_t mine[2];                                 // this thread's first toff words, in flight during the scan
#pragma unroll
        for (int k = 0; k < 2; ++k) mine[k] = (tid + nth * k < P.nbin) ? off[tid + nth * k] : 0u;
        for (int b = tid; b < P.nbin; b += nth) s_cur[b] = tot[b];
        __syncthreads();
        block_scan_inplace(s_cur, P.nbin, s_wsum, &s_carry);
        if (tile == 0) {
            uint32_t *bs = P.bstart + (int64_t)jb * P.nbin;
            for (int b = tid; b < P.nbin; b += nth) bs[b] = (uint32_t)J.base + s_cur[b];
        }
        if (blockIdx.x == 0) {                            // the sentinel, and the bins of a job without rows (it has no tile)
            const BinJob &L = P.j[P.njobs - 1];
            if (tid == 0) P.bstart[(int64_t)P.njobs * P.nbin] = (uint32_t)(L.base + L.n);
            for (int k = 0; k < P.njobs; ++k)
                if (P.j[k].nt == 0)
                    for (int b = tid; b < P.nbin; b += nth) P.bstart[(int64_t)k * P.nbin + b] = (uint32_t)P.j[k].base;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (tid + nth * k < P.nbin) s_cur[tid + nth * k] += (uint32_t)J.base + mine[k];
        for (int b = tid + 2 * nth; b < P.nbin; b += nth) s_cur[b] += (uint32_t)J.base + off[b];
    } else {
        const uint32_t *src = hist + J.hoff + tile;
        for (int b = tid; b < P.nbin; b += nth) s_cur[b] = src[(int64_t)b * J.nt];
    }
    __syncthreads();
    const int64_t i0 = tile * J.tl, i1 = (i0 + J.tl < J.n) ? i0 + J.tl : J.n;
    if (X32 && J.sp) {
        // spatial order: the record is already there (with its original row); it only has to find its bin
        float4 *__restrict__ out = reinterpret_cast<float4 *>(tmp);
        for (int64_t i = i0 + tid; i < i1; i += 8 * nth) {
            float4 r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t ii = i + (int64_t)nth * k;
                r[k] = J.sp[J.row0 + (ii < i1 ? ii : i1 - 1)];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (i + (int64_t)nth * k < i1) {
                    const uint32_t pos = atomicAdd(&s_cur[cell_linear(g, (double)r[k].x, (double)r[k].y, (double)r[k].z) >> P.lg], 1u);
                    if (sizeof(REC) == 16) out[pos] = r[k];
                }
        }
        return;
    }
    if (X32) {
        const int64_t nquad = (i1 - i0 + 3) >> 2;
        for (int64_t qd = tid; qd < nquad; qd += 2 * nth) {
            float4 x[2], y[2], z[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t q = qd + (int64_t)u * nth;
                load_quad(J, i0 + 4 * (q < nquad ? q : nquad - 1), x[u], y[u], z[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t i = i0 + 4 * (qd + (int64_t)u * nth);
                const float xs[4] = {x[u].x, x[u].y, x[u].z, x[u].w}, ys[4] = {y[u].x, y[u].y, y[u].z, y[u].w},
                            zs[4] = {z[u].x, z[u].y, z[u].z, z[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (i + k < i1) {
                        P3 v;
                        v.x = (double)xs[k]; v.y = (double)ys[k]; v.z = (double)zs[k];
                        v.row = (int)(J.row0 + i + k);
                        const uint32_t pos = atomicAdd(&s_cur[cell_linear(g, v.x, v.y, v.z) >> P.lg], 1u);
                        store_rec(tmp, pos, v);
                    }
            }
        }
        return;
    }
    for (int64_t i = i0 + tid; i < i1; i += 4 * nth) {
        P3 v[4];
        uint32_t bin[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t ii = (i + nth * k < i1) ? i + nth * k : i1 - 1;
            load_point<X32>(J, ii, v[k].x, v[k].y, v[k].z);
            v[k].row = (int)(J.row0 + ii);
            bin[k] = cell_linear(g, v[k].x, v[k].y, v[k].z) >> P.lg;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + nth * k < i1) {
                const uint32_t pos = atomicAdd(&s_cur[bin[k]], 1u);
                store_rec(tmp, pos, v[k]);
            }
    }
}

// block-wide exclusive scan of s_cnt[0..nc) in place (any block of whole waves, s_wsum[16]); leaves offsets
__device__ __forceinline__ void block_scan_inplace(uint32_t *s_cnt, int nc, uint32_t *s_wsum, uint32_t *s_carry)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) *s_carry = 0u;
    __syncthreads();
    const int nth = (int)blockDim.x;
    for (int base = 0; base < nc; base += 4 * nth) {
        // four consecutive counters per thread: one 16-byte LDS access each way
        const int e = base + 4 * tid;
        uint32_t c0 = e < nc ? s_cnt[e] : 0u, c1 = e + 1 < nc ? s_cnt[e + 1] : 0u;
        uint32_t c2 = e + 2 < nc ? s_cnt[e + 2] : 0u, c3 = e + 3 < nc ? s_cnt[e + 3] : 0u;
        const uint32_t tot = c0 + c1 + c2 + c3;
        uint32_t inc = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_wsum[w] = inc;
        __syncthreads();
        uint32_t run = *s_carry + inc - tot;
        for (int k = 0; k < w; ++k) run += s_wsum[k];
        if (e < nc) s_cnt[e] = run;
        run += c0;
        if (e + 1 < nc) s_cnt[e + 1] = run;
        run += c1;
        if (e + 2 < nc) s_cnt[e + 2] = run;
        run += c2;
        if (e + 3 < nc) s_cnt[e + 3] = run;
        __syncthreads();
        if (tid == nth - 1) *s_carry = run + c3;
        __syncthreads();
    }
}

// raw records (no widening to fp64 in registers: the bin's records wait in VGPRs while the cells are counted)
__device__ __forceinline__ uint32_t rec_cell(const GridGeom &g, const float4 &r) { return cell_linear(g, (double)r.x, (double)r.y, (double)r.z); }
__device__ __forceinline__ uint32_t rec_cell(const GridGeom &g, const double4 &r) { return cell_linear(g, r.x, r.y, r.z); }
template <typename REC> struct RawRec;
template <> struct RawRec<Rec32> { typedef float4 type; };
template <> struct RawRec<GridRec> { typedef double4 type; };

template <typename REC>
__global__ __launch_bounds__(256, sizeof(REC) == 16 ? 6 : 4) void k_bin_sort(BinPlan P, GridGeom g, const uint32_t *__restrict__ hist, const REC *__restrict__ tmp,
                                                  REC *__restrict__ recs)
{
    extern __shared__ uint32_t s_cnt[];                   // [cells of a bin]
    __shared__ uint32_t s_wsum[16], s_carry;
    const int tid = threadIdx.x;
    const int jb = (P.njobs > 1 && (int)blockIdx.x >= P.nbin) ? 1 : 0;
    const int b = (int)blockIdx.x - jb * P.nbin;
    const BinJob &J = P.j[jb];
    uint32_t s, e, j0;                                    // the bin's records; j0 = first record of this job (cell starts are relative to it)
    if (P.cursor) {
        const uint32_t *bs = P.bstart + (int64_t)jb * P.nbin + b;
        s = bs[0];
        e = bs[1];
        j0 = (uint32_t)J.base;
        if (tid == 0) P.cursor[(int64_t)jb * P.nbin + b] = 0u;   // ready for the next build
    } else {
        s = hist[J.hoff + (int64_t)b * J.nt];
        e = hist[J.hoff + (int64_t)(b + 1) * J.nt];
        j0 = hist[J.hoff];
    }
    const uint32_t m = e - s;
    const int64_t c0 = (int64_t)b << P.lg;
    const int nc = (int)((c0 + (1ll << P.lg) < P.ncells ? c0 + (1ll << P.lg) : P.ncells) - c0);
    for (int c = tid; c < nc; c += 256) s_cnt[c] = 0u;
    __syncthreads();
    if (m <= (uint32_t)kRegRecs) {
        // the whole bin in registers: rank by the counting atomic, place after the scan
        typedef typename RawRec<REC>::type RAW;
        const RAW *__restrict__ src = reinterpret_cast<const RAW *>(tmp);
        RAW *__restrict__ dst = reinterpret_cast<RAW *>(recs);
        RAW v[kRegK];
        uint32_t cell[kRegK], rank[kRegK];
#pragma unroll
        for (int k = 0; k < kRegK; ++k) {
            const uint32_t i = tid + 256u * k;
            if (i < m) v[k] = src[s + i];
        }
#pragma unroll
        for (int k = 0; k < kRegK; ++k)
            if (tid + 256u * k < m) {
                cell[k] = rec_cell(g, v[k]) - (uint32_t)c0;
                rank[k] = atomicAdd(&s_cnt[cell[k]], 1u);
            }
        __syncthreads();
        block_scan_inplace(s_cnt, nc, s_wsum, &s_carry);
        write_cell_starts(J, s_cnt, c0, nc, s - j0, m);
#pragma unroll
        for (int k = 0; k < kRegK; ++k)
            if (tid + 256u * k < m) dst[s + s_cnt[cell[k]] + rank[k]] = v[k];
    } else {
        for (uint32_t i = tid; i < m; i += 256) {
            const P3 v = load_rec(tmp, s + i);
            atomicAdd(&s_cnt[cell_linear(g, v.x, v.y, v.z) - (uint32_t)c0], 1u);
        }
        __syncthreads();
        block_scan_inplace(s_cnt, nc, s_wsum, &s_carry);
        write_cell_starts(J, s_cnt, c0, nc, s - j0, m);
        __syncthreads();                                   // cell_start is out: the offsets become cursors
        for (uint32_t i = tid; i < m; i += 256) {
            const P3 v = load_rec(tmp, s + i);
            const uint32_t pos = atomicAdd(&s_cnt[cell_linear(g, v.x, v.y, v.z) - (uint32_t)c0], 1u);
            store_rec(recs, s + pos, v);
        }
    }
    if (b == P.nbin - 1 && tid == 0) J.cs[P.ncells] = e - j0;   // number of records of the job
    if (blockIdx.x == 0 && tid < P.nzero) P.zero[tid] = 0u;     // saves the caller a memset node
}

static int ceil_log2(int64_t v)
{
    int l = 0;
    while ((1ll << l) < v) ++l;
    return l;
}

int sort_by_cell(pccm_ctx *ctx, const BuildJobs &jobs, const GridGeom &g, int64_t ncells, void *recs, bool rec32, uint32_t *zero, int nzero)
{
    BinPlan P;
    P.zero = zero;
    P.nzero = zero ? nzero : 0;
    P.njobs = jobs.njobs;
    P.ncells = ncells;
    // bins of 2^10 cells while that keeps the bins of a cloud below 8192 (the tiles' LDS histogram): a bin of ~1400 records
    // is sorted from registers; 2^11 cells at 8M + 8M points meant streaming every bin twice (585 instead of 503 us per build)
    int lg = ceil_log2((ncells + 8191) / 8192);
    P.lg = lg < 10 ? 10 : (lg > kMaxLg ? kMaxLg : lg);
    // crowded cells (voxel-brick grids: 8^3 voxels per cell, dozens of points in an occupied one): smaller bins, so that a bin
    // still fits the register path of k_bin_sort and there are enough of them to fill the chip
    {
        int64_t nmax = 0;
        for (int k = 0; k < jobs.njobs; ++k) nmax = jobs.j[k].n > nmax ? jobs.j[k].n : nmax;
        while (P.lg > 6 && (double)nmax * (double)(1ll << P.lg) / (double)ncells > 2000.0 && ((ncells >> (P.lg - 1)) + 1) <= 8192) --P.lg;
    }
    static const int lg_env = [] { const char *e = PCCM_DIAG_ENV("PCCM_BUILD_LG"); return e ? atoi(e) : 0; }();
    if (lg_env >= 8 && lg_env <= kMaxLg && (ncells >> lg_env) < 8192) P.lg = lg_env;
    P.nbin = (int)((ncells + (1ll << P.lg) - 1) >> P.lg);
    if (P.nbin > 8192) return fail(PCCM_E_ARG, "grid of %lld cells is too large", (long long)ncells);
    P.ntiles = 0;
    for (int k = 0; k < 2; ++k) {
        const BuildJob &s = jobs.j[k < jobs.njobs ? k : 0];
        BinJob &d = P.j[k];
        d.x64 = s.x64; d.x32 = s.x32; d.row0 = s.row0; d.cs = s.cs;
        d.sp = rec32 ? reinterpret_cast<const float4 *>(s.sp) : nullptr;
        d.occ = s.occ;
        if (rec32 && k < jobs.njobs && (s.row0 & 3)) return fail(PCCM_E_ARG, "a shard must start on a multiple of 4 rows (it starts on 128-row units)");
        d.n = k < jobs.njobs ? s.n : 0;
        // rows per tile: with bin cursors every (tile, bin) costs one returning atomic and one toff word, so larger tiles
        // are cheaper as long as the tiles still fill the chip: 8192 rows -> 60 us per build at 1M + 1M points (4096: 64, 2048: 74)
        static const int64_t tile_rows = [] { const char *e = getenv("PCCM_BUILD_TILE"); int64_t v = e ? atoll(e) : 8192; return v >= 256 ? v : 8192; }();
        int64_t nt = (d.n + tile_rows - 1) / tile_rows;
        if (nt > 1024) nt = 1024;
        d.nt = nt;
        d.tl = nt > 0 ? ((d.n + nt - 1) / nt + 255) / 256 * 256 : 256;
        if (nt > 0) d.nt = (d.n + d.tl - 1) / d.tl;       // rounding tl up may leave the last tiles empty: drop them
        d.hoff = (int64_t)P.nbin * P.ntiles;
        d.tile0 = P.ntiles;
        d.base = k == 0 ? 0 : P.j[0].n;
        if (k < jobs.njobs) P.ntiles += d.nt;
    }
    P.hlen = (int64_t)P.nbin * P.ntiles;
    const int64_t st = scan_tiles(P.hlen + 1);
    P.nstate = st + 1;
    P.state_off = (P.hlen + 2) / 2 * 2;
    const size_t rsz = rec32 ? sizeof(Rec32) : sizeof(GridRec);
    static const int bin_threads = [] { const char *e = getenv("PCCM_BUILD_THREADS"); int v = e ? atoi(e) : 1024; return (v >= 64 && v <= 1024 && v % 64 == 0) ? v : 1024; }();
    static const bool use_scan = [] { const char *e = getenv("PCCM_BUILD_SCAN"); return e && e[0] == '1'; }();   // A/B: round-2's first form (hist matrix + look-back scan)
    const bool cur = !use_scan && P.nbin <= 8192;
    const int64_t ncur = (int64_t)P.njobs * P.nbin;
    const size_t cur_words = (size_t)(2 * 8192 + 2 * 8192 + 2);          // cursor | bstart (+ sentinel), fixed places
    const size_t hist_bytes = cur ? (cur_words + (size_t)P.hlen) * sizeof(uint32_t)
                                  : (size_t)P.state_off * sizeof(uint32_t) + (size_t)P.nstate * 8;
    int rc;
    const size_t bins_before = ctx->g_bins.bytes;        // ensure() only ever grows a buffer
    if ((rc = ensure(ctx, ctx->g_bins, hist_bytes > cur_words * sizeof(uint32_t) ? hist_bytes : cur_words * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->g_tmp, (size_t)(jobs.total > 0 ? jobs.total : 1) * rsz))) return rc;
    uint32_t *hist = (uint32_t *)ctx->g_bins.p;
    P.cursor = P.toff = P.bstart = nullptr;
    if (cur) {
        P.cursor = hist;
        P.bstart = hist + 2 * 8192;
        P.toff = hist + cur_words;
        if (ctx->g_bins.bytes != bins_before || !ctx->bins_clean) {           // new memory, or a build that did not finish: cursors to zero
            if (ctx->capturing) {
                ctx->capture_failed = true;
                return fail(PCCM_E_STATE, "the grid build's scratch must exist before graph capture: run the sequence once first");
            }
            PCCM_HIP(hipMemsetAsync(hist, 0, (size_t)2 * 8192 * sizeof(uint32_t), ctx->stream));
        }
        ctx->bins_clean = false;
    }
    unsigned long long *state = reinterpret_cast<unsigned long long *>(hist + P.state_off);
    const size_t lds_bins = (size_t)P.nbin * sizeof(uint32_t), lds_cells = ((size_t)1 << P.lg) * sizeof(uint32_t);
    if (P.ntiles > 0 && cur) {
        dim3 tg((unsigned)P.ntiles);
        if (rec32) hipLaunchKernelGGL((k_bin_count<true, true>), tg, dim3(bin_threads), lds_bins, ctx->stream, P, g, hist);
        else hipLaunchKernelGGL((k_bin_count<false, true>), tg, dim3(bin_threads), lds_bins, ctx->stream, P, g, hist);
        if (rec32) hipLaunchKernelGGL((k_bin_scatter<Rec32, true, true>), tg, dim3(bin_threads), lds_bins, ctx->stream, P, g, (const uint32_t *)hist, (Rec32 *)ctx->g_tmp.p);
        else hipLaunchKernelGGL((k_bin_scatter<GridRec, false, true>), tg, dim3(bin_threads), lds_bins, ctx->stream, P, g, (const uint32_t *)hist, (GridRec *)ctx->g_tmp.p);
    } else if (P.ntiles > 0) {
        dim3 tg((unsigned)P.ntiles);
        if (rec32) hipLaunchKernelGGL((k_bin_count<true, false>), tg, dim3(bin_threads), lds_bins, ctx->stream, P, g, hist);
        else hipLaunchKernelGGL((k_bin_count<false, false>), tg, dim3(bin_threads), lds_bins, ctx->stream, P, g, hist);
        hipLaunchKernelGGL(k_scan_lookback, dim3((unsigned)st), dim3(256), 0, ctx->stream, hist, P.hlen + 1, state, st);
        if (rec32) hipLaunchKernelGGL((k_bin_scatter<Rec32, true, false>), tg, dim3(bin_threads), lds_bins, ctx->stream, P, g, (const uint32_t *)hist, (Rec32 *)ctx->g_tmp.p);
        else hipLaunchKernelGGL((k_bin_scatter<GridRec, false, false>), tg, dim3(bin_threads), lds_bins, ctx->stream, P, g, (const uint32_t *)hist, (GridRec *)ctx->g_tmp.p);
    } else if (cur) {
        PCCM_HIP(hipMemsetAsync(P.bstart, 0, (size_t)(ncur + 1) * sizeof(uint32_t), ctx->stream));
    } else {
        PCCM_HIP(hipMemsetAsync(hist, 0, hist_bytes, ctx->stream));
    }
    dim3 bg((unsigned)(P.nbin * P.njobs));
    if (rec32) hipLaunchKernelGGL((k_bin_sort<Rec32>), bg, dim3(256), lds_cells, ctx->stream, P, g, (const uint32_t *)hist, (const Rec32 *)ctx->g_tmp.p, (Rec32 *)recs);
    else hipLaunchKernelGGL((k_bin_sort<GridRec>), bg, dim3(256), lds_cells, ctx->stream, P, g, (const uint32_t *)hist, (const GridRec *)ctx->g_tmp.p, (GridRec *)recs);
    PCCM_HIP(hipGetLastError());
    ctx->bins_clean = true;                                // k_bin_sort has been queued: it leaves the cursors at zero
    return PCCM_OK;
}

}  // namespace pccm
