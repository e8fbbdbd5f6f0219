// Colour columns on the device (gfx950): neighbour gather, colour-space transform, squared differences,
// their column maxima and their column sums in the reference's summation order.
//
// Reference: open_pcc_metric/metric.py:261-290 (transform_colors), :302-333 (ColorMSE),
// :389-427 (ColorHausdorffDistance), cloud_pair.py:114-124 (the colour getters).  Per row i of the
// iterating cloud:  own = T(rgb_own[i]),  other = T(rgb_other[nn(i)]),  diff = scale * (own - other),
// sq = diff * diff;  ColorMSE = np.mean(sq, axis=0),  ColorHausdorffDistance = np.max(sq, axis=0).
//
// k_color_rows   one thread per row; HBM-bound (24 B own + 4 B row + 24 B gathered + 24 B out per row).
// k_color_colsum np.mean(axis=0) of a C-contiguous (N, 3) array is NOT NumPy's pairwise sum: the
//                reduction runs row by row, so every column is a plain left-to-right fp64 sum.  A chain
//                of N dependent adds costs ~8 cycles each on one lane (3.8 ms at N = 1M).  The kernel
//                below reproduces the same N roundings without the chain -- see the comment on it.
#include "pccm_internal.h"

namespace pccm {

// metric.py:270-281; row r of T = fma(m[r][2], c2, fma(m[r][0], c0, m[r][1] * c1)): what np.matmul(M, c)
// evaluates on the authoring host (pinned by tests/golden/*color*), see pccm_color_transform
__constant__ double kColourMatrix[2][9] = {
    {0.2126, 0.7152, 0.0722, -0.1146, -0.3854, 0.5, 0.5, -0.4542, -0.0458},      // "ycc" (BT.709)
    {0.25, 0.5, 0.25, 1, 0, -1, -0.5, 1, -0.5},                                  // "yuv"
};

__device__ __forceinline__ void to_scheme(int scheme, const double c[3], double o[3])
{
    if (scheme == 0) {                      // "rgb": transform_colors returns its input, metric.py:266-267
        o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
        return;
    }
    const double *m = kColourMatrix[scheme - 1];
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = fma(m[3 * r + 2], c[2], fma(m[3 * r], c[0], m[3 * r + 1] * c[1]));
}

__device__ __forceinline__ unsigned long long max_key(double v)
{
    // squares are >= +0 or NaN: the bit pattern orders them, and a NaN ranks above everything (np.max propagates it)
    return isnan(v) ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(v);
}

// what: 0 own rows in the scheme | 1 neighbour rows in the scheme | 2 scale * (own - other) | 3 its square
//       (all AoS [n][3] into `out`) | 4 squares as three columns out[c * n + i] + column maxima (bit keys)
__global__ __launch_bounds__(256) void k_color_rows(const double *__restrict__ own, const double *__restrict__ other,
                                                    const int32_t *__restrict__ rows, int64_t n, int64_t n_other,
                                                    int scheme, double scale, int what, double *__restrict__ out,
                                                    unsigned long long *__restrict__ maxkeys, unsigned int *__restrict__ bad)
{
    __shared__ unsigned long long s_max[4][3];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long key[3] = {0ull, 0ull, 0ull};
    if (i < n) {
        double a[3] = {own[3 * i], own[3 * i + 1], own[3 * i + 2]}, ta[3], tb[3];
        int64_t r = rows[i];
        if (r < 0 || r >= n_other) {        // only possible with caller-supplied rows; reported as PCCM_E_RANGE
            atomicOr(bad, 1u);
            r = 0;
        }
        double b[3] = {other[3 * r], other[3 * r + 1], other[3 * r + 2]};
        to_scheme(scheme, a, ta);
        to_scheme(scheme, b, tb);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double diff = scale * (ta[c] - tb[c]);
            const double sq = diff * diff;
            if (what == 4) {
                out[(int64_t)c * n + i] = sq;
                key[c] = max_key(sq);
            } else {
                out[3 * i + c] = what == 0 ? ta[c] : what == 1 ? tb[c] : what == 2 ? diff : sq;
            }
        }
    }
    if (what != 4) return;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(key[c], off);
            key[c] = o > key[c] ? o : key[c];
        }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 3; ++c) s_max[w][c] = key[c];
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        unsigned long long m = s_max[0][c];
        for (int k = 1; k < 4; ++k) m = s_max[k][c] > m ? s_max[k][c] : m;
        if (m > __hip_atomic_load(&maxkeys[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&maxkeys[c], m);
    }
}

// ---- left-to-right fp64 sum of a non-negative column, without the dependent chain ----------------------
// s_i = fl(s_{i-1} + x_i), x_i >= 0.  While s stays inside one binade [2^e, 2^(e+1)] its unit in the last
// place u = 2^(e-52) is constant and s is a multiple of u, so  fl(s + x) = s + rn_u(x)  where rn_u rounds x
// to the nearest multiple of u -- unless x/u lies exactly half way (then the parity of s decides; such ties
// are ~2^-20 rare).  Hence for a run of elements that keeps s within the binade and contains no tie,
//     s_end = s + u * SUM_i rint(x_i / u)
// and that sum is an exact integer sum (< 2^53), free to be evaluated in any order -- IF the binade of s at the
// start of the run is known.  Round 4 (round 2-3: one workgroup per column walked it chunk by chunk, 0.47 ms per
// 0.8M-row direction on three of 256 CUs): the binade is GUESSED from an ordinary parallel sum and the guess is
// CHECKED by the only serial part that is left, a walk over ready-made totals:
//   k_colsum_approx  plain sum of every 8192-element chunk (any order: a guess needs no more),
//   k_colsum_units   one workgroup per chunk: wave w owns the 512 consecutive elements of sub-chunk w; the binade guess
//                    of a sub-chunk = ilogb(approximate sum of everything in front of it); its total in units of that
//                    binade, and whether every element rounded without a tie and below 2^53 units,
//   k_colsum_chain   one wave per column walks the chunks with the TRUE running sum t: a chunk (or, inside a chunk
//                    that is not accepted whole, a sub-chunk) is accepted iff its guess equals ilogb(t), nothing tied
//                    and t + total stays within the binade; a rejected sub-chunk (the ~log2(N) binade crossings, the
//                    start at t = 0, ties, non-finite values) is redone from memory in groups of 64 elements with
//                    the same test, and a rejected group is summed the plain way, one add after the other.
// The result is the chain's result bit for bit whatever the guesses were (tests: tests/test_gpu_color.py against
// np.add.reduce(axis=0) on random, tie-laden, wide-range and non-finite columns).
constexpr int kSumThreads = 1024, kSumPer = 8, kSumChunk = kSumThreads * kSumPer, kSumWaves = kSumThreads / 64, kSumSub = kSumChunk / kSumWaves;

struct Units {                 // the binade of the running sum
    double inv_u, u, room;     // 2^(52-e), 2^(e-52), (2^(e+1) - s) / u
    int e;
    bool usable;
};

__device__ __forceinline__ Units units_of(double s)
{
    Units q;
    q.usable = s >= 0x1p-900 && s < INFINITY;
    // (ilogb and ldexp by hand: s is a positive normal number here, and 2^(52-e), 2^(e-52) are normal for e in [-900, 1023])
    const long long bits = __double_as_longlong(s);
    q.e = q.usable ? (int)((bits >> 52) & 0x7ff) - 1023 : 0;
    q.inv_u = __longlong_as_double((long long)(52 - q.e + 1023) << 52);
    q.u = __longlong_as_double((long long)(q.e - 52 + 1023) << 52);
    q.room = 0x1p53 - s * q.inv_u;            // exact: s / u is an integer in [2^52, 2^53)
    return q;
}

// x / u rounded to nearest as an integer-valued double; false for a tie, x / u >= 2^53 or a NaN
__device__ __forceinline__ bool unit_round(double x, double inv_u, double &k)
{
    const double y = x * inv_u;               // exact scaling (an underflowing product is < 1/2 either way)
    k = rint(y);
    return y < 0x1p53 && fabs(y - k) != 0.5;  // y - k is exact
}

// lane `l`'s value of x, l wave-uniform: two v_readlane_b32 (a __shfl with a loop counter becomes ds_bpermute_b32 -- an LDS round
// trip per element, which a lone wave cannot hide: 15 k cycles per group of 64 instead of 1.5 k)
__device__ __forceinline__ double lane_value(double x, int l)
{
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (long long)(uint32_t)lo);
}

struct SumSub {                // what k_colsum_units leaves per sub-chunk (512 elements)
    double total;              // SUM rint(x / u) in units of binade e
    int e;                     // the guessed binade (kNoGuess: none)
    int ok;                    // every element rounded without a tie and below 2^53 units
};
struct SumChunk {              // ... and per chunk (8192 elements)
    double total;              // of all sixteen sub-chunks, when they share one binade
    int e;                     // that binade
    int whole;                 // the chunk may be accepted whole: one binade, no tie, no sub-chunk flagged
};
constexpr int kNoGuess = -100000;
// A sub-chunk is FLAGGED when the walk will probably have to redo it from its elements: no guess, a tie, or a binade crossing
// inside it by the approximate sums.  The chain kernel stages the flagged ones (and their chunks' records) in LDS before it
// walks, so that the serial part meets no memory round trip; whatever is rejected without having been flagged is fetched on demand.
constexpr int kStageSubs = 14;             // flagged sub-chunks staged per column (4 KB of LDS each)
constexpr int kStageChunks = 14;           // ... and chunks whose sixteen records are staged
constexpr int kSumMaxChunksLds = 1536;     // chunk records kept in LDS (12.6 M elements; beyond: read from memory as the walk goes)

__global__ __launch_bounds__(kSumThreads) void k_colsum_approx(const double *__restrict__ cols, int64_t n, int64_t nchunks,
                                                               double *__restrict__ approx, uint32_t *__restrict__ nflagged)
{
    __shared__ double s_p[kSumWaves];
    const double *__restrict__ col = cols + (int64_t)blockIdx.y * n;
    const int64_t base = (int64_t)blockIdx.x * kSumChunk;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (blockIdx.x == 0 && tid == 0) nflagged[blockIdx.y] = 0u;      // (the next kernel appends to the column's list)
    double p = 0.0;
#pragma unroll
    for (int j = 0; j < kSumPer; ++j) {
        const int64_t i = base + (int64_t)j * kSumThreads + tid;
        p += i < n ? col[i] : 0.0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off);
    if (lane == 0) s_p[w] = p;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int k = 0; k < kSumWaves; ++k) t += s_p[k];
        approx[(int64_t)blockIdx.y * nchunks + blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(kSumThreads) void k_colsum_units(const double *__restrict__ cols, int64_t n, int64_t nchunks,
                                                              const double *__restrict__ approx, SumSub *__restrict__ subs,
                                                              SumChunk *__restrict__ chunks, uint32_t *__restrict__ nflagged,
                                                              uint32_t *__restrict__ flagged, int flag_cap)
{
    __shared__ double s_red[kSumWaves], s_pre[kSumWaves + 1], s_tot[kSumWaves];
    __shared__ int s_e[kSumWaves], s_good[kSumWaves];
    const double *__restrict__ col = cols + (int64_t)blockIdx.y * n;
    const double *__restrict__ apx = approx + (int64_t)blockIdx.y * nchunks;
    const int64_t base = (int64_t)blockIdx.x * kSumChunk;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // the sub-chunk's elements (consecutive: the walk's order is the index order), issued first
    double x[kSumPer];
#pragma unroll
    for (int j = 0; j < kSumPer; ++j) {
        const int64_t i = base + (int64_t)w * kSumSub + j * 64 + lane;
        x[j] = i < n ? col[i] : 0.0;                  // + 0.0 leaves every partial sum as it is
    }
    // approximate sum of everything in front of this chunk ...
    double before = 0.0;
    for (int64_t k = tid; k < (int64_t)blockIdx.x; k += kSumThreads) before += apx[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
    // ... and of the sub-chunks in front of this one
    double mine = 0.0;
#pragma unroll
    for (int j = 0; j < kSumPer; ++j) mine += x[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0) {
        s_red[w] = before;
        s_pre[w + 1] = mine;
    }
    __syncthreads();
    double start = 0.0;
    for (int k = 0; k < kSumWaves; ++k) start += s_red[k];
    for (int k = 0; k < w; ++k) start += s_pre[k + 1];
    const Units q = units_of(start);                  // wave-uniform
    double part = 0.0;
    bool ok = q.usable;
#pragma unroll
    for (int j = 0; j < kSumPer; ++j) {
        double k;
        ok = unit_round(x[j], q.inv_u, k) && ok;
        part += k;                                    // integers: exact while the total is < 2^53, and a total
    }                                                 // beyond that never rounds back below it (monotone rounding)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    const int okw = __all(ok);
    if (lane == 0) {
        SumSub r;
        r.total = part;
        r.e = q.usable ? q.e : kNoGuess;
        r.ok = okw;
        subs[((int64_t)blockIdx.y * nchunks + blockIdx.x) * kSumWaves + w] = r;
        // flagged: the walk will probably want this sub-chunk's elements (no guess, a tie, or -- by the approximate sums -- the
        // running sum leaves the binade inside it); only sub-chunks that hold elements
        const double end = start + mine;
        const bool crossing = !(end < INFINITY) || !q.usable || ilogb(end) != q.e || part > q.room;
        const bool sus = (base + (int64_t)w * kSumSub < n) && (!okw || crossing);
        if (sus) {
            const uint32_t pos = atomicAdd(&nflagged[blockIdx.y], 1u);
            if (pos < (uint32_t)flag_cap) flagged[(int64_t)blockIdx.y * flag_cap + pos] = (uint32_t)(blockIdx.x * kSumWaves + w);
        }
        s_tot[w] = part;
        s_e[w] = r.e;
        s_good[w] = okw && !sus;
    }
    __syncthreads();
    if (tid == 0) {
        SumChunk c;
        c.total = 0.0;
        c.e = s_e[0];
        c.whole = 1;
        for (int k = 0; k < kSumWaves; ++k) {
            c.total += s_tot[k];                      // (integers: exact below 2^53; beyond, the walk's room test fails anyway)
            c.whole = c.whole && s_good[k] && s_e[k] == c.e && c.e != kNoGuess;
        }
        chunks[(int64_t)blockIdx.y * nchunks + blockIdx.x] = c;
    }
}

__global__ __launch_bounds__(kSumThreads) void k_colsum_chain(const double *__restrict__ cols, int64_t n, int64_t nchunks,
                                                              const SumSub *__restrict__ subs, const SumChunk *__restrict__ chunks,
                                                              const uint32_t *__restrict__ nflagged, const uint32_t *__restrict__ flagged,
                                                              int flag_cap, double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];           // (kStageSubs + 1) x 512 staged elements | kSumMaxChunksLds chunk records
    SumChunk *const s_chunk = reinterpret_cast<SumChunk *>(s_dyn + (kStageSubs + 1) * kSumSub);
    __shared__ SumSub s_sub[kStageChunks][kSumWaves];
    __shared__ uint32_t s_sid[kStageSubs], s_cid[kStageChunks];
    __shared__ int s_nsub, s_nchunk;
    const double *__restrict__ col = cols + (int64_t)blockIdx.x * n;
    const SumSub *__restrict__ sub = subs + (int64_t)blockIdx.x * nchunks * kSumWaves;
    const SumChunk *__restrict__ chk = chunks + (int64_t)blockIdx.x * nchunks;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // ---- everything the walk is likely to need, into LDS: all chunk records, the flagged sub-chunks' elements and their
    //      chunks' sub-chunk records (first come first served: the list is in no particular order)
    const int64_t nlds = nchunks < kSumMaxChunksLds ? nchunks : kSumMaxChunksLds;
    for (int64_t k = tid; k < nlds; k += kSumThreads) s_chunk[k] = chk[k];
    __shared__ uint32_t s_flag[64];
    const uint32_t have = nflagged[blockIdx.x];
    const int cnt = (int)(have < (uint32_t)flag_cap ? have : (uint32_t)flag_cap) < 64 ? (int)(have < (uint32_t)flag_cap ? have : (uint32_t)flag_cap) : 64;
    if (tid < cnt) s_flag[tid] = flagged[(int64_t)blockIdx.x * flag_cap + tid];      // (one coalesced load: a lane walking the list alone
    __syncthreads();                                                                   //  pays a memory round trip per entry)
    if (tid == 0) {
        int ns = 0, nc = 0;
        for (int k = 0; k < cnt; ++k) {
            const uint32_t id = s_flag[k];
            if (ns < kStageSubs) s_sid[ns++] = id;
            const uint32_t c = id / (uint32_t)kSumWaves;
            bool seen = false;
            for (int j = 0; j < nc; ++j) seen = seen || s_cid[j] == c;
            if (!seen && nc < kStageChunks) s_cid[nc++] = c;
        }
        s_nsub = ns;
        s_nchunk = nc;
    }
    __syncthreads();
    for (int k = w; k < s_nsub; k += kSumWaves) {
        const int64_t base = (int64_t)s_sid[k] * kSumSub;
#pragma unroll
        for (int g = 0; g < kSumSub / 64; ++g) {
            const int64_t i = base + g * 64 + lane;
            s_dyn[k * kSumSub + g * 64 + lane] = i < n ? col[i] : 0.0;
        }
    }
    for (int k = tid; k < s_nchunk * kSumWaves; k += kSumThreads) s_sub[k / kSumWaves][k % kSumWaves] = sub[(int64_t)s_cid[k / kSumWaves] * kSumWaves + (k % kSumWaves)];
    __syncthreads();
    if (w != 0) return;
    // ---- the walk (one wave) ---------------------------------------------------------------------------------------------
    double t = 0.0;
    Units q = units_of(t);                            // (ilogb / ldexp are costly: only redone after a plain-sum group)
    const int nsub = s_nsub, nchk = s_nchunk;
    for (int64_t c = 0; c < nchunks; ++c) {
        const SumChunk cc = c < nlds ? s_chunk[c] : chk[c];
        if (q.usable && cc.whole && cc.e == q.e && cc.total <= q.room) {
            t = t + cc.total * q.u;                   // exact: (t / u + total) * u with t / u + total <= 2^53
            q.room -= cc.total;
            if (q.room == 0.0) q = units_of(t);       // (the sum landed on 2^(e+1) exactly: the next binade begins here)
            continue;
        }
        // sub-chunk by sub-chunk: the chunk's sixteen records (lanes 0 .. 15), staged or fetched
        const unsigned long long hitc = __ballot(lane < nchk && s_cid[lane < kStageChunks ? lane : 0] == (uint32_t)c);   // (every lane looks at one entry)
        const int slot = hitc ? __ffsll((long long)hitc) - 1 : -1;
        const int l16 = lane < kSumWaves ? lane : 0;
        const SumSub cur = slot >= 0 ? s_sub[slot][l16] : sub[c * kSumWaves + l16];
        for (int ww = 0; ww < kSumWaves; ++ww) {
            const double tw = __shfl(cur.total, ww);
            const int ew = __shfl(cur.e, ww), okw = __shfl(cur.ok, ww);
            if (q.usable && okw && ew == q.e && tw <= q.room) {
                t = t + tw * q.u;
                q.room -= tw;
                if (q.room == 0.0) q = units_of(t);
                continue;
            }
            // a rejected sub-chunk in groups of 64 elements: from the staged copy, or from memory (all of it in flight at once)
            const int64_t base = c * kSumChunk + (int64_t)ww * kSumSub;
            if (base >= n) break;
            const unsigned long long hits = __ballot(lane < nsub && s_sid[lane < kStageSubs ? lane : 0] == (uint32_t)(c * kSumWaves + ww));
            int st = hits ? __ffsll((long long)hits) - 1 : -1;
            if (st < 0) {                             // not flagged (a guess just beside a binade border): fetched now, into the spare slot
                st = kStageSubs;
#pragma unroll
                for (int g = 0; g < kSumSub / 64; ++g) {
                    const int64_t i = base + g * 64 + lane;
                    s_dyn[st * kSumSub + g * 64 + lane] = i < n ? col[i] : 0.0;
                }
            }
            const double *xp = s_dyn + st * kSumSub;
#pragma unroll 1
            for (int g = 0; g < kSumSub / 64; ++g) {  // (rolled loops: this is cold code, and one wave pays every instruction-cache miss alone)
                const double xv = xp[g * 64 + lane];
                double k;
                const bool okg = unit_round(xv, q.inv_u, k) && q.usable;
                double tg = k;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) tg += __shfl_xor(tg, off);
                if (__all(okg) && tg <= q.room) {
                    t = t + tg * q.u;
                    q.room -= tg;                     // exact (integers below 2^53): still t's distance to 2^(e+1) in units
                    if (q.room == 0.0) q = units_of(t);
                } else {
#pragma unroll 1
                    for (int e = 0; e < 64; ++e) t = t + lane_value(xv, e);   // the reference's own order (+ 0.0 beyond the column's end)
                    q = units_of(t);
                }
            }
        }
    }
    if (lane == 0) out[blockIdx.x] = t;
}

// uchar colours (what PLY / PCD / PTS files hold) widened on the device: k / 255.0, the very division the host readers
// perform (IEEE, correctly rounded on both sides) -- 3 bytes per point over PCIe instead of 24
__global__ __launch_bounds__(256) void k_colors_from_u8(const unsigned char *__restrict__ src, int64_t n3, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n3) out[i] = (double)src[i] / 255.0;
}

int launch_colors_from_u8(pccm_ctx *ctx, const unsigned char *src, int64_t n3, double *out)
{
    ProfScope ps(ctx, PCCM_K_INGEST);
    hipLaunchKernelGGL(k_colors_from_u8, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, ctx->stream, src, n3, out);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

int launch_color_rows(pccm_ctx *ctx, const double *own, const double *other, const int32_t *rows, int64_t n,
                      int64_t n_other, int scheme, double scale, int what, double *out,
                      unsigned long long *maxkeys, unsigned int *bad)
{
    ProfScope ps(ctx, PCCM_K_POINT);
    hipLaunchKernelGGL(k_color_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, own, other, rows, n,
                       n_other, scheme, scale, what, out, maxkeys, bad);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

int launch_color_colsum(pccm_ctx *ctx, const double *cols, int64_t n, double *out3)
{
    ProfScope ps(ctx, PCCM_K_REDUCE);
    const int64_t nchunks = (n + kSumChunk - 1) / kSumChunk;
    constexpr int kFlagCap = 64;          // flagged sub-chunks listed per column (the chain stages the first kStageSubs of them)
    // scratch: [3][nchunks] approximate chunk sums | [3][nchunks] chunk records | [3][nchunks][16] sub-chunk records | counts + lists
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t o_chunk = up((size_t)3 * nchunks * sizeof(double)), o_sub = o_chunk + up((size_t)3 * nchunks * sizeof(SumChunk)),
                 o_cnt = o_sub + up((size_t)3 * nchunks * kSumWaves * sizeof(SumSub)), o_list = o_cnt + 256, total = o_list + up((size_t)3 * kFlagCap * 4);
    int rc = ensure(ctx, ctx->colsum_scratch, total);
    if (rc) return rc;
    char *base = (char *)ctx->colsum_scratch.p;
    double *approx = (double *)base;
    SumChunk *chunks = (SumChunk *)(base + o_chunk);
    SumSub *subs = (SumSub *)(base + o_sub);
    uint32_t *nflag = (uint32_t *)(base + o_cnt), *list = (uint32_t *)(base + o_list);
    const size_t dyn = (size_t)(kStageSubs + 1) * kSumSub * sizeof(double) + (size_t)kSumMaxChunksLds * sizeof(SumChunk);
    static_assert((size_t)(kStageSubs + 1) * kSumSub * sizeof(double) + kSumMaxChunksLds * sizeof(SumChunk) + kStageChunks * kSumWaves * sizeof(SumSub) + 256 <= 96 * 1024,
                  "the chain kernel's LDS stays below 96 KB");
    if (!ctx->colsum_configured) {
        // more than 64 KB of LDS per workgroup needs the opt-in -- a per-DEVICE attribute: kept with the context (which is bound to
        // one device and serialised by its mutex), not in a process-wide flag
        PCCM_HIP(hipFuncSetAttribute((const void *)k_colsum_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        ctx->colsum_configured = true;
    }
    dim3 grid((unsigned)nchunks, 3);
    hipLaunchKernelGGL(k_colsum_approx, grid, dim3(kSumThreads), 0, ctx->stream, cols, n, nchunks, approx, nflag);
    hipLaunchKernelGGL(k_colsum_units, grid, dim3(kSumThreads), 0, ctx->stream, cols, n, nchunks, (const double *)approx, subs, chunks, nflag, list, kFlagCap);
    hipLaunchKernelGGL(k_colsum_chain, dim3(3), dim3(kSumThreads), dyn, ctx->stream, cols, n, nchunks, (const SumSub *)subs, (const SumChunk *)chunks,
                       (const uint32_t *)nflag, (const uint32_t *)list, kFlagCap, out3);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

}  // namespace pccm
