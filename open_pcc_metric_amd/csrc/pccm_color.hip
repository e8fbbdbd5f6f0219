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

// k / 255.0 for every byte k, as the host's and the device's IEEE division give it (the compiler's constant folding is that division)
struct ByteLut {
    double v[256];
    constexpr ByteLut() : v() {
        for (int k = 0; k < 256; ++k) v[k] = (double)k / 255.0;
    }
};
__constant__ ByteLut c_byte_lut = ByteLut();

// what: 0 own rows in the scheme | 1 neighbour rows in the scheme | 2 scale * (own - other) | 3 its square
//       (all AoS [n][3] into `out`) | 4 squares as three columns out[c * n + i] + column maxima (bit keys)
// U8: both clouds' colours are bytes / 255 and come from the packed tables (Cloud::rgb8: 4 bytes per row, the gathered ones out
//     of the L2) through a 2 KB table of the 256 quotients in LDS -- the same doubles rgb64 holds
template <bool U8>
__global__ __launch_bounds__(256) void k_color_rows(const double *__restrict__ own, const double *__restrict__ other,
                                                    const uint32_t *__restrict__ own8, const uint32_t *__restrict__ other8,
                                                    const int32_t *__restrict__ rows, const float4 *__restrict__ recs, int64_t n, int64_t n_other,
                                                    int scheme, double scale, int what, double *__restrict__ out,
                                                    unsigned long long *__restrict__ maxkeys, unsigned int *__restrict__ bad)
{
    __shared__ unsigned long long s_max[4][3];
    __shared__ double s_lut[U8 ? 256 : 1];
    if (U8) {
        s_lut[threadIdx.x] = c_byte_lut.v[threadIdx.x];
        __syncthreads();
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long key[3] = {0ull, 0ull, 0ull};
    if (i < n) {
        double a[3], b[3], ta[3], tb[3];
        int64_t r = rows ? rows[i] : (int64_t)__float_as_int(recs[i].w);     // (the matched record carries the row)
        if (r < 0 || r >= n_other) {        // only possible with caller-supplied rows; reported as PCCM_E_RANGE
            atomicOr(bad, 1u);
            r = 0;
        }
        if (U8) {
            const uint32_t pa = own8[i], pb = other8[r];
            a[0] = s_lut[pa & 255u]; a[1] = s_lut[(pa >> 8) & 255u]; a[2] = s_lut[(pa >> 16) & 255u];
            b[0] = s_lut[pb & 255u]; b[1] = s_lut[(pb >> 8) & 255u]; b[2] = s_lut[(pb >> 16) & 255u];
        } else {
            a[0] = own[3 * i]; a[1] = own[3 * i + 1]; a[2] = own[3 * i + 2];
            b[0] = other[3 * r]; b[1] = other[3 * r + 1]; b[2] = other[3 * r + 2];
        }
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
    if (what != 4 || !maxkeys) return;      // (pccm_color_reduce takes the maxima from the column sums' first pass)
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
//                    binade, and whether every element rounded without a tie and below 2^53 units.  A sub-chunk the walk
//                    will probably reject (no guess, a tie, a crossing by the approximate sums) is listed, and gets the same
//                    record once more for each of its eight groups of 64 elements, every group with a guess of its own,
//   k_colsum_chain   one wave per column walks the chunks with the TRUE running sum t: a chunk (or, inside a chunk
//                    that is not accepted whole, a sub-chunk; inside a rejected sub-chunk, a group) is accepted iff its
//                    guess equals ilogb(t), nothing tied and t + total stays within the binade; a rejected group (the
//                    ~log2(N) binade crossings, the start at t = 0, ties, non-finite values) is summed the plain way, one
//                    add after the other.  Everything the walk touches has been staged in LDS by the other fifteen waves.
// Both directions of a pair go through the same three launches (ColsumJobs): the walk is one wave's latency, whatever runs beside it.
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
    int ok;                    // kRecOk: every element rounded without a tie and below 2^53 units; kRecZero: every element is +0
};
constexpr int kRecOk = 1, kRecZero = 2;    // (a run of zeros leaves ANY running sum as it is: acceptable whatever the binade)
struct SumChunk {              // ... and per chunk (8192 elements)
    double total;              // of all sixteen sub-chunks, when they share one binade
    int e;                     // that binade
    int whole;                 // kRecOk: the chunk may be accepted whole (one binade, no tie, no sub-chunk flagged); kRecZero: all zeros
};
constexpr int kNoGuess = -100000;
// A sub-chunk is FLAGGED when the walk will probably have to redo it from its elements: no guess, a tie, or a binade crossing
// inside it by the approximate sums.  The chain kernel stages the flagged ones (and their chunks' records) in LDS before it
// walks, so that the serial part meets no memory round trip; whatever is rejected without having been flagged is fetched on demand.
constexpr int kStageSubs = 14;             // flagged sub-chunks staged per column (4 KB of LDS each)
constexpr int kStageChunks = 14;           // ... and chunks whose sixteen records are staged
constexpr int kSumMaxChunksLds = 1536;     // chunk records kept in LDS (12.6 M elements; beyond: read from memory as the walk goes)

// Up to two column triples per launch (the two directions of a pair: different lengths, the same three kernels).
struct ColsumJob {
    const double *cols;        // [3][n]
    int64_t n, nchunks;
    double *approx;            // [3][nchunks]
    SumChunk *chunks;          // [3][nchunks]
    SumSub *subs;              // [3][nchunks][16]
    SumSub *groups;            // [3][flag_cap][8]: a flagged sub-chunk's groups of 64 elements, each with a guess of its own
    uint32_t *nflag, *list;    // [3], [3][flag_cap]
    double *out;               // [3]
    unsigned long long *chunkmax, *outmax;   // [3][nchunks] largest bit key (max_key) of every chunk; [3] of the column, or null
    unsigned long long *dbg;   // DIAG builds: [3][16] stamps and counts of the walk
};
struct ColsumJobs {
    ColsumJob j[2];
    int njobs, flag_cap;
};

__global__ __launch_bounds__(kSumThreads) void k_colsum_approx(ColsumJobs jobs)
{
    __shared__ double s_p[kSumWaves];
    const ColsumJob &J = jobs.j[blockIdx.y / 3];
    const int column = blockIdx.y % 3;
    if ((int64_t)blockIdx.x >= J.nchunks) return;
    const int64_t n = J.n;
    const double *__restrict__ col = J.cols + (int64_t)column * n;
    const int64_t base = (int64_t)blockIdx.x * kSumChunk;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    __shared__ unsigned long long s_k[kSumWaves];
    if (blockIdx.x == 0 && tid == 0) J.nflag[column] = 0u;      // (the next kernel appends to the column's list)
    double p = 0.0;
    unsigned long long key = 0ull;                              // the column's maximum rides along (np.max(axis=0): a NaN wins)
#pragma unroll
    for (int j = 0; j < kSumPer; ++j) {
        const int64_t i = base + (int64_t)j * kSumThreads + tid;
        const double x = i < n ? col[i] : 0.0;
        p += x;
        const unsigned long long kx = max_key(x);
        key = kx > key ? kx : key;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        p += __shfl_xor(p, off);
        const unsigned long long o = __shfl_xor(key, off);
        key = o > key ? o : key;
    }
    if (lane == 0) {
        s_p[w] = p;
        s_k[w] = key;
    }
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        unsigned long long m = 0ull;
        for (int k = 0; k < kSumWaves; ++k) {
            t += s_p[k];
            m = s_k[k] > m ? s_k[k] : m;
        }
        J.approx[(int64_t)column * J.nchunks + blockIdx.x] = t;
        J.chunkmax[(int64_t)column * J.nchunks + blockIdx.x] = m;
    }
}

__global__ __launch_bounds__(kSumThreads) void k_colsum_units(ColsumJobs jobs)
{
    __shared__ double s_red[kSumWaves], s_pre[kSumWaves + 1], s_tot[kSumWaves];
    __shared__ int s_e[kSumWaves], s_good[kSumWaves];
    const ColsumJob &J = jobs.j[blockIdx.y / 3];
    const int column = blockIdx.y % 3;
    if ((int64_t)blockIdx.x >= J.nchunks) return;
    const int64_t n = J.n, nchunks = J.nchunks;
    const double *__restrict__ col = J.cols + (int64_t)column * n;
    const double *__restrict__ apx = J.approx + (int64_t)column * nchunks;
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
    bool z = true;
#pragma unroll
    for (int j = 0; j < kSumPer; ++j) z = z && __double_as_longlong(x[j]) == 0ll;
    const int zero = __all(z);
    // flagged: the walk will probably want this sub-chunk's elements (no guess, a tie, or -- by the approximate sums -- the
    // running sum leaves the binade inside it); only sub-chunks that hold elements.  (wave-uniform: every lane holds the sums)
    const double end = start + mine;
    const bool crossing = !(end < INFINITY) || !q.usable || ilogb(end) != q.e || part > q.room;
    const bool sus = (base + (int64_t)w * kSumSub < n) && (!okw || crossing) && !zero;
    uint32_t pos = 0xffffffffu;
    if (sus) {
        if (lane == 0) pos = atomicAdd(&J.nflag[column], 1u);
        pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
        if (pos < (uint32_t)jobs.flag_cap) {
            // the same once more for each of its eight groups of 64 consecutive elements, every group with the binade the
            // approximate sum in front of IT suggests: the walk then redoes a rejected sub-chunk from eight ready-made totals
            // (and sums only the group a crossing falls into element by element)
            // (three passes, the eight groups side by side in each: a reduction per group one after the other made the
            // flagged waves -- and with them their workgroups, and the launch -- 5 us longer)
            double sg[kSumPer], tg[kSumPer];
#pragma unroll
            for (int j = 0; j < kSumPer; ++j) sg[j] = x[j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                for (int j = 0; j < kSumPer; ++j) sg[j] += __shfl_xor(sg[j], off);
            double front = start;
            int eg[kSumPer], fl[kSumPer];
#pragma unroll
            for (int j = 0; j < kSumPer; ++j) {
                const Units qg = units_of(front);
                double k;
                const bool okg = unit_round(x[j], qg.inv_u, k) && qg.usable;
                tg[j] = k;
                eg[j] = qg.usable ? qg.e : kNoGuess;
                fl[j] = (__all(okg) ? kRecOk : 0) | (__all(__double_as_longlong(x[j]) == 0ll) ? kRecZero : 0);
                front += sg[j];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
#pragma unroll
                for (int j = 0; j < kSumPer; ++j) tg[j] += __shfl_xor(tg[j], off);
            if (lane < kSumPer) {
                SumSub r;
                r.total = 0.0;
                r.e = kNoGuess;
                r.ok = 0;
#pragma unroll
                for (int j = 0; j < kSumPer; ++j)
                    if (lane == j) {
                        r.total = (fl[j] & kRecZero) ? 0.0 : tg[j];
                        r.e = eg[j];
                        r.ok = fl[j];
                    }
                J.groups[((int64_t)column * jobs.flag_cap + pos) * kSumPer + lane] = r;
            }
        }
    }
    if (lane == 0) {
        SumSub r;
        r.total = zero ? 0.0 : part;
        r.e = q.usable ? q.e : kNoGuess;
        r.ok = (okw ? kRecOk : 0) | (zero ? kRecZero : 0);
        J.subs[((int64_t)column * nchunks + blockIdx.x) * kSumWaves + w] = r;
        if (pos < (uint32_t)jobs.flag_cap) J.list[(int64_t)column * jobs.flag_cap + pos] = (uint32_t)(blockIdx.x * kSumWaves + w);
        s_tot[w] = r.total;
        s_e[w] = r.e;
        s_good[w] = zero ? 2 : (okw && !sus);         // (2: a sub-chunk of zeros fits every binade)
    }
    __syncthreads();
    if (tid == 0) {
        SumChunk c;
        c.total = 0.0;
        c.e = kNoGuess;
        bool whole = true, zeros = true;
        for (int k = 0; k < kSumWaves; ++k) {
            c.total += s_tot[k];                      // (integers: exact below 2^53; beyond, the walk's room test fails anyway)
            if (s_good[k] == 2) continue;             // zeros: any binade
            if (zeros) c.e = s_e[k];                  // the first sub-chunk that holds something decides the chunk's binade
            zeros = false;
            whole = whole && s_good[k] && s_e[k] == c.e && c.e != kNoGuess;
        }
        c.whole = zeros ? (kRecOk | kRecZero) : (whole ? kRecOk : 0);
        J.chunks[(int64_t)column * nchunks + blockIdx.x] = c;
    }
}

// inclusive prefix sums over the wave's 64 lanes: four DPP row shifts inside the rows of sixteen (a lane whose source lies outside
// its row keeps the 0 it was given), the three row totals through v_readlane.  (Non-negative integer-valued doubles here: exact
// below 2^53, monotone beyond.)
template <int CTRL>
__device__ __forceinline__ double dpp_row(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (long long)(uint32_t)lo);
}

__device__ __forceinline__ double wave_prefix(double v, int lane)
{
    v += dpp_row<0x111>(v);        // row_shr:1
    v += dpp_row<0x112>(v);        // row_shr:2
    v += dpp_row<0x114>(v);        // row_shr:4
    v += dpp_row<0x118>(v);        // row_shr:8
    const double r0 = lane_value(v, 15), r1 = lane_value(v, 31), r2 = lane_value(v, 47);
    const int row = lane >> 4;
    const double front = row == 0 ? 0.0 : row == 1 ? r0 : row == 2 ? r0 + r1 : (r0 + r1) + r2;
    return v + front;
}

// The walk's one step, at every level (chunks, sub-chunks, groups, elements): lanes [lo, hi) hold consecutive records -- `tot`
// the record's total in units of the running sum's binade, `good` that it may be taken so (guess = the sum's binade, nothing
// tied; or all zeros).  Takes the longest prefix of good records whose totals fit below the binade's end -- t + u * (sum of the
// totals) is then what adding them one after the other gives, exactly -- and returns the first lane it did not take (hi: all).
// `again`: the sum landed exactly on the binade's end, so the record that did not fit deserves a second look at this level.
__device__ __forceinline__ int accept_run(double tot, bool good, int lo, int hi, int lane, double &t, Units &q, bool &again)
{
    again = false;
    const bool in = lane >= lo && lane < hi;
    const double p = wave_prefix(in && good ? tot : 0.0, lane);
    const unsigned long long bad = __ballot(in && (!good || !(p <= q.room)));
    const int f = bad ? __ffsll((long long)bad) - 1 : hi;
    if (f > lo) {
        const double s = lane_value(p, f - 1);
        if (s > 0.0) {                                // (zeros only: nothing to add, and q may be unusable)
            t = t + s * q.u;                          // exact: (t / u + s) * u with t / u + s <= 2^53
            q.room -= s;
            if (q.room == 0.0) {                      // (the sum landed on 2^(e+1) exactly: the next binade begins here)
                q = units_of(t);
                again = true;
            }
        }
    }
    return f;
}

__device__ __forceinline__ bool record_good(int flags, int e, const Units &q)
{
    return (flags & kRecZero) || ((flags & kRecOk) && q.usable && e == q.e);
}

__global__ __launch_bounds__(kSumThreads) void k_colsum_chain(ColsumJobs jobs)
{
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];           // (kStageSubs + 1) x 512 staged elements | kSumMaxChunksLds chunk records
    SumChunk *const s_chunk = reinterpret_cast<SumChunk *>(s_dyn + (kStageSubs + 1) * kSumSub);
    __shared__ SumSub s_sub[kStageChunks][kSumWaves], s_grp[kStageSubs][kSumPer];
    __shared__ uint32_t s_sid[kStageSubs], s_cid[kStageChunks];
    __shared__ int s_nsub, s_nchunk;
    const ColsumJob &J = jobs.j[blockIdx.x / 3];
    const int column = blockIdx.x % 3, flag_cap = jobs.flag_cap;
    const int64_t n = J.n, nchunks = J.nchunks;
    const double *__restrict__ col = J.cols + (int64_t)column * n;
    const SumSub *__restrict__ sub = J.subs + (int64_t)column * nchunks * kSumWaves;
    const SumChunk *__restrict__ chk = J.chunks + (int64_t)column * nchunks;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#ifdef PCCM_DIAG
    unsigned long long dg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    dg[0] = __builtin_readcyclecounter();
#define DG(i, v) dg[i] += (v)
#define DGT() __builtin_readcyclecounter()
#else
#define DG(i, v)
#define DGT() 0ull
#endif
    // ---- everything the walk is likely to need, into LDS: all chunk records, the flagged sub-chunks' elements, their groups'
    //      records and their chunks' sub-chunk records (first come first served: the list is in no particular order)
    const int64_t nlds = nchunks < kSumMaxChunksLds ? nchunks : kSumMaxChunksLds;
    for (int64_t k = tid; k < nlds; k += kSumThreads) s_chunk[k] = chk[k];
    __shared__ uint32_t s_flag[64];
    const uint32_t have = J.nflag[column];
    const int capped = (int)(have < (uint32_t)flag_cap ? have : (uint32_t)flag_cap), cnt = capped < 64 ? capped : 64;
    if (tid < cnt) s_flag[tid] = J.list[(int64_t)column * flag_cap + tid];    // (one coalesced load: a lane walking the list alone
    __syncthreads();                                                           //  pays a memory round trip per entry)
    if (tid == 0) {
        int ns = 0, nc = 0;
        for (int k = 0; k < cnt; ++k) {
            const uint32_t id = s_flag[k];
            if (ns < kStageSubs) s_sid[ns++] = id;                            // (staged slot k = list position k)
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
    for (int k = tid; k < s_nsub * kSumPer; k += kSumThreads) s_grp[k / kSumPer][k % kSumPer] = J.groups[((int64_t)column * flag_cap + k / kSumPer) * kSumPer + (k % kSumPer)];
    for (int k = tid; k < s_nchunk * kSumWaves; k += kSumThreads) s_sub[k / kSumWaves][k % kSumWaves] = sub[(int64_t)s_cid[k / kSumWaves] * kSumWaves + (k % kSumWaves)];
    __syncthreads();
    if (w == kSumWaves - 1 && J.outmax) {            // the column's maximum: the last wave, beside the walk
        unsigned long long m = 0ull;
        for (int64_t k = lane; k < nchunks; k += 64) {
            const unsigned long long v = J.chunkmax[(int64_t)column * nchunks + k];
            m = v > m ? v : m;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(m, off);
            m = o > m ? o : m;
        }
        if (lane == 0) J.outmax[column] = m;
    }
    if (w != 0) return;
    DG(1, DGT());
    // ---- the walk (one wave).  Every step takes a RUN of records at once (accept_run): as many chunks as fit below the end of
    //      the binade, then -- inside the chunk that did not fit -- sub-chunks, groups, elements; the element the running sum crosses
    //      into the next binade with is added the plain way, and the walk goes on one level up as soon as a run has been taken.
    double t = 0.0;
    Units q = units_of(t);
    const int nsub = s_nsub, nchk = s_nchunk;
    int64_t c = 0;
    while (c < nchunks) {
        {
            const int64_t ci = c + lane;
            SumChunk cc;
            cc.total = 0.0;
            cc.e = kNoGuess;
            cc.whole = 0;
            if (ci < nchunks) cc = ci < nlds ? s_chunk[ci] : chk[ci];
            const int hi = nchunks - c < 64 ? (int)(nchunks - c) : 64;
            bool again;
            const int f = accept_run(cc.total, record_good(cc.whole, cc.e, q), 0, hi, lane, t, q, again);
            DG(3, f);
            c += f;
            if (f == hi || again) continue;           // (all of them, or the sum is in another binade now: look again)
        }
        [[maybe_unused]] const unsigned long long tc0 = DGT();
        // chunk c sub-chunk by sub-chunk: its sixteen records (lanes 0 .. 15), staged or fetched
        const unsigned long long hitc = __ballot(lane < nchk && s_cid[lane < kStageChunks ? lane : 0] == (uint32_t)c);   // (every lane looks at one entry)
        const int slot = hitc ? __ffsll((long long)hitc) - 1 : -1;
        const int l16 = lane < kSumWaves ? lane : 0;
        const SumSub cur = slot >= 0 ? s_sub[slot][l16] : sub[c * kSumWaves + l16];
        int w0 = 0;
        while (w0 < kSumWaves) {
            if (c * kSumChunk + (int64_t)w0 * kSumSub >= n) break;
            {
                bool again;
                const int f = accept_run(cur.total, record_good(cur.ok, cur.e, q), w0, kSumWaves, lane, t, q, again);
                DG(4, f - w0);
                w0 = f;
                if (f == kSumWaves || again) continue;
                if (c * kSumChunk + (int64_t)w0 * kSumSub >= n) break;
            }
            [[maybe_unused]] const unsigned long long ts0 = DGT();
            // sub-chunk w0 in groups of 64 elements: from the staged copy, or from memory (all of it in flight at once)
            const int64_t base = c * kSumChunk + (int64_t)w0 * kSumSub;
            const unsigned long long hits = __ballot(lane < nsub && s_sid[lane < kStageSubs ? lane : 0] == (uint32_t)(c * kSumWaves + w0));
            int st = hits ? __ffsll((long long)hits) - 1 : -1;
            SumSub mine;                              // lanes 0 .. 7: the groups' ready-made records (none for a sub-chunk fetched now)
            mine.total = 0.0;
            mine.e = kNoGuess;
            mine.ok = 0;
            if (st < 0) {                             // not flagged (a guess just beside a binade border): fetched now, into the spare slot
                st = kStageSubs;
#pragma unroll
                for (int g = 0; g < kSumSub / 64; ++g) {
                    const int64_t i = base + g * 64 + lane;
                    s_dyn[st * kSumSub + g * 64 + lane] = i < n ? col[i] : 0.0;
                }
            } else {
                mine = s_grp[st][lane & (kSumPer - 1)];
            }
            const double *xp = s_dyn + st * kSumSub;
            int g0 = 0;
            while (g0 < kSumPer) {
                {
                    bool again;
                    const int f = accept_run(mine.total, record_good(mine.ok, mine.e, q), g0, kSumPer, lane, t, q, again);
                    DG(5, f - g0);
                    g0 = f;
                    if (f == kSumPer || again) continue;
                }
                // group g0 element by element: runs again, broken by the elements that tie or carry the sum into the next binade
                [[maybe_unused]] const unsigned long long tp0 = DGT();
                const double xv = xp[g0 * 64 + lane];
                int lo = 0, dry = 0;
                while (lo < 64) {
                    if (dry >= 2) {                   // nothing fits (non-finite values, ties in a row): the reference's own order
#pragma unroll 1
                        for (int e = lo; e < 64; ++e) t = t + lane_value(xv, e);
                        q = units_of(t);
                        break;
                    }
                    double k;
                    const bool rounds = unit_round(xv, q.inv_u, k) && q.usable, zero = __double_as_longlong(xv) == 0ll;
                    bool again;
                    const int f = accept_run(zero ? 0.0 : k, zero || rounds, lo, 64, lane, t, q, again);
                    dry = f > lo ? 0 : dry + 1;
                    if (f < 64) {
                        t = t + lane_value(xv, f);
                        q = units_of(t);
                        DG(7, 1);
                    }
                    lo = f + 1;
                }
                DG(6, 1);
                DG(8, DGT() - tp0);
                ++g0;
            }
            DG(9, DGT() - ts0);
            ++w0;
        }
        DG(10, DGT() - tc0);
        ++c;
    }
    if (lane == 0) J.out[column] = t;
#ifdef PCCM_DIAG
    dg[2] = __builtin_readcyclecounter();
    if (lane == 0 && J.dbg)
        for (int k = 0; k < 12; ++k) J.dbg[column * 16 + k] = dg[k];
#endif
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
                      unsigned long long *maxkeys, unsigned int *bad, const uint32_t *own8, const uint32_t *other8, const float4 *recs)
{
    ProfScope ps(ctx, PCCM_K_POINT);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (own8 && other8)
        hipLaunchKernelGGL(k_color_rows<true>, grid, dim3(256), 0, ctx->stream, own, other, own8, other8, rows, recs, n, n_other, scheme, scale, what, out,
                           maxkeys, bad);
    else
        hipLaunchKernelGGL(k_color_rows<false>, grid, dim3(256), 0, ctx->stream, own, other, own8, other8, rows, recs, n, n_other, scheme, scale, what, out,
                           maxkeys, bad);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

// Cloud::rgb8 from the uploaded bytes, or from rgb64 when every value is a byte's quotient (what a reader's k / 255.0 leaves)
__global__ __launch_bounds__(256) void k_rgb8_pack(const unsigned char *__restrict__ src, const double *__restrict__ rgb64, int64_t n,
                                                   uint32_t *__restrict__ out, unsigned int *__restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t p = 0u;
    if (src) {
        p = (uint32_t)src[3 * i] | ((uint32_t)src[3 * i + 1] << 8) | ((uint32_t)src[3 * i + 2] << 16);
    } else {
        bool ok = true;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double v = rgb64[3 * i + c];
            const double k = rint(v * 255.0);
            const int b = (k >= 0.0 && k <= 255.0) ? (int)k : 0;              // (NaN fails both tests)
            ok = ok && __double_as_longlong(c_byte_lut.v[b]) == __double_as_longlong(v);
            p |= (uint32_t)b << (8 * c);
        }
        if (!ok) atomicOr(flag, 1u);
    }
    out[i] = p;
}

int launch_rgb8(pccm_ctx *ctx, Cloud &c, const unsigned char *bytes, unsigned int *flag)
{
    ProfScope ps(ctx, PCCM_K_INGEST);
    hipLaunchKernelGGL(k_rgb8_pack, dim3((unsigned)((c.n + 255) / 256)), dim3(256), 0, ctx->stream, bytes, (const double *)c.rgb64, c.n, c.rgb8, flag);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

int launch_color_colsums(pccm_ctx *ctx, int njobs, const double *const cols[2], const int64_t n[2], double *const out3[2],
                         unsigned long long *const outmax[2])
{
    ProfScope ps(ctx, PCCM_K_REDUCE);
    constexpr int kFlagCap = 64;          // flagged sub-chunks listed per column (the chain stages the first kStageSubs of them)
    // scratch per job: [3][nchunks] approximate chunk sums | [3][nchunks] chunk records | [3][nchunks][16] sub-chunk records |
    //                  [3][kFlagCap][8] group records | counts | lists
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    size_t off[2][7], total = 0;
    int64_t nchunks[2] = {0, 0}, most = 0;
    for (int k = 0; k < njobs; ++k) {
        nchunks[k] = (n[k] + kSumChunk - 1) / kSumChunk;
        most = nchunks[k] > most ? nchunks[k] : most;
        off[k][0] = total;
        off[k][1] = off[k][0] + up((size_t)3 * nchunks[k] * sizeof(double));
        off[k][2] = off[k][1] + up((size_t)3 * nchunks[k] * sizeof(SumChunk));
        off[k][3] = off[k][2] + up((size_t)3 * nchunks[k] * kSumWaves * sizeof(SumSub));
        off[k][4] = off[k][3] + up((size_t)3 * kFlagCap * kSumPer * sizeof(SumSub));
        off[k][5] = off[k][4] + 256;
        off[k][6] = off[k][5] + up((size_t)3 * kFlagCap * 4);
        total = off[k][6] + up((size_t)3 * nchunks[k] * sizeof(unsigned long long));
    }
    const size_t o_dbg = total;
    total += 2 * 3 * 16 * sizeof(unsigned long long);
    int rc = ensure(ctx, ctx->colsum_scratch, total);
    if (rc) return rc;
    char *base = (char *)ctx->colsum_scratch.p;
    ColsumJobs jobs;
    jobs.njobs = njobs;
    jobs.flag_cap = kFlagCap;
    for (int k = 0; k < 2; ++k) {
        ColsumJob &J = jobs.j[k];
        const int s = k < njobs ? k : 0;
        J.cols = cols[s];
        J.n = n[s];
        J.nchunks = nchunks[s];
        J.approx = (double *)(base + off[s][0]);
        J.chunks = (SumChunk *)(base + off[s][1]);
        J.subs = (SumSub *)(base + off[s][2]);
        J.groups = (SumSub *)(base + off[s][3]);
        J.nflag = (uint32_t *)(base + off[s][4]);
        J.list = (uint32_t *)(base + off[s][5]);
        J.out = out3[s];
        J.chunkmax = (unsigned long long *)(base + off[s][6]);
        J.outmax = outmax ? outmax[s] : nullptr;
        J.dbg = (unsigned long long *)(base + o_dbg) + s * 48;
    }
    const size_t dyn = (size_t)(kStageSubs + 1) * kSumSub * sizeof(double) + (size_t)kSumMaxChunksLds * sizeof(SumChunk);
    static_assert((size_t)(kStageSubs + 1) * kSumSub * sizeof(double) + kSumMaxChunksLds * sizeof(SumChunk) + kStageChunks * kSumWaves * sizeof(SumSub) +
                          kStageSubs * kSumPer * sizeof(SumSub) + 512 <= 96 * 1024,
                  "the chain kernel's LDS stays below 96 KB");
    if (!ctx->colsum_configured) {
        // more than 64 KB of LDS per workgroup needs the opt-in -- a per-DEVICE attribute: kept with the context (which is bound to
        // one device and serialised by its mutex), not in a process-wide flag
        PCCM_HIP(hipFuncSetAttribute((const void *)k_colsum_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        ctx->colsum_configured = true;
    }
    dim3 grid((unsigned)most, 3 * njobs);
    hipLaunchKernelGGL(k_colsum_approx, grid, dim3(kSumThreads), 0, ctx->stream, jobs);
    hipLaunchKernelGGL(k_colsum_units, grid, dim3(kSumThreads), 0, ctx->stream, jobs);
    hipLaunchKernelGGL(k_colsum_chain, dim3(3 * njobs), dim3(kSumThreads), dyn, ctx->stream, jobs);
    PCCM_HIP(hipGetLastError());
#ifdef PCCM_DIAG
    if (getenv("PCCM_COLSUM_STAMP")) {
        unsigned long long h[96];
        PCCM_HIP(hipStreamSynchronize(ctx->stream));
        PCCM_HIP(hipMemcpy(h, base + o_dbg, sizeof(h), hipMemcpyDeviceToHost));
        for (int k = 0; k < 3 * njobs; ++k) {
            const unsigned long long *d = h + (k / 3) * 48 + (k % 3) * 16;
            fprintf(stderr, "colsum job %d col %d: stage %llu walk %llu clocks | chunks taken %llu subs %llu groups %llu | groups by element %llu plain adds %llu | element clk %llu sub-loop clk %llu chunk-loop clk %llu\n",
                    k / 3, k % 3, d[1] - d[0], d[2] - d[1], d[3], d[4], d[5], d[6], d[7], d[8], d[9], d[10]);
        }
    }
#endif
    return PCCM_OK;
}

int launch_color_colsum(pccm_ctx *ctx, const double *cols, int64_t n, double *out3)
{
    const double *c[2] = {cols, cols};
    const int64_t nn[2] = {n, n};
    double *o[2] = {out3, out3};
    return launch_color_colsums(ctx, 1, c, nn, o, nullptr);
}

}  // namespace pccm
