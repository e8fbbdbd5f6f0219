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
// and that sum is an exact integer sum (< 2^53), free to be evaluated in any order.  A workgroup takes the
// column in chunks of 8192 elements: every thread scales and rounds its 8 elements, the chunk total is
// reduced, and the chunk is accepted iff no element tied or reached 2^53 units and s + total stays <= 2^(e+1).
// A rejected chunk (binade crossing: ~log2(N) of them; ties; s = 0, tiny or non-finite) is redone by wave 0
// in groups of 64 elements with the same test, and a rejected group is summed the plain way, one add after
// the other.  The result is the chain's result bit for bit (tests: tests/test_gpu_color.py against
// np.add.reduce(axis=0) on random, tie-laden, wide-range and non-finite columns).
constexpr int kSumThreads = 1024, kSumPer = 8, kSumChunk = kSumThreads * kSumPer;

struct Units {                 // the binade of the running sum
    double inv_u, u, room;     // 2^(52-e), 2^(e-52), (2^(e+1) - s) / u
    bool usable;
};

__device__ __forceinline__ Units units_of(double s)
{
    Units q;
    q.usable = s >= 0x1p-900 && s < INFINITY;
    const int e = q.usable ? ilogb(s) : 0;
    q.inv_u = ldexp(1.0, 52 - e);
    q.u = ldexp(1.0, e - 52);
    q.room = 0x1p53 - s * q.inv_u;            // exact: s / u is an integer in [2^52, 2^53)
    return q;
}

// x / u rounded to nearest as an integer-valued double; false for a tie, x / u >= 2^53 or a NaN
__device__ __forceinline__ bool unit_round(double x, const Units &q, double &k)
{
    const double y = x * q.inv_u;             // exact scaling (an underflowing product is < 1/2 either way)
    k = rint(y);
    return y < 0x1p53 && fabs(y - k) != 0.5;  // y - k is exact
}

__global__ __launch_bounds__(kSumThreads) void k_color_colsum(const double *__restrict__ cols, int64_t n,
                                                              double *__restrict__ out)
{
    extern __shared__ double s_x[];                   // kSumChunk doubles (64 KB): the chunk, staged only when rejected
    __shared__ double s_part[2][kSumThreads / 64];
    __shared__ int s_ok[2][kSumThreads / 64];
    __shared__ double s_sum;
    const double *__restrict__ col = cols + (int64_t)blockIdx.x * n;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double s = 0.0;
    double x[kSumPer], nx[kSumPer];
#pragma unroll
    for (int j = 0; j < kSumPer; ++j) {
        const int64_t i = (int64_t)j * kSumThreads + tid;
        nx[j] = i < n ? col[i] : 0.0;                 // + 0.0 leaves every partial sum as it is
    }
    int it = 0;
    for (int64_t base = 0; base < n; base += kSumChunk, ++it) {
#pragma unroll
        for (int j = 0; j < kSumPer; ++j) x[j] = nx[j];
        if (base + kSumChunk < n) {
#pragma unroll
            for (int j = 0; j < kSumPer; ++j) {
                const int64_t i = base + kSumChunk + (int64_t)j * kSumThreads + tid;
                nx[j] = i < n ? col[i] : 0.0;
            }
        }
        const Units q = units_of(s);                  // workgroup-uniform
        double part = 0.0;
        bool ok = q.usable;
#pragma unroll
        for (int j = 0; j < kSumPer; ++j) {
            double k;
            ok = unit_round(x[j], q, k) && ok;
            part += k;                                // integers: exact while the total is < 2^53, and a total
        }                                             // beyond that never rounds back below it (monotone rounding)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        const int okw = __all(ok);
        const int buf = it & 1;
        if (lane == 0) {
            s_part[buf][w] = part;
            s_ok[buf][w] = okw;
        }
        __syncthreads();
        double total = 0.0;
        int okb = 1;
#pragma unroll
        for (int k = 0; k < kSumThreads / 64; ++k) {
            total += s_part[buf][k];
            okb &= s_ok[buf][k];
        }
        if (okb && total <= q.room) {                 // workgroup-uniform
            s = s + total * q.u;                      // exact: (s / u + total) * u with s / u + total <= 2^53
            continue;
        }
        // rejected chunk: wave 0 walks it in groups of 64
#pragma unroll
        for (int j = 0; j < kSumPer; ++j) s_x[j * kSumThreads + tid] = x[j];
        __syncthreads();
        if (w == 0) {
            double t = s;
            const int64_t left = n - base;
            const int groups = (int)((left < kSumChunk ? left : kSumChunk) + 63) / 64;
            Units qg = units_of(t);                     // ilogb / ldexp are costly: only redone after a plain-sum group
            for (int g = 0; g < groups; ++g) {
                double k;
                const bool okg = unit_round(s_x[g * 64 + lane], qg, k) && qg.usable;
                double tot = k;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
                if (__all(okg) && tot <= qg.room) {
                    t = t + tot * qg.u;
                    qg.room -= tot;                     // exact (integers below 2^53): still t's distance to 2^(e+1) in units
                } else {
                    for (int e = 0; e < 64; ++e) t = t + s_x[g * 64 + e];     // the reference's own order
                    qg = units_of(t);
                }
            }
            if (lane == 0) s_sum = t;
        }
        __syncthreads();
        s = s_sum;
    }
    if (tid == 0) out[blockIdx.x] = s;
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
    // 64 KB of dynamic LDS needs the opt-in -- a per-DEVICE attribute: kept with the context (which is bound to one
    // device and serialised by its mutex), not in a process-wide flag
    if (!ctx->colsum_configured) {
        PCCM_HIP(hipFuncSetAttribute((const void *)k_color_colsum, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     kSumChunk * (int)sizeof(double)));
        ctx->colsum_configured = true;
    }
    hipLaunchKernelGGL(k_color_colsum, dim3(3), dim3(kSumThreads), kSumChunk * sizeof(double), ctx->stream, cols, n, out3);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

}  // namespace pccm
