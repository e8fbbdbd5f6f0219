// Ingest, fused per-point error/projection kernel (K3) and the NumPy-ordered leaf reduction (K5).
#include "pccm_internal.h"

namespace pccm {

// ------------------------------------------------------------------------------------------
// Ingest: packed [n][3] f32/f64 rows -> float4 scan copy (padded) + fp64 copy, and three
// statistics: [0] max |coordinate| (as fp64 bits; non-negative doubles order like uint64), [1] number
// of coordinates that do not survive fp64 -> fp32 -> fp64, [2] number of non-finite coordinates,
// [3..5] / [6..8] order keys of the bounding box minimum / maximum per axis (for the grid engine),
// [9] number of coordinates that are not integers (voxelised content has none).
// ------------------------------------------------------------------------------------------
// monotonic map double -> uint64 (so that atomicMin/atomicMax order like the doubles do)
__device__ __forceinline__ unsigned long long order_key(double v)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

template <typename T>
__global__ __launch_bounds__(256) void k_ingest_points(const T *__restrict__ src, int64_t n, int64_t n_pad,
                                                       float *__restrict__ x32, double *__restrict__ x64, float4 *__restrict__ x32r,
                                                       unsigned long long *__restrict__ stats)
{
    __shared__ unsigned long long s_mx[4], s_lo[4][3], s_hi[4][3];
    __shared__ int s_cnt[4][3];
    unsigned long long mx = 0;
    unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0ull, 0ull, 0ull};   // bounding box keys
    int inexact = 0, bad = 0, frac = 0;
    // grid-stride: one workgroup per CU, so that the ten statistics cost a few hundred atomics on the same ten
    // addresses (same-address atomics and the loads that peek at them serialise: 2048 workgroups took 111 us for a
    // million points, 256 take 38) instead of one set per wave
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * 256) {
        if (i < n) {
            double v[3];
            float f[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                v[a] = (double)src[3 * i + a];
                f[a] = (float)v[a];
                inexact += ((double)f[a] != v[a]) ? 1 : 0;
                frac += (floor(v[a]) != v[a]) ? 1 : 0;
                bad += isfinite(v[a]) ? 0 : 1;
                unsigned long long b = (unsigned long long)__double_as_longlong(fabs(v[a]));
                mx = b > mx ? b : mx;
                x64[3 * i + a] = v[a];
                const unsigned long long key = order_key(v[a]);
                lo[a] = key < lo[a] ? key : lo[a];
                hi[a] = key > hi[a] ? key : hi[a];
            }
            float *qd = x32 + (i >> 2) * 12 + (i & 3);
            qd[0] = f[0];
            qd[4] = f[1];
            qd[8] = f[2];
            x32r[i] = make_float4(f[0], f[1], f[2], 0.f);
        } else {
            float *qd = x32 + (i >> 2) * 12 + (i & 3);
            qd[0] = qd[4] = qd[8] = kPadCoord;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
        inexact += __shfl_xor(inexact, off);
        bad += __shfl_xor(bad, off);
        frac += __shfl_xor(frac, off);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            unsigned long long l = __shfl_xor(lo[a], off), h = __shfl_xor(hi[a], off);
            lo[a] = l < lo[a] ? l : lo[a];
            hi[a] = h > hi[a] ? h : hi[a];
        }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_mx[w] = mx;
        s_cnt[w][0] = inexact; s_cnt[w][1] = bad; s_cnt[w][2] = frac;
        for (int a = 0; a < 3; ++a) { s_lo[w][a] = lo[a]; s_hi[w][a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            mx = s_mx[k] > mx ? s_mx[k] : mx;
            inexact += s_cnt[k][0]; bad += s_cnt[k][1]; frac += s_cnt[k][2];
            for (int a = 0; a < 3; ++a) {
                lo[a] = s_lo[k][a] < lo[a] ? s_lo[k][a] : lo[a];
                hi[a] = s_hi[k][a] > hi[a] ? s_hi[k][a] : hi[a];
            }
        }
        // the host reads stats[1], [2] and [9] as flags and the rest as extrema, so an atomic is only issued when it
        // would change the word: after the first few workgroups almost none are (same-address atomics serialise)
        auto peek = [&](int k) { return __hip_atomic_load(&stats[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        if (mx > peek(0)) atomicMax(&stats[0], mx);
        if (inexact && !peek(1)) atomicAdd(&stats[1], (unsigned long long)inexact);
        if (bad && !peek(2)) atomicAdd(&stats[2], (unsigned long long)bad);
        if (frac && !peek(9)) atomicAdd(&stats[9], (unsigned long long)frac);
        for (int a = 0; a < 3; ++a) {
            if (lo[a] < peek(3 + a)) atomicMin(&stats[3 + a], lo[a]);
            if (hi[a] > peek(6 + a)) atomicMax(&stats[6 + a], hi[a]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_ingest_normals(const T *__restrict__ src, int64_t n3,
                                                        double *__restrict__ out, float *__restrict__ out32,
                                                        unsigned long long *__restrict__ stats)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int bad = 0, inexact = 0;
    if (i < n3) {
        double v = (double)src[i];
        out[i] = v;
        const float f = (float)v;
        if (out32) out32[(i / 3) * 4 + (i % 3)] = f;      // stats[1] says whether this copy may stand for `out`
        bad = isfinite(v) ? 0 : 1;
        inexact = ((double)f == v) ? 0 : 1;
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicAdd(&stats[2], 1ull);
    if (__ballot(inexact) && (threadIdx.x & 63) == 0) atomicAdd(&stats[1], 1ull);
}

int launch_ingest_points(pccm_ctx *ctx, const void *src, int dtype, int64_t n, int64_t n_pad, float4 *x32,
                         double *x64, float4 *x32r, unsigned long long *stats)
{
    ProfScope ps(ctx, PCCM_K_INGEST);
    const int64_t blocks = (n_pad + 255) / 256;
    dim3 grid((unsigned)(blocks < 256 ? blocks : 256));
    if (dtype == PCCM_F32)
        hipLaunchKernelGGL((k_ingest_points<float>), grid, dim3(256), 0, ctx->stream, (const float *)src, n, n_pad, (float *)x32, x64, x32r, stats);
    else
        hipLaunchKernelGGL((k_ingest_points<double>), grid, dim3(256), 0, ctx->stream, (const double *)src, n, n_pad, (float *)x32, x64, x32r, stats);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

int launch_ingest_normals(pccm_ctx *ctx, const void *src, int dtype, int64_t n, double *out, float *out32, unsigned long long *stats)
{
    ProfScope ps(ctx, PCCM_K_INGEST);
    const int64_t n3 = 3 * n;
    dim3 grid((unsigned)((n3 + 255) / 256));
    if (dtype == PCCM_F32)
        hipLaunchKernelGGL((k_ingest_normals<float>), grid, dim3(256), 0, ctx->stream, (const float *)src, n3, out, out32, stats);
    else
        hipLaunchKernelGGL((k_ingest_normals<double>), grid, dim3(256), 0, ctx->stream, (const double *)src, n3, out, out32, stats);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

// ------------------------------------------------------------------------------------------
// K3: gather the matched point, error vector e = iter[i] - search[nn(i)] (cloud_pair.py:90-100),
// projection on the other cloud's normal (metric.py:146-153) and its square (metric.py:179).
// The dot product is the FMA chain fma(e2,n2, fma(e1,n1, e0*n0)) that np.dot (OpenBLAS ddot)
// evaluates on FMA-capable hosts; see oracle/pccm_oracle.c for how that was pinned.
// HBM/gather bound: 24 (q) + 4 (idx) + 24 (r, gathered) + 24 (normal) + 8 (out) bytes per row.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_point_metric(const double *__restrict__ q64, int64_t q_begin, int64_t ns,
                                                      const double *__restrict__ r64,
                                                      const int32_t *__restrict__ idx,
                                                      const double *__restrict__ nrm, int metric, int normal_mode,
                                                      double *__restrict__ val, double *__restrict__ err)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ns) return;
    const int64_t gi = q_begin + i;
    const int64_t j = idx[i];
    const double ex = __dsub_rn(q64[3 * gi], r64[3 * j]);
    const double ey = __dsub_rn(q64[3 * gi + 1], r64[3 * j + 1]);
    const double ez = __dsub_rn(q64[3 * gi + 2], r64[3 * j + 2]);
    if (err) {
        err[3 * i] = ex;
        err[3 * i + 1] = ey;
        err[3 * i + 2] = ez;
    }
    if (val) {
        const int64_t k = (normal_mode == PCCM_NORMAL_ROW) ? gi : j;
        double p = __dmul_rn(ex, nrm[3 * k]);
        p = __fma_rn(ey, nrm[3 * k + 1], p);
        p = __fma_rn(ez, nrm[3 * k + 2], p);
        val[i] = (metric == PCCM_METRIC_PROJ) ? p : __dmul_rn(p, p);
    }
}

int launch_point_metric(pccm_ctx *ctx, const Cloud &it, const Cloud &se, const NNResult &res, int metric,
                        int normal_mode, double *out_val, double *out_err)
{
    const int64_t ns = res.end - res.begin;
    if (ns <= 0) return PCCM_OK;
    ProfScope ps(ctx, PCCM_K_POINT);
    dim3 grid((unsigned)((ns + 255) / 256));
    hipLaunchKernelGGL(k_point_metric, grid, dim3(256), 0, ctx->stream, it.xyz64, res.begin, ns, se.xyz64, res.idx,
                       se.nrm64, metric, normal_mode, out_val, out_err);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

// ------------------------------------------------------------------------------------------
// K5 (k_unit_jobs below): per 128-row leaf, eight lanes accumulate rows k, k+8, k+16, ... in order and
// the eight accumulators are combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) -- exactly NumPy's
// pairwise-sum leaf, so that np.sum's tree can be finished bit for bit (pccm_finish_sum / pccm_reduce_total).
// ------------------------------------------------------------------------------------------
// ---- batched forms: several columns per launch, results written straight into pinned host memory ------
// A report needs up to four columns (D1/D2 x left/right).  One k_point_jobs launch evaluates all D2
// columns, one k_unit_jobs launch reduces all columns and stores the per-unit sums/min/max and the raw
// tail values directly into the slots' pinned host buffers (device-visible), so there is no copy node
// and no extra launch per column.
__global__ __launch_bounds__(256) void k_point_jobs(PointJobs jobs)
{
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i0 >= jobs.off[jobs.njobs]) return;
    int jb = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < jobs.njobs && i0 >= jobs.off[k]) jb = k;
    const PointJob &J = jobs.j[jb];
    const int64_t i = i0 - jobs.off[jb];
    const int64_t gi = J.q_begin + i;
    const int64_t j = J.idx[i];
    const double ex = __dsub_rn(J.q64[3 * gi], J.r64[3 * j]);
    const double ey = __dsub_rn(J.q64[3 * gi + 1], J.r64[3 * j + 1]);
    const double ez = __dsub_rn(J.q64[3 * gi + 2], J.r64[3 * j + 2]);
    const int64_t k = (J.normal_mode == PCCM_NORMAL_ROW) ? gi : j;
    double p = __dmul_rn(ex, J.nrm[3 * k]);
    p = __fma_rn(ey, J.nrm[3 * k + 1], p);
    p = __fma_rn(ez, J.nrm[3 * k + 2], p);
    J.val[i] = (J.metric == PCCM_METRIC_PROJ) ? p : __dmul_rn(p, p);
}

int launch_point_jobs(pccm_ctx *ctx, const PointJobs &jobs)
{
    const int64_t total = jobs.off[jobs.njobs];
    if (total <= 0) return PCCM_OK;
    ProfScope ps(ctx, PCCM_K_POINT);
    hipLaunchKernelGGL(k_point_jobs, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, jobs);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

// One job = one per-point array and up to TWO columns reduced from it in the same pass: a plain column (stride 1),
// or fields of the grid engine's 32-byte result records (stride 4 doubles: [0] squared distance, [1] signed
// projection) -- the D1 and D2 columns of a direction then cost one read of its records, 16 of every 32 bytes used.
struct UnitView {               // the fields of a job the inner loop needs, in registers (wave-uniform)
    const double *val;
    int stride, off0, off1, sq0, sq1;
    int defer;                  // UnitJob::defer
    int64_t nrm_rows;           // UnitJob::nrm_rows
    const double *nrm64;
    const float4 *nrm32;
    const float4 *q32;
    int64_t row0;
};

// The two fields of a result record of layout 1 (the matched record {rx, ry, rz, row}, NNOut::layout) for row `row` of the
// iterating cloud: squared distance -- nanoflann's accumulation order, as every search kernel evaluates it (gdist64, pccm_grid.h)
// -- and err . normal[row] -- the FMA chain of emit_result / K3; bit for bit what the searches would have stored.
// defer 4 / 5: the normal of the MATCHED row (--normal-index neighbour; the record carries the row), a gather where 1 / 2 stream;
// nrm_rows bounds it (a row outside the searched cloud cannot be in a record the searches wrote: clamped all the same).
__device__ __forceinline__ void matched_fields(const float4 rec, const float4 *__restrict__ q32, int defer, const double *__restrict__ nrm64,
                                               const float4 *__restrict__ nrm32, int64_t row, bool want_proj, double &d2, double &proj,
                                               int64_t nrm_rows = 0)
{
    const float4 q = q32[row];
    const double qx = (double)q.x, qy = (double)q.y, qz = (double)q.z;
    const double ex = __dsub_rn(qx, (double)rec.x), ey = __dsub_rn(qy, (double)rec.y), ez = __dsub_rn(qz, (double)rec.z);
    d2 = __dadd_rn(__dadd_rn(__dmul_rn(ex, ex), __dmul_rn(ey, ey)), __dmul_rn(ez, ez));
    proj = 0.0;
    if (want_proj && defer != 3) {
        double n0, n1, n2;
        int64_t nr = row;
        if (defer >= 4) {
            nr = (int64_t)__float_as_int(rec.w);
            nr = nr < 0 ? 0 : (nr >= nrm_rows ? nrm_rows - 1 : nr);
        }
        if (defer == 1 || defer == 4) {
            const float4 t = nrm32[nr];
            n0 = (double)t.x; n1 = (double)t.y; n2 = (double)t.z;
        } else {
            n0 = nrm64[3 * nr]; n1 = nrm64[3 * nr + 1]; n2 = nrm64[3 * nr + 2];
        }
        proj = __dmul_rn(ex, n0);
        proj = __fma_rn(ey, n1, proj);
        proj = __fma_rn(ez, n2, proj);
    }
}

__device__ __forceinline__ UnitView unit_view(const UnitJob &J)
{
    UnitView w;
    w.val = J.val;
    w.stride = J.stride;
    w.off0 = J.c[0].off; w.off1 = J.c[1].off;
    w.sq0 = J.c[0].square; w.sq1 = J.c[1].square;
    w.defer = J.defer; w.nrm64 = J.nrm64; w.nrm32 = J.nrm32; w.q32 = J.q32; w.row0 = J.row0; w.nrm_rows = J.nrm_rows;
    return w;
}

__device__ __forceinline__ void unit_load(const UnitView &J, int64_t i, double v[2])      // the load alone (callers batch them)
{
    if (J.stride >= 2 && J.defer) {
        matched_fields(reinterpret_cast<const float4 *>(J.val)[i], J.q32, J.defer, J.nrm64, J.nrm32, J.row0 + i, true, v[0], v[1], J.nrm_rows);
    } else if (J.stride >= 2) {
        const double2 t = *reinterpret_cast<const double2 *>(&J.val[i * J.stride]);
        v[0] = t.x;
        v[1] = t.y;
    } else {
        v[0] = v[1] = J.val[i];
    }
}

__device__ __forceinline__ void unit_pick(const UnitView &J, double v[2])                 // fields -> the job's columns
{
    if (J.stride >= 2) {
        const double x = v[0], y = v[1];
        v[0] = J.off0 ? y : x;
        v[1] = J.off1 ? y : x;
    }
    if (J.sq0) v[0] = __dmul_rn(v[0], v[0]);
    if (J.sq1) v[1] = __dmul_rn(v[1], v[1]);
}

__global__ __launch_bounds__(256) void k_unit_jobs(UnitJobs jobs)
{
    __shared__ double ls[2][32], lmn[2][32], lmx[2][32];
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t unit_threads = jobs.uoff[jobs.njobs];
    if (t < unit_threads) {                                // block-uniform: jobs start at multiples of 256 lanes
        int jb = 0;
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (k < jobs.njobs && t >= jobs.uoff[k]) jb = k;
        jb = __builtin_amdgcn_readfirstlane(jb);              // block-uniform by construction: scalar loads of the job
        const UnitJob &J = jobs.j[jb];
        const UnitView V = unit_view(J);
        const int ncols = J.ncols;
        const int64_t u = (t - jobs.uoff[jb]) >> 3;
        const int k = threadIdx.x & 7, grp = threadIdx.x >> 3;
        const int64_t base = u * kLeaf;
        const bool live = u < J.nunits;
        const int64_t cnt = !live ? 0 : ((J.ns - base < kLeaf) ? J.ns - base : kLeaf);
        double r[2] = {0.0, 0.0}, mn[2] = {INFINITY, INFINITY}, mx[2] = {-INFINITY, -INFINITY};
        if (cnt == kLeaf) {
            double v[kLeaf / 8][2];
            // sixteen independent loads first; the layout test sits outside the loop so that they are issued together
            if (V.stride >= 2 && V.defer) {
#pragma unroll
                for (int j = 0; j < kLeaf / 8; ++j) unit_load(V, base + 8 * j + k, v[j]);
            } else if (V.stride >= 2) {
#pragma unroll
                for (int j = 0; j < kLeaf / 8; ++j) {
                    const double2 t = *reinterpret_cast<const double2 *>(&V.val[(base + 8 * j + k) * V.stride]);
                    v[j][0] = t.x;
                    v[j][1] = t.y;
                }
            } else {
#pragma unroll
                for (int j = 0; j < kLeaf / 8; ++j) v[j][0] = v[j][1] = V.val[base + 8 * j + k];
            }
#pragma unroll
            for (int j = 0; j < kLeaf / 8; ++j) unit_pick(V, v[j]);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                r[c] = v[0][c];
                mn[c] = mx[c] = r[c];
#pragma unroll
                for (int j = 1; j < kLeaf / 8; ++j) {
                    r[c] = __dadd_rn(r[c], v[j][c]);
                    mn[c] = fmin(mn[c], v[j][c]);
                    mx[c] = fmax(mx[c], v[j][c]);
                }
            }
        } else {
            for (int64_t e = k; e < cnt; e += 8) {
                double v[2];
                unit_load(V, base + e, v);
                unit_pick(V, v);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    r[c] = __dadd_rn(r[c], v[c]);
                    mn[c] = fmin(mn[c], v[c]);
                    mx[c] = fmax(mx[c], v[c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                r[c] = __dadd_rn(r[c], __shfl_xor(r[c], off));
                mn[c] = fmin(mn[c], __shfl_xor(mn[c], off));
                mx[c] = fmax(mx[c], __shfl_xor(mx[c], off));
            }
        }
        if (k == 0) {
            for (int c = 0; c < ncols; ++c) {
                double *ou = J.c[c].out_units;
                if (live && ou) {                          // per-leaf results: the sharded exchange needs them
                    ou[u] = r[c];
                    ou[J.nunits + u] = mn[c];
                    ou[2 * J.nunits + u] = mx[c];
                }
            }
            for (int c = 0; c < 2; ++c) {
                ls[c][grp] = r[c];
                lmn[c][grp] = mn[c];
                lmx[c][grp] = mx[c];
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            // this block's 32 leaves = one half of an 8192-row NumPy chunk when the shard starts on a chunk
            // boundary: finish NumPy's pairwise tree for the half here (adjacent pairs, five levels), so the
            // host only adds 2 numbers per chunk instead of walking 64 leaves.  Lanes 0..31: column 0, 32..63: column 1.
            const int c = threadIdx.x >> 5, l = threadIdx.x & 31;
            double s = ls[c][l], a = lmn[c][l], b = lmx[c][l];
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                s = __dadd_rn(s, __shfl_xor(s, off));
                a = fmin(a, __shfl_xor(a, off));
                b = fmax(b, __shfl_xor(b, off));
            }
            if (l == 0 && c < ncols) {
                const int64_t blk = (t - jobs.uoff[jb]) >> 8;
                double *ob = J.c[c].out_blocks;
                ob[blk] = s;
                ob[J.nblocks + blk] = a;
                ob[2 * J.nblocks + blk] = b;
            }
        }
        return;
    }
    // raw values of the last, partial 8192-row chunk (NumPy sums them with its own tree on the host)
    const int64_t c0 = t - unit_threads;
    if (c0 >= jobs.toff[jobs.njobs]) return;
    int jb = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (k < jobs.njobs && c0 >= jobs.toff[k]) jb = k;
    const UnitJob &J = jobs.j[jb];
    const UnitView V = unit_view(J);
    const int64_t e = c0 - jobs.toff[jb];
    double v[2];
    unit_load(V, J.tail_first + e, v);
    unit_pick(V, v);
    for (int c = 0; c < J.ncols; ++c) J.c[c].out_tail[e] = v[c];
}

// The same reduction with the jobs' shape fixed at compile time (every report's jobs share one shape): record stride and
// which field feeds which column are template arguments, so the sixteen loads of a lane are one base address + immediate
// offsets and no value passes through a select.  CFG 0: two columns {field 0, field 1 squared} (D1 + D2 of a direction in
// one pass over its result records), 1: field 0, 2: field 1 squared, 3: field 1.  (The general kernel above spends most of
// its 20 us at 1M + 1M points on per-value selects, 64-bit index products and dependent scalar loads; a kernel of this
// shape streams the same 32 MB in 7 us: scripts/micro/reduce_gap.hip.)
__device__ __forceinline__ double dmin_raw(double a, double b)       // operands are finite: no canonicalisation needed
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double dmax_raw(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int STRIDE, int CFG, int DEFER = 0>      // DEFER: UnitJob::defer of every job (records of layout 1; STRIDE 2)
__global__ __launch_bounds__(256) void k_unit_lean(UnitJobs jobs)
{
    constexpr int NC = CFG == 0 ? 2 : 1;
    __shared__ double ls[NC][32], lmn[NC][32], lmx[NC][32];
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t unit_threads = jobs.uoff[jobs.njobs];
    if ((int64_t)blockIdx.x * 256 < unit_threads) {            // block-uniform: jobs start at multiples of 256 lanes
        int jb = 0;
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (k < jobs.njobs && (int64_t)blockIdx.x * 256 >= jobs.uoff[k]) jb = k;
        const UnitJob &J = jobs.j[jb];
        const double *__restrict__ val = J.val;
        const int64_t ns = J.ns, nunits = J.nunits;
        const int64_t u = (t - jobs.uoff[jb]) >> 3;
        const int k = threadIdx.x & 7, grp = threadIdx.x >> 3;
        const int64_t base = u * kLeaf;
        const bool live = u < nunits;
        const int64_t cnt = !live ? 0 : ((ns - base < kLeaf) ? ns - base : kLeaf);
        double r[NC], mn[NC], mx[NC];
        // column values of one record (e: its index in the shard)
        const double *__restrict__ nrm64 = J.nrm64;
        const float4 *__restrict__ nrm32 = J.nrm32;
        const float4 *__restrict__ q32 = J.q32;
        const int64_t row0 = J.row0, nrm_rows = J.nrm_rows;
        auto cols = [&](const double *p, int64_t e, double out[NC]) {
            if (DEFER) {
                double x, y;
                matched_fields(*reinterpret_cast<const float4 *>(p), q32, DEFER, nrm64, nrm32, row0 + e, CFG != 1, x, y, nrm_rows);
                if (CFG == 0) { out[0] = x; out[1] = __dmul_rn(y, y); }
                else if (CFG == 1) out[0] = x;
                else if (CFG == 2) out[0] = __dmul_rn(y, y);
                else out[0] = y;
            } else if (STRIDE >= 2) {
                const double2 q = *reinterpret_cast<const double2 *>(p);
                if (CFG == 0) { out[0] = q.x; out[1] = __dmul_rn(q.y, q.y); }
                else if (CFG == 1) out[0] = q.x;
                else if (CFG == 2) out[0] = __dmul_rn(q.y, q.y);
                else out[0] = q.y;
            } else {
                out[0] = *p;
            }
        };
        if (cnt == kLeaf) {
            const double *p = val + (base + k) * STRIDE;
            double v[kLeaf / 8][NC];
#pragma unroll
            for (int j = 0; j < kLeaf / 8; ++j) cols(p + (int64_t)j * 8 * STRIDE, base + k + 8 * j, v[j]);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                r[c] = v[0][c];
                mn[c] = mx[c] = v[0][c];
#pragma unroll
                for (int j = 1; j < kLeaf / 8; ++j) {
                    r[c] = __dadd_rn(r[c], v[j][c]);
                    mn[c] = dmin_raw(mn[c], v[j][c]);
                    mx[c] = dmax_raw(mx[c], v[j][c]);
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) { r[c] = 0.0; mn[c] = INFINITY; mx[c] = -INFINITY; }
            for (int64_t e = k; e < cnt; e += 8) {
                double w[NC];
                cols(val + (base + e) * STRIDE, base + e, w);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    r[c] = __dadd_rn(r[c], w[c]);
                    mn[c] = dmin_raw(mn[c], w[c]);
                    mx[c] = dmax_raw(mx[c], w[c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                r[c] = __dadd_rn(r[c], __shfl_xor(r[c], off));
                mn[c] = dmin_raw(mn[c], __shfl_xor(mn[c], off));
                mx[c] = dmax_raw(mx[c], __shfl_xor(mx[c], off));
            }
        }
        if (k == 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                double *ou = J.c[c].out_units;
                if (live && ou) {                          // per-leaf results: the sharded exchange needs them
                    ou[u] = r[c];
                    ou[nunits + u] = mn[c];
                    ou[2 * nunits + u] = mx[c];
                }
                ls[c][grp] = r[c];
                lmn[c][grp] = mn[c];
                lmx[c][grp] = mx[c];
            }
        }
        __syncthreads();
        if (threadIdx.x < 32 * NC) {
            // this block's 32 leaves = one half of an 8192-row NumPy chunk: finish NumPy's pairwise tree for the half here
            const int c = threadIdx.x >> 5, l = threadIdx.x & 31;
            double s = ls[c][l], a = lmn[c][l], b = lmx[c][l];
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) {
                s = __dadd_rn(s, __shfl_xor(s, off));
                a = dmin_raw(a, __shfl_xor(a, off));
                b = dmax_raw(b, __shfl_xor(b, off));
            }
            if (l == 0) {
                const int64_t blk = (t - jobs.uoff[jb]) >> 8;
                double *ob = J.c[c].out_blocks;
                ob[blk] = s;
                ob[J.nblocks + blk] = a;
                ob[2 * J.nblocks + blk] = b;
            }
        }
        return;
    }
    // raw values of the last, partial 8192-row chunk (NumPy sums them with its own tree on the host)
    const int64_t c0 = t - unit_threads;
    if (c0 >= jobs.toff[jobs.njobs]) return;
    int jb = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k)
        if (k < jobs.njobs && c0 >= jobs.toff[k]) jb = k;
    const UnitJob &J = jobs.j[jb];
    const int64_t e = c0 - jobs.toff[jb];
    const double *p = J.val + (J.tail_first + e) * STRIDE;
    double w[NC];
    if (DEFER) {
        double x, y;
        matched_fields(*reinterpret_cast<const float4 *>(p), J.q32, DEFER, J.nrm64, J.nrm32, J.row0 + J.tail_first + e, CFG != 1, x, y, J.nrm_rows);
        if (CFG == 0) { w[0] = x; w[1] = __dmul_rn(y, y); }
        else if (CFG == 1) w[0] = x;
        else if (CFG == 2) w[0] = __dmul_rn(y, y);
        else w[0] = y;
    } else if (STRIDE >= 2) {
        const double2 q = *reinterpret_cast<const double2 *>(p);
        if (CFG == 0) { w[0] = q.x; w[1] = __dmul_rn(q.y, q.y); }
        else if (CFG == 1) w[0] = q.x;
        else if (CFG == 2) w[0] = __dmul_rn(q.y, q.y);
        else w[0] = q.y;
    } else {
        w[0] = *p;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) J.c[c].out_tail[e] = w[c];
}

// a job's shape, or -1: (stride, CFG, defer) as stride * 4 + cfg + 64 * defer
static int job_shape(const UnitJob &J)
{
    int cfg = -1;
    if (J.stride == 1) cfg = (J.ncols == 1 && J.c[0].off == 0 && !J.c[0].square) ? 1 : -1;
    else if (J.ncols == 2) cfg = (J.c[0].off == 0 && !J.c[0].square && J.c[1].off == 1 && J.c[1].square) ? 0 : -1;
    else if (J.c[0].off == 0) cfg = J.c[0].square ? -1 : 1;
    else cfg = J.c[0].square ? 2 : 3;
    if (cfg < 0 || (J.stride != 1 && J.stride != 2 && J.stride != 4)) return -1;
    if (J.defer && J.stride != 2) return -1;
    return J.stride * 4 + cfg + 64 * J.defer;
}

// one launch for jobs of one shape (-1: the general kernel)
static void launch_unit_shape(pccm_ctx *ctx, const UnitJobs &jobs, int shape)
{
    const int64_t total = jobs.uoff[jobs.njobs] + jobs.toff[jobs.njobs];
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    switch (shape) {
    case 1 * 4 + 1: hipLaunchKernelGGL((k_unit_lean<1, 1>), grid, block, 0, ctx->stream, jobs); break;
    case 2 * 4 + 0: hipLaunchKernelGGL((k_unit_lean<2, 0>), grid, block, 0, ctx->stream, jobs); break;
    case 2 * 4 + 1: hipLaunchKernelGGL((k_unit_lean<2, 1>), grid, block, 0, ctx->stream, jobs); break;
    case 2 * 4 + 2: hipLaunchKernelGGL((k_unit_lean<2, 2>), grid, block, 0, ctx->stream, jobs); break;
    case 4 * 4 + 0: hipLaunchKernelGGL((k_unit_lean<4, 0>), grid, block, 0, ctx->stream, jobs); break;
    case 4 * 4 + 1: hipLaunchKernelGGL((k_unit_lean<4, 1>), grid, block, 0, ctx->stream, jobs); break;
    case 4 * 4 + 2: hipLaunchKernelGGL((k_unit_lean<4, 2>), grid, block, 0, ctx->stream, jobs); break;
    case 64 + 2 * 4 + 0: hipLaunchKernelGGL((k_unit_lean<2, 0, 1>), grid, block, 0, ctx->stream, jobs); break;      // matched records, fp32-exact normals
    case 64 + 2 * 4 + 1: hipLaunchKernelGGL((k_unit_lean<2, 1, 1>), grid, block, 0, ctx->stream, jobs); break;
    case 64 + 2 * 4 + 2: hipLaunchKernelGGL((k_unit_lean<2, 2, 1>), grid, block, 0, ctx->stream, jobs); break;
    case 128 + 2 * 4 + 0: hipLaunchKernelGGL((k_unit_lean<2, 0, 2>), grid, block, 0, ctx->stream, jobs); break;     // ... fp64 normals
    case 128 + 2 * 4 + 1: hipLaunchKernelGGL((k_unit_lean<2, 1, 2>), grid, block, 0, ctx->stream, jobs); break;
    case 128 + 2 * 4 + 2: hipLaunchKernelGGL((k_unit_lean<2, 2, 2>), grid, block, 0, ctx->stream, jobs); break;
    case 192 + 2 * 4 + 1: hipLaunchKernelGGL((k_unit_lean<2, 1, 3>), grid, block, 0, ctx->stream, jobs); break;     // ... no normals: distances only
    case 256 + 2 * 4 + 0: hipLaunchKernelGGL((k_unit_lean<2, 0, 4>), grid, block, 0, ctx->stream, jobs); break;     // ... the matched row's normal, fp32-exact
    case 256 + 2 * 4 + 2: hipLaunchKernelGGL((k_unit_lean<2, 2, 4>), grid, block, 0, ctx->stream, jobs); break;
    case 320 + 2 * 4 + 0: hipLaunchKernelGGL((k_unit_lean<2, 0, 5>), grid, block, 0, ctx->stream, jobs); break;     // ... fp64
    case 320 + 2 * 4 + 2: hipLaunchKernelGGL((k_unit_lean<2, 2, 5>), grid, block, 0, ctx->stream, jobs); break;
    default: hipLaunchKernelGGL(k_unit_jobs, grid, block, 0, ctx->stream, jobs); break;      // signed projections (min / max of -0.0 and 0.0: fmin / fmax there), other shapes
    }
}

static bool lean_has(int shape)
{
    switch (shape) {
    case 5: case 8: case 9: case 10: case 16: case 17: case 18: case 72: case 73: case 74: case 136: case 137: case 138: case 201: case 264: case 266:
    case 328: case 330: return true;
    default: return false;
    }
}

int launch_unit_jobs(pccm_ctx *ctx, const UnitJobs &jobs)
{
    const int64_t total = jobs.uoff[jobs.njobs] + jobs.toff[jobs.njobs];
    if (total <= 0) return PCCM_OK;
    ProfScope ps(ctx, PCCM_K_REDUCE);
    static const bool general = [] { const char *e = getenv("PCCM_REDUCE_GENERAL"); return e && e[0] == '1'; }();   // A/B: always the general kernel
    int shape[8], nshapes = 0, first_of[8];
    bool all_lean = !general;
    for (int k = 0; k < jobs.njobs; ++k) {
        shape[k] = job_shape(jobs.j[k]);
        all_lean = all_lean && lean_has(shape[k]);
        bool seen = false;
        for (int j = 0; j < nshapes; ++j) seen = seen || shape[first_of[j]] == shape[k];
        if (!seen) first_of[nshapes++] = k;
    }
    if (!all_lean) {                          // one shape nobody specialised: the general kernel takes the whole batch
        launch_unit_shape(ctx, jobs, -1);
    } else if (nshapes == 1) {
        launch_unit_shape(ctx, jobs, shape[0]);
    } else {
        // jobs of different shapes (the pair's two directions with the projection, the self search without): one specialised launch
        // per shape -- the general kernel costs more than a second launch (53 against 20 + 9 us on the 0.8M-point content pair)
        for (int g = 0; g < nshapes; ++g) {
            UnitJobs sub;
            sub.njobs = 0;
            sub.uoff[0] = sub.toff[0] = 0;
            for (int k = 0; k < jobs.njobs; ++k) {
                if (shape[k] != shape[first_of[g]]) continue;
                sub.j[sub.njobs] = jobs.j[k];
                sub.uoff[sub.njobs + 1] = sub.uoff[sub.njobs] + (jobs.uoff[k + 1] - jobs.uoff[k]);
                sub.toff[sub.njobs + 1] = sub.toff[sub.njobs] + (jobs.toff[k + 1] - jobs.toff[k]);
                sub.njobs++;
            }
            for (int k = sub.njobs; k < 8; ++k) {
                sub.j[k] = sub.j[0];
                sub.uoff[k + 1] = sub.uoff[sub.njobs];
                sub.toff[k + 1] = sub.toff[sub.njobs];
            }
            launch_unit_shape(ctx, sub, shape[first_of[g]]);
        }
    }
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

// Result records -> plain columns (only when a consumer wants them: colour kernels, getters, pccm_nn_fetch).
__global__ __launch_bounds__(256) void k_unpack(const double *__restrict__ rec, int stride, int layout, const float4 *__restrict__ q32, int64_t row0,
                                                int64_t ns, int32_t *__restrict__ idx, double *__restrict__ d2)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ns) return;
    if (layout == 1) {                                     // the matched record: {rx, ry, rz, row}
        const float4 r = reinterpret_cast<const float4 *>(rec)[i];
        double x, y;
        matched_fields(r, q32, 3, nullptr, nullptr, row0 + i, false, x, y);
        if (idx) idx[i] = __float_as_int(r.w);
        d2[i] = x;
        return;
    }
    const double *r = rec + i * stride;
    if (stride == 4 && idx) idx[i] = (int32_t)(__double_as_longlong(r[2]) & 0xffffffffll);      // 16-byte records carry no row
    d2[i] = r[0];
}

int launch_unpack(pccm_ctx *ctx, const double *rec, int stride, int layout, const float4 *q32, int64_t row0, int64_t ns, int32_t *idx, double *d2)
{
    if (ns <= 0) return PCCM_OK;
    hipLaunchKernelGGL(k_unpack, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, ctx->stream, rec, stride, layout, q32, row0, ns, idx, d2);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

// NumPy's DOUBLE pairwise sum over one contiguous run of at most kChunk values.
double np_pairwise_sum(const double *a, int64_t n)
{
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= kLeaf) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

}  // namespace pccm
