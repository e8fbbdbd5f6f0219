// Exact rescan of flagged queries (K2b) and the small device helpers it shares with the brute-force engine: included by
// pccm_brute.hip (k2b_fallback: the brute-force engine's uncertified winners) and pccm_grid.hip (k_grid_tail: the same rescan as
// the second half of the grid engine's one tail launch).
//
// Stands under get_neighbour_cloud(), open_pcc_metric/cloud_pair.py:10-42: the queries no cheaper path could settle get the
// definition -- a scan of the whole searched cloud in the reference's fp64 arithmetic, smallest row on ties.
#pragma once
#include "pccm_internal.h"

namespace pccm {

__device__ __forceinline__ float dist32(float qx, float qy, float qz, float rx, float ry, float rz)
{
    float dx = qx - rx, dy = qy - ry, dz = qz - rz;
    float d = dx * dx;
    d = __builtin_fmaf(dy, dy, d);
    d = __builtin_fmaf(dz, dz, d);
    return d;
}

// fp32 "quad" layout: point j lives in quad j/4 as x[j%4], y[j%4], z[j%4] (12 floats per quad).
__device__ __forceinline__ void load_pt32(const float *__restrict__ p, int64_t j, float &x, float &y, float &z)
{
    const float *qd = p + (j >> 2) * 12 + (j & 3);
    x = qd[0];
    y = qd[4];
    z = qd[8];
}

// The reference's squared distance: nanoflann L2 accumulation order, fp64, no contraction.
__device__ __forceinline__ double dist64(double qx, double qy, double qz, double rx, double ry, double rz)
{
    double dx = __dsub_rn(qx, rx), dy = __dsub_rn(qy, ry), dz = __dsub_rn(qz, rz);
    double d = __dmul_rn(dx, dx);
    d = __dadd_rn(d, __dmul_rn(dy, dy));
    d = __dadd_rn(d, __dmul_rn(dz, dz));
    return d;
}

__device__ __forceinline__ double wave_min_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return v;
}

__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
    return v;
}

// ------------------------------------------------------------------------------------------
// K2b: exact rescan of the queries on a job's flagged list (uncertified fp32 winners of K2; queries the
// grid engine's rings could not settle).  The list length is read on the device: no host sync before the
// launch, and a job with an empty list costs an early exit.  Two regimes:
//  * many flagged queries: one workgroup per query (grid-stride over the list), scanning the whole cloud;
//  * at most kSplitMax of them (a stray point far from everything): the CLOUD is split over the workgroups
//    instead -- every workgroup evaluates its slice against each listed query and leaves a partial
//    (d2, row) minimum; the workgroup that takes the last ticket folds the partials.  1M points: ~0.1 ms
//    instead of the 8 ms one wave needs to walk the cloud alone.
// Filter and arithmetic are the same in both: candidates with d32 <= thr are evaluated in fp64 with the
// reference's expression, lexicographic (d2, row) minimum.
// ------------------------------------------------------------------------------------------
template <bool SELF>
__device__ __forceinline__ void rescan_slice(const RescanJob &J, int64_t i, float thr, int64_t j0, int64_t j1, int tid,
                                             double &bd, int &bj)
{
    float q_x, q_y, q_z;
    load_pt32(J.q32, J.q_begin + i, q_x, q_y, q_z);
    const double qx = J.q64[3 * (J.q_begin + i)], qy = J.q64[3 * (J.q_begin + i) + 1], qz = J.q64[3 * (J.q_begin + i) + 2];
    bd = INFINITY;
    bj = 0x7fffffff;
    for (int64_t j = j0 + tid; j < j1; j += 256) {
        float r_x, r_y, r_z;
        load_pt32(J.r32, j, r_x, r_y, r_z);
        const float d = dist32(q_x, q_y, q_z, r_x, r_y, r_z);
        bool cand = d <= thr;
        if (SELF) cand = cand && (j != J.q_begin + i);
        if (cand) {
            const double e = dist64(qx, qy, qz, J.r64[3 * j], J.r64[3 * j + 1], J.r64[3 * j + 2]);
            if (e < bd) { bd = e; bj = (int)j; }    // j ascending per thread: first hit is the smallest
        }
    }
}

// lexicographic (d2, row) minimum over the workgroup; valid in thread 0
__device__ __forceinline__ void block_lexmin(double &bd, int &bj, double *s_d, int *s_j)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const double m = wave_min_f64(bd);
    const int cj = wave_min_i32(bd == m ? bj : 0x7fffffff);
    __syncthreads();                                 // the previous use of s_d / s_j is over
    if (lane == 0) { s_d[w] = m; s_j[w] = cj; }
    __syncthreads();
    if (tid == 0) {
        bd = s_d[0];
        bj = s_j[0];
        for (int k = 1; k < 4; ++k)
            if (s_d[k] < bd || (s_d[k] == bd && s_j[k] < bj)) { bd = s_d[k]; bj = s_j[k]; }
    }
}

// a rescanned query's answer: plain columns (brute-force engine) or a 32-byte result record with the D2
// projection fused (grid engine; same expression as emit_result in pccm_grid.h / K3 in pccm_point.hip)
__device__ __forceinline__ void rescan_emit(const RescanJob &J, int64_t i, int bj, double bd)
{
    if (!J.rec_out) {
        J.idx_out[i] = bj;
        J.d2_out[i] = bd;
        return;
    }
    if (J.rec_layout == 1) {                        // the matched record: the reduction forms distance and projection (NNOut::layout)
        const bool has = bj >= 0 && bj != 0x7fffffff;
        const double *src = has ? J.r64 + 3 * (int64_t)bj : J.q64 + 3 * (J.q_begin + i);
        reinterpret_cast<float4 *>(J.rec_out)[i] = make_float4((float)src[0], (float)src[1], (float)src[2], __int_as_float(has ? bj : -1));
        return;
    }
    double p = 0.0;
    if (J.nrm && bj >= 0 && bj != 0x7fffffff) {
        const int64_t gi = J.q_begin + i, k = (J.normal_mode == PCCM_NORMAL_ROW) ? gi : (int64_t)bj;
        const double ex = __dsub_rn(J.q64[3 * gi], J.r64[3 * (int64_t)bj]);
        const double ey = __dsub_rn(J.q64[3 * gi + 1], J.r64[3 * (int64_t)bj + 1]);
        const double ez = __dsub_rn(J.q64[3 * gi + 2], J.r64[3 * (int64_t)bj + 2]);
        p = __dmul_rn(ex, J.nrm[3 * k]);
        p = __fma_rn(ey, J.nrm[3 * k + 1], p);
        p = __fma_rn(ez, J.nrm[3 * k + 2], p);
    }
    double *dst = J.rec_out + i * J.rec_stride;
    *reinterpret_cast<double2 *>(dst) = make_double2(bd, p);
    if (J.rec_stride == 4) *reinterpret_cast<double2 *>(dst + 2) = make_double2(__longlong_as_double((long long)(uint32_t)bj), 0.0);
}

// workgroup b of the nb that share job J's list (256 threads each; every one of them calls this)
template <bool SELF>
__device__ __forceinline__ void rescan_body(const RescanJob &J, uint32_t b, uint32_t nb)
{
    __shared__ double s_d[4];
    __shared__ int s_j[4];
    __shared__ uint32_t s_ticket;
    const uint32_t count = __hip_atomic_load(J.nflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (count == 0) return;
    const int tid = threadIdx.x;
    if (count > (uint32_t)kSplitMax) {
        for (uint32_t f = b; f < count; f += nb) {
            const int64_t i = __hip_atomic_load(&J.flagged[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            double bd;
            int bj;
            rescan_slice<SELF>(J, i, __hip_atomic_load(&J.flag_thr[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0, J.nr, tid, bd, bj);
            block_lexmin(bd, bj, s_d, s_j);
            if (tid == 0) rescan_emit(J, i, bj, bd);
        }
        return;
    }
    // split regime: this workgroup's slice of the searched cloud against every listed query
    const int64_t per = (J.nr + nb - 1) / nb;
    const int64_t j0 = (int64_t)b * per, j1 = (j0 + per < J.nr) ? j0 + per : J.nr;
    for (uint32_t f = 0; f < count; ++f) {
        double bd;
        int bj;
        rescan_slice<SELF>(J, __hip_atomic_load(&J.flagged[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&J.flag_thr[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), j0, j1, tid, bd, bj);
        block_lexmin(bd, bj, s_d, s_j);
        if (tid == 0) {
            J.part_d[(size_t)f * nb + b] = bd;
            J.part_j[(size_t)f * nb + b] = bj;
        }
    }
    __threadfence();                                 // partials visible before the ticket is taken
    if (tid == 0) s_ticket = atomicAdd(J.ticket, 1u);
    __syncthreads();
    if (s_ticket != nb - 1) return;
    __threadfence();
    for (uint32_t f = 0; f < count; ++f) {           // last workgroup: fold the partials of every query
        double bd = INFINITY;
        int bj = 0x7fffffff;
        for (uint32_t k = tid; k < nb; k += 256) {
            const double d = __hip_atomic_load(&J.part_d[(size_t)f * nb + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int j = __hip_atomic_load(&J.part_j[(size_t)f * nb + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d < bd || (d == bd && j < bj)) { bd = d; bj = j; }
        }
        block_lexmin(bd, bj, s_d, s_j);
        if (tid == 0) rescan_emit(J, __hip_atomic_load(&J.flagged[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), bj, bd);
    }
    if (tid == 0) *J.ticket = 0u;                    // ready for the next launch
}


}  // namespace pccm
