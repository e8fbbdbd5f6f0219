// Declarations of the LDS-brick kernel (pccm_brick.hip) that its parked variants also use (scripts/attic/: resident workgroups
// that walk the brick list with the next brick's loads in flight -- measured and dropped, round 4).
#pragma once
#include "pccm_grid.h"

namespace pccm {

constexpr int kBXMax = 64;
constexpr int kLcsPitch = kBXMax + 3;                // cell starts per staged run (BX + 2 cells + 1), odd pitch
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr float kBigF = 3.0e38f;
constexpr int kExtra = 21;                           // lanes per run of the staging's second load (3 x 21 <= 64)
constexpr float kFar = 1.0e18f;                      // coordinates of pad records: d2 ~ 3e36, finite, never the winner

struct BrickParams {
    int bx;                 // cells per brick along x
    int nbx, nby, nbz;      // bricks per axis
    int64_t per_job;        // nbx * nby * nbz
    uint32_t total;         // bricks of all jobs
    int cap;                // staged records that fit (< 65536: LDS positions are kept as uint16)
    unsigned long long *stamps;   // diagnostic build only (PCCM_BRICK_STAMP=1): per-phase wave-cycle sums, else null
    // the ring-1 stop rule in fp32 (face32): per axis the cell edge and the two face origins, org - h + slack and org + 2 h - slack,
    // where slack = GridGeom::slack + the worst absolute error of the fp32 evaluation (see launch_brick_query)
    float h32[3], face_lo[3], face_hi[3];
    float org32x, invh32x;  // grid origin and inverse cell edge along x, rounded: where a query's windows are centred (any centre is exact)
};

// Distance from the query coordinate q (cell c of its axis) to the nearer face of the ring-1 cube that has cells behind it --
// face_bound() of pccm_grid.h in fp32, never larger than it: the faces lie at org + (c - 1) h and org + (c + 2) h
__device__ __forceinline__ float face32(float q, int c, int dim, float h, float flo, float fhi)
{
    const float cf = (float)c;
    const float lo = q - __builtin_fmaf(cf, h, flo), hi = __builtin_fmaf(cf, h, fhi) - q;
    const float a = c >= 2 ? lo : INFINITY, b = c <= dim - 3 ? hi : INFINITY;
    return a < b ? a : b;
}

// min of two non-negative floats (or +inf) by their bit patterns
__device__ __forceinline__ float umin_f(float a, float b)
{
    const uint32_t x = __float_as_uint(a), y = __float_as_uint(b);
    return __uint_as_float(x < y ? x : y);
}

// plane sizes the kernels are compiled for: the small one holds the bricks of whole clouds at ~1.4 points per cell with four
// workgroups per CU (4 x 2176 floats + 3.6 KB of tables = 38.4 KB), the middle one the 4 x 4-row bricks of sharded ranks with
// three (4 x 3008 floats + 5.4 KB = 53.6 KB -- 3040 floats no longer fit three times; with the large one -- two per CU -- a rank's search took 40 instead of 32 us at
// 1M points / 8 ranks), the large one is the 64 KB workgroup limit
constexpr int kPlaneSmall = 2176, kPlaneMid = 3008, kPlaneLarge = 3584;

}  // namespace pccm
