// Voxel-brick search for VOXELISED content on gfx950: both clouds integer-valued (every PCC test sequence: 8i, Owlii, MVUB --
// BASELINE.json configs[4]).
//
// Stands under get_neighbour_cloud(), open_pcc_metric/cloud_pair.py:10-42.  Two flavours of one kernel: distances only (round 3:
// the callers that read nothing but the squared distances, cloud_pair.py:102-106 -> metric.py:213-247, 353-386), and -- round
// 4 -- with the matched ROW (cloud_pair.py:34-40: colour metrics, error vectors, D2 projections, where an exact tie decides
// which neighbour's row / error vector is taken): once the nearest distance is known, the voxels at exactly that distance are
// enumerated in the staged bits and the smallest row among their points wins, the rule of every other kernel of the library.
// A voxel's smallest row is one gathered word: `minrow`, one entry per occupied voxel of a cell in voxel order, at the cell's
// first record + the voxel's rank among the brick's set bits (k_vox_bricks).  Only the self search with rows (nobody in the
// metric DAG asks for it) still takes the per-thread lattice search (pccm_lattice.hip).
//
// The pair's grid has cells of exactly 8 x 8 x 8 voxels here (origin on the integer lattice), and every occupied cell of a
// cloud owns a 512-bit occupancy BRICK (bit x + 8 y + 64 z; k_vox_bricks, one pass over the cell-sorted records) plus a brick of
// the voxels that hold more than one point (the self search's "another point at distance 0").  One wave serves one occupied
// cell of the iterating cloud:
//   1. 36 lanes fetch the cell starts of the 3 x 3 x-runs around the cell (which of the 27 cells hold anything, and where their
//      bricks are: a brick lives at the index of its cell's first record -- no slot table, nothing to clear),
//   2. the occupied ones' bricks (64 bytes each) and the cell's queries are fetched together,
//   3. the 27 bricks are transposed in LDS into 24 x 24 x-rows of 24 bits, one word per (y, z),
//   4. every lane takes a query and walks the (dy, dz) rows in order of dy^2 + dz^2 (a 197-entry table in scalar memory: the
//      walk is wave-uniform), per row ONE LDS word and a handful of bit instructions: nearest set bit on either side of the
//      query's x (v_ffbl / v_ffbh), d2 = dx^2 + dy^2 + dz^2 -- integers: exact, no certification -- until dy^2 + dz^2 reaches
//      the best d2 so far.
// A best d2 <= 64 is final: every voxel within 8 of a query of the cell lies inside the staged 24^3.  Anything farther (or
// nothing found) goes to the tail list and through the general tail launch (k_grid_tail), like the brick kernel's.
// Results are matched records {x, y, z of the nearest voxel, its smallest row or -1} (NNOut::layout 1): the reductions form the
// distance (and the row-indexed projection).
// Bound: the launch is short and latency-bound (two dependent round trips per wave, thousands of waves in flight); the per-row
// walk is ~12 VALU instructions.
#include "pccm_grid.h"

namespace pccm {

constexpr int kVoxRowsPadded = 200;                 // 197 rows + padding
// (dy^2 + dz^2) << 16 | (dy + 8) << 8 | (dz + 8), sorted: every (dy, dz) with dy^2 + dz^2 <= 64
__constant__ __attribute__((aligned(16))) uint32_t c_vox_rows[kVoxRowsPadded] = {
    0x808, 0x10708, 0x10807, 0x10809, 0x10908, 0x20707, 0x20709, 0x20907, 0x20909, 0x40608, 0x40806, 0x4080a, 0x40a08, 0x50607, 0x50609,
    0x50706, 0x5070a, 0x50906, 0x5090a, 0x50a07, 0x50a09, 0x80606, 0x8060a, 0x80a06, 0x80a0a, 0x90508, 0x90805, 0x9080b, 0x90b08, 0xa0507,
    0xa0509, 0xa0705, 0xa070b, 0xa0905, 0xa090b, 0xa0b07, 0xa0b09, 0xd0506, 0xd050a, 0xd0605, 0xd060b, 0xd0a05, 0xd0a0b, 0xd0b06, 0xd0b0a,
    0x100408, 0x100804, 0x10080c, 0x100c08, 0x110407, 0x110409, 0x110704, 0x11070c, 0x110904, 0x11090c, 0x110c07, 0x110c09, 0x120505, 0x12050b,
    0x120b05, 0x120b0b, 0x140406, 0x14040a, 0x140604, 0x14060c, 0x140a04, 0x140a0c, 0x140c06, 0x140c0a, 0x190308, 0x190803, 0x19080d, 0x190d08,
    0x190405, 0x19040b, 0x190504, 0x19050c, 0x190b04, 0x190b0c, 0x190c05, 0x190c0b, 0x1a0307, 0x1a0309, 0x1a0703, 0x1a070d, 0x1a0903, 0x1a090d,
    0x1a0d07, 0x1a0d09, 0x1d0306, 0x1d030a, 0x1d0603, 0x1d060d, 0x1d0a03, 0x1d0a0d, 0x1d0d06, 0x1d0d0a, 0x200404, 0x20040c, 0x200c04, 0x200c0c,
    0x220305, 0x22030b, 0x220503, 0x22050d, 0x220b03, 0x220b0d, 0x220d05, 0x220d0b, 0x240208, 0x240802, 0x24080e, 0x240e08, 0x250207, 0x250209,
    0x250702, 0x25070e, 0x250902, 0x25090e, 0x250e07, 0x250e09, 0x280206, 0x28020a, 0x280602, 0x28060e, 0x280a02, 0x280a0e, 0x280e06, 0x280e0a,
    0x290304, 0x29030c, 0x290403, 0x29040d, 0x290c03, 0x290c0d, 0x290d04, 0x290d0c, 0x2d0205, 0x2d020b, 0x2d0502, 0x2d050e, 0x2d0b02, 0x2d0b0e,
    0x2d0e05, 0x2d0e0b, 0x310108, 0x310801, 0x31080f, 0x310f08, 0x320107, 0x320109, 0x320701, 0x32070f, 0x320901, 0x32090f, 0x320f07, 0x320f09,
    0x320303, 0x32030d, 0x320d03, 0x320d0d, 0x340204, 0x34020c, 0x340402, 0x34040e, 0x340c02, 0x340c0e, 0x340e04, 0x340e0c, 0x350106, 0x35010a,
    0x350601, 0x35060f, 0x350a01, 0x350a0f, 0x350f06, 0x350f0a, 0x3a0105, 0x3a010b, 0x3a0501, 0x3a050f, 0x3a0b01, 0x3a0b0f, 0x3a0f05, 0x3a0f0b,
    0x3d0203, 0x3d020d, 0x3d0302, 0x3d030e, 0x3d0d02, 0x3d0d0e, 0x3d0e03, 0x3d0e0d, 0x400008, 0x400800, 0x400810, 0x401008,
    0xffff0808, 0xffff0808, 0xffff0808       // padding to whole quadruples: rows nobody can want (row (0, 0) again, at a distance beyond reach)
};

// Every voxel offset at EXACTLY distance^2 d2 from a query, for d2 <= kVoxNear, as dz + 8 | (dy + 8) << 5 | dx << 10 with dx >= 0 (the
// walk looks at both x + dx and x - dx); c_vox_near_start[d2] .. [d2 + 1] is the list of d2.  The matched-row search of a query
// whose nearest voxel is that close -- nearly all of them on decoded content -- looks at these <= 24 places instead of walking
// every row within reach (section 5 of k_vox_query).
constexpr int kVoxNear = 16, kVoxNearEntries = 153;
__constant__ uint16_t c_vox_near[kVoxNearEntries] = {
    0x108, 0x107, 0xe8, 0x508, 0x128, 0x109, 0xe7, 0x507, 0x127, 0x4e8, 0x528, 0xe9, 0x509, 0x129, 0x4e7, 0x527, 0x4e9, 0x529, 0x106,
    0xc8, 0x908, 0x148, 0x10a, 0xe6, 0x506, 0x126, 0xc7, 0x907, 0x147, 0x4c8, 0x8e8, 0x928, 0x548, 0xc9, 0x909, 0x149, 0xea, 0x50a,
    0x12a, 0x4e6, 0x526, 0x4c7, 0x8e7, 0x927, 0x547, 0x4c9, 0x8e9, 0x929, 0x549, 0x4ea, 0x52a, 0xc6, 0x906, 0x146, 0x8c8, 0x948, 0xca,
    0x90a, 0x14a, 0x105, 0x4c6, 0x8e6, 0x926, 0x546, 0x8c7, 0x947, 0xa8, 0xd08, 0x168, 0x8c9, 0x949, 0x4ca, 0x8ea, 0x92a, 0x54a,
    0x10b, 0xe5, 0x505, 0x125, 0xa7, 0xd07, 0x167, 0x4a8, 0xce8, 0xd28, 0x568, 0xa9, 0xd09, 0x169, 0xeb, 0x50b, 0x12b, 0x4e5, 0x525,
    0x4a7, 0xce7, 0xd27, 0x567, 0x4a9, 0xce9, 0xd29, 0x569, 0x4eb, 0x52b, 0x8c6, 0x946, 0x8ca, 0x94a, 0xc5, 0x905, 0x145, 0xa6, 0xd06,
    0x166, 0x8a8, 0xcc8, 0xd48, 0x968, 0xaa, 0xd0a, 0x16a, 0xcb, 0x90b, 0x14b, 0x4c5, 0x8e5, 0x925, 0x545, 0x4a6, 0xce6, 0xd26, 0x566,
    0x8a7, 0xcc7, 0xd47, 0x967, 0x8a9, 0xcc9, 0xd49, 0x969, 0x4aa, 0xcea, 0xd2a, 0x56a, 0x4cb, 0x8eb, 0x92b, 0x54b, 0x104, 0x88,
    0x1108, 0x188, 0x10c
};
__constant__ uint16_t c_vox_near_start[kVoxNear + 2] = {0, 1, 6, 14, 18, 23, 39, 51, 51, 59, 76, 92, 104, 108, 124, 148, 148, 153};

// The occupied cells of every job in cell order, from the occupancy bitmap the build has written: kVoxListWGs workgroups per job,
// each lists the cells of its segment of the bitmap and counts the bits in front of its segment itself (the whole bitmap is a
// few thousand words: reading it eight times costs less than a second launch or a hand-off between workgroups).
// (No counter is contended: see k_vox_bricks.)
constexpr int kVoxListWGs = 8;
__global__ __launch_bounds__(1024) void k_vox_list(VoxBuild vb)
{
    __shared__ uint32_t s_w[16], s_before[16];
    const VoxBuildJob &J = vb.j[blockIdx.y];
    const int64_t ncells = vb.ncells;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t nwords = (ncells + 31) / 32;
    const uint32_t last_mask = (ncells & 31) ? (1u << (ncells & 31)) - 1u : 0xffffffffu;      // (bits beyond the grid are not cells)
    const int64_t seg = (nwords + kVoxListWGs - 1) / kVoxListWGs;
    const int64_t s0r = (int64_t)blockIdx.x * seg, s0 = s0r < nwords ? s0r : nwords;        // (small grids: the last segments are empty)
    const int64_t s1 = (s0 + seg < nwords) ? s0 + seg : nwords;
    // bits in front of the segment
    uint32_t before = 0u;
    for (int64_t k = tid; k < s0; k += 1024) before += (uint32_t)__popc(J.occ[k]);
    // this thread's words of the segment
    const int64_t per = (seg + 1023) / 1024;
    const int64_t w0 = s0 + (int64_t)tid * per, w1 = (w0 + per < s1) ? w0 + per : s1;
    uint32_t cnt = 0u;
    for (int64_t k = w0; k < w1; ++k) cnt += (uint32_t)__popc(k == nwords - 1 ? J.occ[k] & last_mask : J.occ[k]);
    uint32_t inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
    if (lane == 63) s_w[w] = inc;
    if (lane == 0) s_before[w] = before;
    __syncthreads();
    uint32_t pos = inc - cnt;
    for (int k = 0; k < 16; ++k) pos += s_before[k] + (k < w ? s_w[k] : 0u);
    if (blockIdx.x == kVoxListWGs - 1 && tid == 1023) *J.count = pos + cnt;
    for (int64_t k = w0; k < w1; ++k) {
        uint32_t b = k == nwords - 1 ? J.occ[k] & last_mask : J.occ[k];
        while (b) {
            J.list[pos++] = (uint32_t)(k * 32 + __builtin_ctz(b));
            b &= b - 1u;
        }
    }
}

// One workgroup per tile of 64 consecutive cells of a job's grid (their records are one contiguous piece of the cell-sorted
// array): the occupancy brick and the duplicate brick of every occupied cell, OR-ed together in LDS and written at the index of
// the cell's first record.  Empty tiles leave after one look at their cell starts.  (A list of the occupied cells for the
// search to walk is NOT appended here: appends through one device-scope counter -- one per cell, or one per tile -- cost 45 us,
// returning atomics on one address from eight XCDs serialise at memory.  k_vox_list makes it from the occupancy bitmap.)
constexpr int kVoxTile = 64;
constexpr int kVoxMinCap = 4096;                     // records of a tile whose voxel rows are sorted out in LDS at a time
__global__ __launch_bounds__(256) void k_vox_bricks(VoxBuild vb, GridGeom g)
{
    __shared__ uint32_t s_b[kVoxTile * 32];
    __shared__ uint32_t s_cs[kVoxTile + 1];
    const VoxBuildJob &J = vb.j[blockIdx.y];
    const int tid = threadIdx.x;
    const int64_t c0 = (int64_t)blockIdx.x * kVoxTile;
    const int nc = (int)((c0 + kVoxTile < vb.ncells ? c0 + kVoxTile : vb.ncells) - c0);
    if (tid <= nc) s_cs[tid] = J.cs[c0 + tid];
    for (int k = tid; k < kVoxTile * 32; k += 256) s_b[k] = 0u;
    __syncthreads();
    const uint32_t s = s_cs[0], e = s_cs[nc];
    if (e <= s) return;                                   // (block-uniform)
    const int dimx = g.dim[0], dimy = g.dim[1];
    const int gox = (int)g.org[0], goy = (int)g.org[1], goz = (int)g.org[2];
    const float4 *recs = reinterpret_cast<const float4 *>(J.recs);
    for (uint32_t p = s + (uint32_t)tid; p < e; p += 256u) {
        const float4 r = recs[p];
        const int x = (int)r.x - gox, y = (int)r.y - goy, z = (int)r.z - goz;        // >= 0: the origin is the box's lower corner
        const int64_t lin = ((int64_t)(z >> 3) * dimy + (y >> 3)) * dimx + (x >> 3) - c0;   // 0 .. nc - 1: the records are cell-sorted
        const int v = (x & 7) + 8 * (y & 7) + 64 * (z & 7);
        if ((uint64_t)lin >= (uint64_t)nc) {                              // cannot happen for cell-sorted integer records: no wild LDS write if
            if (vb.err) atomicOr(vb.err, kErrVoxState);                   // it does, and the host hears of it (PCCM_E_STATE behind its next wait)
            continue;
        }
        uint32_t *b = s_b + (int)lin * 32;
        const uint32_t bit = 1u << (v & 31);
        const uint32_t old = atomicOr(&b[v >> 5], bit);
        if (old & bit) atomicOr(&b[16 + (v >> 5)], bit);
    }
    __syncthreads();
    for (int k = tid; k < nc * 32; k += 256) {
        const int j = k >> 5;
        const uint32_t a = s_cs[j];
        if (s_cs[j + 1] > a) J.bricks[(size_t)a * 32 + (k & 31)] = s_b[k];
    }
    if (!J.minrow) return;                                // (block-uniform: distances-only grids carry no rows)
    // ---- the smallest row of every occupied voxel, in voxel order: entry [cell's first record + rank of the voxel among the
    //      brick's set bits] (at most one entry per record: the array is as long as the records).  Set bits in front of every
    //      word of a cell's brick first, then one LDS atomicMin per record; tiles of more than kVoxMinCap records in pieces.
    __shared__ uint16_t s_pre[kVoxTile * 16];
    __shared__ int s_min[kVoxMinCap];
    if (tid < nc) {
        uint32_t run = 0u;
        for (int wv = 0; wv < 16; ++wv) {
            s_pre[tid * 16 + wv] = (uint16_t)run;
            run += (uint32_t)__popc(s_b[tid * 32 + wv]);
        }
    }
    for (uint32_t p0 = s; p0 < e; p0 += (uint32_t)kVoxMinCap) {
        __syncthreads();                                  // (s_pre written / the previous piece copied out)
        for (int k = tid; k < kVoxMinCap; k += 256) s_min[k] = 0x7fffffff;
        __syncthreads();
        for (uint32_t p = s + (uint32_t)tid; p < e; p += 256u) {
            const float4 r = recs[p];
            const int x = (int)r.x - gox, y = (int)r.y - goy, z = (int)r.z - goz;
            const int64_t lin = ((int64_t)(z >> 3) * dimy + (y >> 3)) * dimx + (x >> 3) - c0;
            if ((uint64_t)lin >= (uint64_t)nc) continue;                      // (reported above)
            const int v = (x & 7) + 8 * (y & 7) + 64 * (z & 7);
            const uint32_t rank = (uint32_t)s_pre[(int)lin * 16 + (v >> 5)] + (uint32_t)__popc(s_b[(int)lin * 32 + (v >> 5)] & ((1u << (v & 31)) - 1u));
            const uint32_t dst = s_cs[(int)lin] + rank;                        // < the cell's end: a rank counts distinct voxels of the cell
            if (dst >= p0 && dst - p0 < (uint32_t)kVoxMinCap) atomicMin(&s_min[dst - p0], __float_as_int(r.w));
        }
        __syncthreads();
        const uint32_t pe = (e - p0 < (uint32_t)kVoxMinCap) ? e - p0 : (uint32_t)kVoxMinCap;
        for (uint32_t k = (uint32_t)tid; k < pe; k += 256u) J.minrow[p0 + k] = s_min[k];
    }
}

constexpr int kVoxTies = 12;                        // equidistant nearest voxels a query may have before it is left to the tail kernels

template <bool SELF, bool ROWS>
__global__ __launch_bounds__(64) void k_vox_query(QueryJobs jobs, GridGeom g)
{
    static_assert(!(SELF && ROWS), "the self search with matched rows takes the lattice kernel");
    __shared__ uint16_t s_hit[ROWS ? kVoxTies * 64 : 1];   // ROWS: the lanes' equidistant nearest voxels, packed (section 5)
    __shared__ uint32_t s_cs[40];                 // 9 rows x 4 cell starts of the searched cloud, [36], [37]: the cell's query range
    __shared__ uint32_t s_brick[27 * 16];         // occupancy bricks of the 27 cells (zero: empty / outside)
    __shared__ __attribute__((aligned(16))) uint32_t s_rows[576];   // x-rows of the staged 24^3: bit x + 1 of word [Z * 24 + Y]
    __shared__ uint32_t s_dup[16];                // SELF: the cell's own voxels that hold more than one point
    __shared__ uint16_t s_near[ROWS ? kVoxNearEntries : 1], s_near_start[ROWS ? kVoxNear + 2 : 1];   // ROWS: c_vox_near (lanes index it on their own)
    __shared__ __attribute__((aligned(8))) uint16_t s_pre[ROWS ? 27 * 16 : 4];   // ROWS: set bits in front of every word of the 27 bricks (a voxel's rank)
    const QueryJob &J = jobs.j[blockIdx.y];
    const int lane = threadIdx.x;
    const int dimx = g.dim[0], dimy = g.dim[1], dimz = g.dim[2];
    const uint32_t count = *J.vcount;
    const float4 *__restrict__ qrecs = reinterpret_cast<const float4 *>(J.qrecs);
    const uint32_t *__restrict__ cs = J.cs;
    const int gox = (int)g.org[0], goy = (int)g.org[1], goz = (int)g.org[2];
    const uint4 *__restrict__ tab = reinterpret_cast<const uint4 *>(c_vox_rows);
    // one wave per workgroup, one occupied cell of the iterating cloud per turn (the barriers below cost a single-wave workgroup
    // next to nothing).  (Fetching the next turn's list entry and cell starts while this turn
    // computes was measured: no change -- the turns of 8192 resident waves overlap each other already.)
    // what a lane does in every turn, worked out once: its two quarter-bricks of the staging (brick b = t / 4 of 27, cell-start
    // slot r * 4 + k of s_cs) and its three row quads of the transposition
    int st_cs[2], tr_src[3], tr_dst[3];
    if (ROWS) {
        for (int k = lane; k < kVoxNearEntries; k += 64) s_near[k] = c_vox_near[k];
        if (lane < kVoxNear + 2) s_near_start[lane] = c_vox_near_start[lane];
    }                                             // (the first turn's barriers come before anybody reads them)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = lane + 64 * u, b = t >> 2, r = b / 3, k = b - 3 * r;
        st_cs[u] = t < 27 * 4 ? r * 4 + k : -1;
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = lane + 64 * u, Z = t / 6, Yq = t - 6 * Z;                              // 6 quads of Y x 24 Z
        tr_src[u] = t < 144 ? ((Yq >> 1) + 3 * (Z >> 3)) * 48 + 2 * (Z & 7) + (Yq & 1) : -1;  // brick (r * 3 + 0), word 2 z + (y >> 2)
        tr_dst[u] = Z * 24 + 4 * Yq;
    }
    for (uint32_t it = blockIdx.x; it < count; it += gridDim.x) {
        const uint32_t c = J.vlist[it];
        const uint32_t cyz = c / (uint32_t)dimx;
        const int cx = (int)(c - cyz * (uint32_t)dimx), cy = (int)(cyz % (uint32_t)dimy), cz = (int)(cyz / (uint32_t)dimy);
        // ---- 1. cell starts of the 3 x 3 x-runs (cells cx - 1 .. cx + 1, and the end of the last) + the cell's query range -----
        if (lane < 36) {
            // (a row outside the grid: four zeros = three empty cells; the cell left of the grid takes the start of cell 0 and is
            // empty too; x = dimx is the end of the row's last cell, anything beyond reads as 0 = an empty cell)
            const int r = lane >> 2, k = lane & 3;
            const int y = cy + r % 3 - 1, z = cz + r / 3 - 1, x = max(cx - 1 + k, 0);
            uint32_t v = 0u;
            if (y >= 0 && y < dimy && z >= 0 && z < dimz && x <= dimx) v = cs[((uint32_t)z * dimy + y) * dimx + x];
            s_cs[lane] = v;
        } else if (lane < 38) {
            s_cs[lane] = J.qcs[c + (uint32_t)(lane - 36)];
        }
        __syncthreads();
        // (cell starts never decrease; one that does -- memory the build did not write -- must not become a count of 4 billion queries)
        const uint32_t q0 = s_cs[36], nq = s_cs[37] >= s_cs[36] ? s_cs[37] - s_cs[36] : 0u;
        if (s_cs[37] < s_cs[36] && lane == 0 && jobs.err) atomicOr(jobs.err, kErrVoxState);
        // ---- 2. bricks (a quarter-brick per lane and pass) and the first 64 queries, all in flight together -----------------
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (st_cs[u] < 0) continue;
            const int part = lane & 3;
            const uint32_t s = s_cs[st_cs[u]], e = s_cs[st_cs[u] + 1];
            uint4 w = make_uint4(0u, 0u, 0u, 0u), d = w;
            const bool own = SELF && (lane + 64 * u) >> 2 == 13;              // (a query's own voxel lies in the cell itself)
            if (e > s) {
                const uint4 *src = reinterpret_cast<const uint4 *>(J.vbricks + (size_t)s * 32);
                w = src[part];
                if (own) d = src[4 + part];
            }
            reinterpret_cast<uint4 *>(s_brick)[lane + 64 * u] = w;
            if (own) reinterpret_cast<uint4 *>(s_dup)[part] = d;
        }
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((uint32_t)lane < nq) q = qrecs[q0 + lane];
        __syncthreads();
        // ---- 3. transpose: x-row (Y, Z) of the 24^3 = one byte of each of three bricks.  A word of a brick holds the x-bytes of
        //         four consecutive y at one z: a lane turns three such words (the three bricks along x) into the four row words
        //         (Y .. Y + 3, Z) by byte permutes and writes them with one 16-byte store -- rows lie Y-fastest: [Z * 24 + Y] -----
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (tr_src[u] < 0) continue;
            const uint32_t b0 = s_brick[tr_src[u]], b1 = s_brick[tr_src[u] + 16], b2 = s_brick[tr_src[u] + 32];
            uint32_t r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // bits 1 .. 24 = voxels x = 0 .. 23; bits 0 and 25 are sentinels ("points" at x = -1 and x = 24: at least 9 away
                // from every query of the cell, i.e. beyond the 8 this search vouches for): the bit scans never see an empty word
                const uint32_t lo = __builtin_amdgcn_perm(b1, b0, 0x0c0c0000u | (uint32_t)i | ((uint32_t)(4 + i) << 8));      // {b0.i, b1.i, 0, 0}
                const uint32_t all = __builtin_amdgcn_perm(b2, lo, 0x0c000100u | ((uint32_t)(4 + i) << 16));                   // {lo.0, lo.1, b2.i, 0}
                r[i] = (all << 1) | 0x2000001u;
            }
            *reinterpret_cast<uint4 *>(&s_rows[tr_dst[u]]) = make_uint4(r[0], r[1], r[2], r[3]);
        }
        if (ROWS) {
            // set bits in front of every brick word: a lane counts a quarter-brick, the quarters in front of it are its quad's
            // lower lanes (DPP), four 16-bit counts per 8-byte store
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = lane + 64 * u;
                if (t >= 27 * 4) continue;                // (whole quads: 108 = 27 x 4)
                const uint4 w = reinterpret_cast<const uint4 *>(s_brick)[t];
                const uint32_t c0 = (uint32_t)__popc(w.x), c1 = (uint32_t)__popc(w.y), c2 = (uint32_t)__popc(w.z), c3 = (uint32_t)__popc(w.w);
                const int tot = (int)(c0 + c1 + c2 + c3), part = t & 3;
                const int q0 = __builtin_amdgcn_update_dpp(0, tot, 0x00, 0xf, 0xf, false), q1 = __builtin_amdgcn_update_dpp(0, tot, 0x55, 0xf, 0xf, false),
                          q2 = __builtin_amdgcn_update_dpp(0, tot, 0xaa, 0xf, 0xf, false);      // quad_perm: lane 0 / 1 / 2 of the quad
                const uint32_t b0 = (uint32_t)((part > 0 ? q0 : 0) + (part > 1 ? q1 : 0) + (part > 2 ? q2 : 0));
                const uint32_t b1 = b0 + c0, b2 = b1 + c1, b3 = b2 + c2;
                *reinterpret_cast<uint2 *>(&s_pre[t * 4]) = make_uint2(b0 | (b1 << 16), b2 | (b3 << 16));
            }
        }
        __syncthreads();
        // ---- 4. queries --------------------------------------------------------------------------------------------------
        const int rx0 = gox + 8 * (cx - 1), ry0 = goy + 8 * (cy - 1), rz0 = goz + 8 * (cz - 1);      // voxel (0, 0, 0) of the 24^3
        for (uint32_t qb = 0; qb < nq; qb += 64u) {
            const bool have = qb + (uint32_t)lane < nq;
            if (qb && have) q = qrecs[q0 + qb + lane];      // (not `have ? load : q`: a select between the two ADDRESSES puts q -- and the kernel -- into scratch memory)
            // 8 .. 15 by construction (clamped all the same: a shift count or an LDS index must never leave its range -- and a query
            // that does lie outside its cell raises the device error word: the host fails the search instead of reporting it)
            const int ux = (int)q.x - rx0, uy = (int)q.y - ry0, uz = (int)q.z - rz0;
            if (have && ((((ux | uy | uz) & ~15) != 0) || ((ux & uy & uz & 8) == 0)) && jobs.err) atomicOr(jobs.err, kErrVoxState);
            const int lx = have ? min(max(ux, 8), 15) : 8, ly = have ? min(max(uy, 8), 15) : 8, lz = have ? min(max(uz, 8), 15) : 8;
            const int base = lz * 24 + ly;
            const int L = lx + 1;                           // the query's bit in a row word
            const uint32_t below = (1u << L) - 1u;
            uint32_t own = 0xffffffffu;
            if (SELF) {                                     // the query's own voxel counts only when it holds another point
                const int v = (lx - 8) + 8 * (ly - 8) + 64 * (lz - 8);
                own = ((s_dup[v >> 5] >> (v & 31)) & 1u) ? 0xffffffffu : ~(1u << L);
            }
            // best = d2 << 8 | table index of the row it was found in; an idle lane starts settled
            uint32_t best = have ? 0xffffffffu : 0u;
            // rows four at a time (the table is padded to whole quadruples): one scalar load, four LDS reads in flight
            for (int k4 = 0; k4 < kVoxRowsPadded / 4; ++k4) {
                const uint4 e4 = tab[k4];
                if (__ballot((best >> 8) > (e4.x >> 16)) == 0ull) break;   // every lane's best is within reach of the rows done
                const uint32_t ee[4] = {e4.x, e4.y, e4.z, e4.w};
                uint32_t w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int off = ((int)(ee[u] & 0xffu) - 8) * 24 + ((int)((ee[u] >> 8) & 0xffu) - 8);     // dz * 24 + dy
                    w[u] = s_rows[base + off];
                }
                if (SELF && k4 == 0) w[0] &= own;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    // nearest set bit at or above L, and below it: both scans see a sentinel at worst (v_ffbl_b32 / v_ffbh_u32)
                    const uint32_t up = (uint32_t)__builtin_ctz(w[u] >> L);
                    const uint32_t dn = (uint32_t)(__builtin_clz(w[u] & below) + L - 31);
                    const uint32_t dx = up < dn ? up : dn;
                    const uint32_t key = ((dx * dx + (ee[u] >> 16)) << 8) | (uint32_t)(4 * k4 + u);
                    best = key < best ? key : best;
                }
            }
            if (ROWS) {
                // ---- 5. the matched row: every voxel at exactly the nearest distance, then the smallest row among their points.
                //      (a) the voxels at exactly d2 are noted as hits -- {dy, dz, dx, side} in 16 bits, nothing else:
                //      the lanes hit at different rows, and whatever a hit costs is paid by the whole wave at every row;
                const uint32_t d2 = best >> 8;
                const bool ok = have && d2 <= 64u, close = ok && d2 <= (uint32_t)kVoxNear;
                uint32_t cnt = 0u;
                //      near queries (d2 <= kVoxNear: nearly all on decoded content) go through the list of places at exactly d2;
                {
                    const int st = close ? (int)s_near_start[d2] : 0, len = close ? (int)s_near_start[d2 + 1u] - st : 0;
                    for (int i = 0; __ballot(i < len) != 0ull; ++i) {
                        if (i >= len) continue;
                        const uint32_t code = s_near[st + i];
                        const int dz = (int)(code & 0x1fu) - 8, dy = (int)((code >> 5) & 0x1fu) - 8, dx = (int)(code >> 10);
                        const uint32_t w = s_rows[base + dz * 24 + dy];
                        if ((w >> (L + dx)) & 1u) {                                   // (L + dx <= 20, L - dx >= 5: never a sentinel bit)
                            if (cnt < (uint32_t)kVoxTies) s_hit[cnt * 64u + (uint32_t)lane] = (uint16_t)code;
                            ++cnt;
                        }
                        if (dx && ((w >> (L - dx)) & 1u)) {
                            if (cnt < (uint32_t)kVoxTies) s_hit[cnt * 64u + (uint32_t)lane] = (uint16_t)(code | 0x8000u);
                            ++cnt;
                        }
                    }
                }
                //      the others (a wave rarely has one) walk the rows within reach;
                const bool far = ok && !close;
                for (int k4 = 0; k4 < kVoxRowsPadded / 4; ++k4) {
                    const uint4 e4 = tab[k4];
                    if (__ballot(far && d2 >= (e4.x >> 16)) == 0ull) break;    // (rows are sorted by dy^2 + dz^2: nobody reaches further)
                    const uint32_t ee[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t r2 = ee[u] >> 16;
                        const uint32_t rem = d2 - r2;
                        const uint32_t dx = (uint32_t)(__builtin_sqrtf((float)(int)rem) + 0.5f);
                        if (!(far && r2 <= d2 && dx * dx == rem)) continue;       // no voxel of this row lies at exactly d2 from this query
                        const int dz = (int)(ee[u] & 0xffu) - 8, dy = (int)((ee[u] >> 8) & 0xffu) - 8;
                        const uint32_t w = s_rows[base + dz * 24 + dy];
                        const uint32_t code = (ee[u] & 0x1fu) | (((ee[u] >> 8) & 0x1fu) << 5) | (dx << 10);   // dz + 8 | (dy + 8) << 5 | dx << 10 (dx <= 8)
                        if ((w >> (L + (int)dx)) & 1u) {                           // (L + dx <= 24, L - dx >= 1: never a sentinel bit)
                            if (cnt < (uint32_t)kVoxTies) s_hit[cnt * 64u + (uint32_t)lane] = (uint16_t)code;
                            ++cnt;
                        }
                        if (dx && ((w >> (L - (int)dx)) & 1u)) {
                            if (cnt < (uint32_t)kVoxTies) s_hit[cnt * 64u + (uint32_t)lane] = (uint16_t)(code | 0x8000u);
                            ++cnt;
                        }
                    }
                }
                //      (b) hit j of every lane at once: its voxel, the voxel's rank among the set bits of its cell's brick (the
                //      bits in front of its word from s_pre + those below it in the word), the gather of that voxel's smallest
                //      row -- the gathers of a turn in flight together;
                //      four hits per turn (a query rarely has more): the smallest row so far is all that is carried along;
                //      (c) the smallest row wins (rows are unique: no tie is left).
                int wrow = 0x7fffffff;
                uint32_t wv3 = 0u;
                const uint32_t usable = cnt <= (uint32_t)kVoxTies ? cnt : 0u;
                for (int j0 = 0; j0 < kVoxTies; j0 += 4) {
                    if (__ballot((uint32_t)j0 < usable) == 0ull) break;
                    int rows_of[4];
                    uint32_t vox_of[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const int j = j0 + jj;
                        rows_of[jj] = 0x7fffffff;
                        vox_of[jj] = 0u;
                        if ((uint32_t)j < usable) {
                            const uint32_t h = s_hit[j * 64 + lane];
                            const int dz = (int)(h & 0x1fu) - 8, dy = (int)((h >> 5) & 0x1fu) - 8, dxs = (int)((h >> 10) & 0xfu);
                            const int X = (h & 0x8000u) ? lx - dxs : lx + dxs, Y = ly + dy, Z = lz + dz;      // 0 .. 23 each
                            const int rr = (Y >> 3) + 3 * (Z >> 3), bb = rr * 3 + (X >> 3), v = (X & 7) + 8 * (Y & 7) + 64 * (Z & 7);
                            const uint32_t rank = (uint32_t)s_pre[bb * 16 + (v >> 5)] + (uint32_t)__popc(s_brick[bb * 16 + (v >> 5)] & ((1u << (v & 31)) - 1u));
                            rows_of[jj] = J.vminrow[s_cs[rr * 4 + (X >> 3)] + rank];
                            vox_of[jj] = (uint32_t)X | ((uint32_t)Y << 5) | ((uint32_t)Z << 10);
                        }
                    }
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        if (rows_of[jj] < wrow) {
                            wrow = rows_of[jj];
                            wv3 = vox_of[jj];
                        }
                }
                if (have) {
                    if (ok && cnt >= 1u && cnt <= (uint32_t)kVoxTies && wrow != 0x7fffffff) {
                        store_result_rec(J.out, __float_as_int(q.w), (float)(rx0 + (int)(wv3 & 31u)), (float)(ry0 + (int)((wv3 >> 5) & 31u)),
                                         (float)(rz0 + (int)(wv3 >> 10)), wrow);
                    } else {
                        // nothing within 8 voxels, more equidistant neighbours than the list holds, or (cannot happen) a set bit
                        // without a row: the general kernels decide
                        if (ok && (cnt == 0u || (cnt <= (uint32_t)kVoxTies && wrow == 0x7fffffff)) && jobs.err) atomicOr(jobs.err, kErrVoxState);
                        const uint32_t pos = atomicAdd(&J.counters[1], 1u);
                        reinterpret_cast<float4 *>(J.tail)[pos] = q;
                    }
                }
            } else if (have) {
                const uint32_t d2 = best >> 8;
                const int qrow = __float_as_int(q.w);
                if (d2 <= 64u) {
                    // the winning row again: which side, how far
                    const uint32_t e = c_vox_rows[best & 0xffu];
                    const int dy = (int)((e >> 8) & 0xffu) - 8, dz = (int)(e & 0xffu) - 8;
                    uint32_t w = s_rows[base + dz * 24 + dy];
                    if (SELF && (best & 0xffu) == 0u) w &= own;
                    const int up = __builtin_ctz(w >> L), dn = __builtin_clz(w & below) + L - 31;
                    const int mx = dn < up ? lx - dn : lx + up;
                    store_result_rec(J.out, qrow, (float)(rx0 + mx), (float)(ry0 + ly + dy), (float)(rz0 + lz + dz), -1);
                } else {
                    const uint32_t pos = atomicAdd(&J.counters[1], 1u);
                    reinterpret_cast<float4 *>(J.tail)[pos] = q;
                }
            }
        }
        __syncthreads();                     // the next cell overwrites the staging
    }
}

int launch_vox_bricks(pccm_ctx *ctx, const VoxBuild &vb, const GridGeom &g)
{
    dim3 grid((unsigned)((vb.ncells + kVoxTile - 1) / kVoxTile), (unsigned)vb.njobs);
    hipLaunchKernelGGL(k_vox_bricks, grid, dim3(256), 0, ctx->stream, vb, g);
    hipLaunchKernelGGL(k_vox_list, dim3(kVoxListWGs, (unsigned)vb.njobs), dim3(1024), 0, ctx->stream, vb);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

int launch_vox_query(pccm_ctx *ctx, const QueryJobs &jobs, const GridGeom &g, bool self, bool rows)
{
    // one wave per occupied cell and turn, 32 waves per CU resident: a fixed grid walks the list (its length lives on the device)
    dim3 grid(8192u, (unsigned)jobs.njobs);
    if (self && rows) return fail(PCCM_E_STATE, "the self search with matched rows does not run on voxel bricks");
    if (self) hipLaunchKernelGGL((k_vox_query<true, false>), grid, dim3(64), 0, ctx->stream, jobs, g);
    else if (rows) hipLaunchKernelGGL((k_vox_query<false, true>), grid, dim3(64), 0, ctx->stream, jobs, g);
    else hipLaunchKernelGGL((k_vox_query<false, false>), grid, dim3(64), 0, ctx->stream, jobs, g);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

}  // namespace pccm
