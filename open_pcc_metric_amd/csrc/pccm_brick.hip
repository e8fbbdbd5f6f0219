// LDS-brick ring-1 search for gfx950: the dominant kernel of the default (grid) engine on fp32-exact clouds.
//
// Stands under get_neighbour_cloud(), open_pcc_metric/cloud_pair.py:10-42 (one search_knn_vector_3d call per point there).
//
// One workgroup owns a *brick* of the grid: BX x BY x BZ cells (BX <= 48 cells along the x-fastest axis, BY = 4, BZ = 2).  Its
// queries are the iterating cloud's records in those cells -- eight contiguous runs of the cell-sorted array -- and every
// candidate any of them can have in ring 1 lies in the (BX + 2) x 6 x 4 cells around the brick: 24 contiguous x-runs of the
// searched cloud's cell-sorted array.
//   1. 24 + 8 lanes fetch the run bounds (cell_start) of the searched and the iterating cloud; two wave scans turn them into
//      LDS offsets (every run starts on an even position: odd runs get one far-away pad record),
//   2. wave w copies runs w, w + 8, w + 16 into LDS -- every global load of the phase issued before the first LDS write -- as
//      four PLANES x[PL] | y[PL] | z[PL] | row[PL] (PL a template parameter: the scan reaches the planes through the DS
//      instructions' immediate offsets; ~3 staged records per query), together with their cell starts as LDS byte addresses,
//   3. every lane takes ONE query of the brick (the workgroup is sized to the brick's expected query count) in cell-sorted
//      order; per query nine x-runs of three cells each are scanned a PAIR of candidates per trip: three ds_read_b64 off one
//      address register, six packed-fp32 instructions for both candidates, the nearer of the two into (best, second best,
//      address of the best pair) -- 13 VALU instructions per pair.  The reads are software-pipelined (round 4): the next pair's
//      are issued as soon as this pair's coordinates have been consumed.  A pair may bring a record of a neighbouring cell
//      along: a real point that can only lose or be rejected by the stop rule, so no per-candidate range test exists,
//   4. certification as in pccm_brute.hip K2 -- the fp32 winner is the unique fp64 winner when the second best d32 (and the
//      winner's partner in its pair) lies above thr(best) -- the ring-1 stop rule in fp32 (never more permissive than the fp64
//      one), and ONE 16-byte store per query: the matched record itself {x, y, z, row} (NNOut::layout 1; distance and
//      row-indexed projection are formed by the reduction that reads the records, pccm_point.hip).  Pairs that need the
//      projection on the neighbour's normal keep round 2's epilogue ({d2, projection[, row]} records, normal gathered here).
// Queries that are not certified (near ties) or not settled by ring 1 go to the tail list (k_grid_tail: rings 2-3, then the
// exact rescan, one launch).  A brick whose 24 runs do not fit the LDS budget (clumped data) sends all its queries there.
//
// What bounds it (round 4, PMC busy counters of the launch at 1M + 1M points, profiles/r04): instruction throughput, spread
// over all pipes rather than sitting in one -- per wave 930 VALU (SQ_ACTIVE_INST_VALU = one quad-cycle each: 67 % of the
// launch per SIMD if a vector instruction held its SIMD for four clocks, 43 % at the guide's two clocks for plain and four for
// packed ones), 506 SALU + 123 branches (48 % of the CU's scalar issue), 176 LDS; waves are parked in s_waitcnt / barriers
// 45 % of their cycles and ready-but-not-issued 29 %.  Experiments that moved the time: fewer instructions (round 3, in
// proportion), software-pipelined LDS reads (-5 %); experiments that did not: spreading the first workgroups of a CU over a
// workgroup's lifetime (s_sleep stagger: +0..4 us), two pairs per trip (27 instead of 38 issue slots per four candidates: +-0),
// resident workgroups with the next brick's loads in flight under the scan (scripts/attic/pccm_bstream.hip.txt: 83 against 70 us
// -- 26 more registers cost a workgroup per CU and the per-brick instructions stay).  The time is 41 us + 7.7 ns per workgroup
// over bricks of 16 / 23 / 30 / 45 cells: per-brick set-up (run bounds, staging, window bounds, epilogue: 39 % of the vector
// instructions) is what a shorter kernel has to remove.  Reported against HBM as the contract asks.
// Neighbouring bricks share runs through the XCD's L2 (XCD-aware brick order, as in round 1).
#include "pccm_brick.h"

namespace pccm {

// In-kernel stamps (cdna_hip_programming.md section 7): a SEPARATE instantiation of the kernel, selected by
// PCCM_BRICK_STAMP=1, adds each wave's cycles per phase into bp.stamps; the product kernel (STAMP = false) executes none.
#define BRICK_STAMP(k)                                                                       \
    do {                                                                                     \
        if (STAMP) {                                                                         \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                    \
            t_sum[k] += now_ - t_last;        /* registers; flushed once when the wave ends */ \
            t_last = now_;                                                                   \
        }                                                                                    \
    } while (0)

// the normal of `row`: one aligned 16-byte word when the cloud's normals are fp32-exact (exact widening), else 24 bytes of fp64
__device__ __forceinline__ void load_normal(const NNOut &o, int row, double &a, double &b, double &c)
{
    if (o.nrm32) {
        const float4 t = o.nrm32[row];
        a = (double)t.x; b = (double)t.y; c = (double)t.z;
    } else {
        const double *np = o.nrm + 3 * (int64_t)row;
        a = np[0]; b = np[1]; c = np[2];
    }
}

// N32: every job's normals (if any are fused) are the fp32-exact 16-byte words of nrm32
// PL: floats per LDS plane (compile-time: the scan reaches the y and z planes through the DS instructions' immediate offsets)
template <bool SELF, int BY, int BZ, bool STAMP, int ABL, bool N32, int PL>   // ABL: timing-only ablations (PCCM_BRICK_ABLATE), wrong results
__device__ __forceinline__ void brick_body(const QueryJobs &jobs, const GridGeom &g, const BrickParams &bp, const uint32_t vblock)
{
    if (ABL & 32) return;                                              // timing only: the empty launch
    const int NT = (int)blockDim.x;
    unsigned long long t_last = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long t_start = t_last;
    unsigned long long t_sum[6] = {0, 0, 0, 0, 0, 0};
    constexpr int kNRow = BY * BZ;                   // query rows of a brick
    constexpr int kNRun = (BY + 2) * (BZ + 2);       // staged x-runs
    static_assert(kNRun <= 64 && kNRow <= 64, "one wave scans the run / row lengths");
    // staged records as four planes x[PL], y[PL], z[PL], row[PL]: the scan reads a pair of candidates with three ds_read_b64
    // (2 LDS cycles each, 64 banks: 32 consecutive pairs are conflict-free) and has its packed-fp32 operands (x0 x1), (y0 y1),
    // (z0 z1) in place; rows are read for the winner only.  (Round 2's {x0 x1 y0 y1 | z0 z1 row0 row1} slots took two
    // ds_read_b128 = 8 cycles, each on half of the banks: 37 % of the kernel's LDS cycles were bank conflicts.)
    extern __shared__ __attribute__((aligned(16))) float s_pl[];     // 4 x PL floats
    static_assert(8 * PL * 4 / 2 < 65536 && 12 * PL < 65536, "plane offsets are 16-bit immediates");
    __shared__ uint16_t s_lcs[kNRun * kLcsPitch];
    __shared__ uint32_t s_g0[kNRun], s_len[kNRun], s_base[kNRun + 1], s_qg0[kNRow], s_qoff[kNRow + 1];
    float *const s_x = s_pl, *const s_y = s_pl + PL, *const s_z = s_pl + 2 * PL, *const s_r = s_pl + 3 * PL;
    const uint32_t lds0 = (uint32_t)(uintptr_t)s_x;   // < 8 KB of tables in front, 16-byte aligned; lds0 + 4 PL < 65536: addresses fit uint16
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave-uniform by construction: everything derived from it is scalar
    // XCD-aware order: workgroups b, b+8, ... share an XCD; give every XCD one contiguous eighth of the brick list
    const uint32_t nblk = bp.total, xcd = vblock & 7u, slot = vblock >> 3;
    const uint32_t bq = nblk >> 3, br = nblk & 7u;
    int64_t vb = (int64_t)(xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + slot;
    const int jb = (jobs.njobs > 1 && vb >= bp.per_job) ? 1 : 0;
    if (jb) vb -= bp.per_job;
    if (vb >= bp.per_job) return;
    const QueryJob &J = jobs.j[jb];
    const int dimx = g.dim[0], dimy = g.dim[1], dimz = g.dim[2];
    // brick -> (x, y, z): two 32-bit divisions on wave-uniform values (the 64-bit ones they replace were 360 scalar and 50
    // vector instructions per wave)
    const uint32_t vbu = (uint32_t)vb;
    const uint32_t t_xy = vbu / (uint32_t)bp.nbx;
    const uint32_t t_z = t_xy / (uint32_t)bp.nby;
    const int ibx = (int)(vbu - t_xy * (uint32_t)bp.nbx), iby = (int)(t_xy - t_z * (uint32_t)bp.nby), ibz = (int)t_z;
    const int bx0 = ibx * bp.bx, bx1 = min(bx0 + bp.bx, dimx);
    const int by0 = iby * BY, bz0 = ibz * BZ;
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
            const int y = by0 - 1 + lane % (BY + 2), z = bz0 - 1 + lane / (BY + 2);
            if (y >= 0 && y < dimy && z >= 0 && z < dimz) {
                const uint32_t rowbase = ((uint32_t)z * dimy + y) * dimx;
                g0 = cs[rowbase + sx0];
                len = cs[rowbase + sx1] - g0;
            }
            s_g0[lane] = g0;
            s_len[lane] = len;
        }
        // every run starts on an even position (a whole slot): odd runs are padded with one far-away record, so the
        // slot-wise scan below never picks up a record of a neighbouring run twice
        const uint32_t plen = (len + 1u) & ~1u;
        uint32_t inc = plen;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane < kNRun) s_base[lane] = inc - plen;
        if (lane == kNRun - 1) s_base[kNRun] = inc;
    } else if (w == 1) {
        uint32_t len = 0, q0 = 0;
        if (lane < kNRow) {
            const int y = by0 + lane % BY, z = bz0 + lane / BY;
            if (y < dimy && z < dimz) {
                const uint32_t rowbase = ((uint32_t)z * dimy + y) * dimx;
                q0 = qcs[rowbase + bx0];
                len = qcs[rowbase + bx1] - q0;
            }
            s_qg0[lane] = q0;
        }
        uint32_t inc = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (lane < kNRow) s_qoff[lane] = inc - len;
        if (lane == kNRow - 1) s_qoff[kNRow] = inc;
    }
    __syncthreads();
    BRICK_STAMP(0);                                                    // run bounds + first barrier
    if (ABL & 16) return;                                              // timing only: bounds + one barrier
    const uint32_t T = s_base[kNRun], NQ = s_qoff[kNRow];
    if (NQ == 0) return;                                               // block-uniform

    // query qi of the brick -> its row of the brick and its record
    auto query_row = [&](uint32_t qi) {
        int r = 0;
#pragma unroll
        for (int k = 1; k < kNRow; ++k) r += (qi >= s_qoff[k]) ? 1 : 0;
        return r;
    };
    auto load_query = [&](uint32_t qi, int r) {
        return *reinterpret_cast<const float4 *>(&qbase[s_qg0[r] + (qi - s_qoff[r])]);
    };

    if (T > (uint32_t)bp.cap) {
        // clumped data: more candidates than the LDS budget holds -- the general kernels take this brick's queries
        for (uint32_t qi = tid; qi < NQ; qi += NT) {
            const float4 q = load_query(qi, query_row(qi));
            const uint32_t pos = atomicAdd(&J.counters[1], 1u);
            reinterpret_cast<float4 *>(J.tail)[pos] = q;
        }
        return;
    }

    // the lane's first query is fetched while the brick is staged: its load is issued BEHIND the staging loads (step 2b) --
    // issued first, hipcc copies its row word right behind the load and parks the wave on a full memory round trip before a
    // single staging load has left (round 2's kernel did: ISA, `global_load_dwordx4 ; s_waitcnt vmcnt(0) ; v_mov`)
    uint32_t qi = tid;
    bool have = qi < NQ;
    int rn = 0;
    float4 qn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (have) rn = query_row(qi);

    // ---- 2. stage cell starts (wave w: runs w, w + NT/64, ...) and records (interleaved 64-record pieces) -------
    // Every global load of a batch is issued before the first LDS write that depends on one (a copy loop that loads,
    // waits and writes per record exposes a memory round trip per iteration: the in-kernel stamps put 51 % of the wave
    // cycles of round 2's first version here).
    const NNOut &out = J.out;
    const bool fuse = out.nrm != nullptr;
    const bool fuse_row = fuse && out.normal_mode == PCCM_NORMAL_ROW;
    // row-indexed normal of the lane's current query, AS LOADED (the 16-byte word of nrm32, or the fp64 row): a conversion at
    // the load makes the wave wait for the gather -- a random access -- in front of the barrier (round 2's kernel did);
    // converted in the epilogue instead.  The load is unconditional (a lane without a query, or a job without normals, reads
    // the first query record of the job instead): a conditional one ends in register copies behind the load, i.e. the same wait.
    double n0 = 0.0, n1 = 0.0, n2 = 0.0;
    float4 nraw = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool defer = out.layout == 1;             // the reduction forms err . normal[row]: no normal is read here (NNOut::layout)
    auto gather_row_normal = [&](int row, bool on) {
        if (defer) return;
        if (N32) {
            const float4 *src = on ? out.nrm32 + row : reinterpret_cast<const float4 *>(qbase);
            nraw = *src;
        } else if (on) {
            const double *np = out.nrm + 3 * (int64_t)row;
            n0 = np[0]; n1 = np[1]; n2 = np[2];
        }
    };
    {
        const int nw = NT >> 6;
        // Wave w copies runs w, w + nw, w + 2 nw (a run = the records of BX + 2 consecutive cells, contiguous in the
        // cell-sorted array: lane l takes record l and l + 64 of it -- no search for "which run is position f in", which
        // cost the flat 64-record pieces of the first version a chain of dependent LDS reads per load).
        // (the loop body runs once for every configuration the host picks: 24 runs, 8 waves)
        for (int r0 = w; r0 < kNRun; r0 += 3 * nw) {
            // a. record loads: records 0..63 of each of the three runs, and ONE more load for what the runs have beyond
            //    (a run holds ~69 records here: lanes 0..20 take records 64..84 of the first run, 21..41 of the second,
            //    42..62 of the third; longer runs finish in the loop below)
            float4 rec[3], rex;
            uint32_t rb[3], rl[3], rg[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int r = r0 + u * nw;
                rb[u] = rl[u] = rg[u] = 0u;
                if (r < kNRun) {                                       // wave-uniform
                    rb[u] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_base[r]);      // wave-uniform: scalar registers
                    rl[u] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_len[r]);
                    rg[u] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_g0[r]);
                }
                rec[u] = make_float4(kFar, kFar, kFar, __int_as_float(-1));                // the pad of an odd run
                if (!(ABL & 2) && (uint32_t)lane < rl[u]) rec[u] = *reinterpret_cast<const float4 *>(&srecs[rg[u] + (uint32_t)lane]);
            }
            const int xu = lane / kExtra;                              // 0..2 (lane 63: 3 = nobody)
            const uint32_t xp = 64u + (uint32_t)(lane - xu * kExtra);
            const uint32_t xb = xu == 0 ? rb[0] : xu == 1 ? rb[1] : rb[2];
            const uint32_t xl = xu == 0 ? rl[0] : xu == 1 ? rl[1] : xu == 2 ? rl[2] : 0u;
            const uint32_t xg = xu == 0 ? rg[0] : xu == 1 ? rg[1] : rg[2];
            rex = make_float4(kFar, kFar, kFar, __int_as_float(-1));
            if (!(ABL & 2) && xp < xl) rex = *reinterpret_cast<const float4 *>(&srecs[xg + xp]);
            // b. cell-start loads of this wave's runs
            uint32_t v[3][2];
            bool in[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int r = r0 + u * nw;
                v[u][0] = v[u][1] = 0u;
                in[u] = false;
                if (r < kNRun) {                                       // wave-uniform
                    const int y = by0 - 1 + r % (BY + 2), z = bz0 - 1 + r / (BY + 2);
                    in[u] = y >= 0 && y < dimy && z >= 0 && z < dimz;
                    if (in[u] && !(ABL & 8)) {
                        const uint32_t *src = cs + ((uint32_t)z * dimy + y) * dimx + sx0;
                        if (lane <= ncs) v[u][0] = src[lane];
                        if (lane + 64 <= ncs) v[u][1] = src[lane + 64];
                    }
                }
            }
            if (have && r0 == w) qn = load_query(qi, rn);             // (the youngest load: nothing below waits for it)
            // c. the LDS writes (in the order the loads return)
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const uint32_t plen = (rl[u] + 1u) & ~1u;
                if ((uint32_t)lane < plen) {
                    const uint32_t f = rb[u] + (uint32_t)lane;
                    s_x[f] = rec[u].x;
                    s_y[f] = rec[u].y;
                    s_z[f] = rec[u].z;
                    s_r[f] = rec[u].w;
                }
            }
            if (xp < ((xl + 1u) & ~1u)) {
                const uint32_t f = xb + xp;
                s_x[f] = rex.x;
                s_y[f] = rex.y;
                s_z[f] = rex.z;
                s_r[f] = rex.w;
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                // a run of more than 64 + kExtra records (crowded rows): the rest, piece by piece
                const uint32_t plen = (rl[u] + 1u) & ~1u;
                for (uint32_t p = (uint32_t)lane + 64u + (uint32_t)kExtra; p < plen; p += 64u) {
                    float4 t = make_float4(kFar, kFar, kFar, __int_as_float(-1));
                    if (!(ABL & 2) && p < rl[u]) t = *reinterpret_cast<const float4 *>(&srecs[rg[u] + p]);
                    const uint32_t f = rb[u] + p;
                    s_x[f] = t.x;
                    s_y[f] = t.y;
                    s_z[f] = t.z;
                    s_r[f] = t.w;
                }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int r = r0 + u * nw;
                if (r < kNRun) {
                    // (as LDS byte addresses of the x plane: the scan starts at one and stops at the other without arithmetic)
                    const uint32_t base = lds0 + 4u * rb[u], rebase = base - 4u * rg[u];
                    if (lane <= ncs) s_lcs[r * kLcsPitch + lane] = (uint16_t)(in[u] ? 4u * v[u][0] + rebase : base);
                    if (lane + 64 <= ncs) s_lcs[r * kLcsPitch + lane + 64] = (uint16_t)(in[u] ? 4u * v[u][1] + rebase : base);
                }
            }
        }
        // d. the first query has arrived with the records (loads return in order): its row-indexed normal (quirk Q1)
        //    travels while the workgroup meets at the barrier and scans
        __builtin_amdgcn_sched_barrier(0);
        gather_row_normal(__float_as_int(qn.w), have && fuse_row);
    }
    BRICK_STAMP(1);                                                    // staging issued (+ waits of its loads)
    __syncthreads();
    BRICK_STAMP(2);                                                    // second barrier

    // ---- 3. queries ------------------------------------------------------------------------------------------
    int q_iter = 0;
    while (have) {                                                     // no barrier below: lanes may leave
        const float4 q = qn;
        const int r = rn;
        const int qrow = __float_as_int(q.w);
        const uint32_t qnext = qi + NT;
        const bool hn = qnext < NQ;                                    // a leftover query (rare: the workgroup is sized to the brick)
        const double qx = (double)q.x, qy = (double)q.y, qz = (double)q.z;
        const int ly = r % BY, lz = r / BY;
        // the x-cell the windows are centred on, in fp32 and clamped to the brick: whatever it is, the windows [cx - 1, cx + 1] are
        // what is scanned AND what the stop rule below measures its faces from, so a query that sits a rounding away from a cell
        // border is at worst handed to the tail kernels (the fp64 cell_coord of round 3 cost ten double-precision instructions)
        const int cx = min(max((int)floorf((q.x - bp.org32x) * bp.invh32x), bx0), bx1 - 1);
        const int cy = by0 + ly, cz = bz0 + lz;
        const int ja = max(cx - 1, 0) - sx0, jb2 = min(cx + 2, dimx) - sx0;
        float best = kBigF, second = kBigF;
        uint32_t bestpos = 0xffffffffu;
        const v2f qxx = {q.x, q.x}, qyy = {q.y, q.y}, qzz = {q.z, q.z};
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int run = (lz + dz) * (BY + 2) + (ly + dy);
                const uint32_t fsb = s_lcs[run * kLcsPitch + ja], feb = s_lcs[run * kLcsPitch + jb2];
                // whole pairs from the one that holds the window's first record up to its end: a pair may bring one record of
                // the neighbouring cell of the same run (or the run's pad) along -- a real point of the searched cloud outside
                // the ring-1 cube: if it wins, the stop rule below rejects it; no per-candidate range test is needed
                // (the loop runs on LDS byte addresses of the x plane: the position of the best candidate is kept as its address)
                if constexpr (ABL != 0) {
                // (timing-only ablation builds keep the plain loop: reads, wait, arithmetic)
                for (uint32_t a = fsb & ~7u, ae = (ABL & 1) ? (fsb < feb ? a + 8u : a) : feb; a < ae; a += 8u) {
                    v2f px, py, pz;
                    asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:%4\n\tds_read_b64 %2, %3 offset:%5\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(px), "=&v"(py), "=&v"(pz)
                                 : "v"(a), "n"(4 * PL), "n"(8 * PL)
                                 : "memory");
                    const v2f dx = qxx - px, dy = qyy - py, dz = qzz - pz;
                    v2f dd = dx * dx;
                    dd = __builtin_elementwise_fma(dy, dy, dd);
                    dd = __builtin_elementwise_fma(dz, dz, dd);
                    const float lo = umin_f(dd.x, dd.y);
                    bestpos = __float_as_uint(lo) < __float_as_uint(best) ? a + 8u : bestpos;
                    second = __builtin_amdgcn_fmed3f(best, second, lo);
                    best = umin_f(best, lo);
                }
                } else {
                    // Software-pipelined: three ds_read_b64 off one address register (left to itself hipcc fuses two of them into
                    // a ds_read2st64_b64, which the LDS serves at 8 cycles per wave: MI355X_MICROARCH.md, LDS table), and the
                    // NEXT pair's reads are issued as soon as this pair's coordinates have been consumed by the three
                    // subtractions, so an LDS round trip runs under the other ten instructions of the pair (round 4: 72.6 ->
                    // 68.8 us at 1M + 1M points, 580 -> 544 at 8M; the reads and their wait in one statement -- round 3 -- exposed
                    // a full round trip per pair).  One pair beyond the window is read and dropped: it lies inside the planes
                    // (the budget leaves four positions behind the last run).  The destination registers are tied ("+v") from the
                    // first issue to the last wait, so the compiler never copies them while a read is in flight.
                    uint32_t a = fsb & ~7u;
                    const uint32_t ae = feb;
                    v2f px, py, pz, pr;
                    if (SELF)
                        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:%5\n\tds_read_b64 %2, %4 offset:%6\n\tds_read_b64 %3, %4 offset:%7"
                                     : "=v"(px), "=v"(py), "=v"(pz), "=v"(pr) : "v"(a), "n"(4 * PL), "n"(8 * PL), "n"(12 * PL) : "memory");
                    else
                        asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:%4\n\tds_read_b64 %2, %3 offset:%5"
                                     : "=v"(px), "=v"(py), "=v"(pz) : "v"(a), "n"(4 * PL), "n"(8 * PL) : "memory");
                    while (a < ae) {
                        if (SELF) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(px), "+v"(py), "+v"(pz), "+v"(pr) : : "memory");
                        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(px), "+v"(py), "+v"(pz) : : "memory");
                        // both candidates of the pair at once: v_pk_add / v_pk_mul / v_pk_fma_f32
                        v2f dx = qxx - px, dy = qyy - py, dz = qzz - pz;
                        // (self search: the query's own record is pushed out of the way -- decided HERE, while the row words are
                        // still the ones just waited for; a use of them behind the next issue makes the compiler park them in a
                        // copy that it refills from the registers of the reads in flight)
                        uint32_t k0 = 0u, k1 = 0u;
                        if (SELF) {
                            k0 = __float_as_int(pr.x) == qrow ? __float_as_uint(kBigF) : 0u;
                            k1 = __float_as_int(pr.y) == qrow ? __float_as_uint(kBigF) : 0u;
                            asm volatile("" : "+v"(k0), "+v"(k1));
                        }
                        asm volatile("" : "+v"(dx), "+v"(dy), "+v"(dz));        // the subtractions stay in front of the next reads
                        a += 8u;
                        if (SELF)
                            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:%5\n\tds_read_b64 %2, %4 offset:%6\n\tds_read_b64 %3, %4 offset:%7"
                                         : "+v"(px), "+v"(py), "+v"(pz), "+v"(pr) : "v"(a), "n"(4 * PL), "n"(8 * PL), "n"(12 * PL) : "memory");
                        else
                            asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:%4\n\tds_read_b64 %2, %3 offset:%5"
                                         : "+v"(px), "+v"(py), "+v"(pz) : "v"(a), "n"(4 * PL), "n"(8 * PL) : "memory");
                        v2f dd = dx * dx;
                        dd = __builtin_elementwise_fma(dy, dy, dd);
                        dd = __builtin_elementwise_fma(dz, dz, dd);
                        float d0 = dd.x, d1 = dd.y;
                        if (SELF) {                                              // (bit patterns of non-negative floats order like the floats)
                            d0 = __uint_as_float(max(__float_as_uint(d0), k0));
                            d1 = __uint_as_float(max(__float_as_uint(d1), k1));
                        }
                        // The nearer of the two enters (best, second best, address of the best pair); the farther one is checked
                        // once, after the scan, for the winning pair only: a pair's farther record can be second best without its
                        // nearer one being best or second best only as the winner's own partner.  5 VALU per pair (tracking both
                        // records: 7).  Squared distances are non-negative (or +inf): their order is the order of their bit
                        // patterns, and v_min_u32 needs no canonicalisation of the loop-carried operand (fminf does)
                        const float lo = umin_f(d0, d1);
                        bestpos = __float_as_uint(lo) < __float_as_uint(best) ? a : bestpos;      // (the pair's address + 8)
                        second = __builtin_amdgcn_fmed3f(best, second, lo);
                        best = umin_f(best, lo);
                    }
                    if (SELF) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(px), "+v"(py), "+v"(pz), "+v"(pr) : : "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(px), "+v"(py), "+v"(pz) : : "memory");   // the pair beyond the window
                }
            }
        }
        BRICK_STAMP(5);                                                // the nine scan loops (incl. query/bounds reads)
        // certification (bound derived in pccm_brute.hip: every point that can win or tie in fp64 has
        // d32 <= best (1 + 2^-20)^2 (1 + 2^-30) + 1e-36 for fp32-exact inputs) + the ring-1 stop rule.  In fp32:
        // best (1 + 2^-18) + 2e-36, rounded once, lies above that bound for every best.
        const float thr = __builtin_fmaf(best, 1.0f + 0x1.0p-18f, 2.0e-36f);
        bool settled = false;
        if (ABL & 4) {
            if (best < 0.0f) store_result(out, qrow, 0.0, 0.0, 0);     // never true: keeps the scan alive
            settled = true;
        } else if (bestpos != 0xffffffffu && best < 1.0e30f && second > thr) {          // (a pad record is no neighbour)
            // the winning pair's two records again: which one has d32 == best (the same arithmetic gives the same bits), and
            // is its partner out of the way?  (both at d32 == best: an exact tie, left to the tail kernels like every near tie)
            uint32_t f = (bestpos - 8u - lds0) >> 2;                // the pair's first record (bestpos: its address + 8)
            float partner;
            {
                const v2f px = *reinterpret_cast<const v2f *>(s_x + f), py = *reinterpret_cast<const v2f *>(s_y + f),
                          pz = *reinterpret_cast<const v2f *>(s_z + f);
                const v2f dx = qxx - px, dy = qyy - py, dz = qzz - pz;
                v2f dd = dx * dx;
                dd = __builtin_elementwise_fma(dy, dy, dd);
                dd = __builtin_elementwise_fma(dz, dz, dd);
                float d0 = dd.x, d1 = dd.y;
                if (SELF) {
                    d0 = (__float_as_int(s_r[f]) == qrow) ? kBigF : d0;
                    d1 = (__float_as_int(s_r[f + 1]) == qrow) ? kBigF : d1;
                }
                const bool first = d0 == best;
                partner = first ? d1 : d0;
                f += first ? 0u : 1u;
            }
            // the ring-1 stop rule, in fp32 and never more permissive than settled_by(face_bound(.., 1), d64): the true squared
            // distance is below best (1 + 2^-19) (the certification bound above), L is a lower bound of the face distance
            float L = face32(q.x, cx, dimx, bp.h32[0], bp.face_lo[0], bp.face_hi[0]);
            L = fminf(L, face32(q.y, cy, dimy, bp.h32[1], bp.face_lo[1], bp.face_hi[1]));
            L = fminf(L, face32(q.z, cz, dimz, bp.h32[2], bp.face_lo[2], bp.face_hi[2]));
            settled = partner > thr && L > 0.0f && best * 1.00001f < L * L * 0.99999f;
            const float wx = s_x[f], wy = s_y[f], wz = s_z[f];
            const int wrow = __float_as_int(s_r[f]);
            if (settled && defer) {
                store_result_rec(out, qrow, wx, wy, wz, wrow);              // the matched record as it lies in LDS: one 16-byte store
            } else if (settled) {
                const double rx = (double)wx, ry = (double)wy, rz = (double)wz;
                const double d64 = gdist64(qx, qy, qz, rx, ry, rz);
                double p = 0.0;
                if (fuse) {
                    double e0 = N32 ? (double)nraw.x : n0, e1 = N32 ? (double)nraw.y : n1, e2 = N32 ? (double)nraw.z : n2;
                    if (!fuse_row) {
                        load_normal(out, wrow, e0, e1, e2);
                    }
                    const double ex = __dsub_rn(qx, rx), ey = __dsub_rn(qy, ry), ez = __dsub_rn(qz, rz);
                    p = __dmul_rn(ex, e0);
                    p = __fma_rn(ey, e1, p);
                    p = __fma_rn(ez, e2, p);
                }
                store_result(out, qrow, d64, p, wrow);
            }
        }
        if (!settled) {
            const uint32_t pos = atomicAdd(&J.counters[1], 1u);
            reinterpret_cast<float4 *>(J.tail)[pos] = q;
        }
        if (hn) {                                                      // fetched only now: keeps the scan's registers free
            rn = query_row(qnext);
            qn = load_query(qnext, rn);
            gather_row_normal(__float_as_int(qn.w), fuse_row);
        }
        qi = qnext;
        have = hn;
        if (STAMP) {
            BRICK_STAMP(q_iter == 0 ? 3 : 4);                          // first query of the lane / the leftovers
            ++q_iter;
        }
    }
    if (STAMP && lane == 0) {
        // one slot per wave (same-address atomics from 33 000 waves serialise and keep finished waves in their slots: round 2's
        // flush made the stamped kernel several times slower than the one it was meant to describe)
        unsigned long long *slot = bp.stamps + ((size_t)blockIdx.x * (NT >> 6) + w) * 8;
#pragma unroll
        for (int k = 0; k < 6; ++k) slot[k] = t_sum[k];
        slot[6] = t_start;
        slot[7] = __builtin_amdgcn_s_memtime();
    }
}

// The launch: one workgroup per brick.  (A resident set of workgroups walking the brick list was tried: the loop costs
// 30 registers -- loop-invariant geometry kept live, SGPRs spilled into VGPRs -- i.e. a workgroup per CU, and the dispatch of
// 3900 workgroups is not what the kernel waits for: 148 us against 103.)
// __launch_bounds__(.., 8): 64 VGPRs and <= 80 SGPRs, no spills -- 32 waves (four 512-thread workgroups) resident per CU; the
// kernel lives on occupancy (two workgroups per CU: 148 us, three: 112 us, four: 104 us at 1M points).  With 81-96 SGPRs
// the hardware admits one wave per SIMD less than the compiler's occupancy figure says (MI355X_MICROARCH.md, "Residency
// and cooperative launch").  PCCM_BRICK_V64=0 runs the build without the cap (72 VGPRs, 93 SGPRs) for A/B.
template <bool SELF, int BY, int BZ, int PL, bool STAMP = false, int ABL = 0, bool N32 = true>
__global__ __launch_bounds__(1024, 8) void k_brick_query(QueryJobs jobs, GridGeom g, BrickParams bp)
{
    brick_body<SELF, BY, BZ, STAMP, ABL, N32, PL>(jobs, g, bp, blockIdx.x);
}

template <int BY, int BZ, int PL>
__global__ __launch_bounds__(1024) void k_brick_query_free(QueryJobs jobs, GridGeom g, BrickParams bp)
{
    brick_body<false, BY, BZ, false, 0, false, PL>(jobs, g, bp, blockIdx.x);
}

// brick shape: PCCM_BRICK="BY,BZ[,NT]" picks one of the compiled shapes and optionally forces the workgroup size (A/B runs)
struct BrickShape {
    int by, bz, nt;
};

static BrickShape brick_shape()
{
    static const BrickShape s = [] {
        BrickShape v = {4, 2, 0};
        const char *e = getenv("PCCM_BRICK");
        if (e) {
            int a = 0, b = 0, c = 0;
            const int got = sscanf(e, "%d,%d,%d", &a, &b, &c);
            if (got >= 2) v = {a, b, got == 3 ? c : 0};
        }
        return v;
    }();
    return s;
}

template <int BY, int BZ, int PL>
static void launch_plane(pccm_ctx *ctx, const QueryJobs &jobs, const GridGeom &g, bool self, BrickParams bp, double density_q, int force_nt)
{
    constexpr int kStatic = (BY + 2) * (BZ + 2) * (kLcsPitch * 2 + 12) + (BY * BZ) * 8 + 64;
    static_assert(4 * PL * (int)sizeof(float) + kStatic <= 64 * 1024, "static + dynamic LDS of one workgroup stay under 64 KB");
    // workgroup size: the queries a brick is expected to hold plus 2.5 sigma (Poisson), in whole waves -- but never
    // so large that the workgroups the LDS admits per CU exceed the CU's 32 wave slots (occupancy is what hides the
    // kernel's memory round trips: a few leftover queries per brick cost less than a workgroup per CU)
    const double eq = (double)(BY * BZ) * bp.bx * density_q;
    int nt = ((int)(eq + 2.5 * sqrt(eq)) + 63) / 64 * 64;
    const size_t lds = (size_t)4 * PL * sizeof(float);
    const size_t lds_wg = lds + (size_t)((BY + 2) * (BZ + 2)) * (kLcsPitch * 2 + 16) + 256;
    const int wgs_per_cu = (int)((size_t)160 * 1024 / lds_wg) > 0 ? (int)((size_t)160 * 1024 / lds_wg) : 1;
    const int nt_cap = 64 * (32 / (wgs_per_cu > 16 ? 16 : wgs_per_cu));
    if (nt > nt_cap) nt = nt_cap;
    if (force_nt > 0) nt = force_nt / 64 * 64;
    if (nt < 128) nt = 128;                             // waves 0 and 1 do the bookkeeping of step 1
    if (nt > 1024) nt = 1024;
    bp.total = (uint32_t)(bp.per_job * jobs.njobs);
    dim3 grid(bp.total);
    bp.stamps = nullptr;
#ifdef PCCM_DIAG   // diagnostic builds only (make DIAG=1): in-kernel stamps and the timing-only ablations never ship
    static const bool stamp = [] { const char *e = getenv("PCCM_BRICK_STAMP"); return e && e[0] == '1'; }();
    if (stamp && !self && !ctx->capturing) {
        // diagnostic build: phase shares of the wave cycles, printed per launch (never part of a timed or captured run)
        static unsigned long long *dev = nullptr;
        static size_t dev_slots = 0;
        const size_t slots = (size_t)bp.total * (size_t)(nt >> 6);
        if (slots > dev_slots) {
            if (dev) (void)hipFree(dev);
            dev = nullptr;
            if (hipMalloc((void **)&dev, slots * 8 * sizeof(unsigned long long)) == hipSuccess) dev_slots = slots;
            else dev_slots = 0;
        }
        if (dev) {
            (void)hipMemsetAsync(dev, 0, slots * 8 * sizeof(unsigned long long), ctx->stream);
            bp.stamps = dev;
            hipLaunchKernelGGL((k_brick_query<false, BY, BZ, PL, true>), grid, dim3(nt), lds, ctx->stream, jobs, g, bp);
            std::vector<unsigned long long> h(slots * 8);
            (void)hipMemcpyAsync(h.data(), dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
            (void)hipStreamSynchronize(ctx->stream);
            double sum[6] = {0, 0, 0, 0, 0, 0}, life = 0.0;
            unsigned long long first = ~0ull, last = 0ull;
            size_t waves = 0;
            for (size_t i = 0; i < slots; ++i) {
                const unsigned long long *r = &h[i * 8];
                if (r[7] == 0) continue;                   // a wave of an empty brick
                ++waves;
                for (int k = 0; k < 6; ++k) sum[k] += (double)r[k];
                life += (double)(r[7] - r[6]);
                first = r[6] < first ? r[6] : first;
                last = r[7] > last ? r[7] : last;
            }
            const double w = waves ? (double)waves : 1.0;
            fprintf(stderr, "[pccm] brick stamps (s_memtime ticks per wave, %zu waves, nt %d, cap %d): lifetime %.0f, first start -> last end %.0f | "
                            "bounds+barrier %.0f | staging %.0f | barrier %.0f | scan %.0f | epilogue %.0f | leftovers %.0f\n",
                    waves, nt, bp.cap, life / w, (double)(last - first), sum[0] / w, sum[1] / w, sum[2] / w, sum[5] / w, sum[3] / w, sum[4] / w);
            return;
        }
    }
    static const int ablate = [] { const char *e = getenv("PCCM_BRICK_ABLATE"); return e ? atoi(e) : 0; }();
    if (ablate && !self && BY == 4 && BZ == 2) {        // timing-only builds: results are wrong by construction
        if (ablate == 1) hipLaunchKernelGGL((k_brick_query<false, 4, 2, PL, false, 1>), grid, dim3(nt), lds, ctx->stream, jobs, g, bp);
        else if (ablate == 2) hipLaunchKernelGGL((k_brick_query<false, 4, 2, PL, false, 2>), grid, dim3(nt), lds, ctx->stream, jobs, g, bp);
        else if (ablate == 4) hipLaunchKernelGGL((k_brick_query<false, 4, 2, PL, false, 4>), grid, dim3(nt), lds, ctx->stream, jobs, g, bp);
        else if (ablate == 8) hipLaunchKernelGGL((k_brick_query<false, 4, 2, PL, false, 8>), grid, dim3(nt), lds, ctx->stream, jobs, g, bp);
        else if (ablate == 16) hipLaunchKernelGGL((k_brick_query<false, 4, 2, PL, false, 16>), grid, dim3(nt), lds, ctx->stream, jobs, g, bp);
        else if (ablate == 32) hipLaunchKernelGGL((k_brick_query<false, 4, 2, PL, false, 32>), grid, dim3(nt), lds, ctx->stream, jobs, g, bp);
        else hipLaunchKernelGGL((k_brick_query<false, 4, 2, PL, false, 15>), grid, dim3(nt), lds, ctx->stream, jobs, g, bp);
        return;
    }
#endif
    static const bool v_free = [] { const char *e = PCCM_DIAG_ENV("PCCM_BRICK_V64"); return e && e[0] == '0'; }();
    if (v_free && !self) {     // A/B: no register cap
        hipLaunchKernelGGL((k_brick_query_free<BY, BZ, PL>), grid, dim3(nt), lds, ctx->stream, jobs, g, bp);
        return;
    }
    // fp32-exact normals (file normals) or none fused: the kernel that carries the 16-byte word; fp64 rows (estimated normals) else
    bool n32 = true;
    for (int k = 0; k < jobs.njobs; ++k) n32 = n32 && (jobs.j[k].out.nrm == nullptr || jobs.j[k].out.nrm32 != nullptr);
    if (self) hipLaunchKernelGGL((k_brick_query<true, BY, BZ, PL>), grid, dim3(nt), lds, ctx->stream, jobs, g, bp);     // (no normals in a self search)
    else if (n32) hipLaunchKernelGGL((k_brick_query<false, BY, BZ, PL, false, 0, true>), grid, dim3(nt), lds, ctx->stream, jobs, g, bp);
    else hipLaunchKernelGGL((k_brick_query<false, BY, BZ, PL, false, 0, false>), grid, dim3(nt), lds, ctx->stream, jobs, g, bp);
}

template <int BY, int BZ>
static void launch_shape(pccm_ctx *ctx, const QueryJobs &jobs, const GridGeom &g, bool self, BrickParams bp, double density_s,
                         double density_q, int force_nt)
{
    bp.nby = (g.dim[1] + BY - 1) / BY;
    bp.nbz = (g.dim[2] + BZ - 1) / BZ;
    bp.per_job = (int64_t)bp.nbx * bp.nby * bp.nbz;
    // LDS budget: the expected number of staged records (runs x cells x points per cell) plus a quarter -- occupancy
    // is set by it.  Bricks that hold more (clumped data) hand their queries to the general kernels.
    const double expect = (double)((BY + 2) * (BZ + 2)) * (bp.bx + 2) * density_s;
    const double margin = fmin(0.25 * expect, 8.0 * sqrt(expect));      // a quarter, or eight sigma of a Poisson count
    int cap = (int)(expect + margin) + 64;
    static const int cap_env = [] { const char *e = getenv("PCCM_BRICK_CAP"); return e ? atoi(e) : 0; }();
    if (cap_env > 0) cap = cap_env;
    if (cap < 256) cap = 256;
    // the kernels are compiled for two plane sizes; the budget is the whole plane (it is allocated either way)
    if (cap + 4 <= kPlaneSmall) {
        bp.cap = kPlaneSmall - 4;
        launch_plane<BY, BZ, kPlaneSmall>(ctx, jobs, g, self, bp, density_q, force_nt);
    } else if (cap + 4 <= kPlaneMid) {
        bp.cap = kPlaneMid - 4;
        launch_plane<BY, BZ, kPlaneMid>(ctx, jobs, g, self, bp, density_q, force_nt);
    } else {
        bp.cap = kPlaneLarge - 4;
        launch_plane<BY, BZ, kPlaneLarge>(ctx, jobs, g, self, bp, density_q, force_nt);
    }
}

int launch_brick_query(pccm_ctx *ctx, const QueryJobs &jobs, const GridGeom &g, bool self)
{
    BrickParams bp;
    static const int bx_env = [] { const char *e = PCCM_DIAG_ENV("PCCM_BRICK_BX"); return e ? atoi(e) : 0; }();
    const int bx_max = bx_env > 0 ? bx_env : 48;          // bricks of <= 48 cells: ~65 records per staged run at 1.4 points per cell
    const int nbx = (g.dim[0] + bx_max - 1) / bx_max;
    bp.bx = (g.dim[0] + nbx - 1) / nbx;
    if (bp.bx > kBXMax) bp.bx = kBXMax;
    bp.nbx = (g.dim[0] + bp.bx - 1) / bp.bx;
    bp.nby = bp.nbz = 0;
    bp.per_job = 0;
    bp.total = 0;
    bp.cap = 0;
    bp.stamps = nullptr;
    // The stop rule's faces in fp32.  A face lies at F = org + k h; the kernel forms fmaf((float)c, h32, f32) with f32 the rounded
    // org - h (or org + 2 h) and subtracts the query coordinate.  Against the exact F - q that is off by at most
    //   |c| |h32 - h| + |f32 - f| + the fma's rounding + the subtraction's rounding  <=  4 x 2 M 2^-24,
    // M = the largest coordinate magnitude of the grid's box (+ 2 h).  That error joins GridGeom::slack and is taken off every
    // face distance, so the fp32 bound never exceeds face_bound()'s.
    for (int a = 0; a < 3; ++a) {
        const double top = g.org[a] + (double)g.dim[a] * g.h[a];
        const double M = fmax(fabs(g.org[a]), fabs(top)) + 2.0 * g.h[a];
        const double slack = g.slack[a] + 8.0 * M * 0x1.0p-24;
        bp.h32[a] = (float)g.h[a];
        if (a == 0) {
            bp.org32x = (float)g.org[0];
            bp.invh32x = (float)g.inv_h[0];
        }
        bp.face_lo[a] = (float)(g.org[a] - g.h[a] + slack);
        bp.face_hi[a] = (float)(g.org[a] + 2.0 * g.h[a] - slack);
    }
    const Grid &gr = ctx->grid;
    const int64_t nmax = gr.n[0] > gr.n[1] ? gr.n[0] : gr.n[1];
    const double density = gr.ncells > 0 ? (double)nmax / (double)gr.ncells : 1.5;      // points per cell of the denser cloud
    // ... and of the query list (a shard holds a fraction of its cloud's rows)
    int64_t nqmax = 0;
    for (int k = 0; k < jobs.njobs; ++k) nqmax = jobs.j[k].nq > nqmax ? jobs.j[k].nq : nqmax;
    const double density_q = gr.ncells > 0 ? (double)nqmax / (double)gr.ncells : 1.5;
    BrickShape sh = brick_shape();
    // a shard that holds at most half of its cloud's rows has few queries per brick for the records a brick stages: bricks of
    // 4 x 4 rows stage 2.25 records per owned one instead of 3 (per rank at 8 ranks: 44 -> 34 us at 1M points, 267 -> 200 at 8M);
    // for whole clouds the smaller brick's occupancy wins (89 against 100 us)
    static const bool shape_forced = getenv("PCCM_BRICK") != nullptr;
    if (!shape_forced && density_q <= 0.55 * density) sh = {4, 4, 0};
    if (sh.by == 2 && sh.bz == 2) launch_shape<2, 2>(ctx, jobs, g, self, bp, density, density_q, sh.nt);
    else if (sh.by == 4 && sh.bz == 4) launch_shape<4, 4>(ctx, jobs, g, self, bp, density, density_q, sh.nt);
    else launch_shape<4, 2>(ctx, jobs, g, self, bp, density, density_q, sh.nt);
    PCCM_HIP(hipGetLastError());
    return PCCM_OK;
}

}  // namespace pccm
