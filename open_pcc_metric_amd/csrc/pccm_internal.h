// Internal declarations shared by the translation units of libpccm.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <vector>

#include "pccm.h"

// A/B switches that no test of the shipped library uses (brick length, register cap, build bins, cells per point, ...) exist in
// diagnostic builds only (make DIAG=1): the product reads the environment for the switches tests/test_gpu_ab_paths.py exercises
// and for nothing else.
#ifdef PCCM_DIAG
#define PCCM_DIAG_ENV(name) getenv(name)
#else
#define PCCM_DIAG_ENV(name) (static_cast<const char *>(nullptr))
#endif

namespace pccm {

// ---- geometry of the brute-force scan (K1) ------------------------------------------
constexpr int kScanThreads = 256;   // 4 waves of 64
constexpr int kScanTile = 1024;     // search points staged in LDS per buffer
constexpr int kTileVec = kScanTile / 4 * 3;   // float4 vectors per tile in the quad layout (12 KB)
constexpr int kGranule = 64;        // winner tracking granularity = one wave-wide fp64 rescan
constexpr float kBig32 = 3.0e38f;   // "no candidate" distance (finite: no inf arithmetic in the scan)
constexpr float kPadCoord = 1.0e18f;  // coordinates of padding points: d2 ~ 3e36, never wins
constexpr double kMaxAbsCoord = 1.0e15;

// ---- reduction geometry: NumPy's pairwise sum (numpy/_core/src/umath/loops_utils.h.src) --
constexpr int kLeaf = 128;          // PW_BLOCKSIZE
constexpr int kChunk = 8192;        // NumPy's default ufunc buffer size in elements

struct Cloud {
    int64_t n = 0;
    int64_t n_pad = 0;          // multiple of kScanTile
    float4 *xyz32 = nullptr;    // [n_pad/4][3] quads: x0..3 | y0..3 | z0..3; padding rows = kPadCoord
    double *xyz64 = nullptr;    // [n][3]
    float4 *xyz32r = nullptr;   // [n] {x, y, z, 0}: the fp32 coordinates once more, one aligned 16-byte word per ROW (what the reductions
                                // of matched-record results read next to the record and the normal: three wide loads per row)
    size_t cap32r = 0;
    double *nrm64 = nullptr;    // [n_nrm][3]
    float4 *nrm32 = nullptr;    // [n_nrm] {nx, ny, nz, -}: the same normals in one aligned 16-byte word each, when nrm_exact32
    bool nrm_exact32 = false;   // every component survives fp64 -> fp32 -> fp64 (file normals usually do; estimated ones do not)
    int64_t n_nrm = 0;
    // pccm_set_normals_deferred: the normals are announced (n_nrm, buffers) but still lie in the caller's host array; they cross
    // PCIe when somebody needs them (normals_ready) or at pccm_flush_uploads -- behind the searches, which never read them when
    // results are matched records (NNOut::layout 1)
    const void *nrm_host = nullptr;
    int nrm_host_dtype = 0;
    bool nrm_deferred = false;
    double *rgb64 = nullptr;    // [n_rgb][3] colours as the caller gave them (RGB in [0, 1])
    int64_t n_rgb = 0;
    // colours that are k / 255.0 for bytes k -- what every file holds -- also live as one packed word per row (r | g << 8 |
    // b << 16): the colour kernels gather 4 bytes per row from a table the L2 holds instead of 24 from one it does not
    uint32_t *rgb8 = nullptr;
    bool rgb8_valid = false;
    size_t cap_rgb8 = 0;
    // allocations outlive their content (n / n_nrm / n_rgb say what is there): a context that serves one pair after
    // the other -- the engine pool of _native.py -- does not pay hipFree + hipMalloc per cloud
    // the same points in a spatially coherent order (fp32-exact clouds): Rec32 {x, y, z, original row} sorted along a Z-order
    // curve over the cloud's own bounding box, made once per cloud -- by the first pccm_drop_caches that follows a search, i.e. when
    // the caller shows that the resident clouds will be searched again (a one-shot pair never pays for it).  The per-step grid build reads it instead
    // of xyz32: rows that are neighbours in memory are neighbours in space, so the counting sort's scattered stores fall into a
    // few bins per tile and merge into whole lines (WRITE_SIZE 2x -> ~1x the records).  Results never depend on it.
    void *sp = nullptr;
    size_t cap_sp = 0;
    bool sp_valid = false;
    bool sp_tried = false;      // pccm_drop_caches has already made (or found no use for) the spatial order of this cloud
    size_t cap32 = 0, cap64 = 0, cap_nrm = 0, cap_nrm32 = 0, cap_rgb = 0;
    bool exact32 = true;        // every coordinate survives the fp64 -> fp32 -> fp64 round trip
    bool all_int = false;       // ... and is an integer (voxelised content: exact ties are the rule)
    double maxabs = 0.0;
    double bb_min[3] = {0, 0, 0}, bb_max[3] = {0, 0, 0};   // bounding box (fp64 coordinates)
    uint64_t version = 0;       // bumped by pccm_set_cloud (grid caches key on it)
    double solo_scale = 1.0;    // cell-edge factor of a grid over this cloud alone (grid_ensure_solo), decided for ...
    uint64_t solo_scale_version = ~0ull;   // ... this version of the cloud
};

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct NNResult {
    bool valid = false;
    int64_t begin = 0, end = 0; // shard rows of the iterating cloud
    int32_t *idx = nullptr;     // [end-begin]   plain columns: written by the brute-force engine, or unpacked from `rec`
    double *d2 = nullptr;       // [end-begin]   on demand (ensure_plain)
    int64_t cap = 0;
    DevBuf rec;                 // [end-begin] 32-byte result records {d2, projection, row, -} (grid engine; NNOut in pccm_grid.h)
    bool rec_valid = false;     // `rec` holds the last run's results
    bool plain_valid = false;   // idx / d2 hold them
    bool plain_d2_valid = false;   // d2 alone does (unpacked from records that carry no row)
    int fused_mode = -1;        // normal mode of the projection stored in `rec`, -1: not fused
    int rec_stride = 4;         // doubles per record: 4 = with the matched row, 2 = {d2, projection} only (pccm_nn_want_idx off)
    int rec_layout = 0;         // 0: {d2, projection[, row, -]}; 1 (stride 2): the matched record {rx, ry, rz, row}: distance and
                                //    row-indexed projection are formed by the reduction that reads the records (NNOut::layout)
    bool no_rows = false;       // layout 1 records whose row word is void (voxel-brick search): good for distances only
    int64_t stats[3] = {0, 0, 0};
    uint32_t *nflag_dev = nullptr;  // device counters of the last run: [0] fallback queries, [1] grid tail length
    DevBuf flagged, flag_thr;       // queries handed to the exact rescan (k2b_fallback) and their thresholds
    DevBuf tail;                    // grid engine: unsettled ring-1 queries
};

// Uniform grid over one cloud (grid engine): cells in x-fastest order, points counting-sorted by cell.
struct GridRec {          // 32 B: fp64 position + original row
    double x, y, z;
    int32_t idx, pad;
};

struct PairSignature {           // what decide_scale looks at to tell "a pair like the last one"
    int64_t n[2] = {0, 0};
    int flags[2] = {0, 0};         // exact32 | all_int << 1
    double lo[2][3] = {{0, 0, 0}, {0, 0, 0}}, hi[2][3] = {{0, 0, 0}, {0, 0, 0}};
    bool resembles(const PairSignature &o) const
    {
        for (int k = 0; k < 2; ++k) {
            if (flags[k] != o.flags[k]) return false;
            const double dn = (double)(n[k] - o.n[k]);
            if (dn > 0.02 * (double)o.n[k] || -dn > 0.02 * (double)o.n[k]) return false;
            for (int a = 0; a < 3; ++a) {
                const double ext = o.hi[k][a] - o.lo[k][a], tol = 0.02 * ext;
                const double d0 = lo[k][a] - o.lo[k][a], d1 = hi[k][a] - o.hi[k][a];
                if (!(d0 <= tol && -d0 <= tol && d1 <= tol && -d1 <= tol)) return false;
            }
        }
        return true;
    }
};

struct Grid {                    // one geometry, both clouds (grid engine)
    uint64_t key = 0;              // derived from both Cloud::version values (0 = none)
    uint64_t scale_key = 0;        // clouds the cell-edge scale below was decided for
    double scale = 1.0;            // shrink factor of the volume-rule cell edge (occupancy-adaptive)
    bool boxed = false;            // the grid covers box_lo..box_hi (outliers trimmed) instead of the bounding box
    double box_lo[3] = {0, 0, 0}, box_hi[3] = {0, 0, 0};
    bool hostile = false;          // even so the cells are too crowded: PCCM_ENGINE_AUTO uses the brute engine
    bool coop = true;              // cooperative ring-1 kernel (well-filled x-rows) or the per-thread search (surfaces, lattices)
    double sb = 0.0;               // size-biased points per cell the decision saw
    PairSignature sig;             // of the pair the decisions were last taken (or inherited) for
    uint64_t iso_key = 0;          // pair the isolation count below was taken for
    int64_t isolated[2] = {0, 0};  // points of cloud k with nothing of the other cloud within kMaxRing cells
    bool rec32 = false;            // records are Rec32 (both clouds fp32-exact) instead of GridRec
    int built = 0;                 // bit k: cloud k's records and cell starts are built (a rank that searches one direction
                                   // of a sharded pair builds only the cloud it searches)
    int dim[3] = {1, 1, 1};
    double org[3] = {0, 0, 0};
    double h[3] = {1, 1, 1}, inv_h[3] = {1, 1, 1};   // cell edge per axis
    int64_t ncells = 0, n[2] = {0, 0};
    DevBuf cell_start;             // uint32 [2][ncells + 1]: positions in recs
    bool lattice = false;          // voxelised pair on the per-thread path: pccm_lattice.hip searches it, with the bitmap below
    DevBuf occ;                    // uint32 [2][ncells / 32 + 2]: one bit per cell, set when the cell holds a record
    DevBuf recs;                   // GridRec or Rec32 [n[0] + n[1]]: cloud 0's records, then cloud 1's
    bool vox = false;              // voxel-brick flavour (pccm_vox.hip): cells of 8^3 voxels, searched through the bricks below;
                                   // built for distance-only requests on voxelised pairs, rebuilt with the usual cells otherwise
    DevBuf vbricks;                // uint32 [n[0] + n[1]][32]: occupancy + duplicate brick at the index of a cell's first record
    DevBuf vlist;                  // uint32 [n[0] + n[1]]: occupied cells of cloud 0, then (from n[0]) of cloud 1
    DevBuf vcount;                 // uint32 [2]
    bool vox_rows = false;         // ... and the bricks come with the rows' table below (searches that return the matched row)
    uint64_t vox_rows_pair = 0;    // pair key for which matched rows have been asked for (its later builds keep the table)
    DevBuf vminrow;                // int32 [n[0] + n[1]]: smallest row of every occupied voxel, at the cell's first record + the voxel's
                                   // rank among the set bits of the cell's brick
};

struct ReduceSlot {            // one enqueued reduction (pccm_reduce_prefetch / pccm_reduce)
    bool pending = false;
    int dir = 0, metric = 0, mode = 0;
    uint64_t gen = 0;          // nn generation of `dir` it was computed from
    int64_t n_iter = 0, begin = 0, end = 0, nunits = 0, nblocks = 0, t0 = 0, tail_n = 0;
    bool has_units = false;    // per-leaf results were written (needed by pccm_reduce's exchange vector)
    DevBuf val;
    double *host = nullptr;    // pinned: [3][nunits] leaf sums/min/max | [3][nblocks] half-chunk trees | tail_n raw values
    size_t host_cap = 0;
    hipEvent_t ev = nullptr;
    hipEvent_t wait_ev = nullptr;   // what says "this slot's numbers are on the host": the context's batch event (one record serves
                                    // every slot of a call / of a graph replay; waiting on a later record of it only waits longer)
};

struct ProfSpan {
    hipEvent_t a, b;
    int cls;
};

struct GraphOp {               // host-side effect of one captured call, replayed by pccm_graph_launch
    int kind = 0;              // 0 drop_caches, 1 nn(dir), 2 reduce_prefetch(slot)
    int dir = 0, slot = -1;
    bool rec_valid = false, plain_valid = false;   // kind 1: where the direction's results live once the graph has run
    int fused_mode = -1, rec_stride = 4, rec_layout = 0;
    bool no_rows = false;
    ReduceSlot snap;           // kind 2: the slot's bookkeeping at capture time (pointers are not owned)
};

struct GraphRec {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<GraphOp> ops;
    uint64_t epoch = 0;        // pccm_ctx::epoch it was captured under
    bool valid = false;
};

}  // namespace pccm

struct pccm_ctx {
    // one context = one caller at a time: every entry point holds this for its whole duration (recursive: entry points
    // call each other), so threads that share a context by mistake are serialised instead of corrupting it
    std::recursive_mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t copy_stream = nullptr;     // deferred uploads (pccm_set_normals_deferred): they run beside the main stream's kernels
    pccm::DevBuf staging2;                 // ... through a staging buffer of their own
    // Large transfers between the CALLER's arrays and the device can go through pinned buffers of the context's own
    // (pccm_set_io_staged; off by default): the runtime otherwise pins the caller's pages itself and keep the mapping cached, and when the caller later
    // frees such an array the driver evicts and restores every queue of the process -- 13-27 ms during which a running kernel
    // stands still (measured: DESIGN.md section 4).  [0]: uploads on the main stream, [1]: on the copy stream, [2]: downloads.
    void *pin[3] = {nullptr, nullptr, nullptr};      // (at most 64 MB each: larger transfers reuse the buffer window by window)
    size_t pin_cap[3] = {0, 0, 0};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};      // the last upload's copies out of pin[0] / pin[1] have been issued up to here
    bool pin_ev_set[2] = {false, false};
    bool io_staged = false;
    pccm::Cloud cloud[2];
    // query-axis shard per direction (pccm_set_shard / pccm_set_shard_dir): this context owns the rows shard_of(n, rank,
    // world) of the iterating cloud; world 0 = none of them (another group of ranks searches that direction)
    int shard_rank[3] = {0, 0, 0}, shard_world[3] = {1, 1, 1};
    bool sharded() const { return shard_world[0] != 1 || shard_world[1] != 1 || shard_world[2] != 1; }
    pccm::NNResult nn[3];
    // scratch
    pccm::DevBuf part_b1, part_g, part_b2, val, stats, staging, counters;
    pccm::DevBuf rescan_part;             // k2b_fallback's split regime: partial minima per (query, workgroup)
    pccm::DevBuf tail_sync;               // k_grid_tail: retired-entry counts + ticket, for the normal and the self pass
    bool tail_sync_clean = false;
    pccm::DevBuf color_cols, color_idx;   // colour pass: squares as three columns / caller-supplied neighbour rows
    pccm::DevBuf colsum_scratch;          // ... the column sums' chunk guesses and sub-chunk totals (pccm_color.hip)
    bool colsum_configured = false;       // k_colsum_chain's LDS opt-in was set on this context's device
    // pccm_color_reduce works on BOTH directions when it can (the same launches serve two column triples; the walk behind
    // NumPy's summation order is one wave's latency however many run side by side) and keeps the other direction's answer
    // here until that direction is asked for -- or a search, new colours or another scheme make it stale
    struct ColorMemo {
        bool valid = false, range_bad = false;
        int dir = 0, scheme = 0;
        double scale = 0.0, sum[3] = {0, 0, 0}, max[3] = {0, 0, 0};
        uint64_t gen = 0, rgb_gen = 0;
    } color_memo;
    uint64_t rgb_gen = 1;                 // bumped by every colour upload
    pccm::Grid grid;
    pccm::DevBuf g_cell_of, g_rank, g_hist, g_blocksum, g_qrecs;   // grid-engine scratch (g_qrecs: cell-sorted shard rows)
    pccm::DevBuf g_bins, g_tmp;            // grid build: per-tile bin histogram + scan state; bin-partitioned records
    hipEvent_t batch_ev = nullptr;   // recorded once behind every batch of reductions (ReduceSlot::wait_ev)
    // device error word (pinned host memory the kernels can write): a kernel that meets a state it cannot be in -- a cell start
    // that contradicts the occupancy brick (pccm_vox.hip), a tail wait that ran out (k_grid_tail) -- sets a bit instead of
    // answering wrongly in silence; every call that hands results to the caller checks it behind its wait (check_device_errors)
    uint32_t *host_err = nullptr;
    bool bins_clean = false;   // the build's bin cursors (head of g_bins) are zero on the stream
    int want_idx = 1;                      // pccm_nn_want_idx: searches store the matched row with every result
    int fuse_mode[3] = {-1, -1, -1};       // pccm_nn_fuse: normal mode of the D2 projection fused into the search, per direction
    pccm::ReduceSlot slots[8];
    uint64_t nn_gen[3] = {1, 1, 1};
    // hipGraph capture of a step (pccm_graph_*): epoch changes whenever inputs, shard or any device
    // buffer a captured kernel may reference changes, which invalidates every recorded graph
    uint64_t epoch = 1;
    bool capturing = false, capture_failed = false;
    std::vector<pccm::GraphOp> cap_ops;
    std::vector<pccm::GraphRec> graphs;
    // profiling
    bool prof_on = false;
    std::vector<pccm::ProfSpan> spans;
    std::vector<hipEvent_t> event_pool;
    double prof_ms[PCCM_K_COUNT] = {0};
    int64_t prof_n[PCCM_K_COUNT] = {0};
};

namespace pccm {

// error plumbing ---------------------------------------------------------------------
int fail(int code, const char *fmt, ...);
#define PCCM_HIP(expr)                                                                        \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return ::pccm::fail(_e == hipErrorOutOfMemory ? PCCM_E_OOM : PCCM_E_HIP, "%s: %s", \
                                #expr, hipGetErrorString(_e));                                \
    } while (0)

int ensure(pccm_ctx *ctx, DevBuf &b, size_t bytes);
int grow(void **p, size_t &cap, size_t bytes);      // (re)allocate *p to hold `bytes`; keeps a buffer that is large enough

struct ProfScope {   // records a HIP-event pair around a launch group when profiling is on
    pccm_ctx *ctx;
    int cls;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(pccm_ctx *c, int k);
    ~ProfScope();
};

// kernel launchers (each returns PCCM_OK or an error) -----------------------------------
int launch_ingest_points(pccm_ctx *ctx, const void *src, int dtype, int64_t n, int64_t n_pad, float4 *x32,
                         double *x64, float4 *x32r, unsigned long long *stats /*[3] device*/);
int launch_ingest_normals(pccm_ctx *ctx, const void *src, int dtype, int64_t n, double *out, float *out32,
                          unsigned long long *stats);

// brute-force engine: fills res.idx / res.d2 for rows [res.begin, res.end) of `it` searched in `se`
int nn_brute(pccm_ctx *ctx, const Cloud &it, const Cloud &se, bool self, NNResult &res);
// grid engine
int nn_grid(pccm_ctx *ctx, int ndirs, const int *dirs, int force_idx = 0);   // force_idx: records carry the matched row whatever pccm_nn_want_idx says
void grid_release(pccm_ctx *ctx);
void grid_invalidate(pccm_ctx *ctx);
int grid_ensure(pccm_ctx *ctx, bool need64 = false, int need_mask = 3);
int grid_ensure_solo(pccm_ctx *ctx, int which);    // GridRec grid over cloud `which` alone, cells sized for it (normal estimation)   // need64: GridRec records wanted (pccm_normals.hip reads them)
int spatial_order(pccm_ctx *ctx, Cloud &c);      // fills Cloud::sp (ingest; no-op for clouds that are not fp32-exact)
int grid_decide(pccm_ctx *ctx, bool *hostile);   // geometry decision for the current pair (cached per pair)
int grid_prefers_brute(pccm_ctx *ctx, bool *yes); // builds the grid if needed; isolation verdict (cached per pair)
int estimate_normals(pccm_ctx *ctx, int which, int k);
int tie_exposure(pccm_ctx *ctx, int dir, const Cloud &it, const Cloud &se, const NNResult &res, int normal_mode, double out[8]);
int check_device_errors(pccm_ctx *ctx);
int normals_ready(pccm_ctx *ctx, Cloud &c);       // uploads normals announced by pccm_set_normals_deferred (no-op otherwise)           // PCCM_E_STATE when a kernel raised the context's device error word
// exact rescan of the flagged queries of njobs <= 2 results (k2b_fallback)
int launch_fallback(pccm_ctx *ctx, int njobs, const Cloud *const *its, const Cloud *const *ses, NNResult *const *ress, bool self);

constexpr int kSplitMax = 32;   // flagged queries up to which the rescan splits the cloud instead of the list
struct RescanJob {              // flagged queries of one result (k2b_fallback)
    const float *q32, *r32;     // fp32 quad layouts of the iterating / searched cloud
    const double *q64, *r64;
    int64_t q_begin, nr;
    const int32_t *flagged;     // rows relative to q_begin
    const float *flag_thr;      // fp32 filter threshold of each
    const uint32_t *nflag;      // list length (device)
    int32_t *idx_out;           // plain outputs (brute-force engine) ...
    double *d2_out;
    double *rec_out;            // ... or result records with the projection fused (grid engine), when non-null
    int rec_stride;             // doubles per record (NNResult::rec_stride)
    int rec_layout;             // NNResult::rec_layout
    const double *nrm;          // normals for the fused projection, or null
    int normal_mode;
    double *part_d;             // split regime: [kSplitMax][gridDim.x] partial minima
    int32_t *part_j;
    uint32_t *ticket;           // split regime: workgroups done (self-resetting)
};
struct RescanJobs {
    RescanJob j[2];
    int njobs;
};

struct PointJob {               // one D2 / PROJ column (k_point_jobs)
    const double *q64, *r64, *nrm;
    const int32_t *idx;
    int64_t q_begin;
    int metric, normal_mode;
    double *val;
};
struct PointJobs {
    PointJob j[4];
    int njobs;
    int64_t off[5];             // prefix sums of the jobs' row counts
};
struct UnitCol {                // one column reduced from a job's array
    int off;                    // field of the 32-byte result record (0: squared distance, 1: projection); 0 for plain columns
    int square;                 // reduce value^2 (the D2 column from the records' signed projection; metric.py:179)
    double *out_units;          // pinned host memory [3][nunits] per-leaf sum/min/max, or null
    double *out_blocks;         // pinned host memory [3][nblocks] per-32-leaf tree sum/min/max
    double *out_tail;           // pinned host memory [tail_n]
};
struct UnitJob {                // one per-point array to reduce (k_unit_jobs): up to two columns per pass
    const double *val;          // plain column (stride 1) or the result records (stride 2 or 4 doubles)
    int stride;
    // records of layout 1 (the matched record {rx, ry, rz, row}, 16 bytes): field 0 = the squared distance to row row0 + i of the
    // iterating cloud (q32), field 1 = err . normal[row0 + i] (metric.py:146-153), both formed here -- the rows and the searched
    // cloud's row-indexed normals are read in row order, i.e. coalesced, where the search would have gathered the normal
    int defer;                  // 0: no; 1: normals as 16-byte fp32-exact words (nrm32); 2: as fp64 rows (nrm64); 3: no normals (field 0 only);
                                // 4 / 5: as 1 / 2 with the normal of the MATCHED row (the record's row: --normal-index neighbour)
    int64_t nrm_rows;           // rows of the searched cloud's normals (bounds the gather of 4 / 5)
    const double *nrm64;
    const float4 *nrm32;
    const float4 *q32;          // iterating cloud, one fp32 word per row (Cloud::xyz32r)
    int64_t row0;               // row of the cloud the shard's first record belongs to
    int ncols;
    UnitCol c[2];
    int64_t ns, nunits;
    int64_t tail_first, tail_n; // rows [tail_first, tail_first + tail_n) are copied out raw
    int64_t nblocks;            // ceil(nunits / 32)
};
struct UnitJobs {
    UnitJob j[8];
    int njobs;
    int64_t uoff[9];            // prefix sums of 8 * nunits, each rounded up to a multiple of 256
    int64_t toff[9];            // prefix sums of tail_n
};
// the rescan jobs of njobs <= 2 results (scratch allocated), for k2b_fallback or the rescan half of k_grid_tail
int rescan_jobs(pccm_ctx *ctx, int njobs, const Cloud *const *its, const Cloud *const *ses, NNResult *const *ress, bool self, RescanJobs *out);
constexpr unsigned kRescanCap = 512;   // most workgroups that ever share one job's list (sizes the split regime's partials)
int launch_point_jobs(pccm_ctx *ctx, const PointJobs &jobs);
// result records -> plain columns (q32 / row0: the iterating cloud's rows, for records of layout 1)
int launch_unpack(pccm_ctx *ctx, const double *rec, int stride, int layout, const float4 *q32, int64_t row0, int64_t ns, int32_t *idx, double *d2);
int launch_unit_jobs(pccm_ctx *ctx, const UnitJobs &jobs);

int launch_point_metric(pccm_ctx *ctx, const Cloud &it, const Cloud &se, const NNResult &res, int metric,
                        int normal_mode, double *out_val /*[ns]*/, double *out_err /*[ns][3] or null*/);

double np_pairwise_sum(const double *a, int64_t n);

// minimal-OBB frame search (pccm_obb.hip)
int launch_obb_frames(pccm_ctx *ctx, const double *verts, int64_t nv, const double *tri, int64_t nt, double *ext_out, double *vol_out);
int launch_extreme_rows(pccm_ctx *ctx, const double *x64, int64_t n, const float *dirs, int ndirs, unsigned long long *best);
int launch_outside_planes(pccm_ctx *ctx, const double *x64, int64_t n, const double *planes, int nplanes, double margin,
                          int32_t *rows_out, unsigned int *count);

// colour columns (pccm_color.hip)
int launch_rgb8(pccm_ctx *ctx, Cloud &c, const unsigned char *bytes, unsigned int *flag);   // packs Cloud::rgb8 (from bytes, or from rgb64: *flag set when a value is no k / 255.0)
int launch_color_rows(pccm_ctx *ctx, const double *own, const double *other, const int32_t *rows, int64_t n,
                      int64_t n_other, int scheme, double scale, int what, double *out,
                      unsigned long long *maxkeys, unsigned int *bad, const uint32_t *own8 = nullptr, const uint32_t *other8 = nullptr,
                      const float4 *recs = nullptr);   // recs: matched records {x, y, z, row} instead of `rows`
int launch_color_colsum(pccm_ctx *ctx, const double *cols, int64_t n, double *out3);
int launch_color_colsums(pccm_ctx *ctx, int njobs, const double *const cols[2], const int64_t n[2], double *const out3[2],
                         unsigned long long *const outmax[2]);   // outmax: [3] bit keys of the columns' maxima per job, or null
int launch_colors_from_u8(pccm_ctx *ctx, const unsigned char *src, int64_t n3, double *out);

}  // namespace pccm
