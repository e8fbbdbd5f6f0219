// C ABI of libpccm.so (include/pccm.h): context, ingest, nn dispatch, getters, reductions,
// profiling.  Host-side only; kernels live in pccm_brute.hip / pccm_grid.hip / pccm_point.hip.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <thread>

#include "pccm_internal.h"

namespace pccm {

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int ensure(pccm_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.bytes && b.p) return PCCM_OK;
    if (ctx->capturing) {
        ctx->capture_failed = true;
        return fail(PCCM_E_STATE, "a buffer must grow during graph capture: run the same call sequence once before capturing");
    }
    ctx->epoch++;
    if (b.p) {
        PCCM_HIP(hipStreamSynchronize(ctx->stream));
        PCCM_HIP(hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes;
    PCCM_HIP(hipMalloc(&b.p, want));
    b.bytes = want;
    return PCCM_OK;
}

int grow(void **p, size_t &cap, size_t bytes)
{
    if (*p && cap >= bytes) return PCCM_OK;
    if (*p) {
        (void)hipFree(*p);
        *p = nullptr;
        cap = 0;
    }
    PCCM_HIP(hipMalloc(p, bytes ? bytes : 1));
    cap = bytes;
    return PCCM_OK;
}

// Called behind a wait for the GPU by everything that hands results out: a kernel that met a state it cannot be in has set a
// bit of the context's error word instead of answering wrongly.  The word is cleared when reported; the results are not usable.
int check_device_errors(pccm_ctx *ctx)
{
    if (!ctx->host_err) return PCCM_OK;
    const uint32_t e = __atomic_exchange_n(ctx->host_err, 0u, __ATOMIC_RELAXED);
    if (!e) return PCCM_OK;
    for (int d = 0; d < 3; ++d) { ctx->nn[d].valid = false; ctx->nn_gen[d]++; }
    for (auto &s : ctx->slots) s.pending = false;
    return fail(PCCM_E_STATE, "a search kernel reported an inconsistent state (device error word 0x%x: %s%s): the results of this search were "
                              "dropped, run it again", e, (e & 1u) ? "the tail launch's wait for its own workgroups ran out; " : "",
                (e & 2u) ? "a voxel brick contradicts its cell start" : "");
}

static hipEvent_t take_event(pccm_ctx *ctx)
{
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(pccm_ctx *c, int k) : ctx(c), cls(k)
{
    // not while capturing: on ROCm 7.2 event-record nodes replayed by a hipGraph return meaningless
    // (negative) elapsed times, so kernels are timed in eager launches only
    if (!ctx->prof_on || ctx->capturing) return;
    a = take_event(ctx);
    b = take_event(ctx);
    if (a) (void)hipEventRecord(a, ctx->stream);
}

ProfScope::~ProfScope()
{
    if (!ctx->prof_on || ctx->capturing || !a || !b) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->spans.push_back({a, b, cls});
}

static int collect_spans(pccm_ctx *ctx)
{
    if (ctx->spans.empty()) return PCCM_OK;
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &s : ctx->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            ctx->prof_ms[s.cls] += ms;
            ctx->prof_n[s.cls] += 1;
        }
        ctx->event_pool.push_back(s.a);
        ctx->event_pool.push_back(s.b);
    }
    ctx->spans.clear();
    return PCCM_OK;
}

static void free_buf(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

static void free_cloud(Cloud &c)
{
    if (c.xyz32) (void)hipFree(c.xyz32);
    if (c.xyz64) (void)hipFree(c.xyz64);
    if (c.xyz32r) (void)hipFree(c.xyz32r);
    c.xyz32r = nullptr;
    c.cap32r = 0;
    if (c.nrm64) (void)hipFree(c.nrm64);
    if (c.nrm32) (void)hipFree(c.nrm32);
    if (c.rgb64) (void)hipFree(c.rgb64);
    if (c.rgb8) (void)hipFree(c.rgb8);
    if (c.sp) (void)hipFree(c.sp);
    c.sp = nullptr;
    c.cap_sp = 0;
    c.sp_valid = false;
    c.xyz32 = nullptr;
    c.xyz64 = nullptr;
    c.nrm64 = nullptr;
    c.nrm32 = nullptr;
    c.rgb64 = nullptr;
    c.rgb8 = nullptr;
    c.cap_rgb8 = 0;
    c.rgb8_valid = false;
    c.cap32 = c.cap64 = c.cap_nrm = c.cap_nrm32 = c.cap_rgb = 0;
    c.n = c.n_pad = c.n_nrm = c.n_rgb = 0;
}

// forget the content, keep the allocations
static void drop_cloud(Cloud &c)
{
    c.n = c.n_pad = c.n_nrm = c.n_rgb = 0;
    c.nrm_deferred = false;
    c.nrm_host = nullptr;
    c.sp_valid = c.sp_tried = false;
    c.rgb8_valid = false;
}

static void free_nn(NNResult &r)
{
    free_buf(r.flagged);
    free_buf(r.flag_thr);
    free_buf(r.tail);
    free_buf(r.rec);
    r.rec_valid = r.plain_valid = false;
    if (r.idx) (void)hipFree(r.idx);
    if (r.d2) (void)hipFree(r.d2);
    r.idx = nullptr;
    r.d2 = nullptr;
    r.cap = 0;
    r.valid = false;
}

static void shard_of(int64_t n, int rank, int world, int64_t *b, int64_t *e)
{
    if (world <= 0) {                     // this context owns no rows of the direction
        *b = 0;
        *e = 0;
        return;
    }
    // shards start on whole 8192-row chunks of NumPy's sum whenever every rank can have one (the ranks then exchange
    // one number per chunk: pccm_reduce_chunks_many), else on 128-row leaves (pccm_reduce's per-leaf exchange vector)
    const int64_t unit = n >= (int64_t)world * kChunk ? kChunk : kLeaf;
    const int64_t units = (n + unit - 1) / unit;
    int64_t u0 = units * rank / world, u1 = units * (rank + 1) / world;
    int64_t lo = u0 * unit, hi = u1 * unit;
    *b = lo < n ? lo : n;
    *e = hi < n ? hi : n;
}

static int dir_clouds(pccm_ctx *ctx, int dir, const Cloud **it, const Cloud **se)
{
    if (dir == PCCM_DIR_LEFT) { *it = &ctx->cloud[0]; *se = &ctx->cloud[1]; }
    else if (dir == PCCM_DIR_RIGHT) { *it = &ctx->cloud[1]; *se = &ctx->cloud[0]; }
    else if (dir == PCCM_DIR_SELF) { *it = &ctx->cloud[0]; *se = &ctx->cloud[0]; }
    else return fail(PCCM_E_ARG, "bad direction %d", dir);
    if ((*it)->n <= 0 || (*se)->n <= 0) return fail(PCCM_E_STATE, "clouds are not set");
    return PCCM_OK;
}

// ---- transfers between caller memory and the device ---------------------------------------------------------------------------
// A copy on a few host threads (one thread moves ~10 GB/s, the link 50)
static void host_copy(void *dst, const void *src, size_t bytes)
{
    constexpr size_t kPiece = 1u << 20;
    const int nt = bytes >= 8 * kPiece ? 4 : bytes >= 2 * kPiece ? 2 : 1;
    if (nt == 1) {
        memcpy(dst, src, bytes);
        return;
    }
    std::thread th[3];
    const size_t part = (bytes / nt + 63) & ~(size_t)63;
    for (int k = 1; k < nt; ++k) {
        const size_t off = (size_t)k * part, len = k + 1 < nt ? part : bytes - off;
        th[k - 1] = std::thread([=] { memcpy((char *)dst + off, (const char *)src + off, len); });
    }
    memcpy(dst, src, part);
    for (int k = 1; k < nt; ++k) th[k - 1].join();
}

static int pin_ensure(pccm_ctx *ctx, int which, size_t bytes)
{
    if (ctx->pin_cap[which] >= bytes) return PCCM_OK;
    if (ctx->pin[which]) {
        PCCM_HIP(hipDeviceSynchronize());                       // (a copy out of the old buffer may be in flight)
        (void)hipHostFree(ctx->pin[which]);
        ctx->pin[which] = nullptr;
        ctx->pin_cap[which] = 0;
    }
    const size_t cap = (bytes + (bytes >> 2) + 4095) & ~(size_t)4095;
    if (hipHostMalloc(&ctx->pin[which], cap, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        ctx->pin[which] = nullptr;
        return fail(PCCM_E_OOM, "hipHostMalloc of %zu bytes (transfer staging) failed", cap);
    }
    ctx->pin_cap[which] = cap;
    return PCCM_OK;
}

constexpr size_t kStagedFrom = 32u << 10;        // (smaller transfers go through the runtime's own bounce buffers)
constexpr size_t kPinWindow = 64u << 20;         // most pinned memory one of the three buffers holds: larger transfers reuse it window by window

// host -> device on stream `st` (the context's main or copy stream)
static int h2d(pccm_ctx *ctx, void *dev, const void *host, size_t bytes, hipStream_t st)
{
    if (!ctx->io_staged || bytes < kStagedFrom) {
        PCCM_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st));
        return PCCM_OK;
    }
    const int which = st == ctx->stream ? 0 : 1;
    if (!ctx->pin_ev[which]) PCCM_HIP(hipEventCreateWithFlags(&ctx->pin_ev[which], hipEventDisableTiming));
    if (ctx->pin_ev_set[which]) PCCM_HIP(hipEventSynchronize(ctx->pin_ev[which]));      // the previous upload has left the buffer
    int rc = pin_ensure(ctx, which, bytes < kPinWindow ? bytes : kPinWindow);
    if (rc) return rc;
    constexpr size_t kChunk = 4u << 20;                          // the copy of piece k + 1 runs beside the DMA of piece k
    static_assert(kPinWindow % kChunk == 0, "whole pieces per window");
    for (size_t off = 0; off < bytes; off += kChunk) {
        const size_t len = bytes - off < kChunk ? bytes - off : kChunk, at = off % kPinWindow;
        if (off && at == 0) {                                    // the window is full: its pieces must have left before it is refilled
            PCCM_HIP(hipEventRecord(ctx->pin_ev[which], st));
            PCCM_HIP(hipEventSynchronize(ctx->pin_ev[which]));
        }
        host_copy((char *)ctx->pin[which] + at, (const char *)host + off, len);
        PCCM_HIP(hipMemcpyAsync((char *)dev + off, (char *)ctx->pin[which] + at, len, hipMemcpyHostToDevice, st));
    }
    PCCM_HIP(hipEventRecord(ctx->pin_ev[which], st));
    ctx->pin_ev_set[which] = true;
    return PCCM_OK;
}

// device -> host on the main stream; the data are in `host` when the call returns only in staged mode -- callers synchronise the
// stream behind it either way
static int d2h(pccm_ctx *ctx, void *host, const void *dev, size_t bytes)
{
    if (!ctx->io_staged || bytes < kStagedFrom) {
        PCCM_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return PCCM_OK;
    }
    int rc = pin_ensure(ctx, 2, bytes < kPinWindow ? bytes : kPinWindow);
    if (rc) return rc;
    for (size_t off = 0; off < bytes; off += kPinWindow) {
        const size_t len = bytes - off < kPinWindow ? bytes - off : kPinWindow;
        PCCM_HIP(hipMemcpyAsync(ctx->pin[2], (const char *)dev + off, len, hipMemcpyDeviceToHost, ctx->stream));
        PCCM_HIP(hipStreamSynchronize(ctx->stream));
        host_copy((char *)host + off, ctx->pin[2], len);
    }
    return PCCM_OK;
}

static int upload(pccm_ctx *ctx, const void *src, size_t bytes, int on_device, const void **dev_src)
{
    if (on_device) {
        *dev_src = src;
        return PCCM_OK;
    }
    int rc = ensure(ctx, ctx->staging, bytes);
    if (rc) return rc;
    rc = h2d(ctx, ctx->staging.p, src, bytes, ctx->stream);
    if (rc) return rc;
    *dev_src = ctx->staging.p;
    return PCCM_OK;
}

// upload + widening copy + validation of one cloud's normals on stream `st` (staging buffer `stage` for host sources)
static int ingest_normals(pccm_ctx *ctx, Cloud &c, int which, const void *nrm, int64_t n, int dtype, int on_device, hipStream_t st, DevBuf &stage)
{
    const size_t esz = dtype == PCCM_F32 ? 4 : 8;
    const void *dsrc = nrm;
    if (!on_device) {
        int rc = ensure(ctx, stage, (size_t)n * 3 * esz);
        if (rc) return rc;
        rc = h2d(ctx, stage.p, nrm, (size_t)n * 3 * esz, st);
        if (rc) return rc;
        dsrc = stage.p;
    }
    unsigned long long *stats = (unsigned long long *)ctx->stats.p + (st == ctx->stream ? 0 : 12);
    PCCM_HIP(hipMemsetAsync(stats, 0, 3 * sizeof(unsigned long long), st));
    hipStream_t keep = ctx->stream;
    ctx->stream = st;                                  // (the launcher takes the context's stream)
    int rc = launch_ingest_normals(ctx, dsrc, dtype, n, c.nrm64, (float *)c.nrm32, stats);
    ctx->stream = keep;
    if (rc) return rc;
    unsigned long long h[3];
    PCCM_HIP(hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, st));
    PCCM_HIP(hipStreamSynchronize(st));
    if (h[2] != 0) {
        c.n_nrm = 0;
        return fail(PCCM_E_ARG, "normals of cloud %d are not finite", which);      // n_nrm stays 0: no normals
    }
    c.n_nrm = n;
    c.nrm_exact32 = h[1] == 0;      // the search's fused projection then gathers 16 bytes per normal instead of 24 unaligned ones
    return PCCM_OK;
}

int normals_ready(pccm_ctx *ctx, Cloud &c)
{
    if (!c.nrm_deferred) return PCCM_OK;
    if (ctx->capturing) {
        ctx->capture_failed = true;
        return fail(PCCM_E_STATE, "deferred normals must be uploaded before graph capture: call pccm_flush_uploads first");
    }
    if (!ctx->copy_stream) PCCM_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    c.nrm_deferred = false;
    const void *src = c.nrm_host;
    c.nrm_host = nullptr;
    const int64_t n = c.n_nrm;
    c.n_nrm = 0;
    return ingest_normals(ctx, c, (int)(&c - ctx->cloud), src, n, c.nrm_host_dtype, 0, ctx->copy_stream, ctx->staging2);
}


}  // namespace pccm

using namespace pccm;

#define CHECK_CTX(ctx)                                                                            \
    if (!(ctx)) return fail(PCCM_E_ARG, "null context");                                          \
    std::lock_guard<std::recursive_mutex> ctx_guard_((ctx)->mu);                                  \
    do {                                                                                          \
        hipError_t _e = hipSetDevice((ctx)->device);                                              \
        if (_e != hipSuccess) return fail(PCCM_E_HIP, "hipSetDevice: %s", hipGetErrorString(_e)); \
    } while (0)

static void graph_free(GraphRec &g);
static int ensure_plain(pccm_ctx *ctx, NNResult &res, bool need_idx = true);

#define NOT_CAPTURING(ctx)                                                                          \
    do {                                                                                           \
        if ((ctx)->capturing) {                                                                    \
            (ctx)->capture_failed = true;                                                          \
            return fail(PCCM_E_STATE, "%s is not allowed between pccm_graph_begin and pccm_graph_end", __func__); \
        }                                                                                          \
    } while (0)

extern "C" {

int pccm_version(void) { return PCCM_VERSION; }

const char *pccm_last_error(void) { return g_err; }

int pccm_device_count(int *n)
{
    if (!n) return fail(PCCM_E_ARG, "null pointer");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *n = c;
    return PCCM_OK;
}

int pccm_ctx_create(int device, void *hip_stream, pccm_ctx **out)
{
    if (!out) return fail(PCCM_E_ARG, "null pointer");
    *out = nullptr;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) {
        (void)hipGetLastError();
        return fail(PCCM_E_NODEV, "no HIP device is visible: libpccm has no CPU path");
    }
    if (device < 0 || device >= c) return fail(PCCM_E_ARG, "device %d out of range (0..%d)", device, c - 1);
    PCCM_HIP(hipSetDevice(device));
    pccm_ctx *ctx = new (std::nothrow) pccm_ctx();
    if (!ctx) return fail(PCCM_E_OOM, "host allocation failed");
    ctx->device = device;
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            return fail(PCCM_E_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
        }
        ctx->own_stream = true;
    }
    int rc = ensure(ctx, ctx->counters, 16 * sizeof(uint32_t));      // [0..5] per-direction counters, [8..11] rescan tickets, [12..15] tail-launch counts
    if (!rc && hipHostMalloc((void **)&ctx->host_err, 64, hipHostMallocDefault) != hipSuccess) {
        ctx->host_err = nullptr;
        rc = fail(PCCM_E_OOM, "hipHostMalloc of the device error word failed");
    }
    if (!rc) *ctx->host_err = 0u;
    if (!rc && hipEventCreateWithFlags(&ctx->batch_ev, hipEventDisableTiming) != hipSuccess) rc = fail(PCCM_E_HIP, "hipEventCreate failed");
    if (!rc) rc = ensure(ctx, ctx->stats, 32 * sizeof(unsigned long long));      // [0..9] the main stream's scratch, [12..14] the copy stream's, [16..22] colour reduction of the other direction
    if (!rc && hipMemsetAsync(ctx->counters.p, 0, 16 * sizeof(uint32_t), ctx->stream) != hipSuccess)
        rc = fail(PCCM_E_HIP, "hipMemsetAsync failed");
    if (rc) {
        pccm_ctx_destroy(ctx);
        return rc;
    }
    for (int d = 0; d < 3; ++d) ctx->nn[d].nflag_dev = (uint32_t *)ctx->counters.p + 2 * d;
    *out = ctx;
    return PCCM_OK;
}

int pccm_ctx_destroy(pccm_ctx *ctx)
{
    if (!ctx) return PCCM_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &s : ctx->spans) {
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->batch_ev) (void)hipEventDestroy(ctx->batch_ev);
    ctx->batch_ev = nullptr;
    if (ctx->host_err) (void)hipHostFree(ctx->host_err);
    ctx->host_err = nullptr;
    for (int k = 0; k < 2; ++k) free_cloud(ctx->cloud[k]);
    for (int d = 0; d < 3; ++d) free_nn(ctx->nn[d]);
    DevBuf *bufs[] = {&ctx->part_b1, &ctx->part_g, &ctx->part_b2, &ctx->val, &ctx->stats, &ctx->staging, &ctx->staging2,
                      &ctx->counters, &ctx->color_cols, &ctx->color_idx, &ctx->colsum_scratch, &ctx->rescan_part, &ctx->tail_sync};
    for (DevBuf *b : bufs) free_buf(*b);
    for (auto &g : ctx->graphs) graph_free(g);
    for (auto &s : ctx->slots) {
        free_buf(s.val);
        if (s.host) (void)hipHostFree(s.host);
        if (s.ev) (void)hipEventDestroy(s.ev);
        s.ev = s.wait_ev = nullptr;
    }
    grid_release(ctx);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (auto &pbuf : ctx->pin)
        if (pbuf) (void)hipHostFree(pbuf);
    for (auto &pe : ctx->pin_ev)
        if (pe) (void)hipEventDestroy(pe);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PCCM_OK;
}

int pccm_set_cloud(pccm_ctx *ctx, int which, const void *xyz, int64_t n, int dtype, int on_device)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (which != 0 && which != 1) return fail(PCCM_E_ARG, "cloud index must be 0 or 1");
    if (!xyz || n <= 0) return fail(PCCM_E_ARG, "empty cloud (the reference cannot evaluate one either)");
    if (n > 0x7fffff00LL) return fail(PCCM_E_ARG, "more than 2^31 points per cloud are not supported");
    if (dtype != PCCM_F32 && dtype != PCCM_F64) return fail(PCCM_E_ARG, "dtype must be PCCM_F32 or PCCM_F64");
    Cloud &c = ctx->cloud[which];
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    drop_cloud(c);                                   // its normals and colours go with it
    // (a new cloud 1 leaves the self search of cloud 0 -- cloud_pair.py:108-109 -- as valid as it was: one reference cloud
    // against several decoded ones, BASELINE configs[4], keeps it, see CloudPair.with_reconst)
    for (int d = 0; d < 3; ++d) {
        if (which == 1 && d == PCCM_DIR_SELF) continue;
        ctx->nn[d].valid = false;
        ctx->nn_gen[d]++;
    }
    ctx->epoch++;
    c.version++;
    const int64_t n_pad = (n + kScanTile - 1) / kScanTile * kScanTile;
    int rc = grow((void **)&c.xyz32, c.cap32, (size_t)n_pad * 3 * sizeof(float));
    if (!rc) rc = grow((void **)&c.xyz64, c.cap64, (size_t)n * 3 * sizeof(double));
    if (!rc) rc = grow((void **)&c.xyz32r, c.cap32r, (size_t)n * sizeof(float4));
    if (rc) return rc;
    const size_t esz = dtype == PCCM_F32 ? 4 : 8;
    const void *dsrc = nullptr;
    rc = upload(ctx, xyz, (size_t)n * 3 * esz, on_device, &dsrc);
    if (rc) return rc;
    unsigned long long *stats = (unsigned long long *)ctx->stats.p;
    PCCM_HIP(hipMemsetAsync(stats, 0, 10 * sizeof(unsigned long long), ctx->stream));
    PCCM_HIP(hipMemsetAsync(stats + 3, 0xff, 3 * sizeof(unsigned long long), ctx->stream));
    rc = launch_ingest_points(ctx, dsrc, dtype, n, n_pad, c.xyz32, c.xyz64, c.xyz32r, stats);
    if (rc) return rc;
    unsigned long long h[10];
    PCCM_HIP(hipMemcpyAsync(h, stats, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    double maxabs;
    memcpy(&maxabs, &h[0], sizeof(double));
    if (h[2] != 0 || !(maxabs <= kMaxAbsCoord)) {
        drop_cloud(c);
        return fail(PCCM_E_ARG, "cloud %d has non-finite coordinates or |x| > 1e15", which);
    }
    c.n = n;
    c.n_pad = n_pad;
    c.maxabs = maxabs;
    c.exact32 = (h[1] == 0);
    c.all_int = (h[1] == 0 && h[9] == 0);
    for (int k = 0; k < 3; ++k) {
        auto unkey = [](unsigned long long b) {
            b = (b >> 63) ? (b & 0x7fffffffffffffffull) : ~b;
            double v;
            memcpy(&v, &b, sizeof(v));
            return v;
        };
        c.bb_min[k] = unkey(h[3 + k]);
        c.bb_max[k] = unkey(h[6 + k]);
    }
    return PCCM_OK;
}

int pccm_set_normals(pccm_ctx *ctx, int which, const void *nrm, int64_t n, int dtype, int on_device)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (which != 0 && which != 1) return fail(PCCM_E_ARG, "cloud index must be 0 or 1");
    if (!nrm || n <= 0) return fail(PCCM_E_ARG, "empty normals");
    if (dtype != PCCM_F32 && dtype != PCCM_F64) return fail(PCCM_E_ARG, "dtype must be PCCM_F32 or PCCM_F64");
    Cloud &c = ctx->cloud[which];
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    for (int d = 0; d < 3; ++d) ctx->nn_gen[d]++;      // pending D2 reductions used the old normals
    ctx->epoch++;
    c.n_nrm = 0;
    c.nrm_exact32 = false;
    c.nrm_deferred = false;
    c.nrm_host = nullptr;
    int rc = grow((void **)&c.nrm64, c.cap_nrm, (size_t)n * 3 * sizeof(double));
    if (rc) return rc;
    if ((rc = grow((void **)&c.nrm32, c.cap_nrm32, (size_t)n * sizeof(float4)))) return rc;
    return ingest_normals(ctx, c, which, nrm, n, dtype, on_device, ctx->stream, ctx->staging);
}

int pccm_set_normals_deferred(pccm_ctx *ctx, int which, const void *nrm, int64_t n, int dtype)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (which != 0 && which != 1) return fail(PCCM_E_ARG, "cloud index must be 0 or 1");
    if (!nrm || n <= 0) return fail(PCCM_E_ARG, "empty normals");
    if (dtype != PCCM_F32 && dtype != PCCM_F64) return fail(PCCM_E_ARG, "dtype must be PCCM_F32 or PCCM_F64");
    Cloud &c = ctx->cloud[which];
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    for (int d = 0; d < 3; ++d) ctx->nn_gen[d]++;
    ctx->epoch++;
    c.nrm_exact32 = false;
    int rc = grow((void **)&c.nrm64, c.cap_nrm, (size_t)n * 3 * sizeof(double));
    if (rc) return rc;
    if ((rc = grow((void **)&c.nrm32, c.cap_nrm32, (size_t)n * sizeof(float4)))) return rc;
    c.n_nrm = n;                                       // announced: the searches know that (and how many) normals exist
    c.nrm_host = nrm;
    c.nrm_host_dtype = dtype;
    c.nrm_deferred = true;
    return PCCM_OK;
}

int pccm_set_io_staged(pccm_ctx *ctx, int on)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    ctx->io_staged = on != 0;
    return PCCM_OK;
}

int pccm_flush_uploads(pccm_ctx *ctx)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    for (int k = 0; k < 2; ++k) {
        int rc = normals_ready(ctx, ctx->cloud[k]);
        if (rc) return rc;
    }
    return PCCM_OK;
}

int pccm_set_colors(pccm_ctx *ctx, int which, const void *rgb, int64_t n, int dtype, int on_device)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (which != 0 && which != 1) return fail(PCCM_E_ARG, "cloud index must be 0 or 1");
    if (!rgb || n <= 0) return fail(PCCM_E_ARG, "empty colours");
    if (dtype != PCCM_F32 && dtype != PCCM_F64) return fail(PCCM_E_ARG, "dtype must be PCCM_F32 or PCCM_F64");
    Cloud &c = ctx->cloud[which];
    if (c.n == 0) return fail(PCCM_E_STATE, "set cloud %d before its colours", which);
    if (n != c.n) return fail(PCCM_E_ARG, "cloud %d has %lld points but %lld colours", which, (long long)c.n, (long long)n);
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    c.n_rgb = 0;
    ctx->rgb_gen++;
    int rc = grow((void **)&c.rgb64, c.cap_rgb, (size_t)n * 3 * sizeof(double));
    if (rc) return rc;
    const void *dsrc = nullptr;
    rc = upload(ctx, rgb, (size_t)n * 3 * (dtype == PCCM_F32 ? 4 : 8), on_device, &dsrc);
    if (rc) return rc;
    unsigned long long *stats = (unsigned long long *)ctx->stats.p;
    PCCM_HIP(hipMemsetAsync(stats, 0, 3 * sizeof(unsigned long long), ctx->stream));
    rc = launch_ingest_normals(ctx, dsrc, dtype, n, c.rgb64, nullptr, stats);      // same widening copy; non-finite values are
    if (rc) return rc;                                                    // allowed here (NumPy propagates them)
    // colours that are bytes / 255 (nearly all are) also as packed words: Cloud::rgb8
    c.rgb8_valid = false;
    if ((rc = grow((void **)&c.rgb8, c.cap_rgb8, (size_t)n * sizeof(uint32_t)))) return rc;
    PCCM_HIP(hipMemsetAsync(stats + 4, 0, sizeof(unsigned long long), ctx->stream));
    if ((rc = launch_rgb8(ctx, c, nullptr, (unsigned int *)(stats + 4)))) return rc;
    unsigned long long not_bytes = 1;
    PCCM_HIP(hipMemcpyAsync(&not_bytes, stats + 4, sizeof(not_bytes), hipMemcpyDeviceToHost, ctx->stream));
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    c.rgb8_valid = not_bytes == 0;
    c.n_rgb = n;
    return PCCM_OK;
}

int pccm_set_colors_u8(pccm_ctx *ctx, int which, const unsigned char *rgb, int64_t n)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (which != 0 && which != 1) return fail(PCCM_E_ARG, "cloud index must be 0 or 1");
    if (!rgb || n <= 0) return fail(PCCM_E_ARG, "empty colours");
    Cloud &c = ctx->cloud[which];
    if (c.n == 0) return fail(PCCM_E_STATE, "set cloud %d before its colours", which);
    if (n != c.n) return fail(PCCM_E_ARG, "cloud %d has %lld points but %lld colours", which, (long long)c.n, (long long)n);
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    c.n_rgb = 0;
    ctx->rgb_gen++;
    int rc = grow((void **)&c.rgb64, c.cap_rgb, (size_t)n * 3 * sizeof(double));
    if (rc) return rc;
    const void *dsrc = nullptr;
    rc = upload(ctx, rgb, (size_t)n * 3, 0, &dsrc);
    if (rc) return rc;
    rc = launch_colors_from_u8(ctx, (const unsigned char *)dsrc, n * 3, c.rgb64);
    if (rc) return rc;
    c.rgb8_valid = false;
    if ((rc = grow((void **)&c.rgb8, c.cap_rgb8, (size_t)n * sizeof(uint32_t)))) return rc;
    if ((rc = launch_rgb8(ctx, c, (const unsigned char *)dsrc, nullptr))) return rc;
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    c.rgb8_valid = true;
    c.n_rgb = n;
    return PCCM_OK;
}

// common front end of the two colour calls: operands of direction `dir` and the neighbour rows to use
// (*drecs: matched records that carry the rows -- the kernel reads the row out of the record, no unpacked copy is made for it)
static int color_operands(pccm_ctx *ctx, int dir, int scheme, const int32_t *rows, int64_t nrows,
                          const Cloud **own, const Cloud **other, const int32_t **drows, const float4 **drecs)
{
    *drecs = nullptr;
    if (dir != PCCM_DIR_LEFT && dir != PCCM_DIR_RIGHT) return fail(PCCM_E_ARG, "colour metrics exist for directions 0 and 1");
    if (scheme < 0 || scheme > 2) return fail(PCCM_E_ARG, "unknown colour scheme %d", scheme);
    const Cloud &it = ctx->cloud[dir == PCCM_DIR_LEFT ? 0 : 1], &se = ctx->cloud[dir == PCCM_DIR_LEFT ? 1 : 0];
    if (it.n_rgb <= 0 || se.n_rgb <= 0) return fail(PCCM_E_STATE, "both clouds need colours (pccm_set_colors)");
    if (rows) {
        if (nrows != it.n) return fail(PCCM_E_ARG, "%lld neighbour rows for %lld points", (long long)nrows, (long long)it.n);
        int rc = ensure(ctx, ctx->color_idx, (size_t)nrows * sizeof(int32_t));
        if (rc) return rc;
        { int rch = h2d(ctx, ctx->color_idx.p, rows, (size_t)nrows * sizeof(int32_t), ctx->stream); if (rch) return rch; }
        *drows = (const int32_t *)ctx->color_idx.p;
    } else {
        NNResult &res = ctx->nn[dir];
        if (!res.valid) return fail(PCCM_E_STATE, "run pccm_nn for direction %d first", dir);
        if (res.begin != 0 || res.end != it.n)
            return fail(PCCM_E_STATE, "the search of direction %d was sharded: pass the gathered neighbour rows", dir);
        if (!res.plain_valid && res.rec_valid && res.rec_layout == 1 && res.rec_stride == 2 && !res.no_rows) {
            *drows = nullptr;
            *drecs = (const float4 *)res.rec.p;
        } else {
            int rc = ensure_plain(ctx, res);
            if (rc) return rc;
            *drows = res.idx;
        }
    }
    *own = &it;
    *other = &se;
    return PCCM_OK;
}

int pccm_color_reduce(pccm_ctx *ctx, int dir, int scheme, double scale, const int32_t *rows, int64_t nrows,
                      double sum_out[3], double max_out[3])
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (!sum_out || !max_out) return fail(PCCM_E_ARG, "null output");
    const Cloud *own[2], *other[2];
    const int32_t *drows[2];
    const float4 *drecs[2];
    int rc = color_operands(ctx, dir, scheme, rows, nrows, &own[0], &other[0], &drows[0], &drecs[0]);
    if (rc) return rc;
    pccm_ctx::ColorMemo &memo = ctx->color_memo;
    if (!rows && memo.valid && memo.dir == dir && memo.scheme == scheme && memo.scale == scale && memo.gen == ctx->nn_gen[dir] &&
        memo.rgb_gen == ctx->rgb_gen) {
        memo.valid = false;                               // (answered once: the host side keeps what it was given)
        if (memo.range_bad) return fail(PCCM_E_RANGE, "a neighbour row is outside the other cloud");
        memcpy(max_out, memo.max, sizeof(memo.max));
        memcpy(sum_out, memo.sum, sizeof(memo.sum));
        return PCCM_OK;
    }
    memo.valid = false;
    // the other direction rides along when it could be asked for the same way: the pair's own rows, an unsharded result
    int njobs = 1;
    const int sib = dir == PCCM_DIR_LEFT ? PCCM_DIR_RIGHT : PCCM_DIR_LEFT;
    if (!rows && ctx->nn[sib].valid && ctx->nn[sib].begin == 0 && ctx->nn[sib].end == ctx->cloud[sib == PCCM_DIR_LEFT ? 0 : 1].n &&
        color_operands(ctx, sib, scheme, nullptr, 0, &own[1], &other[1], &drows[1], &drecs[1]) == PCCM_OK)
        njobs = 2;
    const int64_t n[2] = {own[0]->n, njobs == 2 ? own[1]->n : 0};
    rc = ensure(ctx, ctx->color_cols, (size_t)(n[0] + n[1]) * 3 * sizeof(double));
    if (rc) return rc;
    // stats scratch per job: [0..2] column maxima as bit keys, [3..5] column sums, [6] range flag; the second job at word 16
    unsigned long long *small[2] = {(unsigned long long *)ctx->stats.p, (unsigned long long *)ctx->stats.p + 16};
    const double *cols[2] = {(const double *)ctx->color_cols.p, (const double *)ctx->color_cols.p + 3 * n[0]};
    double *sums[2] = {(double *)(small[0] + 3), (double *)(small[1] + 3)};
    PCCM_HIP(hipMemsetAsync(small[0], 0, (njobs == 2 ? 23 : 7) * sizeof(unsigned long long), ctx->stream));      // (one fill for both jobs' words)
    for (int k = 0; k < njobs; ++k) {
        const bool bytes = own[k]->rgb8_valid && other[k]->rgb8_valid;
        rc = launch_color_rows(ctx, own[k]->rgb64, other[k]->rgb64, drows[k], n[k], other[k]->n, scheme, scale, 4, (double *)cols[k], nullptr,
                               (unsigned int *)(small[k] + 6), bytes ? own[k]->rgb8 : nullptr, bytes ? other[k]->rgb8 : nullptr, drecs[k]);
        if (rc) return rc;
    }
    rc = launch_color_colsums(ctx, njobs, cols, n, sums, small);      // (+ the columns' maxima as bit keys into small[k][0..2])
    if (rc) return rc;
    unsigned long long h[2][7];
    for (int k = 0; k < njobs; ++k) PCCM_HIP(hipMemcpyAsync(h[k], small[k], sizeof(h[k]), hipMemcpyDeviceToHost, ctx->stream));
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    if (njobs == 2) {
        memo.valid = true;
        memo.range_bad = h[1][6] != 0;
        memo.dir = sib;
        memo.scheme = scheme;
        memo.scale = scale;
        memo.gen = ctx->nn_gen[sib];
        memo.rgb_gen = ctx->rgb_gen;
        memcpy(memo.max, h[1], sizeof(memo.max));
        memcpy(memo.sum, h[1] + 3, sizeof(memo.sum));
    }
    if (h[0][6]) return fail(PCCM_E_RANGE, "a neighbour row is outside the other cloud");
    memcpy(max_out, h[0], 3 * sizeof(double));
    memcpy(sum_out, h[0] + 3, 3 * sizeof(double));
    return PCCM_OK;
}

int pccm_obb_frames(pccm_ctx *ctx, const double *verts, int64_t nv, const double *tri, int64_t nt, double ext_out[3], double *vol_out)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (!verts || !tri || !ext_out || nv <= 0 || nt <= 0) return fail(PCCM_E_ARG, "bad argument");
    // scratch: [nv][3] vertices | [nt][9] triangles | [nt][3] extents | [nt] volumes
    const size_t bytes = ((size_t)nv * 3 + (size_t)nt * 13) * sizeof(double);
    int rc = ensure(ctx, ctx->color_cols, bytes);
    if (rc) return rc;
    double *dv = (double *)ctx->color_cols.p, *dt = dv + 3 * nv, *de = dt + 9 * nt, *dvol = de + 3 * nt;
    // (two uploads through one pinned buffer: the second waits for the first's copy out of it)
    if ((rc = h2d(ctx, dv, verts, (size_t)nv * 3 * sizeof(double), ctx->stream))) return rc;
    if ((rc = h2d(ctx, dt, tri, (size_t)nt * 9 * sizeof(double), ctx->stream))) return rc;
    rc = launch_obb_frames(ctx, dv, nv, dt, nt, de, dvol);
    if (rc) return rc;
    std::vector<double> ext((size_t)nt * 3), vol((size_t)nt);
    PCCM_HIP(hipMemcpyAsync(ext.data(), de, ext.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCCM_HIP(hipMemcpyAsync(vol.data(), dvol, vol.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    int64_t best = -1;
    for (int64_t t = 0; t < nt; ++t)                      // first smallest finite volume (np.argmin's choice)
        if (vol[t] < INFINITY && (best < 0 || vol[t] < vol[best])) best = t;
    if (best < 0) return fail(PCCM_E_ARG, "degenerate convex hull: no triangle spans a box of finite volume");
    for (int k = 0; k < 3; ++k) ext_out[k] = ext[3 * best + k];
    if (vol_out) *vol_out = vol[best];
    return PCCM_OK;
}

int pccm_extreme_rows(pccm_ctx *ctx, int which, const float *dirs, int ndirs, int32_t *rows_out)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if ((which != 0 && which != 1) || !dirs || !rows_out || ndirs <= 0 || ndirs > 1024) return fail(PCCM_E_ARG, "bad argument (1..1024 directions)");
    const Cloud &c = ctx->cloud[which];
    if (c.n <= 0) return fail(PCCM_E_STATE, "cloud %d is not set", which);
    int rc = ensure(ctx, ctx->color_cols, (size_t)ndirs * (3 * sizeof(float) + sizeof(unsigned long long)));
    if (rc) return rc;
    unsigned long long *best = (unsigned long long *)ctx->color_cols.p;
    float *ddirs = (float *)(best + ndirs);
    PCCM_HIP(hipMemsetAsync(best, 0, (size_t)ndirs * sizeof(unsigned long long), ctx->stream));
    PCCM_HIP(hipMemcpyAsync(ddirs, dirs, (size_t)ndirs * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    rc = launch_extreme_rows(ctx, c.xyz64, c.n, ddirs, ndirs, best);
    if (rc) return rc;
    std::vector<unsigned long long> h((size_t)ndirs);
    PCCM_HIP(hipMemcpyAsync(h.data(), best, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < ndirs; ++k) rows_out[k] = (int32_t)(h[k] & 0xffffffffull);
    return PCCM_OK;
}

int pccm_rows_outside(pccm_ctx *ctx, int which, const double *planes, int nplanes, double margin, int32_t *rows_out, int64_t *count)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if ((which != 0 && which != 1) || !planes || !rows_out || !count || nplanes <= 0 || !(margin >= 0.0)) return fail(PCCM_E_ARG, "bad argument");
    const Cloud &c = ctx->cloud[which];
    if (c.n <= 0) return fail(PCCM_E_STATE, "cloud %d is not set", which);
    int rc = ensure(ctx, ctx->color_cols, (size_t)nplanes * 4 * sizeof(double) + (size_t)c.n * sizeof(int32_t) + 16);
    if (rc) return rc;
    double *dpl = (double *)ctx->color_cols.p;
    unsigned int *dcount = (unsigned int *)(dpl + 4 * (size_t)nplanes);
    int32_t *drows = (int32_t *)(dcount + 4);
    PCCM_HIP(hipMemcpyAsync(dpl, planes, (size_t)nplanes * 4 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PCCM_HIP(hipMemsetAsync(dcount, 0, sizeof(unsigned int), ctx->stream));
    rc = launch_outside_planes(ctx, c.xyz64, c.n, dpl, nplanes, margin, drows, dcount);
    if (rc) return rc;
    unsigned int hc = 0;
    PCCM_HIP(hipMemcpyAsync(&hc, dcount, sizeof(hc), hipMemcpyDeviceToHost, ctx->stream));
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    if (hc) {
        if ((rc = d2h(ctx, rows_out, drows, (size_t)hc * sizeof(int32_t)))) return rc;
        PCCM_HIP(hipStreamSynchronize(ctx->stream));
    }
    *count = (int64_t)hc;
    return PCCM_OK;
}

int pccm_seq_colsum(pccm_ctx *ctx, const double *cols, int64_t n, double out[3])
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (!cols || !out || n <= 0) return fail(PCCM_E_ARG, "bad argument");
    int rc = ensure(ctx, ctx->color_cols, (size_t)n * 3 * sizeof(double));
    if (rc) return rc;
    if ((rc = h2d(ctx, ctx->color_cols.p, cols, (size_t)n * 3 * sizeof(double), ctx->stream))) return rc;
    double *dsum = (double *)ctx->stats.p + 3;
    rc = launch_color_colsum(ctx, (const double *)ctx->color_cols.p, n, dsum);
    if (rc) return rc;
    PCCM_HIP(hipMemcpyAsync(out, dsum, 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    return PCCM_OK;
}

int pccm_color_rows(pccm_ctx *ctx, int dir, int scheme, double scale, int what, const int32_t *rows, int64_t nrows,
                    double *out)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (!out) return fail(PCCM_E_ARG, "null output");
    if (what < 0 || what > 3) return fail(PCCM_E_ARG, "what must be 0..3");
    const Cloud *own, *other;
    const int32_t *drows;
    const float4 *drecs;
    int rc = color_operands(ctx, dir, scheme, rows, nrows, &own, &other, &drows, &drecs);
    if (rc) return rc;
    const int64_t n = own->n;
    rc = ensure(ctx, ctx->color_cols, (size_t)n * 3 * sizeof(double));
    if (rc) return rc;
    unsigned long long *small = (unsigned long long *)ctx->stats.p;
    PCCM_HIP(hipMemsetAsync(small + 6, 0, sizeof(unsigned long long), ctx->stream));
    const bool bytes = own->rgb8_valid && other->rgb8_valid;
    rc = launch_color_rows(ctx, own->rgb64, other->rgb64, drows, n, other->n, scheme, scale, what, (double *)ctx->color_cols.p,
                           small, (unsigned int *)(small + 6), bytes ? own->rgb8 : nullptr, bytes ? other->rgb8 : nullptr, drecs);
    if (rc) return rc;
    unsigned long long flag = 0;
    if ((rc = d2h(ctx, out, ctx->color_cols.p, (size_t)n * 3 * sizeof(double)))) return rc;
    PCCM_HIP(hipMemcpyAsync(&flag, small + 6, sizeof(flag), hipMemcpyDeviceToHost, ctx->stream));
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    if (flag) return fail(PCCM_E_RANGE, "a neighbour row is outside the other cloud");
    return PCCM_OK;
}

int pccm_estimate_normals(pccm_ctx *ctx, int which, int knn)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (which != 0 && which != 1) return fail(PCCM_E_ARG, "cloud index must be 0 or 1");
    return estimate_normals(ctx, which, knn);
}

int pccm_get_normals(pccm_ctx *ctx, int which, double *out)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (which != 0 && which != 1) return fail(PCCM_E_ARG, "cloud index must be 0 or 1");
    if (!out) return fail(PCCM_E_ARG, "null pointer");
    Cloud &c = ctx->cloud[which];
    { int rcn = normals_ready(ctx, c); if (rcn) return rcn; }
    if (c.n_nrm <= 0) return fail(PCCM_E_STATE, "cloud %d has no normals", which);
    { int rcd = d2h(ctx, out, c.nrm64, (size_t)c.n_nrm * 3 * sizeof(double)); if (rcd) return rcd; }
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    return PCCM_OK;
}

int pccm_set_shard(pccm_ctx *ctx, int rank, int world)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (world < 1 || rank < 0 || rank >= world) return fail(PCCM_E_ARG, "bad shard %d of %d", rank, world);
    for (int d = 0; d < 3; ++d) {
        ctx->shard_rank[d] = rank;
        ctx->shard_world[d] = world;
    }
    ctx->epoch++;
    for (int d = 0; d < 3; ++d) { ctx->nn[d].valid = false; ctx->nn_gen[d]++; }
    return PCCM_OK;
}

int pccm_set_shard_dir(pccm_ctx *ctx, int dir, int rank, int world)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (dir < 0 || dir > 2) return fail(PCCM_E_ARG, "bad direction %d", dir);
    if (world < 0 || (world > 0 && (rank < 0 || rank >= world))) return fail(PCCM_E_ARG, "bad shard %d of %d", rank, world);
    ctx->shard_rank[dir] = world > 0 ? rank : 0;
    ctx->shard_world[dir] = world;
    ctx->epoch++;
    ctx->nn[dir].valid = false;
    ctx->nn_gen[dir]++;
    return PCCM_OK;
}

int pccm_shard_range(pccm_ctx *ctx, int dir, int64_t *begin, int64_t *end)
{
    CHECK_CTX(ctx);
    if (!begin || !end) return fail(PCCM_E_ARG, "null pointer");
    const Cloud *it, *se;
    int rc = dir_clouds(ctx, dir, &it, &se);
    if (rc) return rc;
    shard_of(it->n, ctx->shard_rank[dir], ctx->shard_world[dir], begin, end);
    return PCCM_OK;
}

// shard range, result buffers and bookkeeping of one direction; *trivial = 1 when nothing is left to compute
static int prepare_nn(pccm_ctx *ctx, int dir, int *trivial)
{
    const Cloud *it, *se;
    int rc = dir_clouds(ctx, dir, &it, &se);
    if (rc) return rc;
    NNResult &res = ctx->nn[dir];
    res.valid = false;
    ctx->nn_gen[dir]++;
    shard_of(it->n, ctx->shard_rank[dir], ctx->shard_world[dir], &res.begin, &res.end);
    const int64_t ns = res.end - res.begin;
    if (ctx->capturing) {
        GraphOp op;
        op.kind = 1;
        op.dir = dir;
        ctx->cap_ops.push_back(op);
    }
    if (ns > res.cap) {
        if (ctx->capturing) {
            ctx->capture_failed = true;
            return fail(PCCM_E_STATE, "result buffers must grow during graph capture: run pccm_nn once before capturing");
        }
        ctx->epoch++;
        PCCM_HIP(hipStreamSynchronize(ctx->stream));
        if (res.idx) (void)hipFree(res.idx);
        if (res.d2) (void)hipFree(res.d2);
        res.idx = nullptr;
        res.d2 = nullptr;
        res.cap = 0;
        PCCM_HIP(hipMalloc((void **)&res.idx, (size_t)ns * sizeof(int32_t)));
        PCCM_HIP(hipMalloc((void **)&res.d2, (size_t)ns * sizeof(double)));
        res.cap = ns;
    }
    {
        int rc2 = ensure(ctx, res.rec, (size_t)(ns > 0 ? ns : 1) * sizeof(double4));   // 32-byte result records (grid engine)
        if (rc2) return rc2;
    }
    res.rec_valid = false;
    res.no_rows = false;
    res.plain_valid = res.plain_d2_valid = ns <= 0;          // an empty shard has nothing to unpack
    res.fused_mode = -1;
    res.stats[0] = res.stats[1] = res.stats[2] = 0;
    *trivial = 0;
    if (dir == PCCM_DIR_SELF && it->n < 2) {
        // Open3D's compute_nearest_neighbor_distance returns zeros for fewer than two points
        if (ns > 0) {
            PCCM_HIP(hipMemsetAsync(res.idx, 0xff, (size_t)ns * sizeof(int32_t), ctx->stream));
            PCCM_HIP(hipMemsetAsync(res.d2, 0, (size_t)ns * sizeof(double), ctx->stream));
        }
        PCCM_HIP(hipMemsetAsync(res.nflag_dev, 0, 2 * sizeof(uint32_t), ctx->stream));
        res.plain_valid = res.plain_d2_valid = true;
        *trivial = 1;
    }
    return PCCM_OK;
}

static int pick_engine(int engine)
{
    if (engine != PCCM_ENGINE_AUTO) return engine;
    const char *e = getenv("PCCM_ENGINE");
    if (e && !strcmp(e, "brute")) return PCCM_ENGINE_BRUTE;
    return PCCM_ENGINE_GRID;
}

static int run_nn(pccm_ctx *ctx, int ndirs, const int *dirs, int engine)
{
    const bool automatic = engine == PCCM_ENGINE_AUTO && !getenv("PCCM_ENGINE");
    engine = pick_engine(engine);
    if (automatic && ctx->cloud[0].n > 0 && ctx->cloud[1].n > 0) {
        // distributions no uniform grid can separate (see decide_scale) go to the engine whose cost is flat
        bool hostile = false;
        int rc = grid_decide(ctx, &hostile);
        if (rc) return rc;
        if (!hostile && (rc = grid_prefers_brute(ctx, &hostile))) return rc;
        if (hostile) engine = PCCM_ENGINE_BRUTE;
    }
    if (engine != PCCM_ENGINE_BRUTE && engine != PCCM_ENGINE_GRID) return fail(PCCM_E_ARG, "unknown engine %d", engine);
    int todo[3], ntodo = 0, rc;
    for (int k = 0; k < ndirs; ++k) {
        int trivial = 0;
        if ((rc = prepare_nn(ctx, dirs[k], &trivial))) return rc;
        if (!trivial) todo[ntodo++] = dirs[k];
    }
    if (engine == PCCM_ENGINE_GRID) {
        if (ntodo > 0 && (rc = nn_grid(ctx, ntodo, todo))) return rc;
    } else {
        for (int k = 0; k < ntodo; ++k) {
            const Cloud *it, *se;
            if ((rc = dir_clouds(ctx, todo[k], &it, &se))) return rc;
            ctx->nn[todo[k]].plain_valid = ctx->nn[todo[k]].plain_d2_valid = true;   // the brute-force engine writes the plain columns
            if ((rc = nn_brute(ctx, *it, *se, todo[k] == PCCM_DIR_SELF, ctx->nn[todo[k]]))) return rc;
        }
    }
    for (int k = 0; k < ndirs; ++k) ctx->nn[dirs[k]].valid = true;
    return PCCM_OK;
}

int pccm_nn(pccm_ctx *ctx, int dir, int engine)
{
    CHECK_CTX(ctx);
    if (dir < 0 || dir > 2) return fail(PCCM_E_ARG, "bad direction %d", dir);
    return run_nn(ctx, 1, &dir, engine);
}

int pccm_nn_pair(pccm_ctx *ctx, int engine)
{
    CHECK_CTX(ctx);
    const int dirs[2] = {PCCM_DIR_LEFT, PCCM_DIR_RIGHT};
    return run_nn(ctx, 2, dirs, engine);
}

static int need_nn(pccm_ctx *ctx, int dir, const Cloud **it, const Cloud **se, NNResult **res)
{
    int rc = dir_clouds(ctx, dir, it, se);
    if (rc) return rc;
    *res = &ctx->nn[dir];
    if (!(*res)->valid) return fail(PCCM_E_STATE, "pccm_nn(dir=%d) has not run for the current clouds/shard", dir);
    return PCCM_OK;
}

// the plain idx / d2 columns of a result: the grid engine leaves 32-byte records, unpacked here when somebody
// wants columns (getters, colour kernels, the separate point kernel)
static int ensure_plain(pccm_ctx *ctx, NNResult &res, bool need_idx)
{
    if (res.plain_valid || (!need_idx && res.plain_d2_valid)) return PCCM_OK;
    if (!res.rec_valid) return fail(PCCM_E_STATE, "no nearest-neighbour result to read");
    if (need_idx && ((res.rec_stride != 4 && res.rec_layout != 1) || res.no_rows)) {   // (matched records carry the row, the voxel-brick search's excepted)
        // the search ran without the matched rows (pccm_nn_want_idx off) and now somebody asks for them: run it again
        // for this direction with the rows on -- same results, 32-byte records; the clouds and the grid are resident
        if (ctx->capturing) {
            ctx->capture_failed = true;
            return fail(PCCM_E_STATE, "matched rows are needed during graph capture: switch pccm_nn_want_idx on before the search");
        }
        const int dir = (int)(&res - ctx->nn);
        int rc = nn_grid(ctx, 1, &dir, /*force_idx=*/1);
        if (rc) return rc;
    }
    const bool rows = (res.rec_stride == 4 || res.rec_layout == 1) && !res.no_rows;
    const int udir = (int)(&res - ctx->nn);
    const Cloud &uit = ctx->cloud[udir == PCCM_DIR_RIGHT ? 1 : 0];
    int rc = launch_unpack(ctx, (const double *)res.rec.p, res.rec_stride, res.rec_layout, uit.xyz32r, res.begin, res.end - res.begin,
                           rows ? res.idx : nullptr, res.d2);
    if (rc) return rc;
    res.plain_d2_valid = true;
    res.plain_valid = rows;
    return PCCM_OK;
}

int pccm_nn_want_idx(pccm_ctx *ctx, int on)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if ((ctx->want_idx != 0) != (on != 0)) {
        ctx->want_idx = on ? 1 : 0;
        ctx->epoch++;                                        // captured searches carry the old record layout
    }
    return PCCM_OK;
}

int pccm_nn_fuse(pccm_ctx *ctx, int dir, int normal_mode)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (dir != PCCM_DIR_LEFT && dir != PCCM_DIR_RIGHT) return fail(PCCM_E_ARG, "the projection exists for directions 0 and 1");
    if (normal_mode != -1 && normal_mode != PCCM_NORMAL_ROW && normal_mode != PCCM_NORMAL_NEIGHBOUR)
        return fail(PCCM_E_ARG, "bad normal mode %d", normal_mode);
    if (ctx->fuse_mode[dir] != normal_mode) {
        ctx->fuse_mode[dir] = normal_mode;
        ctx->epoch++;                                        // captured searches carry the old choice
    }
    return PCCM_OK;
}

int pccm_nn_fetch(pccm_ctx *ctx, int dir, int32_t *idx, double *d2)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    const Cloud *it, *se;
    NNResult *res;
    int rc = need_nn(ctx, dir, &it, &se, &res);
    if (rc) return rc;
    if ((rc = ensure_plain(ctx, *res, idx != nullptr))) return rc;
    const int64_t ns = res->end - res->begin;
    if (ns > 0 && idx) { int rcd = d2h(ctx, idx, res->idx, (size_t)ns * sizeof(int32_t)); if (rcd) return rcd; }
    if (ns > 0 && d2) { int rcd = d2h(ctx, d2, res->d2, (size_t)ns * sizeof(double)); if (rcd) return rcd; }
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    return check_device_errors(ctx);
}

static int check_normals(pccm_ctx *ctx, const Cloud &it, const Cloud &se, const NNResult &res, int normal_mode)
{
    if (normal_mode != PCCM_NORMAL_ROW && normal_mode != PCCM_NORMAL_NEIGHBOUR)
        return fail(PCCM_E_ARG, "bad normal mode %d", normal_mode);
    {   // whoever checks the normals is about to read them: announced ones (pccm_set_normals_deferred) cross PCIe now
        int rcn = normals_ready(ctx, const_cast<Cloud &>(se));
        if (rcn) return rcn;
    }
    if (se.n_nrm <= 0) return fail(PCCM_E_STATE, "the searched cloud has no normals (pccm_set_normals)");
    // sharded: the test is on the whole iterating cloud, so that every rank raises (or none does) -- a per-shard test
    // would let the low ranks walk into the exchange while the last one raises
    if (normal_mode == PCCM_NORMAL_ROW && (ctx->sharded() ? it.n : res.end) > se.n_nrm)
        return fail(PCCM_E_RANGE, "index %lld is out of bounds for axis 0 with size %lld (row-indexed normals, reference quirk Q1)",
                    (long long)se.n_nrm, (long long)se.n_nrm);
    if (normal_mode == PCCM_NORMAL_NEIGHBOUR && se.n_nrm != se.n)
        return fail(PCCM_E_ARG, "neighbour-indexed normals need one normal per point");
    return PCCM_OK;
}

int pccm_error_vectors(pccm_ctx *ctx, int dir, double *out)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (!out) return fail(PCCM_E_ARG, "null pointer");
    const Cloud *it, *se;
    NNResult *res;
    int rc = need_nn(ctx, dir, &it, &se, &res);
    if (rc) return rc;
    const int64_t ns = res->end - res->begin;
    if (ns <= 0) return PCCM_OK;
    if ((rc = ensure_plain(ctx, *res))) return rc;
    if ((rc = ensure(ctx, ctx->val, (size_t)ns * 3 * sizeof(double)))) return rc;
    if ((rc = launch_point_metric(ctx, *it, *se, *res, PCCM_METRIC_D1, PCCM_NORMAL_ROW, nullptr, (double *)ctx->val.p))) return rc;
    { int rcd = d2h(ctx, out, ctx->val.p, (size_t)ns * 3 * sizeof(double)); if (rcd) return rcd; }
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    return PCCM_OK;
}

// device pointer to the shard's per-point metric (computing it into ctx->val when needed)
static int metric_on_device(pccm_ctx *ctx, int dir, int metric, int normal_mode, const double **dev, int64_t *ns_out,
                            const Cloud **it_out, NNResult **res_out, DevBuf *valbuf = nullptr)
{
    DevBuf &vb = valbuf ? *valbuf : ctx->val;
    const Cloud *it, *se;
    NNResult *res;
    int rc = need_nn(ctx, dir, &it, &se, &res);
    if (rc) return rc;
    const int64_t ns = res->end - res->begin;
    *ns_out = ns;
    *it_out = it;
    *res_out = res;
    if ((rc = ensure_plain(ctx, *res, metric != PCCM_METRIC_D1))) return rc;
    if (metric == PCCM_METRIC_D1) {
        *dev = res->d2;
        return PCCM_OK;
    }
    if (metric != PCCM_METRIC_D2 && metric != PCCM_METRIC_PROJ) return fail(PCCM_E_ARG, "bad metric %d", metric);
    if (dir == PCCM_DIR_SELF) return fail(PCCM_E_ARG, "point-to-plane is not defined for the self search");
    if ((rc = check_normals(ctx, *it, *se, *res, normal_mode))) return rc;
    if ((rc = ensure(ctx, vb, (size_t)(ns > 0 ? ns : 1) * sizeof(double)))) return rc;
    if ((rc = launch_point_metric(ctx, *it, *se, *res, metric, normal_mode, (double *)vb.p, nullptr))) return rc;
    *dev = (const double *)vb.p;
    return PCCM_OK;
}

int pccm_tie_exposure(pccm_ctx *ctx, int dir, int normal_mode, double out[8])
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (!out) return fail(PCCM_E_ARG, "null pointer");
    if (dir != PCCM_DIR_LEFT && dir != PCCM_DIR_RIGHT) return fail(PCCM_E_ARG, "tie exposure exists for directions 0 and 1");
    const Cloud *it, *se;
    NNResult *res;
    int rc = need_nn(ctx, dir, &it, &se, &res);
    if (rc) return rc;
    if (normal_mode >= 0 && (rc = check_normals(ctx, *it, *se, *res, normal_mode))) return rc;
    if ((rc = ensure_plain(ctx, *res, true))) return rc;
    return tie_exposure(ctx, dir, *it, *se, *res, normal_mode, out);
}

int pccm_point_metric(pccm_ctx *ctx, int dir, int metric, int normal_mode, double *out)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (!out) return fail(PCCM_E_ARG, "null pointer");
    const double *dev;
    int64_t ns;
    const Cloud *it;
    NNResult *res;
    int rc = metric_on_device(ctx, dir, metric, normal_mode, &dev, &ns, &it, &res);
    if (rc) return rc;
    if (ns > 0) { int rcd = d2h(ctx, out, dev, (size_t)ns * sizeof(double)); if (rcd) return rcd; }
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    return PCCM_OK;
}

int64_t pccm_xvec_len(int64_t n_iter)
{
    if (n_iter <= 0) return 0;
    return (n_iter / kChunk) * (kChunk / kLeaf) + (n_iter % kChunk);
}

// ---- reductions: enqueue (prefetch) and consume ---------------------------------------------------------
// A reduction is enqueued into a slot: point kernel (D2/PROJ) -> per-unit sums/min/max -> async copy of
// the unit arrays and of the shard's raw tail values into pinned host memory -> event.  pccm_reduce()
// consumes a slot (enqueuing it first when nobody prefetched it), so a caller that prefetches every
// column it will need waits for the GPU once per step instead of once per column.
// bookkeeping + buffers of one slot; the kernels are launched for all new slots together (slots_launch)
static int slot_prepare(pccm_ctx *ctx, ReduceSlot &s, int dir, int metric, int normal_mode, bool want_units, PointJobs &pj,
                        UnitJobs &uj)
{
    const Cloud *it, *se;
    NNResult *res;
    int rc = need_nn(ctx, dir, &it, &se, &res);
    if (rc) return rc;
    const int64_t ns = res->end - res->begin;
    // where the column lives: a field of the grid engine's 32-byte result records (squared distance, or the signed
    // projection fused into the search by pccm_nn_fuse), or a plain column (brute-force engine; unfused projection)
    const double *dev = nullptr;
    int stride = 1, square = 0, defer = 0;
    if (metric == PCCM_METRIC_D1) {
        if (res->rec_valid) {
            dev = (const double *)res->rec.p;
            stride = res->rec_stride;
            if (res->rec_layout == 1) defer = 3;             // matched records: the reduction forms the distance (NNOut::layout)
        } else {
            if ((rc = ensure_plain(ctx, *res, false))) return rc;
            dev = res->d2;
        }
    } else {
        if (metric != PCCM_METRIC_D2 && metric != PCCM_METRIC_PROJ) return fail(PCCM_E_ARG, "bad metric %d", metric);
        if (dir == PCCM_DIR_SELF) return fail(PCCM_E_ARG, "point-to-plane is not defined for the self search");
        if ((rc = check_normals(ctx, *it, *se, *res, normal_mode))) return rc;
        if (res->rec_valid && !res->no_rows &&
            (res->fused_mode == normal_mode || (res->rec_layout == 1 && (normal_mode == PCCM_NORMAL_ROW || normal_mode == PCCM_NORMAL_NEIGHBOUR)))) {
            dev = (const double *)res->rec.p + 1;
            stride = res->rec_stride;
            square = metric == PCCM_METRIC_D2 ? 1 : 0;       // metric.py:179: the square of the stored projection
            // ... which this reduction forms itself from a matched record (NNOut::layout): with the normal of the query's row
            // (streamed) or of the matched row the record carries (gathered: what a separate point pass would gather too)
            if (res->rec_layout == 1) defer = (se->nrm_exact32 ? 1 : 2) + (normal_mode == PCCM_NORMAL_NEIGHBOUR ? 3 : 0);
        } else {
            if ((rc = ensure_plain(ctx, *res))) return rc;
            if ((rc = ensure(ctx, s.val, (size_t)(ns > 0 ? ns : 1) * sizeof(double)))) return rc;
            dev = (const double *)s.val.p;
            if (ns > 0) {
                if (pj.njobs >= 4) return fail(PCCM_E_ARG, "at most four unfused point-to-plane columns per call");
                PointJob &P = pj.j[pj.njobs];
                P.q64 = it->xyz64; P.r64 = se->xyz64; P.nrm = se->nrm64; P.idx = res->idx;
                P.q_begin = res->begin; P.metric = metric; P.normal_mode = normal_mode; P.val = (double *)s.val.p;
                pj.off[pj.njobs + 1] = pj.off[pj.njobs] + ns;
                pj.njobs++;
            }
        }
    }
    s.dir = dir; s.metric = metric; s.mode = normal_mode;
    s.gen = ctx->nn_gen[dir];
    s.n_iter = it->n; s.begin = res->begin; s.end = res->end;
    s.nunits = ns > 0 ? (ns + kLeaf - 1) / kLeaf : 0;
    s.nblocks = (s.nunits + 31) / 32;
    s.has_units = want_units;
    const int64_t nfull = it->n / kChunk, full_rows = nfull * kChunk;
    s.t0 = res->begin > full_rows ? res->begin : full_rows;
    s.tail_n = s.t0 < res->end ? res->end - s.t0 : 0;
    const size_t need = (size_t)(3 * s.nunits + 3 * s.nblocks + s.tail_n + 1) * sizeof(double);
    if (ctx->capturing && (need > s.host_cap || !s.ev)) {
        ctx->capture_failed = true;
        return fail(PCCM_E_STATE, "a reduction slot must be allocated during graph capture: run the sequence once first");
    }
    if (need > s.host_cap) {
        ctx->epoch++;
        if (s.host) (void)hipHostFree(s.host);
        s.host = nullptr; s.host_cap = 0;
        PCCM_HIP(hipHostMalloc((void **)&s.host, need, hipHostMallocDefault));
        s.host_cap = need;
    }
    if (!s.ev) PCCM_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    if (s.nunits > 0) {
        UnitCol col;
        col.off = (stride >= 2 && metric != PCCM_METRIC_D1) ? 1 : 0;
        col.square = square;
        col.out_units = want_units ? s.host : nullptr;
        col.out_blocks = s.host + 3 * s.nunits;
        col.out_tail = s.host + 3 * s.nunits + 3 * s.nblocks;
        // a second column over the same result records rides along with the job that already reads them
        const double *base = stride >= 2 ? (const double *)res->rec.p : dev;
        UnitJob *host_job = nullptr;
        static const bool merge = [] {
            const char *e = PCCM_DIAG_ENV("PCCM_REDUCE_MERGE");
            return !(e && e[0] == '0');
        }();
        if (stride >= 2 && merge)
            for (int k = 0; k < uj.njobs; ++k)
                if (uj.j[k].stride >= 2 && uj.j[k].val == base && uj.j[k].ncols == 1 &&
                    (uj.j[k].defer == defer || uj.j[k].defer == 3 || defer == 3))      // (one normal per job: row- and neighbour-indexed D2 do not share one)
                    host_job = &uj.j[k];
        if (host_job) {
            host_job->c[1] = col;
            host_job->ncols = 2;
            if (defer && defer != 3 && (host_job->defer == 0 || host_job->defer == 3)) host_job->defer = defer;   // (3: distances only so far)
        } else {
            if (uj.njobs >= 8) return fail(PCCM_E_ARG, "too many columns in one reduction batch");
            UnitJob &U = uj.j[uj.njobs];
            U.val = base; U.stride = stride; U.ncols = 1;
            U.defer = defer; U.nrm64 = se->nrm64; U.nrm32 = se->nrm32; U.nrm_rows = se->n_nrm; U.q32 = it->xyz32r; U.row0 = res->begin;
            U.c[0] = col; U.c[1] = col;
            U.ns = ns; U.nunits = s.nunits;
            U.tail_first = s.t0 - res->begin; U.tail_n = s.tail_n;
            U.nblocks = s.nblocks;
            const int64_t lanes = (s.nunits * 8 + 255) / 256 * 256;
            uj.uoff[uj.njobs + 1] = uj.uoff[uj.njobs] + lanes;
            uj.toff[uj.njobs + 1] = uj.toff[uj.njobs] + s.tail_n;
            uj.njobs++;
        }
    }
    return PCCM_OK;
}

static ReduceSlot *slot_find(pccm_ctx *ctx, int dir, int metric, int normal_mode, bool need_units = false)
{
    for (auto &s : ctx->slots)
        if (s.pending && (s.has_units || !need_units) && s.dir == dir && s.metric == metric && (metric == PCCM_METRIC_D1 || s.mode == normal_mode) &&
            s.gen == ctx->nn_gen[dir])
            return &s;
    return nullptr;
}

// A slot for a new reduction: an idle or stale one; failing that, a pending one that does NOT belong to the batch being
// assembled (`fresh`): its unconsumed result is given up (a later pccm_reduce recomputes it) -- never a slot of the current
// batch, whose host buffers an earlier job of the same launch is about to write.
static ReduceSlot *slot_free(pccm_ctx *ctx, ReduceSlot *const *fresh, int nfresh)
{
    for (auto &s : ctx->slots)
        if (!s.pending || s.gen != ctx->nn_gen[s.dir]) return &s;
    for (auto &s : ctx->slots) {
        bool mine = false;
        for (int k = 0; k < nfresh; ++k) mine = mine || fresh[k] == &s;
        if (!mine) return &s;
    }
    return nullptr;
}

static int prefetch_many(pccm_ctx *ctx, int n, const int *dirs, const int *metrics, const int *normal_modes, bool want_units);

int pccm_reduce_prefetch_many(pccm_ctx *ctx, int n, const int *dirs, const int *metrics, const int *normal_modes)
{
    CHECK_CTX(ctx);
    // per-leaf results cross PCIe only when a sharded exchange will need them: shards that start and end on whole chunks
    // exchange chunk sums (pccm_reduce_chunks_many), which the block results already hold
    bool units = false;
    if (ctx->sharded() && dirs)
        for (int k = 0; k < n && !units; ++k) {
            if (dirs[k] < 0 || dirs[k] > 2) continue;                 // reported by prefetch_many
            const NNResult &res = ctx->nn[dirs[k]];
            const Cloud &it = ctx->cloud[dirs[k] == PCCM_DIR_RIGHT ? 1 : 0];
            units = res.end > res.begin && (res.begin % kChunk != 0 || (res.end % kChunk != 0 && res.end != it.n));
        }
    return prefetch_many(ctx, n, dirs, metrics, normal_modes, units);
}

static int prefetch_many(pccm_ctx *ctx, int n, const int *dirs, const int *metrics, const int *normal_modes, bool want_units)
{
    if (n < 0 || n > 8 || (n > 0 && (!dirs || !metrics || !normal_modes))) return fail(PCCM_E_ARG, "1..8 requests expected");
    PointJobs pj;
    UnitJobs uj;
    pj.njobs = 0; pj.off[0] = 0;
    uj.njobs = 0; uj.uoff[0] = 0; uj.toff[0] = 0;
    ReduceSlot *fresh[8];
    int nfresh = 0;
    // first, whatever may change the layout of a direction's result records: a projection that was not fused into the
    // search needs the matched rows, and if the search left them out it is repeated (ensure_plain) -- with the rows, and
    // with the projection fused when the normals have arrived meanwhile.  Only then are the columns of this batch bound to
    // the records (a job bound earlier would read 16-byte records through a 32-byte stride).
    for (int k = 0; k < n; ++k) {
        if (dirs[k] < 0 || dirs[k] > 2) return fail(PCCM_E_ARG, "bad direction %d", dirs[k]);
        if (metrics[k] == PCCM_METRIC_D1 || dirs[k] == PCCM_DIR_SELF) continue;
        NNResult &res = ctx->nn[dirs[k]];
        if (!res.valid || !res.rec_valid || ((res.fused_mode == normal_modes[k] || res.rec_stride == 4 || res.rec_layout == 1) && !res.no_rows)) continue;
        if (slot_find(ctx, dirs[k], metrics[k], normal_modes[k], want_units)) continue;
        int rc = ensure_plain(ctx, res, true);
        if (rc) return rc;
    }
    for (int k = 0; k < n; ++k) {
        if (dirs[k] < 0 || dirs[k] > 2) return fail(PCCM_E_ARG, "bad direction %d", dirs[k]);
        if (slot_find(ctx, dirs[k], metrics[k], normal_modes[k], want_units)) continue;
        ReduceSlot *s = slot_free(ctx, fresh, nfresh);
        if (!s) return fail(PCCM_E_STATE, "no free reduction slot: more than 8 live columns in one batch");
        if (s->pending && !ctx->capturing && s->wait_ev) PCCM_HIP(hipEventSynchronize(s->wait_ev));
        s->pending = false;
        int rc = slot_prepare(ctx, *s, dirs[k], metrics[k], normal_modes[k], want_units, pj, uj);
        if (rc) return rc;
        s->pending = true;                 // so that slot_free/slot_find see it while the batch is assembled
        fresh[nfresh++] = s;
    }
    if (nfresh == 0) return PCCM_OK;
    int rc;
    if ((rc = launch_point_jobs(ctx, pj))) return rc;
    if ((rc = launch_unit_jobs(ctx, uj))) return rc;
    for (int k = 0; k < nfresh; ++k) {
        ReduceSlot &s = *fresh[k];
        if (ctx->capturing) {
            GraphOp op;
            op.kind = 2;
            op.dir = s.dir;
            op.slot = (int)(&s - ctx->slots);
            op.snap = s;
            ctx->cap_ops.push_back(op);
        } else {
            s.wait_ev = ctx->batch_ev;
        }
    }
    if (!ctx->capturing) PCCM_HIP(hipEventRecord(ctx->batch_ev, ctx->stream));      // one record for the whole batch
    return PCCM_OK;
}

int pccm_reduce_prefetch(pccm_ctx *ctx, int dir, int metric, int normal_mode)
{
    return pccm_reduce_prefetch_many(ctx, 1, &dir, &metric, &normal_mode);
}

int pccm_reduce(pccm_ctx *ctx, int dir, int metric, int normal_mode, double *xvec, double *minmax)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (!xvec || !minmax) return fail(PCCM_E_ARG, "null pointer");
    if (dir < 0 || dir > 2) return fail(PCCM_E_ARG, "bad direction %d", dir);
    ReduceSlot *s = slot_find(ctx, dir, metric, normal_mode, true);
    if (!s) {
        int rc = prefetch_many(ctx, 1, &dir, &metric, &normal_mode, true);
        if (rc) return rc;
        s = slot_find(ctx, dir, metric, normal_mode, true);
        if (!s) return fail(PCCM_E_STATE, "reduction slot lost");
    }
    if (s->wait_ev) PCCM_HIP(hipEventSynchronize(s->wait_ev));
    { int rce = check_device_errors(ctx); if (rce) return rce; }
    s->pending = false;
    const int64_t n = s->n_iter;
    const int64_t xlen = pccm_xvec_len(n);
    memset(xvec, 0, (size_t)xlen * sizeof(double));
    minmax[0] = INFINITY;
    minmax[1] = -INFINITY;
    const int64_t nunits = s->nunits;
    const int64_t nfull = n / kChunk, full_rows = nfull * kChunk;
    const double *usum = s->host, *umin = usum + nunits, *umax = usum + 2 * nunits;
    for (int64_t u = 0; u < nunits; ++u) {
        const int64_t row = s->begin + u * kLeaf;    // shard boundaries are multiples of kLeaf
        if (row < full_rows) xvec[row / kLeaf] = usum[u];
        if (umin[u] < minmax[0]) minmax[0] = umin[u];
        if (umax[u] > minmax[1]) minmax[1] = umax[u];
    }
    if (s->tail_n > 0)   // raw values of the last, partial 8192-row chunk that fall into this shard
        memcpy(xvec + nfull * (kChunk / kLeaf) + (s->t0 - full_rows), s->host + 3 * nunits + 3 * s->nblocks,
               (size_t)s->tail_n * sizeof(double));
    return PCCM_OK;
}

static double leaf_tree(const double *l, int cnt)
{
    if (cnt == 1) return l[0];
    return leaf_tree(l, cnt / 2) + leaf_tree(l + cnt / 2, cnt / 2);
}

int pccm_finish_sum(const double *xvec, int64_t n_iter, double *sum)
{
    if (!xvec || !sum || n_iter < 0) return fail(PCCM_E_ARG, "bad argument");
    const int64_t nfull = n_iter / kChunk, tail = n_iter % kChunk;
    const int lpc = kChunk / kLeaf;
    double s = 0.0;
    bool first = true;
    for (int64_t c = 0; c < nfull; ++c) {
        double cs = leaf_tree(xvec + c * lpc, lpc);
        s = first ? cs : s + cs;
        first = false;
    }
    if (tail) {
        double ts = np_pairwise_sum(xvec + nfull * lpc, tail);
        s = first ? ts : s + ts;
    }
    *sum = s;
    return PCCM_OK;
}

// one column's total from its slot (unsharded): np.sum = chunks of 8192 rows in sequence, each chunk = NumPy's pairwise tree
// = (tree of its first 32 leaves) + (tree of its last 32 leaves), and the GPU already finished both halves (begin = 0 here)
static int total_from_slot(pccm_ctx *ctx, int dir, int metric, int normal_mode, double out[3])
{
    if (dir < 0 || dir > 2) return fail(PCCM_E_ARG, "bad direction %d", dir);
    ReduceSlot *s = slot_find(ctx, dir, metric, normal_mode);
    if (!s) {
        int rc = pccm_reduce_prefetch(ctx, dir, metric, normal_mode);
        if (rc) return rc;
        s = slot_find(ctx, dir, metric, normal_mode);
        if (!s) return fail(PCCM_E_STATE, "reduction slot lost");
    }
    if (s->wait_ev) PCCM_HIP(hipEventSynchronize(s->wait_ev));
    { int rce = check_device_errors(ctx); if (rce) return rce; }
    s->pending = false;
    const int64_t n = s->n_iter, nunits = s->nunits, nblocks = s->nblocks;
    const int64_t nfull = n / kChunk;
    const double *bsum = s->host + 3 * nunits, *bmin = bsum + nblocks, *bmax = bsum + 2 * nblocks;
    double total = 0.0;
    bool first = true;
    for (int64_t c = 0; c < nfull; ++c) {
        const double cs = bsum[2 * c] + bsum[2 * c + 1];
        total = first ? cs : total + cs;
        first = false;
    }
    if (s->tail_n > 0) {
        const double ts = np_pairwise_sum(s->host + 3 * nunits + 3 * nblocks, s->tail_n);
        total = first ? ts : total + ts;
    }
    double mn = INFINITY, mx = -INFINITY;
    for (int64_t b = 0; b < nblocks; ++b) {
        mn = bmin[b] < mn ? bmin[b] : mn;
        mx = bmax[b] > mx ? bmax[b] : mx;
    }
    out[0] = total;
    out[1] = mn;
    out[2] = mx;
    return PCCM_OK;
}

int64_t pccm_cvec_len(int64_t n_iter)
{
    if (n_iter <= 0) return 0;
    return n_iter / kChunk + n_iter % kChunk;
}

// one column's chunk vector from its slot: a number per full 8192-row chunk this shard owns (the GPU finished both
// halves of the chunk's pairwise tree) + the raw values of the last, partial chunk; zero elsewhere
static int chunks_from_slot(pccm_ctx *ctx, int dir, int metric, int normal_mode, double *cvec, double minmax[2])
{
    if (dir < 0 || dir > 2) return fail(PCCM_E_ARG, "bad direction %d", dir);
    ReduceSlot *s = slot_find(ctx, dir, metric, normal_mode);
    if (!s) {
        int rc = prefetch_many(ctx, 1, &dir, &metric, &normal_mode, false);
        if (rc) return rc;
        s = slot_find(ctx, dir, metric, normal_mode);
        if (!s) return fail(PCCM_E_STATE, "reduction slot lost");
    }
    if (s->wait_ev) PCCM_HIP(hipEventSynchronize(s->wait_ev));
    { int rce = check_device_errors(ctx); if (rce) return rce; }
    s->pending = false;
    const int64_t n = s->n_iter, nunits = s->nunits, nblocks = s->nblocks;
    const int64_t nfull = n / kChunk, full_rows = nfull * kChunk;
    memset(cvec, 0, (size_t)pccm_cvec_len(n) * sizeof(double));
    minmax[0] = INFINITY;
    minmax[1] = -INFINITY;
    if (s->end <= s->begin) return PCCM_OK;                  // this rank owns no rows of the direction
    if (s->begin % kChunk != 0 || (s->end % kChunk != 0 && s->end != n))
        return fail(PCCM_E_STATE, "rows [%lld, %lld) do not start and end on 8192-row chunks: use pccm_reduce", (long long)s->begin,
                    (long long)s->end);
    const double *bsum = s->host + 3 * nunits, *bmin = bsum + nblocks, *bmax = bsum + 2 * nblocks;
    const int64_t c0 = s->begin / kChunk;
    const int64_t owned = ((s->end < full_rows ? s->end : full_rows) - s->begin) / kChunk;
    for (int64_t c = 0; c < owned; ++c) cvec[c0 + c] = bsum[2 * c] + bsum[2 * c + 1];
    if (s->tail_n > 0)
        memcpy(cvec + nfull + (s->t0 - full_rows), s->host + 3 * nunits + 3 * nblocks, (size_t)s->tail_n * sizeof(double));
    for (int64_t b = 0; b < nblocks; ++b) {
        minmax[0] = bmin[b] < minmax[0] ? bmin[b] : minmax[0];
        minmax[1] = bmax[b] > minmax[1] ? bmax[b] : minmax[1];
    }
    return PCCM_OK;
}

int pccm_reduce_chunks_many(pccm_ctx *ctx, int n, const int *dirs, const int *metrics, const int *normal_modes, double *cvecs, double *minmax)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (n < 0 || n > 8 || (n > 0 && (!dirs || !metrics || !normal_modes || !cvecs || !minmax))) return fail(PCCM_E_ARG, "1..8 requests expected");
    int rc = prefetch_many(ctx, n, dirs, metrics, normal_modes, false);       // whatever is not enqueued yet, in one batch
    if (rc) return rc;
    for (int k = 0; k < n; ++k) {
        const Cloud *it, *se;
        if ((rc = dir_clouds(ctx, dirs[k], &it, &se))) return rc;
        if ((rc = chunks_from_slot(ctx, dirs[k], metrics[k], normal_modes[k], cvecs, minmax + 2 * k))) return rc;
        cvecs += pccm_cvec_len(it->n);
    }
    return PCCM_OK;
}

int pccm_finish_chunks(const double *cvec, int64_t n_iter, double *sum)
{
    if (!cvec || !sum || n_iter < 0) return fail(PCCM_E_ARG, "bad argument");
    const int64_t nfull = n_iter / kChunk, tail = n_iter % kChunk;
    double s = 0.0;
    bool first = true;
    for (int64_t c = 0; c < nfull; ++c) {                    // np.sum: the chunks one after the other
        s = first ? cvec[c] : s + cvec[c];
        first = false;
    }
    if (tail) {
        const double ts = np_pairwise_sum(cvec + nfull, tail);
        s = first ? ts : s + ts;
    }
    *sum = s;
    return PCCM_OK;
}

int pccm_reduce_total(pccm_ctx *ctx, int dir, int metric, int normal_mode, double out[3])
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (!out) return fail(PCCM_E_ARG, "null pointer");
    if (ctx->sharded()) return fail(PCCM_E_STATE, "pccm_reduce_total needs the whole column on this GPU (world = 1)");
    return total_from_slot(ctx, dir, metric, normal_mode, out);
}

int pccm_reduce_total_many(pccm_ctx *ctx, int n, const int *dirs, const int *metrics, const int *normal_modes, double *out)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (n < 0 || n > 8 || (n > 0 && (!dirs || !metrics || !normal_modes || !out))) return fail(PCCM_E_ARG, "1..8 requests expected");
    if (ctx->sharded()) return fail(PCCM_E_STATE, "pccm_reduce_total_many needs the whole columns on this GPU (world = 1)");
    int rc = prefetch_many(ctx, n, dirs, metrics, normal_modes, false);       // whatever is not enqueued yet, in one batch
    if (rc) return rc;
    for (int k = 0; k < n; ++k)
        if ((rc = total_from_slot(ctx, dirs[k], metrics[k], normal_modes[k], out + 3 * k))) return rc;
    return PCCM_OK;
}

// Host helper for the colour metrics (metric.py:261-290): out[r] = M * rgb[r] for the BT.709 "ycc" (1) or
// the "yuv" (2) matrix.  The reference maps every row with np.matmul(M, c); on the authoring host
// (NumPy 2.2.6 / OpenBLAS dgemv) that evaluates each component as fma(m2*c2, fma(m0*c0, m1*c1)) -- pinned
// by tests/golden/*color* -- and that is the order used here.
int pccm_color_transform(const double *rgb, int64_t n, int scheme, double *out)
{
    static const double kYcc[9] = {0.2126, 0.7152, 0.0722, -0.1146, -0.3854, 0.5, 0.5, -0.4542, -0.0458};
    static const double kYuv[9] = {0.25, 0.5, 0.25, 1, 0, -1, -0.5, 1, -0.5};
    if (!rgb || !out || n < 0) return fail(PCCM_E_ARG, "bad argument");
    const double *m = scheme == 1 ? kYcc : (scheme == 2 ? kYuv : nullptr);
    if (!m) return fail(PCCM_E_ARG, "unknown colour scheme %d", scheme);
    for (int64_t r = 0; r < n; ++r) {
        const double c0 = rgb[3 * r], c1 = rgb[3 * r + 1], c2 = rgb[3 * r + 2];
        for (int i = 0; i < 3; ++i) out[3 * r + i] = fma(m[3 * i + 2], c2, fma(m[3 * i], c0, m[3 * i + 1] * c1));
    }
    return PCCM_OK;
}

// Host helper for the PCD reader (io.py): liblzf decompression (binary_compressed bodies).  Format: a control byte c;
// c < 32: c + 1 literal bytes follow; otherwise a back reference of length (c >> 5) + 2 (7 in the field: one more
// length byte is added) at distance (((c & 31) << 8) | next byte) + 1.
int pccm_lzf_decompress(const unsigned char *in, int64_t in_len, unsigned char *out, int64_t out_cap, int64_t *out_len)
{
    if (!in || !out || !out_len || in_len < 0 || out_cap < 0) return fail(PCCM_E_ARG, "bad argument");
    int64_t ip = 0, op = 0;
    while (ip < in_len) {
        const unsigned c = in[ip++];
        if (c < 32) {
            const int64_t run = (int64_t)c + 1;
            if (ip + run > in_len || op + run > out_cap) return fail(PCCM_E_ARG, "corrupt LZF stream (literal run)");
            memcpy(out + op, in + ip, (size_t)run);
            ip += run;
            op += run;
        } else {
            int64_t len = c >> 5;
            if (len == 7) {
                if (ip >= in_len) return fail(PCCM_E_ARG, "corrupt LZF stream (length)");
                len += in[ip++];
            }
            if (ip >= in_len) return fail(PCCM_E_ARG, "corrupt LZF stream (offset)");
            const int64_t ref = op - ((int64_t)(c & 31) << 8) - in[ip++] - 1;
            len += 2;
            if (ref < 0 || op + len > out_cap) return fail(PCCM_E_ARG, "corrupt LZF stream (back reference)");
            for (int64_t k = 0; k < len; ++k) out[op + k] = out[ref + k];      // may overlap: byte by byte
            op += len;
        }
    }
    *out_len = op;
    return PCCM_OK;
}

int pccm_drop_caches(pccm_ctx *ctx)
{
    CHECK_CTX(ctx);
    if (ctx->capturing) {
        GraphOp op;
        op.kind = 0;
        ctx->cap_ops.push_back(op);
    } else if (ctx->grid.key != 0) {
        // The caller drops a search structure it has used: it is going to search the same resident clouds AGAIN (a sequence
        // of reports, the bench's steps).  That is when the spatial order pays -- one more counting sort per cloud now, every
        // later rebuild reads coherent rows (Cloud::sp) -- and a pair that is searched once (the command line, `end_to_end`)
        // never pays for it.
        for (int k = 0; k < 2; ++k) {
            Cloud &c = ctx->cloud[k];
            if (c.n <= 0 || c.sp_valid || c.sp_tried) continue;
            c.sp_tried = true;
            int rc = spatial_order(ctx, c);
            if (rc) return rc;
            if (c.sp_valid) ctx->epoch++;                    // graphs captured before carry the row-order build
        }
    }
    grid_invalidate(ctx);
    return PCCM_OK;
}

// ---- hipGraph capture of a call sequence ------------------------------------------------------------------
// A report over resident clouds is ~45 small launches; issued eagerly the host cannot feed the GPU fast
// enough (MI355X_MICROARCH.md: ~3.5 us per launch).  pccm_graph_begin/end capture the sequence
// {pccm_drop_caches, pccm_nn, pccm_reduce_prefetch}* on the context's stream into a hipGraph;
// pccm_graph_launch replays it with one launch and re-applies the host-side bookkeeping of every call.
static void graph_free(GraphRec &g)
{
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
    g.exec = nullptr;
    g.graph = nullptr;
    g.valid = false;
    g.ops.clear();
}

static int graph_replay(pccm_ctx *ctx, GraphRec &g)
{

    for (auto &op : g.ops) {
        if (op.kind == 1) {
            ctx->nn_gen[op.dir]++;
            ctx->nn[op.dir].valid = true;
            ctx->nn[op.dir].rec_valid = op.rec_valid;
            ctx->nn[op.dir].plain_valid = op.plain_valid;
            ctx->nn[op.dir].fused_mode = op.fused_mode;
            ctx->nn[op.dir].rec_stride = op.rec_stride;
            ctx->nn[op.dir].rec_layout = op.rec_layout;
            ctx->nn[op.dir].no_rows = op.no_rows;
            ctx->nn[op.dir].plain_d2_valid = op.plain_valid;
        } else if (op.kind == 2) {
            ReduceSlot &s = ctx->slots[op.slot];
            if (s.pending && s.gen == ctx->nn_gen[s.dir] && s.wait_ev) PCCM_HIP(hipEventSynchronize(s.wait_ev));   // still in use by someone else
            s.dir = op.snap.dir; s.metric = op.snap.metric; s.mode = op.snap.mode;
            s.n_iter = op.snap.n_iter; s.begin = op.snap.begin; s.end = op.snap.end;
            s.nunits = op.snap.nunits; s.nblocks = op.snap.nblocks; s.has_units = op.snap.has_units;
            s.t0 = op.snap.t0; s.tail_n = op.snap.tail_n;
            s.gen = ctx->nn_gen[s.dir];
            s.pending = true;
            s.wait_ev = ctx->batch_ev;
        }
    }
    PCCM_HIP(hipGraphLaunch(g.exec, ctx->stream));
    PCCM_HIP(hipEventRecord(ctx->batch_ev, ctx->stream));          // one record for every reduction of the graph
    return PCCM_OK;
}

int pccm_graph_begin(pccm_ctx *ctx)
{
    CHECK_CTX(ctx);
    if (ctx->capturing) return fail(PCCM_E_STATE, "already capturing");
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &s : ctx->slots) s.pending = false;      // nothing outside the graph may be half-consumed
    ctx->cap_ops.clear();
    ctx->capture_failed = false;
    PCCM_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    ctx->capturing = true;
    return PCCM_OK;
}

int pccm_graph_end(pccm_ctx *ctx, int *graph_id)
{
    CHECK_CTX(ctx);
    if (!graph_id) return fail(PCCM_E_ARG, "null pointer");
    if (!ctx->capturing) return fail(PCCM_E_STATE, "pccm_graph_begin was not called");
    ctx->capturing = false;
    GraphRec g;
    hipError_t e = hipStreamEndCapture(ctx->stream, &g.graph);
    if (e != hipSuccess || ctx->capture_failed || !g.graph) {
        (void)hipGetLastError();
        if (g.graph) (void)hipGraphDestroy(g.graph);
        // whatever the captured calls recorded on the host never ran on the GPU
        for (int d = 0; d < 3; ++d) { ctx->nn[d].valid = false; ctx->nn_gen[d]++; }
        for (auto &s : ctx->slots) s.pending = false;
        grid_invalidate(ctx);
        return fail(PCCM_E_STATE, "graph capture failed (%s); the context is usable, results were invalidated",
                    e != hipSuccess ? hipGetErrorString(e) : "a captured call reported an error");
    }
    e = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g.graph);
        for (int d = 0; d < 3; ++d) { ctx->nn[d].valid = false; ctx->nn_gen[d]++; }
        for (auto &s : ctx->slots) s.pending = false;
        grid_invalidate(ctx);
        return fail(PCCM_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    }
    g.ops = ctx->cap_ops;
    for (auto &op : g.ops)
        if (op.kind == 1) {                                // the state the captured sequence leaves behind
            op.rec_valid = ctx->nn[op.dir].rec_valid;
            op.plain_valid = ctx->nn[op.dir].plain_valid;
            op.fused_mode = ctx->nn[op.dir].fused_mode;
            op.rec_stride = ctx->nn[op.dir].rec_stride;
            op.rec_layout = ctx->nn[op.dir].rec_layout;
            op.no_rows = ctx->nn[op.dir].no_rows;
        }
    g.epoch = ctx->epoch;
    g.valid = true;
    // the captured calls changed the host bookkeeping but nothing ran yet: run the graph once now
    PCCM_HIP(hipGraphLaunch(g.exec, ctx->stream));
    for (auto &op : g.ops)
        if (op.kind == 2) ctx->slots[op.slot].wait_ev = ctx->batch_ev;
    PCCM_HIP(hipEventRecord(ctx->batch_ev, ctx->stream));
    int id = -1;
    for (size_t k = 0; k < ctx->graphs.size(); ++k)
        if (!ctx->graphs[k].valid && !ctx->graphs[k].exec) { id = (int)k; break; }
    if (id < 0) { ctx->graphs.emplace_back(); id = (int)ctx->graphs.size() - 1; }
    ctx->graphs[id] = g;
    *graph_id = id;
    return PCCM_OK;
}

int pccm_graph_launch(pccm_ctx *ctx, int graph_id)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (graph_id < 0 || graph_id >= (int)ctx->graphs.size() || !ctx->graphs[graph_id].valid)
        return fail(PCCM_E_ARG, "unknown graph %d", graph_id);
    GraphRec &g = ctx->graphs[graph_id];
    if (g.epoch != ctx->epoch) {
        graph_free(g);
        return fail(PCCM_E_STATE, "graph %d is stale: inputs, shard or buffers changed since it was captured", graph_id);
    }
    return graph_replay(ctx, g);
}

int pccm_graph_destroy(pccm_ctx *ctx, int graph_id)
{
    CHECK_CTX(ctx);
    if (graph_id < 0 || graph_id >= (int)ctx->graphs.size()) return fail(PCCM_E_ARG, "unknown graph %d", graph_id);
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    graph_free(ctx->graphs[graph_id]);
    return PCCM_OK;
}

int pccm_ctx_reset(pccm_ctx *ctx)
{
    CHECK_CTX(ctx);
    if (ctx->capturing) {                              // an abandoned capture: end it, discard what it recorded
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(ctx->stream, &g);
        if (g) (void)hipGraphDestroy(g);
        ctx->capturing = false;
        ctx->capture_failed = false;
        ctx->cap_ops.clear();
    }
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    ctx->io_staged = false;                            // (the next owner of the context says what it wants)
    for (int k = 0; k < 2; ++k) {
        drop_cloud(ctx->cloud[k]);
        ctx->cloud[k].version++;
    }
    for (int d = 0; d < 3; ++d) {
        ctx->shard_rank[d] = 0;
        ctx->shard_world[d] = 1;
    }
    for (int d = 0; d < 3; ++d) { ctx->nn[d].valid = false; ctx->nn_gen[d]++; }
    for (auto &s : ctx->slots) s.pending = false;
    for (auto &g : ctx->graphs) graph_free(g);
    ctx->graphs.clear();
    ctx->epoch++;
    grid_invalidate(ctx);                              // the geometry decisions stay: the next pair may inherit them
    int rc = collect_spans(ctx);
    ctx->prof_on = false;
    for (int k = 0; k < PCCM_K_COUNT; ++k) { ctx->prof_ms[k] = 0.0; ctx->prof_n[k] = 0; }
    return rc;
}

int pccm_sync(pccm_ctx *ctx)
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    return check_device_errors(ctx);
}

int pccm_profile_enable(pccm_ctx *ctx, int on)
{
    CHECK_CTX(ctx);
    int rc = collect_spans(ctx);
    ctx->prof_on = on != 0;
    return rc;
}

int pccm_profile_reset(pccm_ctx *ctx)
{
    CHECK_CTX(ctx);
    int rc = collect_spans(ctx);
    for (int k = 0; k < PCCM_K_COUNT; ++k) {
        ctx->prof_ms[k] = 0.0;
        ctx->prof_n[k] = 0;
    }
    return rc;
}

int pccm_profile_get(pccm_ctx *ctx, int kernel_class, double *ms_total, int64_t *launches)
{
    CHECK_CTX(ctx);
    if (kernel_class < 0 || kernel_class >= PCCM_K_COUNT || !ms_total || !launches) return fail(PCCM_E_ARG, "bad argument");
    int rc = collect_spans(ctx);
    if (rc) return rc;
    *ms_total = ctx->prof_ms[kernel_class];
    *launches = ctx->prof_n[kernel_class];
    return PCCM_OK;
}

int pccm_nn_stats(pccm_ctx *ctx, int dir, int64_t out[3])
{
    CHECK_CTX(ctx);
    NOT_CAPTURING(ctx);
    if (!out) return fail(PCCM_E_ARG, "null pointer");
    const Cloud *it, *se;
    NNResult *res;
    const bool tail = (dir & PCCM_STATS_TAIL) != 0;
    dir &= ~PCCM_STATS_TAIL;
    int rc = need_nn(ctx, dir, &it, &se, &res);
    if (rc) return rc;
    if (tail) {
        uint32_t nt = 0;
        PCCM_HIP(hipMemcpyAsync(&nt, res->nflag_dev + 1, sizeof(nt), hipMemcpyDeviceToHost, ctx->stream));
        PCCM_HIP(hipStreamSynchronize(ctx->stream));
        out[0] = nt;
        out[1] = out[2] = 0;
        return PCCM_OK;
    }
    uint32_t nf = 0;
    PCCM_HIP(hipMemcpyAsync(&nf, res->nflag_dev, sizeof(nf), hipMemcpyDeviceToHost, ctx->stream));
    PCCM_HIP(hipStreamSynchronize(ctx->stream));
    out[0] = nf;
    out[1] = res->stats[1];
    out[2] = res->stats[2];
    return PCCM_OK;
}

}  // extern "C"
