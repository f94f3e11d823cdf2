// Context, device memory and per-kernel event profiling for libuwip.so.
#include "uwip_internal.hpp"
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <ctime>

UWIP_API const char *uwip_version(void) { return "uwip-mi355x 0.5 (gfx950)"; }

UWIP_API int uwip_device_count(int *count)
{
    if (!count) return UWIP_ERR_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return UWIP_ERR_HIP; }
    *count = n;
    return UWIP_OK;
}

// UWIP_TRACE_ALLOC=1: every device / pinned allocation of the library is reported on stderr with its address range,
// so that a faulting address in a GPU memory-access fault report can be attributed to a buffer (or to none of ours).
static bool trace_alloc()
{
    static const bool on = [] { const char *e = std::getenv("UWIP_TRACE_ALLOC"); return e && *e && *e != '0'; }();
    return on;
}
void uwip_trace_range(const uwip_ctx *ctx, const char *kind, const char *name, const void *p, size_t bytes)
{
    if (!trace_alloc()) return;
    std::fprintf(stderr, "[uwip alloc] ctx=%p stream=%p %s %s [%p, %p) %zu B\n", (const void *)ctx,
                 ctx ? (const void *)ctx->stream : nullptr, kind, name, p, (const void *)((const char *)p + bytes), bytes);
}

// Wait for an event WITHOUT burning a core.  Measured on this runtime (tools/ubench/wait_cpu.hip, ROCm 7.2): for a 100 ms
// kernel hipStreamSynchronize and hipEventSynchronize cost 100 ms of thread CPU each -- also on an event created with
// hipEventBlockingSync, which is ignored; only the process-wide hipSetDeviceFlags(hipDeviceScheduleBlockingSync) makes them
// sleep (0.9 ms CPU), and a library has no business changing its host's device flags.  So: poll hipEventQuery, a short
// spin first (a wait that is nearly over costs nothing extra), then nanosleep with a doubling interval up to
// `max_sleep_us` (0.6 ms of CPU per 100 ms at 200 us).
hipError_t uwip_event_wait(hipEvent_t ev, int max_sleep_us)
{
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {                                            // ~30 us of polling
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        if ((t.tv_sec - t0.tv_sec) * 1000000000ll + (t.tv_nsec - t0.tv_nsec) > 30000) break;
    }
    long ns = 20000;
    for (;;) {
        timespec ts{0, ns};
        nanosleep(&ts, nullptr);
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        ns = std::min<long>(ns * 2, (long)max_sleep_us * 1000);
    }
}

hipError_t uwip_stream_wait(uwip_ctx *ctx)
{
    static const bool env_spin = [] { const char *e = std::getenv("UWIP_SPIN_WAIT"); return e && *e && *e != '0'; }();
    if (ctx->spin_wait || env_spin) return hipStreamSynchronize(ctx->stream);
    if (!ctx->wait_ev) {
        hipError_t e = hipEventCreateWithFlags(&ctx->wait_ev, hipEventDisableTiming);
        if (e != hipSuccess) { ctx->wait_ev = nullptr; return e; }
    }
    hipError_t e = hipEventRecord(ctx->wait_ev, ctx->stream);
    if (e != hipSuccess) return e;
    return uwip_event_wait(ctx->wait_ev, 200);
}

UWIP_API int uwip_ctx_create_ex(int device, void *stream, unsigned flags, uwip_ctx **out)
{
    if (!out) return UWIP_ERR_INVALID;
    *out = nullptr;
    if (flags & ~(unsigned)(UWIP_CTX_STREAM_GIVEN | UWIP_CTX_SPIN_WAIT)) return UWIP_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return UWIP_ERR_HIP;  // no CPU fallback
    if (device < 0 || device >= n) return UWIP_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return UWIP_ERR_HIP;
    uwip_ctx *ctx = new (std::nothrow) uwip_ctx();
    if (!ctx) return UWIP_ERR_NOMEM;
    ctx->device = device;
    ctx->spin_wait = (flags & UWIP_CTX_SPIN_WAIT) != 0;
    if (stream || (flags & UWIP_CTX_STREAM_GIVEN)) {
        // a NULL handle with UWIP_CTX_STREAM_GIVEN is the device's default (null) stream itself
        ctx->stream = (hipStream_t)stream;
        ctx->own_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return UWIP_ERR_HIP;
        }
        ctx->own_stream = true;
    }
    *out = ctx;
    return UWIP_OK;
}

UWIP_API int uwip_ctx_create(int device, void *stream, uwip_ctx **out) { return uwip_ctx_create_ex(device, stream, 0u, out); }

UWIP_API int uwip_ctx_destroy(uwip_ctx *ctx)
{
    if (!ctx) return UWIP_OK;
    (void)hipSetDevice(ctx->device);
    (void)uwip_stream_wait(ctx);
    for (auto &p : ctx->prof_pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->wait_ev) (void)hipEventDestroy(ctx->wait_ev);
    for (auto &kv : ctx->ws) if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    for (auto &kv : ctx->hs) if (kv.second.ptr) (void)hipHostFree(kv.second.ptr);
    for (auto &kv : ctx->tables) if (kv.second.ptr) (void)hipFree(kv.second.ptr);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return UWIP_OK;
}

UWIP_API const char *uwip_last_error(const uwip_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

UWIP_API int uwip_sync(uwip_ctx *ctx)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    UWIP_HIP(ctx, uwip_stream_wait(ctx));
    return UWIP_OK;
}

UWIP_API int uwip_malloc(uwip_ctx *ctx, size_t bytes, void **d_ptr)
{
    if (!ctx || !d_ptr) return UWIP_ERR_INVALID;
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    *d_ptr = nullptr;
    if (bytes == 0) return UWIP_OK;
    UWIP_HIP(ctx, hipMalloc(d_ptr, bytes));
    uwip_trace_range(ctx, "device", "uwip_malloc", *d_ptr, bytes);
    return UWIP_OK;
}

UWIP_API int uwip_free(uwip_ctx *ctx, void *d_ptr)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    if (!d_ptr) return UWIP_OK;
    UWIP_HIP(ctx, uwip_stream_wait(ctx));
    UWIP_HIP(ctx, hipFree(d_ptr));
    return UWIP_OK;
}

UWIP_API int uwip_memcpy_h2d(uwip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    if (bytes == 0) return UWIP_OK;
    UWIP_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    UWIP_HIP(ctx, uwip_stream_wait(ctx));
    return UWIP_OK;
}

UWIP_API int uwip_memcpy_d2h(uwip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    if (bytes == 0) return UWIP_OK;
    UWIP_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    UWIP_HIP(ctx, uwip_stream_wait(ctx));
    return UWIP_OK;
}

UWIP_API int uwip_host_alloc(uwip_ctx *ctx, size_t bytes, void **h_ptr)
{
    if (!ctx || !h_ptr) return UWIP_ERR_INVALID;
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    *h_ptr = nullptr;
    if (bytes == 0) return UWIP_OK;
    UWIP_HIP(ctx, hipHostMalloc(h_ptr, bytes, hipHostMallocDefault));
    uwip_trace_range(ctx, "pinned", "uwip_host_alloc", *h_ptr, bytes);
    return UWIP_OK;
}

UWIP_API int uwip_host_free(uwip_ctx *ctx, void *h_ptr)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    if (!h_ptr) return UWIP_OK;
    UWIP_HIP(ctx, uwip_stream_wait(ctx));   // copies still queued on this context's stream
    UWIP_HIP(ctx, hipHostFree(h_ptr));
    return UWIP_OK;
}

UWIP_API int uwip_memcpy_h2d_async(uwip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    if (bytes == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, d_dst && h_src, "null buffer");
    UWIP_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return UWIP_OK;
}

UWIP_API int uwip_memcpy_d2h_async(uwip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    if (bytes == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, h_dst && d_src, "null buffer");
    UWIP_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return UWIP_OK;
}

void *uwip_ws(uwip_ctx *ctx, const char *name, size_t bytes)
{
    uwip_ws_buf &b = ctx->ws[name];
    if (b.bytes >= bytes && b.ptr) return b.ptr;
    if (b.ptr) {
        // grow: the old buffer may still be in use by queued kernels
        (void)uwip_stream_wait(ctx);
        (void)hipFree(b.ptr);
        b.ptr = nullptr; b.bytes = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes;
    hipError_t e = hipMalloc(&b.ptr, want);
    if (e != hipSuccess) {
        b.ptr = nullptr;
        ctx->fail(UWIP_ERR_NOMEM, "workspace hipMalloc", hipGetErrorString(e));
        return nullptr;
    }
    b.bytes = want;
    uwip_trace_range(ctx, "device", name, b.ptr, want);
    return b.ptr;
}

void *uwip_host_ws(uwip_ctx *ctx, const char *name, size_t bytes)
{
    uwip_ws_buf &b = ctx->hs[name];
    if (b.bytes >= bytes && b.ptr) return b.ptr;
    if (b.ptr) {
        (void)uwip_stream_wait(ctx);
        (void)hipHostFree(b.ptr);
        b.ptr = nullptr; b.bytes = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes;
    hipError_t e = hipHostMalloc(&b.ptr, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        b.ptr = nullptr;
        ctx->fail(UWIP_ERR_NOMEM, "pinned hipHostMalloc", hipGetErrorString(e));
        return nullptr;
    }
    b.bytes = want;
    uwip_trace_range(ctx, "pinned", name, b.ptr, want);
    return b.ptr;
}

const void *uwip_table_find(uwip_ctx *ctx, const std::string &key, size_t *bytes)
{
    auto it = ctx->tables.find(key);
    if (it == ctx->tables.end()) return nullptr;
    if (bytes) *bytes = it->second.bytes;
    return it->second.ptr;
}

const void *uwip_table_put(uwip_ctx *ctx, const std::string &key, const void *host, size_t bytes)
{
    uwip_ws_buf b;
    if (hipMalloc(&b.ptr, bytes ? bytes : 16) != hipSuccess) {
        ctx->fail(UWIP_ERR_NOMEM, "table hipMalloc");
        return nullptr;
    }
    b.bytes = bytes;
    // on the context's own stream (not the null stream, which another thread may be using), and drained before
    // `host` -- usually a local std::vector of the caller -- goes away
    if (bytes && (hipMemcpyAsync(b.ptr, host, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                  uwip_stream_wait(ctx) != hipSuccess)) {
        (void)hipFree(b.ptr);
        ctx->fail(UWIP_ERR_HIP, "table upload");
        return nullptr;
    }
    uwip_trace_range(ctx, "device", key.c_str(), b.ptr, bytes ? bytes : 16);
    ctx->tables[key] = b;
    return b.ptr;
}

int uwip_lds_optin(uwip_ctx *ctx, const char *name, const void *func, size_t bytes)
{
    if (ctx->lds_optin[name]) return UWIP_OK;
    UWIP_HIP(ctx, hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    ctx->lds_optin[name] = true;
    return UWIP_OK;
}

// ---- profiling ---------------------------------------------------------

static hipEvent_t take_event(uwip_ctx *ctx)
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

uwip_kscope::uwip_kscope(uwip_ctx *c, const char *name) : ctx(c)
{
    if (!ctx || !ctx->prof) return;
    auto it = ctx->prof_index.find(name);
    if (it == ctx->prof_index.end()) {
        rec = (int)ctx->prof_recs.size();
        ctx->prof_recs.push_back(uwip_prof_rec{name, 0.0, 0});
        ctx->prof_index[name] = rec;
    } else {
        rec = it->second;
    }
    a = take_event(ctx);
    b = take_event(ctx);
    if (!a || !b) { rec = -1; return; }
    (void)hipEventRecord(a, ctx->stream);
}

uwip_kscope::~uwip_kscope()
{
    if (rec < 0) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->prof_pending.push_back({rec, a, b});
    if (ctx->prof_pending.size() > 4096) (void)uwip_prof_flush(ctx);
}

int uwip_prof_flush(uwip_ctx *ctx)
{
    if (ctx->prof_pending.empty()) return UWIP_OK;
    UWIP_HIP(ctx, uwip_stream_wait(ctx));
    for (auto &p : ctx->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ctx->prof_recs[p.rec].total_ms += ms;
            ctx->prof_recs[p.rec].launches += 1;
        }
        ctx->event_pool.push_back(p.a);
        ctx->event_pool.push_back(p.b);
    }
    ctx->prof_pending.clear();
    return UWIP_OK;
}

UWIP_API int uwip_prof_enable(uwip_ctx *ctx, int on)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    int rc = uwip_prof_flush(ctx);
    ctx->prof = on != 0;
    return rc;
}

UWIP_API int uwip_prof_reset(uwip_ctx *ctx)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    int rc = uwip_prof_flush(ctx);
    ctx->prof_recs.clear();
    ctx->prof_index.clear();
    return rc;
}

UWIP_API int uwip_prof_count(uwip_ctx *ctx, int *n)
{
    if (!ctx || !n) return UWIP_ERR_INVALID;
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    int rc = uwip_prof_flush(ctx);
    *n = (int)ctx->prof_recs.size();
    return rc;
}

UWIP_API int uwip_prof_get(uwip_ctx *ctx, int index, char *name, size_t name_cap,
                           double *total_ms, uint64_t *launches)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    if (index < 0 || index >= (int)ctx->prof_recs.size()) return ctx->fail(UWIP_ERR_INVALID, "prof index out of range");
    const uwip_prof_rec &r = ctx->prof_recs[index];
    if (name && name_cap) {
        std::strncpy(name, r.name.c_str(), name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (total_ms) *total_ms = r.total_ms;
    if (launches) *launches = r.launches;
    return UWIP_OK;
}
