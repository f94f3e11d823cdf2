// Internal definitions shared by the HIP translation units of libuwip.so.
// gfx950 (MI355X) only: wave = 64 lanes, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>
#include "../../include/uwip.h"

#define UWIP_API extern "C" __attribute__((visibility("default")))

struct uwip_prof_rec {
    std::string name;
    double total_ms = 0.0;
    uint64_t launches = 0;
};

struct uwip_ws_buf {
    void *ptr = nullptr;
    size_t bytes = 0;
};

struct uwip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // named, grow-only device workspaces: no hipMalloc in steady state
    std::map<std::string, uwip_ws_buf> ws;
    // named pinned host staging buffers
    std::map<std::string, uwip_ws_buf> hs;
    // immutable device tables keyed by their geometry (strip / cell lists)
    std::map<std::string, uwip_ws_buf> tables;
    // kernels whose > 64 KiB dynamic-LDS opt-in (hipFuncSetAttribute) has been made on this device
    std::map<std::string, bool> lds_optin;
    int ov_last_frames = 0;        // batch size of the most recent uwip_overlap_detect (debug taps)
    // the parameters the most recent uwip_aclahe_auto_ex chose: on the host (the synchronous forms) or still on the device
    // (UWIP_ACLAHE_ASYNC: workspace "auto.par", [n][4] int32) -- uwip_aclahe_last_params
    int aclahe_last_n = 0;
    bool aclahe_last_on_device = false;
    std::vector<int32_t> aclahe_last_host;      // [n][2] = BS, CL
    // the pair list uwip_overlap_match last uploaded (a stream of batches sends the same one every time: no re-upload,
    // and no host wait for the staging buffer)
    std::vector<int32_t> ov_pairs_host;
    const void *ov_pairs_dev = nullptr;
    // profiling
    bool prof = false;
    std::vector<uwip_prof_rec> prof_recs;
    std::map<std::string, int> prof_index;
    struct pending_t { int rec; hipEvent_t a, b; };
    std::vector<pending_t> prof_pending;
    std::vector<hipEvent_t> event_pool;
    // host waits: an event recorded behind the stream's work and polled with sleeps in between, so the calling thread
    // does not burn a core until the stream has drained (hipStreamSynchronize / hipEventSynchronize spin on this runtime,
    // blocking-sync events included: eight sub-batch threads of a rank burned eight cores, measured host_cpu_s_per_step
    // 1.59 s per 0.177 s step); UWIP_CTX_SPIN_WAIT keeps the spinning wait
    hipEvent_t wait_ev = nullptr;
    bool spin_wait = false;

    int fail(int code, const char *what, const char *detail = nullptr)
    {
        err = what;
        if (detail) { err += ": "; err += detail; }
        return code;
    }
};

// Wait on the host until everything queued on the context's stream has finished (sleeping, see uwip_ctx::wait_ev).
hipError_t uwip_stream_wait(uwip_ctx *ctx);
hipError_t uwip_event_wait(hipEvent_t ev, int max_sleep_us);      // hipEventQuery + nanosleep back-off (ctx.hip)
void *uwip_ws(uwip_ctx *ctx, const char *name, size_t bytes);       // nullptr on failure (ctx->err set)
void *uwip_host_ws(uwip_ctx *ctx, const char *name, size_t bytes);  // pinned host
// Cached immutable device table: uploaded once (blocking) the first time `key` is seen.
const void *uwip_table_find(uwip_ctx *ctx, const std::string &key, size_t *bytes);
const void *uwip_table_put(uwip_ctx *ctx, const std::string &key, const void *host, size_t bytes);
int uwip_prof_flush(uwip_ctx *ctx);
// UWIP_TRACE_ALLOC=1: report an allocation's address range on stderr (attributing a GPU fault address to a buffer)
void uwip_trace_range(const uwip_ctx *ctx, const char *kind, const char *name, const void *p, size_t bytes);
// clahe.hip: in-place 8-bit BGR -> HSV -> BGR (an HSV letter of histretch)
int uwip_hsv_roundtrip(uwip_ctx *ctx, const uwip_batch_u8 *img);
// winfilter15.hip: 15x15 window max and/or min of interleaved 3-channel u8 frames -> planar [F][3][H][W]
bool uwip_winfilter15_ok(const uint8_t *img, size_t step, size_t fs, int H, int W, int w);
// stats (optional, both filters only): stats[f*stride + {0..3}] = min, max of all channels, min, max of channel 2,
// by atomicMin / atomicMax onto the caller's 255 / 0 initial values
int uwip_winfilter15(uwip_ctx *ctx, const uint8_t *img, size_t step, size_t fs, int F, int H, int W, uint8_t *out_max,
                     uint8_t *out_min, int *stats = nullptr, int stats_stride = 0, unsigned long long *redsum = nullptr);
// guided_filter_ws.hip: the wave-strip guided filter (guide u8 x3, P [F][np][H][W] -> Q, AB [F*np][4][H][W] scratch)
// Optional 8-bit source of p (bgdehaze's transmission): plane ip of frame f is max(1 - normv(m) / B_ip, tmin) of the u8
// plane m = planes[(f * nplanes + ip)][H][W], and 1 where the w x w window around the pixel leaves the image;
// B_ip = sc[f * sc_stride + b_off + ip], normv by the frame's gnorm pair.
struct uwip_gf_pu8 {
    const uint8_t *planes = nullptr;
    int nplanes = 0;
    const double *sc = nullptr;
    int sc_stride = 0, b_off = 0;
    double tmin = 0.0;
    int w = 0, pad = 0;
};
// Optional fused scene recovery (bgdehaze D5) in the second kernel: Q receives J_ip = (normv(I_ip) - B_ip) / q + B_ip
// instead of q, and part[(z * nb + block) * 3 + {0,1,2}] the per-block min / max / sum of J (z = f * np + ip; `part` and
// `nb` are filled in by uwip_gf_wave_strip).  Aligned path only.
struct uwip_gf_recover {
    const double *sc = nullptr;
    int sc_stride = 0, b_off = 0;
    double *part = nullptr;
    int nb = 0;
};
// true when the 8-bit p source can be used for this geometry (aligned rows, W and r multiples of 4, np = 2)
bool uwip_gf_pu8_ok(const uint8_t *guide, size_t step, size_t fs, const uint8_t *planes, int np, int W, int r);
int uwip_gf_wave_strip(uwip_ctx *ctx, const uint8_t *guide, size_t step, size_t fs, const int *gnorm, int gstride,
                       const double *P, double *Q, double *AB, int F, int np, int H, int W, int r, double eps,
                       const uwip_gf_pu8 *pu8 = nullptr, uwip_gf_recover *rec = nullptr);
// Test / measurement hooks (UWIP_ACLAHE_TEST_FORCE_CL, UWIP_DIAG_GF_ONLY) are dead unless the process was started with
// UWIP_TEST_HOOKS=1 in its environment: read ONCE, at the first call; a product process never looks at the hook variables.
inline bool uwip_test_hooks()
{
    static const bool on = [] { const char *e = std::getenv("UWIP_TEST_HOOKS"); return e && *e == '1'; }();
    return on;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (context, kernel)
int uwip_lds_optin(uwip_ctx *ctx, const char *name, const void *func, size_t bytes);

#define UWIP_HIP(ctx, expr)                                                          \
    do {                                                                             \
        hipError_t e__ = (expr);                                                     \
        if (e__ != hipSuccess) return (ctx)->fail(UWIP_ERR_HIP, #expr, hipGetErrorString(e__)); \
    } while (0)

#define UWIP_REQUIRE(ctx, cond, msg)                                   \
    do {                                                               \
        if (!(cond)) return (ctx)->fail(UWIP_ERR_INVALID, msg, #cond); \
    } while (0)

// RAII launch bracket: when profiling is on, records events around a kernel.
struct uwip_kscope {
    uwip_ctx *ctx;
    int rec = -1;
    hipEvent_t a = nullptr, b = nullptr;
    uwip_kscope(uwip_ctx *c, const char *name);
    ~uwip_kscope();
};

// First statement of every entry point that allocates, copies or launches: makes the context's device the calling
// thread's current one (hipSetDevice is thread-local state; a second host thread or a device != 0 would otherwise
// hipMalloc / launch on whatever device that thread last used).
static inline int uwip_enter(uwip_ctx *ctx)
{
    if (!ctx) return UWIP_ERR_INVALID;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return ctx->fail(UWIP_ERR_HIP, "hipSetDevice", hipGetErrorString(e));
    return UWIP_OK;
}

static inline int uwip_check_batch(uwip_ctx *ctx, const uwip_batch_u8 *b, int channels /*0=any*/)
{
    if (int rc0 = uwip_enter(ctx)) return rc0;
    UWIP_REQUIRE(ctx, b != nullptr, "null batch");
    UWIP_REQUIRE(ctx, b->rows >= 0 && b->cols >= 0 && b->frames >= 0, "negative extent");
    UWIP_REQUIRE(ctx, b->channels == 1 || b->channels == 3, "channels must be 1 or 3");
    if (channels) UWIP_REQUIRE(ctx, b->channels == channels, "wrong channel count");
    UWIP_REQUIRE(ctx, b->step >= (size_t)b->cols * b->channels, "step smaller than a row");
    if (b->frames > 1) UWIP_REQUIRE(ctx, b->frame_stride >= b->step * (size_t)b->rows, "frame_stride smaller than a frame");
    if ((size_t)b->rows * b->cols * b->frames > 0) UWIP_REQUIRE(ctx, b->data != nullptr, "null data");
    UWIP_REQUIRE(ctx, (uint64_t)b->rows * (uint64_t)b->cols < (1ull << 31), "frame too large");
    return UWIP_OK;
}

static inline bool uwip_batch_empty(const uwip_batch_u8 *b)
{
    return (size_t)b->rows * b->cols * b->frames == 0;
}

static inline unsigned uwip_cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
