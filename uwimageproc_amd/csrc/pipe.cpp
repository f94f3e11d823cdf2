// uwip_pipe: the per-frame chain  bgdehaze -> histretch -> aclahe -> videostrip-overlap  as one object over the C ABI
// (include/uwip.h, "the whole per-frame chain").  The reference runs the four tools back to back over files:
// modules/bgdehaze/main.py:14-20, modules/histretch/src/histretch.cpp:217-254, modules/aclahe/src/aclahe.cpp:152-218 with
// python/ACLAHE.py:9-129 + python/main.py:19-20, modules/videostrip/src/main.cpp:300-394.  Everything here is host code that
// calls the library's own entry points; what it adds is the state the chain carries between steps (rounds 1-4 kept that in
// Python, uwimageproc_amd/pipeline.py): the feature-slot carry, the throttle, the double-buffered host front end.
#include "uwip_internal.hpp"
#include <cstring>
#include <deque>

struct uwip_pipe {
    uwip_ctx *ctx = nullptr;
    uwip_pipe_config cfg{};
    std::string err;
    size_t frame_bytes = 0, plane_bytes = 0;
    uint8_t *v = nullptr, *v_out = nullptr;            // [F][H][W]: V of the stretched frames, and its CLAHE
    uwip_features *feats = nullptr;                    // slot 0 = the previous batch's last frame, 1..F = this batch
    bool have_prev = false;
    std::vector<int32_t> pair_q, pair_t;
    // resident form: events of the steps still queued
    std::deque<hipEvent_t> inflight;
    std::vector<hipEvent_t> ev_pool;
    // host-buffer form
    uwip_copier *copier = nullptr;
    bool own_copier = false;
    uint8_t *staging = nullptr;
    bool own_staging = false;
    uint8_t *src[2] = {nullptr, nullptr}, *work[2] = {nullptr, nullptr};
    float *ratio[2] = {nullptr, nullptr};
    int32_t *info[2] = {nullptr, nullptr};
    uint64_t t_up[2] = {0, 0}, t_dn[2] = {0, 0}, t_rt[2] = {0, 0};
    uint64_t k = 0;
    const void *pending = nullptr;                     // host buffer whose upload into src[k % 2] has been requested
    // most recent step's results (uwip_pipe_device_results)
    const uint8_t *last_frames = nullptr;
    const float *last_ratio = nullptr;
    const int32_t *last_info = nullptr;

    int fail(int code, const char *what)
    {
        err = what;
        return code;
    }
    int from_ctx(int rc)
    {
        if (rc) err = ctx->err;
        return rc;
    }
};

namespace {

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

bool cfg_ok(const uwip_pipe_config &c)
{
    return c.frames >= 1 && c.rows >= 1 && c.cols >= 1 && c.max_in_flight >= 1 && c.videoWidth >= 0 && c.videoHeight >= 0 &&
           std::memchr(c.letters, 0, sizeof c.letters) != nullptr && (uint64_t)c.rows * (uint64_t)c.cols < (1ull << 31);
}

uwip_batch_u8 batch_of(void *data, const uwip_pipe_config &c, int channels, int frames = -1)
{
    uwip_batch_u8 b;
    b.data = data;
    b.step = (size_t)c.cols * channels;
    b.frame_stride = b.step * c.rows;
    b.rows = c.rows; b.cols = c.cols; b.channels = channels;
    b.frames = frames < 0 ? c.frames : frames;
    return b;
}

int check_io(uwip_pipe *p, const uwip_batch_u8 *b, const char *what)
{
    if (!b || b->rows != p->cfg.rows || b->cols != p->cfg.cols || b->frames != p->cfg.frames || b->channels != 3 || !b->data)
        return p->fail(UWIP_ERR_INVALID, what);
    return UWIP_OK;
}

// the chain itself (pipeline stage definitions: uwip.h)
int run_stages(uwip_pipe *p, unsigned stages, const uwip_batch_u8 *in, const uwip_batch_u8 *out, float *d_ratio, int32_t *d_info)
{
    uwip_ctx *ctx = p->ctx;
    const uwip_pipe_config &c = p->cfg;
    int rc = UWIP_OK;
    if ((stages & UWIP_PIPE_DEHAZE) && (stages & UWIP_PIPE_HISTRETCH))
        // chained: the kernel that writes the dehazed bytes hands the stretch its histogram
        rc = uwip_dehaze_histretch(ctx, in, out, c.w, (int)c.dehaze_flags, c.letters, c.lo, c.hi, c.histretch_flags);
    else if (stages & UWIP_PIPE_DEHAZE)
        rc = uwip_dehaze(ctx, in, out, c.w, (int)c.dehaze_flags, nullptr, nullptr, nullptr);
    else if (stages & UWIP_PIPE_HISTRETCH)
        rc = uwip_histretch_ex(ctx, out, c.letters, c.lo, c.hi, c.histretch_flags);
    if (rc) return p->from_ctx(rc);
    if (stages & UWIP_PIPE_ACLAHE) {
        const uwip_batch_u8 vb = batch_of(p->v, c, 1), ob = batch_of(p->v_out, c, 1);
        // V of HSV (aclahe.cpp:152-154) -> sweep, parameter choice, final CLAHE (ACLAHE.py:9-129, python/main.py:19-20) ->
        // back to BGR (the stub of aclahe.cpp:216)
        if ((rc = uwip_bgr_to_v(ctx, out, &vb))) return p->from_ctx(rc);
        if ((rc = uwip_aclahe_auto_ex(ctx, &vb, &ob, c.residual_rule, c.aclahe_flags, nullptr, nullptr))) return p->from_ctx(rc);
        if ((rc = uwip_hsv_replace_v(ctx, out, &ob, out))) return p->from_ctx(rc);
    }
    if (stages & UWIP_PIPE_OVERLAP) {
        if (!d_ratio) return p->fail(UWIP_ERR_INVALID, "the overlap stage needs d_ratio");
        if (!p->have_prev) {
            // first batch: frame 0 is its own key frame (main.cpp:284-297 takes the first frame as key frame)
            uwip_batch_u8 first = *out;
            first.frames = 1;
            if ((rc = uwip_overlap_detect_ex(ctx, &first, p->feats, 0, c.detect_flags))) return p->from_ctx(rc);
            p->have_prev = true;
        } else {
            // the previous batch's last frame becomes the key frame of this batch's first frame (kframe's cached
            // keypoints / descriptors, videostrip.hpp:62-68)
            if ((rc = uwip_features_copy(ctx, p->feats, c.frames, p->feats, 0))) return p->from_ctx(rc);
        }
        if ((rc = uwip_overlap_detect_ex(ctx, out, p->feats, 1, c.detect_flags))) return p->from_ctx(rc);
        rc = uwip_overlap_match_ex(ctx, p->feats, p->feats, p->pair_q.data(), p->pair_t.data(), c.frames,
                                   c.videoWidth ? c.videoWidth : c.cols, c.videoHeight ? c.videoHeight : c.rows, c.seed, c.match_flags,
                                   d_ratio, d_info, nullptr, nullptr, nullptr);
        if (rc) return p->from_ctx(rc);
    }
    return UWIP_OK;
}

int ensure_host_state(uwip_pipe *p)
{
    const uwip_pipe_config &c = p->cfg;
    if (!p->copier) {
        int rc = uwip_copier_create(p->ctx->device, &p->copier);
        if (rc) return p->fail(rc, "uwip_copier_create failed");
        p->own_copier = true;
    }
    if (!p->src[0]) {
        uint8_t *base = (uint8_t *)c.d_staging;
        if (!base) {
            void *d = nullptr;
            int rc = uwip_malloc(p->ctx, uwip_pipe_staging_bytes(&c), &d);
            if (rc) return p->from_ctx(rc);
            p->staging = base = (uint8_t *)d;
            p->own_staging = true;
        }
        const size_t fb = p->frame_bytes;
        p->src[0] = base; p->src[1] = base + fb; p->work[0] = base + 2 * fb; p->work[1] = base + 3 * fb;
        uint8_t *r = base + align256(4 * fb);
        p->ratio[0] = (float *)r; p->ratio[1] = (float *)r + c.frames;
        uint8_t *i = r + align256(2 * sizeof(float) * (size_t)c.frames);
        p->info[0] = (int32_t *)i; p->info[1] = (int32_t *)i + 8 * (size_t)c.frames;
    }
    return UWIP_OK;
}

int copier_rc(uwip_pipe *p, int rc)
{
    if (rc) p->err = std::string("copier: ") + uwip_copier_last_error(p->copier);
    return rc;
}

}  // namespace

UWIP_API int uwip_pipe_config_default(uwip_pipe_config *cfg, int frames, int rows, int cols)
{
    if (!cfg) return UWIP_ERR_INVALID;
    std::memset(cfg, 0, sizeof *cfg);
    cfg->frames = frames; cfg->rows = rows; cfg->cols = cols;
    std::strcpy(cfg->letters, "RGB");                   // histretch -c=RGB (histretch.cpp:154)
    cfg->lo = 2; cfg->hi = 98;                          // histretch.cpp:154,236,247
    cfg->w = 15;                                        // main.py:28-29
    cfg->dehaze_flags = UWIP_DEHAZE_FULL;               // adaptiveExp_map as written: S unguarded (BGDehaze.py:83)
    cfg->histretch_flags = 0;
    cfg->residual_rule = 0;                             // OpenCV 3.4.x (INSTALL.md:47-63)
    cfg->aclahe_flags = UWIP_ACLAHE_PREFILTER | UWIP_ACLAHE_ASYNC;     // ParametrosACLAHE (ACLAHE.py:15), no host wait
    cfg->detect_flags = 0;
    cfg->match_flags = 0;                               // >= 4 good matches (videostrip.cpp:252-272)
    cfg->videoWidth = 0; cfg->videoHeight = 0;          // = cols, rows (main.cpp:238-239)
    cfg->seed = 1;
    cfg->max_in_flight = 2;
    cfg->d_staging = nullptr;
    return UWIP_OK;
}

UWIP_API size_t uwip_pipe_staging_bytes(const uwip_pipe_config *cfg)
{
    if (!cfg || !cfg_ok(*cfg)) return 0;
    const size_t fb = (size_t)cfg->frames * cfg->rows * cfg->cols * 3;
    return align256(4 * fb) + align256(2 * sizeof(float) * (size_t)cfg->frames) + align256(2 * 8 * sizeof(int32_t) * (size_t)cfg->frames);
}

UWIP_API int uwip_pipe_create(uwip_ctx *ctx, const uwip_pipe_config *cfg, uwip_copier *copier, uwip_pipe **out)
{
    if (!out) return UWIP_ERR_INVALID;
    *out = nullptr;
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    UWIP_REQUIRE(ctx, cfg && cfg_ok(*cfg), "bad pipe configuration");
    UWIP_REQUIRE(ctx, (cfg->aclahe_flags & ~(unsigned)(UWIP_ACLAHE_PREFILTER | UWIP_ACLAHE_HOST_SELECT | UWIP_ACLAHE_ASYNC)) == 0 &&
                          (cfg->dehaze_flags & ~(unsigned)(UWIP_DEHAZE_FULL | UWIP_DEHAZE_GUARD_S)) == 0, "unknown stage flag");
    uwip_pipe *p = new (std::nothrow) uwip_pipe;
    if (!p) return ctx->fail(UWIP_ERR_NOMEM, "uwip_pipe");
    p->ctx = ctx;
    p->cfg = *cfg;
    p->copier = copier;
    p->plane_bytes = (size_t)cfg->frames * cfg->rows * cfg->cols;
    p->frame_bytes = p->plane_bytes * 3;
    void *d = nullptr;
    int rc = uwip_malloc(ctx, 2 * p->plane_bytes, &d);
    if (!rc) {
        p->v = (uint8_t *)d;
        p->v_out = p->v + p->plane_bytes;
        rc = uwip_features_create(ctx, cfg->frames + 1, &p->feats);
    }
    if (rc) {
        uwip_free(ctx, p->v);
        delete p;
        return rc;
    }
    p->pair_q.resize(cfg->frames);
    p->pair_t.resize(cfg->frames);
    for (int i = 0; i < cfg->frames; ++i) { p->pair_q[i] = i + 1; p->pair_t[i] = i; }
    *out = p;
    return UWIP_OK;
}

UWIP_API int uwip_pipe_sync(uwip_pipe *p)
{
    if (!p) return UWIP_ERR_INVALID;
    int rc = uwip_sync(p->ctx);
    if (rc) return p->from_ctx(rc);
    for (hipEvent_t e : p->inflight) p->ev_pool.push_back(e);
    p->inflight.clear();
    if (p->copier)
        for (int s = 0; s < 2; ++s) {
            const uint64_t t[3] = {p->t_up[s], p->t_dn[s], p->t_rt[s]};
            for (uint64_t x : t)
                if (x && (rc = uwip_copier_wait(p->copier, x))) return copier_rc(p, rc);
        }
    return UWIP_OK;
}

UWIP_API int uwip_pipe_destroy(uwip_pipe *p)
{
    if (!p) return UWIP_OK;
    (void)uwip_pipe_sync(p);
    (void)hipSetDevice(p->ctx->device);
    for (hipEvent_t e : p->ev_pool) (void)hipEventDestroy(e);
    if (p->own_copier) uwip_copier_destroy(p->copier);
    if (p->own_staging) uwip_free(p->ctx, p->staging);
    uwip_features_destroy(p->feats);
    uwip_free(p->ctx, p->v);
    delete p;
    return UWIP_OK;
}

UWIP_API const char *uwip_pipe_last_error(const uwip_pipe *p) { return p ? p->err.c_str() : "null pipe"; }

UWIP_API int uwip_pipe_stages(uwip_pipe *p, unsigned stages, const uwip_batch_u8 *in, const uwip_batch_u8 *out, float *d_ratio,
                              int32_t *d_info)
{
    if (!p) return UWIP_ERR_INVALID;
    if (int rc_e = uwip_enter(p->ctx)) return p->from_ctx(rc_e);
    if (stages & ~UWIP_PIPE_ALL) return p->fail(UWIP_ERR_INVALID, "unknown stage");
    if (int rc = check_io(p, out, "`out` is not a 3-channel batch of the configured geometry")) return rc;
    if (stages & UWIP_PIPE_DEHAZE) {
        if (int rc = check_io(p, in, "`in` is not a 3-channel batch of the configured geometry")) return rc;
        if (in->data == out->data) return p->fail(UWIP_ERR_INVALID, "the dehaze stage is not in place: `in` and `out` must be distinct buffers");
    }
    const int rc = run_stages(p, stages, in, out, d_ratio, d_info);
    if (!rc) { p->last_frames = (const uint8_t *)out->data; p->last_ratio = d_ratio; p->last_info = d_info; }
    return rc;
}

UWIP_API int uwip_pipe_step(uwip_pipe *p, const uwip_batch_u8 *in, const uwip_batch_u8 *out, float *d_ratio, int32_t *d_info)
{
    if (!p) return UWIP_ERR_INVALID;
    if (int rc_e = uwip_enter(p->ctx)) return p->from_ctx(rc_e);
    // Nothing in a step waits on the host, so a caller that loops would queue steps without bound and end up spinning inside
    // the runtime once its hardware queue is full: the wait for the oldest queued step polls its event and sleeps in between.
    while ((int)p->inflight.size() >= p->cfg.max_in_flight) {
        hipEvent_t e = p->inflight.front();
        const hipError_t he = uwip_event_wait(e, 1000);
        if (he != hipSuccess) return p->fail(UWIP_ERR_HIP, hipGetErrorString(he));
        p->inflight.pop_front();
        p->ev_pool.push_back(e);
    }
    int rc = uwip_pipe_stages(p, UWIP_PIPE_ALL, in, out, d_ratio, d_info);
    if (rc) return rc;
    hipEvent_t e = nullptr;
    if (!p->ev_pool.empty()) { e = p->ev_pool.back(); p->ev_pool.pop_back(); }
    else if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return p->fail(UWIP_ERR_HIP, "hipEventCreate");
    if (hipEventRecord(e, p->ctx->stream) != hipSuccess) { p->ev_pool.push_back(e); return p->fail(UWIP_ERR_HIP, "hipEventRecord"); }
    p->inflight.push_back(e);
    return UWIP_OK;
}

UWIP_API int uwip_pipe_step_host(uwip_pipe *p, const void *h_in, void *h_out, float *h_ratio, const void *h_prefetch, uint64_t tickets[3])
{
    if (!p) return UWIP_ERR_INVALID;
    if (int rc_e = uwip_enter(p->ctx)) return p->from_ctx(rc_e);
    if (!h_in || !h_out) return p->fail(UWIP_ERR_INVALID, "null host buffer");
    int rc = ensure_host_state(p);
    if (rc) return rc;
    // Two source and two result buffers in device memory, so that batch k + 1 arrives and batch k leaves while the kernels of
    // batch k / k + 1 run.  Every hand-over is a ticket waited for on the host: no stream ever waits for another one on the
    // device (a barrier packet behind a DMA copy stalls whatever shares its hardware queue; DESIGN.md, host-buffer mode).
    const uint64_t k = p->k;
    const int slot = (int)(k & 1);
    const size_t fb = p->frame_bytes;
    if (p->pending != h_in) {
        // nobody prefetched this batch.  src[slot] was last read by batch k - 2's dehaze, which precedes everything queued now
        rc = uwip_copier_upload(p->copier, k >= 2 ? p->ctx : nullptr, p->src[slot], h_in, fb, &p->t_up[slot]);
        if (rc) return copier_rc(p, rc);
    }
    p->pending = nullptr;
    const uint64_t t_in = p->t_up[slot];
    if (h_prefetch) {
        // src[1 - slot] was last read by batch k - 1's dehaze: the upload starts when the stream has finished batch k - 1
        rc = uwip_copier_upload(p->copier, k >= 1 ? p->ctx : nullptr, p->src[1 - slot], h_prefetch, fb, &p->t_up[1 - slot]);
        if (rc) return copier_rc(p, rc);
        p->pending = h_prefetch;
    }
    if ((rc = uwip_copier_wait(p->copier, t_in))) return copier_rc(p, rc);                 // batch k is in device memory
    if ((rc = uwip_copier_wait(p->copier, p->t_dn[slot]))) return copier_rc(p, rc);        // batch k - 2's frames have left work[slot]
    if ((rc = uwip_copier_wait(p->copier, p->t_rt[slot]))) return copier_rc(p, rc);        // ... and its ratios
    const uwip_batch_u8 in = batch_of(p->src[slot], p->cfg, 3), out = batch_of(p->work[slot], p->cfg, 3);
    if ((rc = run_stages(p, UWIP_PIPE_DEHAZE | UWIP_PIPE_HISTRETCH | UWIP_PIPE_ACLAHE, &in, &out, nullptr, nullptr))) return rc;
    // the enhanced frames are final here (the overlap stage only reads them): they leave under its kernels
    if ((rc = uwip_copier_download(p->copier, p->ctx, h_out, p->work[slot], fb, &p->t_dn[slot]))) return copier_rc(p, rc);
    if ((rc = run_stages(p, UWIP_PIPE_OVERLAP, nullptr, &out, p->ratio[slot], p->info[slot]))) return rc;
    p->t_rt[slot] = 0;
    if (h_ratio && (rc = uwip_copier_download(p->copier, p->ctx, h_ratio, p->ratio[slot], sizeof(float) * (size_t)p->cfg.frames, &p->t_rt[slot])))
        return copier_rc(p, rc);
    p->last_frames = p->work[slot]; p->last_ratio = p->ratio[slot]; p->last_info = p->info[slot];
    if (tickets) { tickets[0] = t_in; tickets[1] = p->t_dn[slot]; tickets[2] = p->t_rt[slot]; }
    p->k = k + 1;
    return UWIP_OK;
}

UWIP_API int uwip_pipe_wait(uwip_pipe *p, uint64_t ticket)
{
    if (!p) return UWIP_ERR_INVALID;
    if (!ticket) return UWIP_OK;
    if (!p->copier) return p->fail(UWIP_ERR_INVALID, "no copy has been requested on this pipe");
    return copier_rc(p, uwip_copier_wait(p->copier, ticket));
}

UWIP_API int uwip_pipe_reset(uwip_pipe *p)
{
    if (!p) return UWIP_ERR_INVALID;
    p->have_prev = false;
    return UWIP_OK;
}

UWIP_API int uwip_pipe_last_params(uwip_pipe *p, int32_t *h_bs, int32_t *h_cl)
{
    if (!p) return UWIP_ERR_INVALID;
    return p->from_ctx(uwip_aclahe_last_params(p->ctx, h_bs, h_cl, p->cfg.frames));
}

UWIP_API int uwip_pipe_device_results(uwip_pipe *p, const uint8_t **d_v, const uint8_t **d_frames, const float **d_ratio,
                                      const int32_t **d_info)
{
    if (!p) return UWIP_ERR_INVALID;
    if (d_v) *d_v = p->v;
    if (d_frames) *d_frames = p->last_frames;
    if (d_ratio) *d_ratio = p->last_ratio;
    if (d_info) *d_info = p->last_info;
    return UWIP_OK;
}
