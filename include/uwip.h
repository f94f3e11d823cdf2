/*
 * uwip.h -- C ABI of the MI355X-native (gfx950) implementation of the
 * uwimageproc per-frame hot path:  bgdehaze -> histretch -> aclahe ->
 * videostrip-overlap.
 *
 * The reference has no FFI/plugin layer: its only seam is ordinary C++ calls
 * into modules/common/preprocessing.h and modules/videostrip/include/
 * videostrip.hpp, switched at compile time by USE_GPU and at run time by
 * -cuda=0/1 (modules/histretch/src/histretch.cpp:72,121-141).  The `...GPU`
 * twins there (imgChannelStretchGPU, calcOverlapGPU, calcBlurGPU) are the
 * precedent for an accelerator back end behind the same call sites; this
 * header is what such a back end binds to.  Every entry point cites the
 * reference interface it replaces.  INTEGRATION.md shows the reference-side
 * stubs.
 *
 * Conventions
 *   - plain C: opaque context, raw pointers, sizes; no C++/torch types.
 *   - every function returns an int status (UWIP_OK == 0); the message for
 *     the last failure is uwip_last_error(ctx).
 *   - images are batches of equally sized 8-bit frames in DEVICE memory,
 *     described by uwip_batch_u8, whose (data, step, rows, cols, channels)
 *     map field-for-field onto a cv::Mat of type CV_8UC1 / CV_8UC3
 *     (BGR interleaved, as cv::imread returns).  frames == 1 is one cv::Mat.
 *   - pointers named d_* are device pointers, h_* host pointers.
 *   - work is enqueued on the context's HIP stream and is asynchronous
 *     unless the function has a host-side result; uwip_sync() drains it.
 *   - there is NO CPU fallback: without a HIP device every compute entry
 *     point fails with UWIP_ERR_HIP.
 */
#ifndef UWIP_H
#define UWIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UWIP_OK               0
#define UWIP_ERR_INVALID      1   /* bad argument / shape */
#define UWIP_ERR_HIP          2   /* HIP runtime failure (incl. no device) */
#define UWIP_ERR_UNSUPPORTED  3   /* valid request outside the hot path */
#define UWIP_ERR_NOMEM        4

typedef struct uwip_ctx uwip_ctx;

/* One batch of frames in device memory (cv::Mat fields + batch extent). */
typedef struct uwip_batch_u8 {
    void   *data;          /* device pointer: frame 0, row 0 (cv::Mat::data) */
    size_t  step;          /* bytes between rows (cv::Mat::step)             */
    size_t  frame_stride;  /* bytes between consecutive frames               */
    int32_t rows, cols;    /* cv::Mat::rows / cols                           */
    int32_t channels;      /* 1 (CV_8UC1) or 3 (CV_8UC3, BGR)                */
    int32_t frames;        /* batch size (1 == a single cv::Mat)             */
} uwip_batch_u8;

/* ---- context, memory, profiling -------------------------------------- */

/* Replaces cuda::getCudaEnabledDeviceCount / cuda::setDevice(0)
 * (histretch.cpp:122-134, videostrip/src/main.cpp:188).  `stream` may be a
 * caller-owned hipStream_t (e.g. the framework's current stream) or NULL to
 * let the context create its own. */
int uwip_device_count(int *count);
int uwip_ctx_create(int device, void *stream, uwip_ctx **out);
/* Same with flags.  UWIP_CTX_STREAM_GIVEN: `stream` is used as it is even when
 * it is NULL -- the handle of the device's default ("null") stream is 0, which
 * uwip_ctx_create cannot tell from "no stream given"; a host framework whose
 * current stream is the default one (torch's is, until a side stream is made
 * current) passes this flag so that its own work and the library's stay in one
 * stream order.
 *
 * Threading: a uwip_ctx (and every uwip_features made from it) is
 * single-threaded -- one host thread at a time may be inside calls on it; the
 * library keeps no state outside contexts, so different contexts may be used
 * from different threads concurrently (one context per host thread / stream,
 * as bench.py does).  Every entry point makes the context's device current for
 * the calling thread (hipSetDevice) before it allocates, copies or launches. */
#define UWIP_CTX_STREAM_GIVEN 1u
/* Host waits.  Wherever the library waits on the host for its stream (uwip_sync, the ACLAHE parameter choice, staging
 * reuse, the copier's lanes) the calling thread polls an event with sleeps in between (20 ... 200 us) by default:
 * hipStreamSynchronize and hipEventSynchronize spin on this runtime -- also on hipEventBlockingSync events; only the
 * process-wide hipDeviceScheduleBlockingSync device flag makes them sleep, which is the host application's to set --
 * and a rank with eight sub-batch threads then burns eight cores doing nothing (measured: 1.59 CPU-seconds per 0.177 s
 * step), which a node's CPU quota does not have for eight ranks.  UWIP_CTX_SPIN_WAIT (or the environment variable
 * UWIP_SPIN_WAIT=1) keeps the spinning wait: up to 200 us less wake-up latency per wait for one core per waiting thread. */
#define UWIP_CTX_SPIN_WAIT 2u
int uwip_ctx_create_ex(int device, void *stream, unsigned flags, uwip_ctx **out);
int uwip_ctx_destroy(uwip_ctx *ctx);
const char *uwip_last_error(const uwip_ctx *ctx);
const char *uwip_version(void);
int uwip_sync(uwip_ctx *ctx);

/* Replaces GpuMat::upload / download (histretch.cpp:174-175,212-213). */
int uwip_malloc(uwip_ctx *ctx, size_t bytes, void **d_ptr);
int uwip_free(uwip_ctx *ctx, void *d_ptr);
int uwip_memcpy_h2d(uwip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int uwip_memcpy_d2h(uwip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* The same transfers enqueued on the context's stream without waiting
 * (GpuMat::upload / download with a cuda::Stream): the host buffer must be
 * page-locked -- take it from uwip_host_alloc (cuda::HostMem, PAGE_LOCKED) --
 * and stay untouched until uwip_sync().  This is the host-buffer front end of
 * the timed region of histretch.cpp:165-216 (upload ... download). */
int uwip_host_alloc(uwip_ctx *ctx, size_t bytes, void **h_ptr);
int uwip_host_free(uwip_ctx *ctx, void *h_ptr);
int uwip_memcpy_h2d_async(uwip_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int uwip_memcpy_d2h_async(uwip_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);

/* Copy engine for a stream of frame batches (the double-buffered form of
 * GpuMat::upload ... download around the timed region, histretch.cpp:165-216;
 * SURVEY.md section 7 "PCIe feed").  One per device / rank: an upload lane and a
 * download lane, each a host thread with a stream of its own that serves the
 * first queued request whose dependency has completed (requests that depend on
 * the same context are served in their order, and so are two requests of which
 * one writes bytes -- host or device -- the other reads or writes; one whose
 * stream is still busy does not hold back the ready requests of other
 * contexts).  A request with `after` != NULL starts once
 * everything queued on that context's stream AT THE TIME OF THE CALL has
 * finished (the lane waits for it on the host, so a copy is only handed to the
 * DMA engine when it can run: no hardware queue ever holds a barrier packet
 * that waits for a copy or for another stream -- with more streams than the
 * runtime has hardware queues such packets stall unrelated kernels).  A ticket
 * names the request: uwip_copier_wait blocks the calling host thread until
 * that copy has completed (ticket 0: returns at once), after which the host
 * buffer may be reused (upload) or read (download), and the device buffer may
 * be read by kernels queued from then on.  Requests may come from any thread.
 * uwip_copier_create leaves the caller's current HIP device as it found it.
 * Failure is sticky: after the first HIP error in a lane every later submit /
 * wait / query returns UWIP_ERR_HIP (uwip_copier_last_error says what failed;
 * queued requests are completed without copying so that no waiter blocks) and
 * the copier must be destroyed and created anew -- there is no reset. */
typedef struct uwip_copier uwip_copier;
int uwip_copier_create(int device, uwip_copier **out);
int uwip_copier_destroy(uwip_copier *c);                 /* drains both lanes first */
int uwip_copier_upload(uwip_copier *c, uwip_ctx *after, void *d_dst, const void *h_src, size_t bytes, uint64_t *ticket);
int uwip_copier_download(uwip_copier *c, uwip_ctx *after, void *h_dst, const void *d_src, size_t bytes, uint64_t *ticket);
int uwip_copier_wait(uwip_copier *c, uint64_t ticket);
int uwip_copier_query(uwip_copier *c, uint64_t ticket, int *done);
const char *uwip_copier_last_error(const uwip_copier *c);

/* Per-kernel hipEvent timing on the context's stream (replaces the
 * getTickCount stopwatch, histretch.cpp:165,257-261).  Enabling it brackets
 * every kernel launch with events; totals are read back per kernel name. */
int uwip_prof_enable(uwip_ctx *ctx, int on);
int uwip_prof_reset(uwip_ctx *ctx);
int uwip_prof_count(uwip_ctx *ctx, int *n);
int uwip_prof_get(uwip_ctx *ctx, int index, char *name, size_t name_cap,
                  double *total_ms, uint64_t *launches);

/* ---- histretch (H1-H4) ------------------------------------------------- */

/* numChannel / numSpace, preprocessing.cpp:147-161 (host, pure). */
int uwip_numChannel(char c);
int uwip_numSpace(char c);

/* getHistogram, preprocessing.cpp:25-34 (cv::calcHist, 256 bins), for every
 * frame and channel of the batch in one pass.  d_hist: [frames][channels][256]
 * uint32 counts (the reference stores the same counts as CV_32F). */
int uwip_getHistogram(uwip_ctx *ctx, const uwip_batch_u8 *img, uint32_t *d_hist);

/* Percentile search + stretch LUT, preprocessing.cpp:82-100, for `nplanes`
 * histograms of rows*cols-pixel planes.  d_lut: [nplanes][256] bytes;
 * d_bounds (may be NULL): [nplanes][2] int32 = (lower, higher) bins. */
int uwip_stretch_lut(uwip_ctx *ctx, const uint32_t *d_hist, int nplanes, int rows, int cols,
                     int lo, int hi, uint8_t *d_lut, int32_t *d_bounds);

/* In-place LUT application (the fused `img += b; img *= m`,
 * preprocessing.cpp:99-100).  d_lut: [frames][channels][256]. */
int uwip_apply_lut(uwip_ctx *ctx, const uwip_batch_u8 *img, const uint8_t *d_lut);

/* imgChannelStretch(Mat, Mat, lo, hi), preprocessing.cpp:74-105, on lane
 * `channel` of every frame, in place (the reference passes the same Mat as
 * input and output). */
int uwip_imgChannelStretch(uwip_ctx *ctx, const uwip_batch_u8 *img, int channel, int lo, int hi);

/* The per-letter loop of histretch.cpp:217-254 on BGR frames, in place:
 * for each letter in order, stretch plane numChannel(letter).  Unknown
 * letters are skipped as the reference does.  Letters of the other colour
 * spaces -- HSV ('H','S','V'), HLS ('h','s','l'), Lab ('L','a','b'), YCrCb
 * ('Y','C','X') -- do what the reference's code does with them (SURVEY.md
 * B-3): the stretch lands in a split copy and the image receives the 8-bit
 * colour round trip cvtColor(BGR2xxx) -> cvtColor(xxx2BGR), applied in letter
 * order between the BGR letters' stretches.
 * uwip_histretch_ex with UWIP_HISTRETCH_FIXED_ORDER runs the evident intent
 * instead: convert (histretch.cpp:232), split / stretch / MERGE (:234-236,
 * :240), then convert back (:238) -- the stretch is kept.
 * The 8-bit conversions restate OpenCV 3.x's color.cpp (parity unpinned).  Lab -> BGR exists in two forms there: OpenCV
 * 3.4.x (the version INSTALL.md:47-63 pins) converts 8-bit Lab with the integer Lab2RGBinteger, OpenCV 3.2 (the version the
 * module READMEs name) with the float Lab2RGB_f + inverse-gamma spline.  Default = 3.4.x, as for CLAHE's residual rule and
 * the 3 x 3 blur's rounding; UWIP_HISTRETCH_OPENCV32 / opencv_rule = 1 selects the 3.2 form. */
#define UWIP_HISTRETCH_FIXED_ORDER 1u
#define UWIP_HISTRETCH_OPENCV32    2u
int uwip_histretch_ex(uwip_ctx *ctx, const uwip_batch_u8 *img, const char *letters, int lo, int hi,
                      unsigned flags);
/* cv::cvtColor(src, dst, COLOR_BGR2{HSV,HLS,Lab,YCrCb}) (to_bgr = 0) or COLOR_{..}2BGR (to_bgr = 1) on 8UC3 batches;
 * space = uwip_numSpace's index 1..4 (histretch.cpp:155-156).  dst may alias src. */
int uwip_cvtColor(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst, int space, int to_bgr);
/* opencv_rule: 0 = OpenCV 3.4.x, 1 = OpenCV 3.2 (only COLOR_Lab2BGR differs). */
int uwip_cvtColor_ex(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst, int space, int to_bgr, int opencv_rule);
int uwip_histretch(uwip_ctx *ctx, const uwip_batch_u8 *img, const char *letters, int lo, int hi);

/* ---- aclahe (C1-C4) ----------------------------------------------------- */

/* V plane of cvtColor(BGR2HSV)+split, aclahe.cpp:152-154 (V = max(B,G,R)). */
int uwip_bgr_to_v(uwip_ctx *ctx, const uwip_batch_u8 *bgr, const uwip_batch_u8 *v);

/* cv::CLAHE::{setClipLimit,setTilesGridSize,apply}, aclahe.cpp:184-187, on
 * 8UC1 planes.  residual_rule: 0 = OpenCV 3.4.x, 1 = OpenCV 3.2. */
int uwip_clahe(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst,
               double clipLimit, int gx, int gy, int residual_rule);
/* Same with per-frame parameters (host arrays of length src->frames), the
 * "for resulting CL/BS, apply classic clahe" step of aclahe.cpp:215. */
int uwip_clahe_per_frame(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst,
                         const double *h_clipLimit, const int32_t *h_grid, int residual_rule);
/* Tile LUT stage tap for parity tests: d_luts [frames][gy*gx][256]. */
int uwip_clahe_luts(uwip_ctx *ctx, const uwip_batch_u8 *src, double clipLimit, int gx, int gy,
                    int residual_rule, uint8_t *d_luts);

/* aclaheEntropy, aclahe.cpp:228-248: d_entropy [frames] float. */
int uwip_entropy(uwip_ctx *ctx, const uwip_batch_u8 *src, float *d_entropy);

/* The sweep of aclahe.cpp:160-193: grid in {2,4,8,16,32} x clip limit in
 * {0,0.5,...,25}; d_entropy: [frames][5][51] float (clean table). */
int uwip_aclahe_sweep(uwip_ctx *ctx, const uwip_batch_u8 *src, int residual_rule,
                      float *d_entropy);
/* The same with a tap (may be NULL): d_hist [frames][5][51][256] uint32, the histogram of the CLAHE output of every
 * (grid, clip limit) -- what calcHist sees at aclahe.cpp:236 -- for exact parity checks of the sweep. */
int uwip_aclahe_sweep_hist(uwip_ctx *ctx, const uwip_batch_u8 *src, int residual_rule,
                           float *d_entropy, uint32_t *d_hist);

/* Parameter choice of modules/aclahe/python/ACLAHE.py:66-129 (host, pure; the
 * reference does it with scipy, functions.py:49-93; the C++ module stops at
 * comment stubs, aclahe.cpp:209-218).
 * uwip_aclahe_knee: DerivadaY + DerivadaX + Curvatura on one 49-sample curve
 *   (h_xs49 = clip limits 0.5..24.5, h_ys49 = entropies) -> arg-max index, or
 *   -1 where scipy's curve_fit would raise.
 * uwip_aclahe_select: h_entropy [frames][5][51] (the sweep table) -> per frame
 *   h_bs (block size) and h_cl (clip limit = the largest of the five knee
 *   indices, used as a clip limit as the reference does); h_knee (may be NULL)
 *   [frames][5].  When 2*CL lies outside the swept grid the BS choice falls
 *   back to the last swept clip limit; uwip_aclahe_auto evaluates it exactly. */
int uwip_aclahe_knee(const float *h_xs49, const float *h_ys49, int32_t *index);
/* The same choice made ON THE DEVICE, from the table the sweep leaves in HBM (no copy to the host, no host computation):
 * one wavefront per (frame, block size) runs the knee stage -- the same source as uwip_aclahe_select
 * (csrc/lm_core.hpp), the 49 samples of a curve on 49 lanes, MINPACK's summation order kept, so host and device agree
 * bit for bit -- and one thread per frame the choice of ACLAHE.py:92-125.
 *   d_entropy  device, [frames][5][51] (uwip_aclahe_sweep's output)
 *   d_par      device, [frames][4] int32 = {BS, CL, need_eval, 0}; need_eval = 1 when 2 * CL lies outside the swept grid
 *              (BS then comes from the last swept clip limit; uwip_aclahe_auto_ex evaluates such frames exactly)
 *   d_knee     device, [frames][5] int32, may be NULL: the five knee indices (-1 where curve_fit would raise) */
int uwip_aclahe_select_device(uwip_ctx *ctx, const float *d_entropy, int frames, int32_t *d_par, int32_t *d_knee);
/* The persistent host pool uwip_aclahe_select spreads its frames over (one per process, created on first use):
 * its size is the CPU budget of this rank minus the calling thread, at most 16 -- budget = min(CPUs in the affinity
 * mask, cgroup CPU quota) / ranks on the node (UWIP_RANKS_ON_NODE, else the launcher's LOCAL_WORLD_SIZE, else 1);
 * UWIP_HOST_THREADS overrides the size.  Any argument may be NULL. */
int uwip_host_pool_info(int *workers, double *cpu_budget, int *ranks_on_node);
int uwip_aclahe_select(const float *h_entropy, int frames, int32_t *h_bs, int32_t *h_cl,
                       int32_t *h_knee);

/* The whole aclahe stage on 8UC1 planes: sweep (aclahe.cpp:160-193), parameter
 * choice (ACLAHE.py:66-129) and the final createCLAHE(CL,(BS,BS)).apply
 * (python/main.py:19-20).  h_bs / h_cl (may be NULL): the chosen parameters.
 * Synchronises the stream once (the choice is a host decision). */
int uwip_aclahe_auto(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst,
                     int residual_rule, int32_t *h_bs, int32_t *h_cl);
/* The (BS, CL) the most recent uwip_aclahe_auto / uwip_aclahe_auto_ex on this context chose, h_bs / h_cl [frames] (frames =
 * that call's frame count).  After a UWIP_ACLAHE_ASYNC call this waits for the stream and copies them from the device. */
int uwip_aclahe_last_params(uwip_ctx *ctx, int32_t *h_bs, int32_t *h_cl, int frames);
/* cv2.GaussianBlur(img, (3,3), 0) on 8UC1 planes, ACLAHE.py:15 (fixed [1 2 1]/4 kernel, BORDER_REFLECT_101; rounding of
 * the /16: rule 0 = half up, OpenCV 3.4.x's 8-bit fixed-point path; rule 1 = half to even, OpenCV 3.2's float path).
 * Not in place. */
int uwip_GaussianBlur3(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst, int rounding_rule);
/* The two forms of the aclahe stage the reference holds:
 *   flags = 0                      the C++ driver, modules/aclahe/src/aclahe.cpp:152-187: the sweep runs on the plane itself
 *                                  (= uwip_aclahe_auto);
 *   flags = UWIP_ACLAHE_PREFILTER  ParametrosACLAHE, modules/aclahe/python/ACLAHE.py:9-129: the sweep (:40-47) and the
 *                                  block-size search (:102-112) run on GaussianBlur(img,(3,3),0) (:15), the final
 *                                  createCLAHE(CL,(BS,BS)).apply on the unfiltered image (python/main.py:19-20). */
#define UWIP_ACLAHE_PREFILTER 1u
/*   UWIP_ACLAHE_HOST_SELECT        the parameter choice by uwip_aclahe_select on the host (the sweep table is copied to
 *                                  page-locked memory, the host pool fits the curves) instead of uwip_aclahe_select_device,
 *                                  which is the default for batches of more than 4 frames (smaller ones take the host form
 *                                  for its latency: 0.15 ms per curve on a core against 1.3 ms on one wavefront); same
 *                                  parameters -- the two forms agree bit for bit.  Environment UWIP_ACLAHE_SELECT=host |
 *                                  device forces one form process-wide. */
#define UWIP_ACLAHE_HOST_SELECT 2u
/*   UWIP_ACLAHE_ASYNC              nothing comes back to the host and the call does not wait: the choice is made on the device,
 *                                  the final per-frame CLAHE is launched from the device-side parameters (the kernels of all
 *                                  five grids are launched over the batch, blocks of frames that chose another grid exit),
 *                                  and a frame whose clip limit leaves the swept grid gets its exact block-size search from a
 *                                  device kernel.  h_bs / h_cl must be NULL; uwip_aclahe_last_params fetches the parameters
 *                                  later (it waits for the stream).  Same image, same parameters as the other forms.  Batches
 *                                  the library gives to the host form (<= 4 frames, UWIP_ACLAHE_SELECT=host) run synchronously
 *                                  under this flag too. */
#define UWIP_ACLAHE_ASYNC 4u
int uwip_aclahe_auto_ex(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst,
                        int residual_rule, unsigned flags, int32_t *h_bs, int32_t *h_cl);

/* "transform back image" (aclahe.cpp:216 stub): cvtColor(BGR2HSV), replace V by
 * v_new (the CLAHE'd plane), cvtColor(HSV2BGR), all 8-bit.  bgr_out may alias bgr. */
int uwip_hsv_replace_v(uwip_ctx *ctx, const uwip_batch_u8 *bgr, const uwip_batch_u8 *v_new,
                       const uwip_batch_u8 *bgr_out);

/* ---- bgdehaze (D1-D6) ---------------------------------------------------- */
/* All real-valued results are float64, as in the reference.  The input is the
 * uint8 BGR frame cv2.imread returns; normI = (I - min)/(max - min)
 * (modules/bgdehaze/main.py:17) is formed inside. */

/* Background_light(normI, w), BGDehaze.py:14-26.  d_B: [frames][3] (BGR);
 * d_idx (may be NULL): [frames][2] row-major pixel indices of the two minima.
 * Ties take the first index (the reference's order is unspecified, B-9). */
int uwip_dehaze_background_light(uwip_ctx *ctx, const uwip_batch_u8 *in, int w, double *d_B,
                                 int32_t *d_idx);

/* transmission_map(normI, 15) with B injected, BGDehaze.py:28-37.
 * d_B: [frames][3]; d_t: [frames][2][rows][cols] (blue, green), before the
 * 0.2 clamp. */
int uwip_dehaze_transmission(uwip_ctx *ctx, const uwip_batch_u8 *in, const double *d_B, double *d_t);

/* guided_filter(I, p, r, eps), guidedfilter.py:54-103, for the guide form both
 * call sites use: an 8-bit 3-channel image normalised by its global min/max.
 * d_p, d_q: [frames][rows][cols].  Needs rows, cols >= 2r+1. */
int uwip_guided_filter(uwip_ctx *ctx, const uwip_batch_u8 *guide, const double *d_p, int r, double eps,
                       double *d_q);

/* generate_results(), modules/bgdehaze/main.py:14-20: uint8 BGR in -> uint8
 * BGR out.  flags: UWIP_DEHAZE_FULL runs adaptiveExp_map (BGDehaze.py:71-89),
 * without it the chain stops after RC_correction (:59-69).  As written, a 0/0
 * in the exposure map S (:83) turns the whole frame into NaN -> black
 * (SURVEY.md B-11); that is reproduced unless UWIP_DEHAZE_GUARD_S is set, which
 * substitutes S = 1 there (a documented deviation).
 * `out` may be NULL when only taps are wanted.
 * Optional taps / injection (device pointers, may be NULL):
 *   d_B_inject   [frames][3]              use this background light (parity tests)
 *   d_refined_t  [frames][2][rows][cols]  refined_t (:39-48)
 *   d_float_out  [frames][rows][cols][3]  the float64 image before *255 */
#define UWIP_DEHAZE_FULL     1
#define UWIP_DEHAZE_GUARD_S  2
int uwip_dehaze(uwip_ctx *ctx, const uwip_batch_u8 *in, const uwip_batch_u8 *out, int w, int flags,
                const double *d_B_inject, double *d_refined_t, double *d_float_out);

/* bgdehaze -> histretch chained on one batch (BASELINE.json configs[2]; the reference runs the two tools back to back
 * over files: modules/bgdehaze/main.py:14-20, then modules/histretch/src/histretch.cpp:218-254 on its output).  Same
 * result as uwip_dehaze(in, out, ...) followed by uwip_histretch_ex(out, ...); with UWIP_DEHAZE_FULL the kernel
 * that writes the dehazed bytes also counts them, so the stretch starts from a finished histogram instead of
 * reading the image once more. */
int uwip_dehaze_histretch(uwip_ctx *ctx, const uwip_batch_u8 *in, const uwip_batch_u8 *out, int w, int dehaze_flags,
                          const char *letters, int lo, int hi, unsigned histretch_flags);

/* ---- videostrip overlap (V1-V5) -------------------------------------------- */
/* calcOverlap(keyframe*, Mat), modules/videostrip/src/videostrip.cpp:192-289, split
 * at the point where the reference caches a key frame's keypoints/descriptors in
 * `struct keyframe` (videostrip.hpp:62-68; kframe->new_img, :209-213):
 *   uwip_overlap_detect   resize to 640 wide (main.cpp:242,311) + BGR2GRAY + detect + describe
 *                         for a batch of frames, into slots of an opaque feature set;
 *   uwip_overlap_match    kNN(2) match (:229-231), ratio test (:233-242), homography (:270),
 *                         overlapArea (:280) for a list of (object slot, key slot) pairs.
 * The detector/descriptor/matcher are the AKAZE-style / M-LDB / Hamming design that
 * BASELINE.json's north_star names in place of the reference's OpenCV-contrib SURF +
 * L2 matcher (DESIGN.md "overlap stage"); the dense distance matrix runs on i8 MFMA. */
typedef struct uwip_features uwip_features;

#define UWIP_MAX_KEYPOINTS 2048
/* keypoint record returned by uwip_features_download (32 bytes) */
typedef struct uwip_keypoint {
    float   x, y;        /* sub-pixel position in the 640-wide working image */
    float   response;    /* scale-normalised determinant of the Hessian */
    int32_t level;       /* evolution level 0..3 */
    int32_t xi, yi;      /* integer extremum position */
    float   co, si;      /* unit vector of the dominant orientation ((1, 0) with UWIP_OVERLAP_UPRIGHT) */
} uwip_keypoint;

int uwip_features_create(uwip_ctx *ctx, int max_frames, uwip_features **out);
int uwip_features_destroy(uwip_features *feats);
/* copy one slot's cached keypoints/descriptors (kframe->keypoints/descriptors) to another slot */
int uwip_features_copy(uwip_ctx *ctx, const uwip_features *src, int src_slot, uwip_features *dst,
                       int dst_slot);
/* working size for a rows x cols frame: cv::resize(frame, Size(), f, f), f = 640/cols */
int uwip_overlap_working_size(int rows, int cols, int *orows, int *ocols);
/* cv::resize(frame, res_frame, Size(), f, f) by itself (main.cpp:242,287,311; INTER_LINEAR, 8UC3 fixed point): the frame
 * the reference hands to calcBlur (main.cpp:338,355).  dst: rows x cols of uwip_overlap_working_size. */
int uwip_resize_bgr(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst);
/* frames: full-resolution CV_8UC3 BGR (resized inside) or CV_8UC1 planes already at the
 * working size.  Fills slots [first_slot, first_slot + frames->frames). */
int uwip_overlap_detect(uwip_ctx *ctx, const uwip_batch_u8 *frames, uwip_features *feats, int first_slot);
/* Same with flags.  By default keypoints carry a dominant orientation and the descriptor is sampled in the keypoint's own
 * frame, as the reference's SURF::create(400) is oriented (upright = false, videostrip.cpp:206-208): the overlap ratio
 * holds under any in-plane rotation of the camera.  UWIP_OVERLAP_UPRIGHT skips the orientation estimate (SURF's
 * `upright` parameter): rotation tolerance then ends near 20 degrees. */
#define UWIP_OVERLAP_UPRIGHT 1u
/* The detector threshold is FIXED by default (1e-3 on the scale-normalised determinant of the Hessian), as SURF's
 * hessianThreshold is in the reference (SURF::create(400), videostrip.cpp:200,206).  UWIP_OVERLAP_RELATIVE_THRESHOLD (opt-in, a
 * DEVIATION): the threshold follows the frame's own contrast, 1e-3 * min(1, (k / 0.5)^2), k = the frame's contrast factor
 * (70th percentile of the gradient magnitude).  The response scales with the square of the contrast, and raw frames of
 * turbid water -- what the reference's videostrip is run on -- have none above a fixed threshold (the reference's own
 * photograph PIS_T1A_259: no keypoint at all at 1e-3, so calcOverlap answers -2.0 whatever the motion; 130 keypoints with
 * the relative one).  It also finds a few dozen "keypoints" in pure sensor noise: combine it with UWIP_OVERLAP_MIN6.
 * (Rounds 3-4 had the relative threshold as the default; UWIP_OVERLAP_FIXED_THRESHOLD is still accepted and names the
 * default.) */
#define UWIP_OVERLAP_FIXED_THRESHOLD 2u
#define UWIP_OVERLAP_RELATIVE_THRESHOLD 16u
int uwip_overlap_detect_ex(uwip_ctx *ctx, const uwip_batch_u8 *frames, uwip_features *feats, int first_slot, unsigned flags);
/* parity taps: one slot's keypoints (uwip_keypoint[2048]) / packed 64-byte descriptors; and the
 * scale-space images of frame `frame` of the most recent uwip_overlap_detect call (host buffers
 * of rows*cols floats, any may be NULL). */
int uwip_features_download(uwip_ctx *ctx, const uwip_features *feats, int slot, void *h_kps,
                           uint8_t *h_desc, int32_t *h_count);
/* the opposite direction: fill one slot from host keypoints (uwip_keypoint[count]) and packed
 * descriptors ([count][64]); the reference's `struct keyframe` members are public and caller-fillable
 * (videostrip.hpp:62-68).  rows/cols: the working size the keypoints refer to. */
int uwip_features_upload(uwip_ctx *ctx, uwip_features *feats, int slot, int rows, int cols,
                         const void *h_kps, const uint8_t *h_desc, int32_t count);
int uwip_overlap_debug_level(uwip_ctx *ctx, int frame, int level, int rows, int cols, float *h_Lt,
                             float *h_Lx, float *h_Ly, float *h_Ldet, float *h_kcontrast);
/* h_pair_q / h_pair_t: host arrays of slot indices (query = object frame in fq, train = key
 * frame in ft).  d_ratio [npairs]: the overlap ratio, or -2.0 when fewer than 4 good matches
 * survive or no homography is found (videostrip.cpp:252-256,272).  videoWidth / videoHeight
 * are the reference's globals (main.cpp:238-239).  Optional device outputs:
 * d_info [npairs][8] = {nkp_obj, nkp_key, ngood, ninliers, overlap pixels, 0,0,0};
 * d_H [npairs][9]; d_match_idx / d_match_dist [npairs][2048][2] (the kNN(2) result). */
int uwip_overlap_match(uwip_ctx *ctx, const uwip_features *fq, const uwip_features *ft,
                       const int32_t *h_pair_q, const int32_t *h_pair_t, int npairs, int videoWidth,
                       int videoHeight, uint32_t seed, float *d_ratio, int32_t *d_info, double *d_H,
                       int32_t *d_match_idx, int32_t *d_match_dist);
/* Same with flags.  The default is the reference's rule (videostrip.cpp:252-256,270-272): -2.0 only when fewer than 4 good
 * matches survive the ratio test or no homography is found; whatever findHomography returns for >= 4 good matches is
 * used -- i.e. any hypothesis with >= 4 inliers (any solvable sample of four good matches) yields an overlap value.
 * UWIP_OVERLAP_MIN6 (opt-in, a DEVIATION from the reference): a homography supported by fewer than 6 RANSAC inliers is
 * reported as none (-2.0) -- four matches always fit one exactly, and with the contrast-relative detector threshold two
 * frames of pure sensor noise now and then produce four chance matches.  (Rounds 3-4 had the two the other way round:
 * UWIP_OVERLAP_MIN4 is still accepted and now names the default.) */
#define UWIP_OVERLAP_MIN4 4u
#define UWIP_OVERLAP_MIN6 8u
int uwip_overlap_match_ex(uwip_ctx *ctx, const uwip_features *fq, const uwip_features *ft,
                          const int32_t *h_pair_q, const int32_t *h_pair_t, int npairs, int videoWidth,
                          int videoHeight, uint32_t seed, unsigned flags, float *d_ratio, int32_t *d_info,
                          double *d_H, int32_t *d_match_idx, int32_t *d_match_dist);
/* overlapArea(Mat H), videostrip.cpp:291-319, for n row-major 3x3 double homographies. */
int uwip_overlapArea(uwip_ctx *ctx, const double *d_H, int n, int videoWidth, int videoHeight,
                     float *d_ratio, int32_t *d_count);
/* calcBlur(Mat frame), videostrip.cpp:170-184, on BGR frames (the reference passes the
 * resized frame, main.cpp:338,355): d_blur [frames]. */
int uwip_calcBlur(uwip_ctx *ctx, const uwip_batch_u8 *frames, float *d_blur);

/* ---- the whole per-frame chain ----------------------------------------------------------------------------- */
/* bgdehaze -> histretch -> aclahe -> videostrip-overlap on batches of frames, as ONE object: what the reference runs as
 * four tools back to back over files -- modules/bgdehaze/main.py:14-20, modules/histretch/src/histretch.cpp:217-254,
 * modules/aclahe/src/aclahe.cpp:152-218 (+ python/ACLAHE.py:9-129, python/main.py:19-20), modules/videostrip/src/
 * main.cpp:300-394 (calcOverlap of every frame against its predecessor, videostrip.cpp:192-289).  A pipe owns what the
 * chain carries from step to step: the V planes, the feature slots (slot 0 = the previous batch's last frame, i.e. the
 * `struct keyframe` cache of videostrip.hpp:62-68 carried across batches), the pair list, a throttle (at most
 * `max_in_flight` steps queued on the stream: nothing in a step waits on the host, so an unthrottled caller would queue
 * without bound), and -- for the host-buffer form -- two source and two result buffers in device memory with their copy
 * tickets (GpuMat::upload ... download around the timed region, histretch.cpp:165-216, double-buffered).
 * Stage definitions: dehaze = uwip_dehaze(w, dehaze_flags); histretch = uwip_histretch_ex(letters, lo, hi,
 * histretch_flags) (the two chained through uwip_dehaze_histretch when both run); aclahe = uwip_bgr_to_v +
 * uwip_aclahe_auto_ex(residual_rule, aclahe_flags) + uwip_hsv_replace_v; overlap = uwip_overlap_detect_ex(detect_flags)
 * + uwip_overlap_match_ex(videoWidth, videoHeight, seed, match_flags) of frame i against frame i - 1 (frame 0 against the
 * previous batch's last frame; in the very first batch against itself, main.cpp:284-297 takes the first frame as key frame).
 * uwip_pipe_config_default fills in the REFERENCE's rules: letters "RGB", 2 / 98 percent, w = 15, UWIP_DEHAZE_FULL (S
 * unguarded, BGDehaze.py:83), UWIP_ACLAHE_PREFILTER | UWIP_ACLAHE_ASYNC, OpenCV 3.4.x rounding, >= 4 good matches
 * (videostrip.cpp:252-272), videoWidth x videoHeight = cols x rows (main.cpp:238-239), seed 1, two steps in flight.
 * A pipe is bound to the context it was made from (its stream, its thread rule); destroy it before the context. */
typedef struct uwip_pipe uwip_pipe;
typedef struct uwip_pipe_config {
    int32_t  frames, rows, cols;       /* batch geometry of every step: `frames` CV_8UC3 BGR frames of rows x cols */
    char     letters[16];              /* histretch -c letters, NUL-terminated */
    int32_t  lo, hi;                   /* histretch percentiles */
    int32_t  w;                        /* bgdehaze window */
    uint32_t dehaze_flags;             /* UWIP_DEHAZE_* */
    uint32_t histretch_flags;          /* UWIP_HISTRETCH_* */
    int32_t  residual_rule;            /* 0 = OpenCV 3.4.x, 1 = OpenCV 3.2 (CLAHE redistribution, 3x3 blur rounding) */
    uint32_t aclahe_flags;             /* UWIP_ACLAHE_* */
    uint32_t detect_flags;             /* UWIP_OVERLAP_UPRIGHT | UWIP_OVERLAP_RELATIVE_THRESHOLD */
    uint32_t match_flags;              /* UWIP_OVERLAP_MIN6 */
    int32_t  videoWidth, videoHeight;  /* the reference's globals; 0 = cols / rows */
    uint32_t seed;                     /* RANSAC seed */
    int32_t  max_in_flight;            /* steps queued before uwip_pipe_step waits for the oldest (>= 1) */
    void    *d_staging;                /* host-buffer form: caller-owned device memory of uwip_pipe_staging_bytes() bytes,
                                          or NULL = the pipe allocates it at the first uwip_pipe_step_host.  Layout:
                                          src[2][frames][rows][cols][3] u8, work[2][...] u8, ratio[2][frames] f32 (256-byte
                                          aligned), info[2][frames][8] i32 */
} uwip_pipe_config;
int uwip_pipe_config_default(uwip_pipe_config *cfg, int frames, int rows, int cols);
size_t uwip_pipe_staging_bytes(const uwip_pipe_config *cfg);
/* copier: the copy engine the host-buffer form uses (shared by all pipes of a rank), or NULL = the pipe makes its own at
 * the first uwip_pipe_step_host. */
int uwip_pipe_create(uwip_ctx *ctx, const uwip_pipe_config *cfg, uwip_copier *copier, uwip_pipe **out);
int uwip_pipe_destroy(uwip_pipe *p);          /* drains the stream and the pipe's outstanding copies first */
const char *uwip_pipe_last_error(const uwip_pipe *p);
/* One step on frames resident in device memory: in -> out (distinct buffers of the configured geometry), d_ratio [frames]
 * f32 = the overlap ratio of every frame against its predecessor (-2.0 / the value, videostrip.cpp:252-289), d_info (may
 * be NULL) [frames][8] as uwip_overlap_match.  Asynchronous: returns when the step is queued (after waiting, if need be,
 * until fewer than max_in_flight earlier steps are unfinished); uwip_pipe_sync drains. */
int uwip_pipe_step(uwip_pipe *p, const uwip_batch_u8 *in, const uwip_batch_u8 *out, float *d_ratio, int32_t *d_info);
/* The same, stage by stage (parity tests tap the chain between stages; a tool runs a part of it): `stages` is a mask of
 * UWIP_PIPE_*.  Without UWIP_PIPE_DEHAZE `in` is ignored (may be NULL) and `out` is processed in place; d_ratio / d_info
 * are used by UWIP_PIPE_OVERLAP only.  Not throttled. */
#define UWIP_PIPE_DEHAZE    1u
#define UWIP_PIPE_HISTRETCH 2u
#define UWIP_PIPE_ACLAHE    4u
#define UWIP_PIPE_OVERLAP   8u
#define UWIP_PIPE_ALL       15u
int uwip_pipe_stages(uwip_pipe *p, unsigned stages, const uwip_batch_u8 *in, const uwip_batch_u8 *out, float *d_ratio,
                     int32_t *d_info);
/* One step from / to page-locked host memory (uwip_host_alloc): h_in -> upload -> the four stages -> download -> h_out
 * (frames) and, when h_ratio != NULL, h_ratio [frames] f32.  h_prefetch (may be NULL) is the NEXT step's h_in: its upload
 * is requested now and runs under this step's kernels.  Returns without waiting for the device; tickets[0] = the upload of
 * h_in (h_in may be refilled once it is complete), tickets[1] = the download of the frames (they leave as soon as the
 * aclahe stage is done, under the overlap kernels), tickets[2] = the download of the ratios (0 when h_ratio is NULL):
 * uwip_pipe_wait blocks the calling thread until that copy is complete.  The call itself waits (on the host, sleeping) for
 * the upload of h_in and for the download of the step before last, whose device buffer it reuses. */
int uwip_pipe_step_host(uwip_pipe *p, const void *h_in, void *h_out, float *h_ratio, const void *h_prefetch,
                        uint64_t tickets[3]);
int uwip_pipe_wait(uwip_pipe *p, uint64_t ticket);
int uwip_pipe_sync(uwip_pipe *p);             /* the stream and every outstanding copy of this pipe */
/* Forget the carried key frame: the next step's frame 0 is its own key frame again (a new video). */
int uwip_pipe_reset(uwip_pipe *p);
/* (BS, CL) the aclahe stage of the most recent step chose, h_bs / h_cl [frames] (waits for the stream). */
int uwip_pipe_last_params(uwip_pipe *p, int32_t *h_bs, int32_t *h_cl);
/* Device pointers of the most recent step's results, for callers that chain further device work: the V planes the
 * aclahe stage left ([frames][rows][cols] u8, the unfiltered V of the stretched frames), and -- host-buffer form -- the
 * frames / ratios / info of that step inside the staging area.  Any argument may be NULL. */
int uwip_pipe_device_results(uwip_pipe *p, const uint8_t **d_v, const uint8_t **d_frames, const float **d_ratio,
                             const int32_t **d_info);

#ifdef __cplusplus
}
#endif
#endif /* UWIP_H */
