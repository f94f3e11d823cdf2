// uwip.hpp -- C++ host shim over the C ABI (include/uwip.h) with the reference's own
// function names and argument meaning, on a cv::Mat-shaped POD (uw::Mat).  When the
// reference is built against OpenCV, a cv::Mat maps onto uw::Mat field for field
// (data, step, rows, cols, channels); see INTEGRATION.md.  Host buffers go in and out;
// device staging is internal.  There is no CPU path: every call needs a HIP device.
//
//   modules/common/preprocessing.h:38,66,112,115   getHistogram, imgChannelStretch, numChannel, numSpace
//   modules/videostrip/include/videostrip.hpp:62-68,84,98,118   keyframe, calcOverlap, calcBlur, overlapArea
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "uwip.h"

namespace uw {

struct Mat {                      // the cv::Mat fields this path uses (8-bit, 1 or 3 channels, BGR)
    uint8_t *data = nullptr;
    size_t step = 0;
    int rows = 0, cols = 0, chans = 0;
    bool empty() const { return !data || rows <= 0 || cols <= 0; }
    int channels() const { return chans; }
};

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

class Context {
public:
    explicit Context(int device = 0)
    {
        int rc = uwip_ctx_create(device, nullptr, &ctx_);
        if (rc != UWIP_OK) throw Error(rc, "uwip_ctx_create failed: no HIP device (there is no CPU fallback)");
    }
    ~Context() { uwip_ctx_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    uwip_ctx *get() const { return ctx_; }
    void check(int rc) const { if (rc != UWIP_OK) throw Error(rc, uwip_last_error(ctx_)); }
    static int deviceCount() { int n = 0; uwip_device_count(&n); return n; }   // cuda::getCudaEnabledDeviceCount
private:
    uwip_ctx *ctx_ = nullptr;
};

// The rank's copy engine (uwip_copier): upload / download of frame batches between page-locked host buffers and HBM on
// two lanes of their own, hand-overs by ticket (GpuMat::upload / download with a cuda::Stream around the timed region,
// histretch.cpp:165-216, for a stream of batches).  `after`: the copy starts once everything queued on that context's
// stream so far has finished.
class Copier {
public:
    explicit Copier(int device = 0)
    {
        int rc = uwip_copier_create(device, &c_);
        if (rc != UWIP_OK) throw Error(rc, "uwip_copier_create failed: no HIP device (there is no CPU fallback)");
    }
    ~Copier() { uwip_copier_destroy(c_); }
    Copier(const Copier &) = delete;
    Copier &operator=(const Copier &) = delete;
    uint64_t upload(void *d_dst, const void *h_src, size_t bytes, Context *after = nullptr)
    {
        uint64_t t = 0;
        check(uwip_copier_upload(c_, after ? after->get() : nullptr, d_dst, h_src, bytes, &t));
        return t;
    }
    uint64_t download(void *h_dst, const void *d_src, size_t bytes, Context *after = nullptr)
    {
        uint64_t t = 0;
        check(uwip_copier_download(c_, after ? after->get() : nullptr, h_dst, d_src, bytes, &t));
        return t;
    }
    void wait(uint64_t ticket) { check(uwip_copier_wait(c_, ticket)); }
    bool done(uint64_t ticket) { int d = 0; check(uwip_copier_query(c_, ticket, &d)); return d != 0; }
private:
    void check(int rc) const { if (rc != UWIP_OK) throw Error(rc, uwip_copier_last_error(c_)); }
    uwip_copier *c_ = nullptr;
};

// device copy of a host Mat (the role cv::cuda::GpuMat plays in the reference's CUDA branches)
class DeviceMat {
public:
    DeviceMat(Context &c, const Mat &m) : c_(c)
    {
        b_.rows = m.rows; b_.cols = m.cols; b_.channels = m.chans; b_.frames = 1;
        b_.step = (size_t)m.cols * m.chans;                      // packed on the device
        b_.frame_stride = b_.step * m.rows;
        c_.check(uwip_malloc(c_.get(), b_.frame_stride, &b_.data));
        upload(m);
    }
    DeviceMat(Context &c, int rows, int cols, int chans) : c_(c)
    {
        b_.rows = rows; b_.cols = cols; b_.channels = chans; b_.frames = 1;
        b_.step = (size_t)cols * chans; b_.frame_stride = b_.step * rows;
        c_.check(uwip_malloc(c_.get(), b_.frame_stride, &b_.data));
    }
    ~DeviceMat() { uwip_free(c_.get(), b_.data); }
    DeviceMat(const DeviceMat &) = delete;
    DeviceMat &operator=(const DeviceMat &) = delete;
    void upload(const Mat &m)                                     // GpuMat::upload
    {
        if (m.step == b_.step) c_.check(uwip_memcpy_h2d(c_.get(), b_.data, m.data, b_.frame_stride));
        else for (int y = 0; y < m.rows; ++y)
            c_.check(uwip_memcpy_h2d(c_.get(), (uint8_t *)b_.data + (size_t)y * b_.step, m.data + (size_t)y * m.step, b_.step));
    }
    void download(Mat &m) const                                   // GpuMat::download
    {
        if (m.step == b_.step) c_.check(uwip_memcpy_d2h(c_.get(), m.data, b_.data, b_.frame_stride));
        else for (int y = 0; y < m.rows; ++y)
            c_.check(uwip_memcpy_d2h(c_.get(), m.data + (size_t)y * m.step, (const uint8_t *)b_.data + (size_t)y * b_.step, b_.step));
    }
    const uwip_batch_u8 *batch() const { return &b_; }
private:
    Context &c_;
    uwip_batch_u8 b_{};
};

// ---- modules/common/preprocessing.h ---------------------------------------------------------
inline int numChannel(char c) { return uwip_numChannel(c); }
inline int numSpace(char c) { return uwip_numSpace(c); }

// void getHistogram(cv::Mat*, cv::Mat*): 256 float counts of an 8UC1 plane
inline void getHistogram(Context &c, const Mat &img, float hist[256])
{
    DeviceMat d(c, img);
    void *dh = nullptr;
    c.check(uwip_malloc(c.get(), sizeof(uint32_t) * 256 * img.chans, &dh));
    c.check(uwip_getHistogram(c.get(), d.batch(), (uint32_t *)dh));
    std::vector<uint32_t> h(256 * (size_t)img.chans);
    c.check(uwip_memcpy_d2h(c.get(), h.data(), dh, h.size() * 4));
    uwip_free(c.get(), dh);
    for (int i = 0; i < 256; ++i) hist[i] = (float)h[i];
}

// void imgChannelStretch(cv::Mat imgOriginal, cv::Mat imgStretched, int lo = 0, int hi = 100): in place
// (both Mats share pixels at every reference call site); `channel` picks the lane of a packed image.
inline void imgChannelStretch(Context &c, Mat imgOriginal, Mat imgStretched, int lowerPercentile = 0, int higherPercentile = 100,
                              int channel = 0)
{
    if (imgOriginal.data != imgStretched.data) throw Error(UWIP_ERR_INVALID, "imgStretched must share pixels with imgOriginal");
    DeviceMat d(c, imgOriginal);
    c.check(uwip_imgChannelStretch(c.get(), d.batch(), channel, lowerPercentile, higherPercentile));
    d.download(imgStretched);
}

// the per-letter loop of histretch.cpp:217-254 on a BGR image, in place
inline void histretch(Context &c, Mat src, const std::string &cChannel, int min_percent = 2, int max_percent = 98, bool fixed_order = false,
                      bool opencv32 = false)
{
    DeviceMat d(c, src);
    c.check(uwip_histretch_ex(c.get(), d.batch(), cChannel.c_str(), min_percent, max_percent,
                              (fixed_order ? UWIP_HISTRETCH_FIXED_ORDER : 0u) | (opencv32 ? UWIP_HISTRETCH_OPENCV32 : 0u)));
    d.download(src);
}

// ---- modules/videostrip/include/videostrip.hpp -------------------------------------------------
struct keyframe {                 // videostrip.hpp:62-68 (keypoints/descriptors live in a device feature slot)
    bool new_img = true;
    Mat img;                      // reference frame (full resolution; the 640-wide resize happens inside)
    uwip_features *feats = nullptr;
};

class Videostrip {
public:
    int videoWidth = 0, videoHeight = 0;          // the reference's globals (main.cpp:45-46)
    unsigned match_flags = 0;                     // 0 = the reference's ">= 4 good matches" rule (videostrip.cpp:252-272); UWIP_OVERLAP_MIN6 = >= 6 inliers
    unsigned detect_flags = 0;                    // 0 = fixed detector threshold (SURF::create(400) is fixed, videostrip.cpp:206); UWIP_OVERLAP_RELATIVE_THRESHOLD
    explicit Videostrip(Context &c) : c_(c)
    {
        c_.check(uwip_features_create(c_.get(), 1, &obj_));
        c_.check(uwip_malloc(c_.get(), 128, &scratch_));
    }
    ~Videostrip() { uwip_features_destroy(obj_); uwip_free(c_.get(), scratch_); }
    void initKeyframe(keyframe &k) { if (!k.feats) c_.check(uwip_features_create(c_.get(), 1, &k.feats)); }
    static void releaseKeyframe(keyframe &k) { uwip_features_destroy(k.feats); k.feats = nullptr; }

    // float calcOverlap(keyframe* kframe, Mat img_object): -1 on empty input, -2.0 when no homography
    float calcOverlap(keyframe *kframe, const Mat &img_object, uint32_t seed = 1)
    {
        if (img_object.empty() || kframe->img.empty()) { std::printf(" --(!) Error reading images \n"); return -1.f; }
        initKeyframe(*kframe);
        if (kframe->new_img) {
            DeviceMat k(c_, kframe->img);
            c_.check(uwip_overlap_detect_ex(c_.get(), k.batch(), kframe->feats, 0, detect_flags));
            kframe->new_img = false;
        }
        DeviceMat o(c_, img_object);
        c_.check(uwip_overlap_detect_ex(c_.get(), o.batch(), obj_, 0, detect_flags));
        int32_t q = 0, t = 0;
        c_.check(uwip_overlap_match_ex(c_.get(), obj_, kframe->feats, &q, &t, 1, videoWidth, videoHeight, seed, match_flags,
                                       (float *)scratch_, nullptr, nullptr, nullptr, nullptr));
        float r = 0.f;
        c_.check(uwip_memcpy_d2h(c_.get(), &r, scratch_, 4));
        if (r == -2.0f) std::printf("[WARN] Not enough good matches!\n");
        return r;
    }
    // cv::resize(frame, res_frame, cv::Size(), hResizeFactor, hResizeFactor), main.cpp:242,287,311: `res` receives the
    // 640-wide BGR frame (its buffer is (re)allocated into `store`)
    void resize(const Mat &frame, Mat &res, std::vector<uint8_t> &store)
    {
        int oh = 0, ow = 0;
        c_.check(uwip_overlap_working_size(frame.rows, frame.cols, &oh, &ow));
        store.resize((size_t)oh * ow * 3);
        res.data = store.data(); res.rows = oh; res.cols = ow; res.chans = 3; res.step = (size_t)ow * 3;
        DeviceMat s(c_, frame), d(c_, oh, ow, 3);
        c_.check(uwip_resize_bgr(c_.get(), s.batch(), d.batch()));
        d.download(res);
    }
    // float calcBlur(Mat frame): frame = the resized BGR frame (main.cpp:338,355)
    float calcBlur(const Mat &frame)
    {
        DeviceMat d(c_, frame);
        c_.check(uwip_calcBlur(c_.get(), d.batch(), (float *)scratch_));
        float r = 0.f;
        c_.check(uwip_memcpy_d2h(c_.get(), &r, scratch_, 4));
        return r;
    }
    // float overlapArea(Mat H): H row-major 3x3 doubles
    float overlapArea(const double H[9])
    {
        double *dH = (double *)((uint8_t *)scratch_ + 16);
        c_.check(uwip_memcpy_h2d(c_.get(), dH, H, sizeof(double) * 9));
        c_.check(uwip_overlapArea(c_.get(), dH, 1, videoWidth, videoHeight, (float *)scratch_, nullptr));
        float r = 0.f;
        c_.check(uwip_memcpy_d2h(c_.get(), &r, scratch_, 4));
        return r;
    }
private:
    Context &c_;
    uwip_features *obj_ = nullptr;
    void *scratch_ = nullptr;
};

// ---- the reference's own signatures, on a process-wide default context -------------------------------------
// A call site of the reference switches by name alone (the `...GPU` twins of preprocessing.h:96 /
// videostrip.hpp:84-118 are the precedent): same names, same argument order and defaults, the same globals
// (videoWidth / videoHeight / hResizeFactor, main.cpp:45-49), cv::Mat replaced by the field-compatible uw::Mat.
// Single-threaded, device 0, like the reference.
namespace ref {
inline Context &defaultContext() { static Context c(0); return c; }
inline Videostrip &defaultVideostrip() { static Videostrip v(defaultContext()); return v; }
inline int videoWidth = 0, videoHeight = 0;
inline float hResizeFactor = 1.f;

inline int numChannel(char c) { return uwip_numChannel(c); }                         // preprocessing.h:112
inline int numSpace(char c) { return uwip_numSpace(c); }                             // preprocessing.h:115
// void getHistogram(cv::Mat *imgOriginal, cv::Mat *histogram): the 256 x 1 CV_32F histogram as float[256]
inline void getHistogram(Mat *imgOriginal, float histogram[256]) { uw::getHistogram(defaultContext(), *imgOriginal, histogram); }
// void imgChannelStretch(cv::Mat imgOriginal, cv::Mat imgStretched, int lowerPercentile = 0, int higherPercentile = 100)
inline void imgChannelStretch(Mat imgOriginal, Mat imgStretched, int lowerPercentile = 0, int higherPercentile = 100)
{
    uw::imgChannelStretch(defaultContext(), imgOriginal, imgStretched, lowerPercentile, higherPercentile, 0);
}
inline void imgChannelStretchGPU(Mat imgOriginal, Mat imgStretched, int lowerPercentile = 0, int higherPercentile = 100)
{
    imgChannelStretch(imgOriginal, imgStretched, lowerPercentile, higherPercentile);
}
// float calcOverlap(keyframe *kframe, cv::Mat img_object) / float calcBlur(cv::Mat frame) / float overlapArea(cv::Mat H)
inline float calcOverlap(keyframe *kframe, Mat img_object)
{
    Videostrip &v = defaultVideostrip();
    v.videoWidth = videoWidth; v.videoHeight = videoHeight;
    return v.calcOverlap(kframe, img_object);
}
inline float calcOverlapGPU(keyframe *kframe, Mat img_object) { return calcOverlap(kframe, img_object); }
inline float calcBlur(Mat frame) { return defaultVideostrip().calcBlur(frame); }
inline float calcBlurGPU(Mat frame) { return calcBlur(frame); }
inline float overlapArea(const double H[9])
{
    Videostrip &v = defaultVideostrip();
    v.videoWidth = videoWidth; v.videoHeight = videoHeight;
    return v.overlapArea(H);
}
}  // namespace ref

}  // namespace uw
