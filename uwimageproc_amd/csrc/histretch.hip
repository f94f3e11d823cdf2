// histretch hot path for gfx950.
//
// Three kernels replace the reference's 5-7 full-image passes per letter
// (split, calcHist, add, convertTo, merge; modules/histretch/src/histretch.cpp:
// 245-249 and modules/common/preprocessing.cpp:74-105):
//   k_hist_u8        one coalesced pass over packed rows, per-wave LDS
//                    histograms for all channels at once (H1)
//   k_compose_luts   percentile search + LUT per letter on the 256-bin
//                    histograms only; repeated letters are folded by pushing
//                    the histogram through the LUT, so the image is never
//                    re-read between letters (H2, H4)
//   k_apply_lut      one in-place pass, 16 bytes per lane (H2's add+convertTo)
// Algorithmic HBM bytes for "RGB" on a W*H frame: 3N read + 3N read + 3N write.
#include "uwip_internal.hpp"
#include "device_utils.hpp"

namespace {

struct SpanGeom {
    size_t frame_stride;   // bytes between frames
    size_t span_stride;    // bytes between spans of one frame
    uint32_t nspans;       // spans per frame (1 when rows are contiguous)
    uint32_t span_bytes;   // payload bytes per span (multiple of the channel count)
    uint32_t cps;          // 16-byte chunks per span, ceil(span_bytes/16)
};

// 2-bit channel id of byte j in a 16-byte chunk whose first byte has channel
// phase p (packed BGR: channel = byte offset mod 3, and (16*c) mod 3 == c mod 3).
constexpr uint32_t make_chan_pattern(int p)
{
    uint32_t pat = 0;
    for (int j = 0; j < 16; ++j) pat |= (uint32_t)((p + j) % 3) << (2 * j);
    return pat;
}

__device__ __forceinline__ uint32_t chan_pattern(uint32_t p)
{
    constexpr uint32_t P0 = make_chan_pattern(0), P1 = make_chan_pattern(1), P2 = make_chan_pattern(2);
    return p == 0 ? P0 : (p == 1 ? P1 : P2);
}

template <bool VEC>
__device__ __forceinline__ void load_chunk(const uint8_t *p, uint32_t nvalid, uint32_t w[4])
{
    if (VEC && nvalid == 16) {
        const uint4 v = *reinterpret_cast<const uint4 *>(p);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
        w[0] = w[1] = w[2] = w[3] = 0;
        for (uint32_t j = 0; j < nvalid; ++j) w[j >> 2] |= (uint32_t)p[j] << ((j & 3) * 8);
    }
}

template <bool VEC>
__device__ __forceinline__ void store_chunk(uint8_t *p, uint32_t nvalid, const uint32_t w[4])
{
    if (VEC && nvalid == 16) {
        *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (uint32_t j = 0; j < nvalid; ++j) p[j] = (uint8_t)(w[j >> 2] >> ((j & 3) * 8));
    }
}

// ---- H1: histogram of every channel of every frame ----------------------
template <int C, bool VEC>
__global__ __launch_bounds__(256) void k_hist_u8(const uint8_t *__restrict__ data, SpanGeom g,
                                                 uint32_t units_per_block,
                                                 uint32_t *__restrict__ hist)
{
    __shared__ uint32_t sh[4 * C * 256];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 4 * C * 256; i += 256) sh[i] = 0;
    __syncthreads();
    uint32_t *my = sh + wave * C * 256;

    const uint32_t f = blockIdx.y;
    const uint8_t *fbase = data + (size_t)f * g.frame_stride;
    const uint32_t total = g.nspans * g.cps;
    const uint32_t u0 = blockIdx.x * units_per_block;
    const uint32_t u1 = min(u0 + units_per_block, total);
    for (uint32_t u = u0 + tid; u < u1; u += 256) {
        uint32_t s = 0, c = u;
        if (g.nspans != 1) { s = u / g.cps; c = u - s * g.cps; }
        const uint8_t *p = fbase + (size_t)s * g.span_stride + (size_t)c * 16;
        const uint32_t nvalid = min(16u, g.span_bytes - c * 16u);
        uint32_t w[4];
        load_chunk<VEC>(p, nvalid, w);
        const uint32_t pat = (C == 3) ? chan_pattern(c % 3u) : 0u;
        if (nvalid == 16) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t b = (w[j >> 2] >> ((j & 3) * 8)) & 255u;
                const uint32_t ch = (C == 3) ? ((pat >> (2 * j)) & 3u) : 0u;
                atomicAdd(&my[ch * 256 + b], 1u);
            }
        } else {
            for (uint32_t j = 0; j < nvalid; ++j) {
                const uint32_t b = (w[j >> 2] >> ((j & 3) * 8)) & 255u;
                const uint32_t ch = (C == 3) ? ((pat >> (2 * j)) & 3u) : 0u;
                atomicAdd(&my[ch * 256 + b], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t *out = hist + (size_t)f * C * 256;
    for (int i = tid; i < C * 256; i += 256) {
        const uint32_t s = sh[i] + sh[C * 256 + i] + sh[2 * C * 256 + i] + sh[3 * C * 256 + i];
        if (s) atomicAdd(&out[i], s);
    }
}

// ---- H2: percentile search + LUT on one 256-bin histogram ----------------
// Follows preprocessing.cpp:82-100 in float32.  The reference's serial loop
//   while (sum < hi*norm) { if (sum < lo*norm) lower++; higher++; sum += h[i++]; }
// is evaluated from the exclusive prefix sums: the loop runs for i < I where I is
// the first bin whose exclusive prefix is >= hi*norm (bounded at 256), so
// higher = I-1 and lower = -1 + #{ i < I : prefix_i < lo*norm }.  With at most
// 2^24 pixels every partial float sum is an exact integer, hence identical to the
// integer scan; larger planes take the serial float path.
// Must be called by all 256 threads; cnt = this thread's bin count.
// s_scratch: >= 16 uint32 of LDS.  Returns this thread's LUT entry.
__device__ __forceinline__ uint32_t stretch_lut_entry(uint32_t cnt, const uint32_t *s_hist,
                                                      int rows, int cols, int lo, int hi,
                                                      uint32_t *s_scratch, int *lower_out,
                                                      int *higher_out)
{
    const int v = threadIdx.x;
    const float norm = (float)((double)(rows * cols) / 100.0);
    const float thr_hi = (float)hi * norm, thr_lo = (float)lo * norm;
    int lower, higher;
    if ((uint64_t)rows * (uint64_t)cols <= (1ull << 24)) {
        const uint32_t incl = block256_incl_scan_u32(cnt, s_scratch);
        const float excl = (float)(incl - cnt);
        const bool run_hi = excl < thr_hi;           // monotone: true for bins < I
        const bool run_lo = run_hi && (excl < thr_lo);
        const uint32_t I = block256_sum_u32(run_hi ? 1u : 0u, s_scratch + 4);
        const uint32_t L = block256_sum_u32(run_lo ? 1u : 0u, s_scratch + 8);
        higher = (int)I - 1;
        lower = (int)L - 1;
    } else {
        __syncthreads();
        if (v == 0) {
            float lw = -1.0f, hg = -1.0f, sum = 0.0f;
            int i = 0;
            while (sum < thr_hi && i < 256) {
                if (sum < thr_lo) lw += 1.0f;
                hg += 1.0f;
                sum += (float)s_hist[i];
                i++;
            }
            s_scratch[12] = (uint32_t)(int)lw;
            s_scratch[13] = (uint32_t)(int)hg;
        }
        __syncthreads();
        lower = (int)s_scratch[12];
        higher = (int)s_scratch[13];
    }
    const float m = (float)(255.0 / ((double)higher - (double)lower));
    const int a = v - lower;                                  // img += b  (saturating, b = -lower)
    const float s = (float)min(max(a, 0), 255);
    if (lower_out) *lower_out = lower;
    if (higher_out) *higher_out = higher;
    return sat_u8_rne(s * m);                                 // img *= m  (convertTo)
}

__global__ __launch_bounds__(256) void k_stretch_lut(const uint32_t *__restrict__ hist, int rows,
                                                     int cols, int lo, int hi,
                                                     uint8_t *__restrict__ lut,
                                                     int32_t *__restrict__ bounds)
{
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_scratch[16];
    const int v = threadIdx.x;
    const size_t plane = blockIdx.x;
    const uint32_t cnt = hist[plane * 256 + v];
    s_hist[v] = cnt;
    int lower, higher;
    const uint32_t e = stretch_lut_entry(cnt, s_hist, rows, cols, lo, hi, s_scratch, &lower, &higher);
    lut[plane * 256 + v] = (uint8_t)e;
    if (bounds && v == 0) { bounds[plane * 2] = lower; bounds[plane * 2 + 1] = higher; }
}

// ---- H4: fold an ordered list of letters into one LUT per channel ---------
struct LetterList {
    int n;
    int8_t plane[64];
};

template <int C>
__global__ __launch_bounds__(256) void k_compose_luts(const uint32_t *__restrict__ hist, int rows,
                                                      int cols, int lo, int hi, LetterList L,
                                                      uint8_t *__restrict__ lut)
{
    __shared__ uint32_t s_hist[C * 256];
    __shared__ uint32_t s_new[256];
    __shared__ uint32_t s_step[256];
    __shared__ uint32_t s_scratch[16];
    const int v = threadIdx.x;
    const size_t f = blockIdx.x;
    uint32_t tot[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { s_hist[c * 256 + v] = hist[(f * C + c) * 256 + v]; tot[c] = v; }
    __syncthreads();
    for (int k = 0; k < L.n; ++k) {
        const int ch = L.plane[k];
        const uint32_t cnt = s_hist[ch * 256 + v];
        const uint32_t e = stretch_lut_entry(cnt, s_hist + ch * 256, rows, cols, lo, hi, s_scratch,
                                             nullptr, nullptr);
        s_step[v] = e;
        s_new[v] = 0;
        __syncthreads();
        if (cnt) atomicAdd(&s_new[e], cnt);          // histogram of the stretched plane
#pragma unroll
        for (int c = 0; c < C; ++c) if (c == ch) tot[c] = s_step[tot[c]];
        __syncthreads();
        s_hist[ch * 256 + v] = s_new[v];
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < C; ++c) lut[(f * C + c) * 256 + v] = (uint8_t)tot[c];
}

// ---- H2 apply: in-place LUT over packed rows -----------------------------
template <int C, bool VEC>
__global__ __launch_bounds__(256) void k_apply_lut(uint8_t *__restrict__ data, SpanGeom g,
                                                   uint32_t units_per_block,
                                                   const uint8_t *__restrict__ lut)
{
    __shared__ uint8_t s_lut[C * 256];
    const int tid = threadIdx.x;
    const uint32_t f = blockIdx.y;
    for (int i = tid; i < C * 256; i += 256) s_lut[i] = lut[(size_t)f * C * 256 + i];
    __syncthreads();
    uint8_t *fbase = data + (size_t)f * g.frame_stride;
    const uint32_t total = g.nspans * g.cps;
    const uint32_t u0 = blockIdx.x * units_per_block;
    const uint32_t u1 = min(u0 + units_per_block, total);
    for (uint32_t u = u0 + tid; u < u1; u += 256) {
        uint32_t s = 0, c = u;
        if (g.nspans != 1) { s = u / g.cps; c = u - s * g.cps; }
        uint8_t *p = fbase + (size_t)s * g.span_stride + (size_t)c * 16;
        const uint32_t nvalid = min(16u, g.span_bytes - c * 16u);
        uint32_t w[4], o[4] = {0, 0, 0, 0};
        load_chunk<VEC>(p, nvalid, w);
        const uint32_t pat = (C == 3) ? chan_pattern(c % 3u) : 0u;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t b = (w[j >> 2] >> ((j & 3) * 8)) & 255u;
            const uint32_t ch = (C == 3) ? ((pat >> (2 * j)) & 3u) : 0u;
            o[j >> 2] |= (uint32_t)s_lut[ch * 256 + b] << ((j & 3) * 8);
        }
        store_chunk<VEC>(p, nvalid, o);
    }
}

__global__ void k_identity_lut(uint8_t *lut, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) lut[i] = (uint8_t)(i & 255);
}

// -------------------------------------------------------------------------
struct SpanPlan {
    SpanGeom g;
    bool vec;
    uint32_t units_per_block;
    uint32_t blocks;
};

int make_plan(uwip_ctx *ctx, const uwip_batch_u8 *b, SpanPlan *p)
{
    const size_t rowbytes = (size_t)b->cols * b->channels;
    const bool contiguous = (b->step == rowbytes);
    const uint64_t frame_bytes = (uint64_t)rowbytes * b->rows;
    UWIP_REQUIRE(ctx, frame_bytes < (1ull << 32), "frame larger than 4 GiB");
    p->g.frame_stride = b->frame_stride;
    if (contiguous) {
        p->g.nspans = 1;
        p->g.span_bytes = (uint32_t)frame_bytes;
        p->g.span_stride = 0;
    } else {
        p->g.nspans = (uint32_t)b->rows;
        p->g.span_bytes = (uint32_t)rowbytes;
        p->g.span_stride = b->step;
    }
    p->g.cps = (p->g.span_bytes + 15u) / 16u;
    const uintptr_t base = (uintptr_t)b->data;
    p->vec = (base % 16 == 0) && (b->frames <= 1 || b->frame_stride % 16 == 0) &&
             (contiguous || b->step % 16 == 0);
    const uint64_t total = (uint64_t)p->g.nspans * p->g.cps;
    UWIP_REQUIRE(ctx, total < (1ull << 32), "too many chunks");
    // >= 16 chunks per thread, at most 256 blocks per frame
    uint32_t nb = (uint32_t)((total + 4095) / 4096);
    if (nb < 1) nb = 1;
    if (nb > 256) nb = 256;
    p->units_per_block = (uint32_t)((total + nb - 1) / nb);
    p->blocks = nb;
    return UWIP_OK;
}

int launch_hist(uwip_ctx *ctx, const uwip_batch_u8 *img, uint32_t *d_hist)
{
    SpanPlan p;
    int rc = make_plan(ctx, img, &p);
    if (rc) return rc;
    const int C = img->channels;
    UWIP_HIP(ctx, hipMemsetAsync(d_hist, 0, sizeof(uint32_t) * 256 * (size_t)C * img->frames, ctx->stream));
    dim3 grid(p.blocks, (unsigned)img->frames);
    const uint8_t *data = (const uint8_t *)img->data;
    uwip_kscope ks(ctx, "k_hist_u8");
    if (C == 3) {
        if (p.vec) k_hist_u8<3, true><<<grid, 256, 0, ctx->stream>>>(data, p.g, p.units_per_block, d_hist);
        else k_hist_u8<3, false><<<grid, 256, 0, ctx->stream>>>(data, p.g, p.units_per_block, d_hist);
    } else {
        if (p.vec) k_hist_u8<1, true><<<grid, 256, 0, ctx->stream>>>(data, p.g, p.units_per_block, d_hist);
        else k_hist_u8<1, false><<<grid, 256, 0, ctx->stream>>>(data, p.g, p.units_per_block, d_hist);
    }
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

int launch_apply(uwip_ctx *ctx, const uwip_batch_u8 *img, const uint8_t *d_lut)
{
    SpanPlan p;
    int rc = make_plan(ctx, img, &p);
    if (rc) return rc;
    const int C = img->channels;
    dim3 grid(p.blocks, (unsigned)img->frames);
    uint8_t *data = (uint8_t *)img->data;
    uwip_kscope ks(ctx, "k_apply_lut");
    if (C == 3) {
        if (p.vec) k_apply_lut<3, true><<<grid, 256, 0, ctx->stream>>>(data, p.g, p.units_per_block, d_lut);
        else k_apply_lut<3, false><<<grid, 256, 0, ctx->stream>>>(data, p.g, p.units_per_block, d_lut);
    } else {
        if (p.vec) k_apply_lut<1, true><<<grid, 256, 0, ctx->stream>>>(data, p.g, p.units_per_block, d_lut);
        else k_apply_lut<1, false><<<grid, 256, 0, ctx->stream>>>(data, p.g, p.units_per_block, d_lut);
    }
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

// have_hist: "histretch.hist" already holds this image's histogram (the dehaze writer counted it)
int stretch_planes(uwip_ctx *ctx, const uwip_batch_u8 *img, const LetterList &L, int lo, int hi, bool have_hist = false)
{
    if (uwip_batch_empty(img) || L.n == 0) return UWIP_OK;
    const int C = img->channels;
    const size_t nplanes = (size_t)C * img->frames;
    uint32_t *d_hist = (uint32_t *)uwip_ws(ctx, "histretch.hist", nplanes * 256 * sizeof(uint32_t));
    uint8_t *d_lut = (uint8_t *)uwip_ws(ctx, "histretch.lut", nplanes * 256);
    if (!d_hist || !d_lut) return UWIP_ERR_NOMEM;
    int rc = have_hist ? UWIP_OK : launch_hist(ctx, img, d_hist);
    if (rc) return rc;
    {
        uwip_kscope ks(ctx, "k_compose_luts");
        if (C == 3) k_compose_luts<3><<<img->frames, 256, 0, ctx->stream>>>(d_hist, img->rows, img->cols, lo, hi, L, d_lut);
        else k_compose_luts<1><<<img->frames, 256, 0, ctx->stream>>>(d_hist, img->rows, img->cols, lo, hi, L, d_lut);
        UWIP_HIP(ctx, hipGetLastError());
    }
    return launch_apply(ctx, img, d_lut);
}

}  // namespace

int uwip_launch_hist_internal(uwip_ctx *ctx, const uwip_batch_u8 *img, uint32_t *d_hist)
{
    return launch_hist(ctx, img, d_hist);
}

// ---- exported entry points -------------------------------------------------

UWIP_API int uwip_numChannel(char c)
{
    if (c == 'R' || c == 'H' || c == 'h' || c == 'L' || c == 'Y') return 0;
    if (c == 'G' || c == 'S' || c == 's' || c == 'a' || c == 'C') return 1;
    if (c == 'B' || c == 'V' || c == 'l' || c == 'b' || c == 'X') return 2;
    return -1;
}

UWIP_API int uwip_numSpace(char c)
{
    if (c == 'R' || c == 'G' || c == 'B') return 0;
    if (c == 'H' || c == 'S' || c == 'V') return 1;
    if (c == 'h' || c == 's' || c == 'l') return 2;
    if (c == 'L' || c == 'a' || c == 'b') return 3;
    if (c == 'Y' || c == 'C' || c == 'X') return 4;
    return -1;
}

UWIP_API int uwip_getHistogram(uwip_ctx *ctx, const uwip_batch_u8 *img, uint32_t *d_hist)
{
    int rc = uwip_check_batch(ctx, img, 0);
    if (rc) return rc;
    UWIP_REQUIRE(ctx, d_hist != nullptr || img->frames == 0, "null histogram buffer");
    if (img->frames == 0) return UWIP_OK;
    if (uwip_batch_empty(img)) {
        UWIP_HIP(ctx, hipMemsetAsync(d_hist, 0, sizeof(uint32_t) * 256 * (size_t)img->channels * img->frames, ctx->stream));
        return UWIP_OK;
    }
    return launch_hist(ctx, img, d_hist);
}

UWIP_API int uwip_stretch_lut(uwip_ctx *ctx, const uint32_t *d_hist, int nplanes, int rows, int cols,
                              int lo, int hi, uint8_t *d_lut, int32_t *d_bounds)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    UWIP_REQUIRE(ctx, nplanes >= 0 && rows >= 0 && cols >= 0, "negative extent");
    if (nplanes == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, d_hist && d_lut, "null buffer");
    UWIP_REQUIRE(ctx, (uint64_t)rows * (uint64_t)cols < (1ull << 31), "plane too large");
    uwip_kscope ks(ctx, "k_stretch_lut");
    k_stretch_lut<<<nplanes, 256, 0, ctx->stream>>>(d_hist, rows, cols, lo, hi, d_lut, d_bounds);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

UWIP_API int uwip_apply_lut(uwip_ctx *ctx, const uwip_batch_u8 *img, const uint8_t *d_lut)
{
    int rc = uwip_check_batch(ctx, img, 0);
    if (rc) return rc;
    if (uwip_batch_empty(img)) return UWIP_OK;
    UWIP_REQUIRE(ctx, d_lut != nullptr, "null LUT");
    return launch_apply(ctx, img, d_lut);
}

UWIP_API int uwip_imgChannelStretch(uwip_ctx *ctx, const uwip_batch_u8 *img, int channel, int lo, int hi)
{
    int rc = uwip_check_batch(ctx, img, 0);
    if (rc) return rc;
    UWIP_REQUIRE(ctx, channel >= 0 && channel < img->channels, "channel out of range");
    LetterList L{};
    L.n = 1;
    L.plane[0] = (int8_t)channel;
    return stretch_planes(ctx, img, L, lo, hi);
}

namespace {
// cvtColor(BGR2YCrCb) followed by cvtColor(YCrCb2BGR), 8-bit, in place (OpenCV 3.4 RGB2YCrCb_i / YCrCb2RGB_i: shift 14,
// delta 128; forward 1868 / 9617 / 4899, 11682, 9241; inverse 22987, -11698, -5636, 29049).  parity unpinned.
__device__ __forceinline__ int descale14(int x) { return (x + (1 << 13)) >> 14; }
__device__ __forceinline__ int sat8(int x) { return min(max(x, 0), 255); }
__global__ __launch_bounds__(256) void k_ycrcb_roundtrip(uint8_t *__restrict__ img, size_t step, size_t fs, int rows, int cols)
{
    const int f = blockIdx.z, y = blockIdx.y;
    uint8_t *row = img + (size_t)f * fs + (size_t)y * step;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < cols; x += gridDim.x * 256) {
        const int b = row[3 * x], g = row[3 * x + 1], r = row[3 * x + 2];
        const int Y = descale14(b * 1868 + g * 9617 + r * 4899);
        const int Cr = sat8(descale14((r - Y) * 11682 + (128 << 14)));
        const int Cb = sat8(descale14((b - Y) * 9241 + (128 << 14)));
        const int Ys = sat8(Y);
        row[3 * x] = (uint8_t)sat8(Ys + descale14((Cb - 128) * 29049));
        row[3 * x + 1] = (uint8_t)sat8(Ys + descale14((Cb - 128) * -5636 + (Cr - 128) * -11698));
        row[3 * x + 2] = (uint8_t)sat8(Ys + descale14((Cr - 128) * 22987));
    }
}
}  // namespace

int uwip_cvt_space_internal(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst, int space, int dir, int opencv_rule);   // colorspace.hip

uint32_t *uwip_histretch_hist_ws(uwip_ctx *ctx, const uwip_batch_u8 *img)
{
    return (uint32_t *)uwip_ws(ctx, "histretch.hist", (size_t)img->channels * img->frames * 256 * sizeof(uint32_t));
}

int uwip_histretch_internal(uwip_ctx *ctx, const uwip_batch_u8 *img, const char *letters, int lo, int hi, unsigned flags,
                            bool hist_is_fresh)
{
    int rc = uwip_check_batch(ctx, img, 3);
    if (rc) return rc;
    UWIP_REQUIRE(ctx, letters != nullptr, "null letters");
    UWIP_REQUIRE(ctx, (flags & ~(unsigned)(UWIP_HISTRETCH_FIXED_ORDER | UWIP_HISTRETCH_OPENCV32)) == 0, "unknown flag");
    const bool fixed = (flags & UWIP_HISTRETCH_FIXED_ORDER) != 0;
    const int cv_rule = (flags & UWIP_HISTRETCH_OPENCV32) ? 1 : 0;        // Lab -> BGR: 3.4.x integer form / 3.2 float form
    // Runs of BGR letters are composed into one LUT pass.  A letter of another colour space is, as written in the
    // reference (histretch.cpp:230-241), the 8-bit colour round trip of the image: the stretch goes to a
    // split copy and cvtColor(dst -> src) converts the unstretched planes back before the merge.  With
    // UWIP_HISTRETCH_FIXED_ORDER the evident intent runs instead: convert, stretch the letter's plane, convert back.
    LetterList L{};
    L.n = 0;
    auto flush = [&]() -> int {
        if (L.n == 0) return UWIP_OK;
        const int r2 = stretch_planes(ctx, img, L, lo, hi, hist_is_fresh);
        L.n = 0;
        hist_is_fresh = false;
        return r2;
    };
    for (const char *c = letters; *c; ++c) {
        const int sp = uwip_numSpace(*c);
        if (sp == -1) continue;                        // "not recognized, skipping" (histretch.cpp:252)
        if (sp == 0) {
            if (L.n == 64) { rc = flush(); if (rc) return rc; }
            L.plane[L.n++] = (int8_t)uwip_numChannel(*c);
            continue;
        }
        rc = flush();
        if (rc) return rc;
        hist_is_fresh = false;                         // the image changes below
        if (uwip_batch_empty(img)) continue;
        if (fixed) {
            uwip_batch_u8 tmp = *img;
            tmp.data = uwip_ws(ctx, "histretch.space", (size_t)img->rows * img->cols * 3 * img->frames);
            if (!tmp.data) return UWIP_ERR_NOMEM;
            tmp.step = (size_t)img->cols * 3; tmp.frame_stride = tmp.step * img->rows;
            rc = uwip_cvt_space_internal(ctx, img, &tmp, sp, 0, cv_rule);                 // cvtColor(src, dst, BGR2xxx)   :232
            if (rc) return rc;
            LetterList one{};
            one.n = 1; one.plane[0] = (int8_t)uwip_numChannel(*c);
            rc = stretch_planes(ctx, &tmp, one, lo, hi);                         // split / imgChannelStretch / merge   :234-236,240
            if (rc) return rc;
            rc = uwip_cvt_space_internal(ctx, &tmp, img, sp, 1, cv_rule);                 // cvtColor(dst, src, xxx2BGR) AFTER the merge
            if (rc) return rc;
        } else if (sp == 1) {
            rc = uwip_hsv_roundtrip(ctx, img);
            if (rc) return rc;
        } else if (sp == 4) {
            UWIP_REQUIRE(ctx, img->rows <= 65535 && img->frames <= 65535, "too many rows/frames for one launch");
            uwip_kscope ks(ctx, "k_ycrcb_roundtrip");
            k_ycrcb_roundtrip<<<dim3(uwip_cdiv(img->cols, 256), (unsigned)img->rows, (unsigned)img->frames), 256, 0, ctx->stream>>>(
                (uint8_t *)img->data, img->step, img->frame_stride, img->rows, img->cols);
            UWIP_HIP(ctx, hipGetLastError());
        } else {
            rc = uwip_cvt_space_internal(ctx, img, img, sp, 2, cv_rule);                  // HLS / Lab: the 8-bit round trip
            if (rc) return rc;
        }
    }
    return flush();
}

UWIP_API int uwip_histretch_ex(uwip_ctx *ctx, const uwip_batch_u8 *img, const char *letters, int lo, int hi, unsigned flags)
{
    return uwip_histretch_internal(ctx, img, letters, lo, hi, flags, false);
}

UWIP_API int uwip_histretch(uwip_ctx *ctx, const uwip_batch_u8 *img, const char *letters, int lo, int hi)
{
    return uwip_histretch_ex(ctx, img, letters, lo, hi, 0u);
}
