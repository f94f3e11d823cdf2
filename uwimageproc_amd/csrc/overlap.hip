// videostrip overlap path for gfx950.
//
// calcOverlap (modules/videostrip/src/videostrip.cpp:192-289) = resize ->
// gray -> detect+describe -> brute-force kNN(2) -> ratio test -> RANSAC
// homography -> overlapArea.  The reference delegates detect/describe/match to
// OpenCV-contrib SURF + L2 BFMatcher; BASELINE.json's north_star replaces them
// with an AKAZE-style detector, binary descriptors and a Hamming matcher whose
// dense distance matrix runs on MFMA.  The algorithm is specified in
// DESIGN.md ("overlap stage") and restated independently in
// oracle/uwip_oracle_overlap.c; every float kernel here keeps that
// specification's operation order (-ffp-contract=off), so keypoints,
// descriptors and matches are bit-exact against the oracle.
//
// Everything is batched: one launch covers all frames (grid.z), because a
// 640x360 working image is far too small to fill 256 CUs on its own.
#include "uwip_internal.hpp"
#include "device_utils.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

constexpr int NLEV = 4;
constexpr int MAXKP = 2048;
constexpr int DESC_BYTES = 64;     // packed bits
constexpr int DESC_K = 512;        // unpacked 0/1 bytes for the i8 MFMA
constexpr int DESC_NIBW = 64;      // the same 512 bits as FP4 E2M1 nibbles: 64 dwords = 256 bytes (the f8f6f4 MFMA operand)
constexpr int BORDER = 8;
constexpr float DTHRESH = 0.001f;
constexpr int MIN_INLIERS = 4;       // default = the reference's rule: whatever findHomography returns for >= 4 good matches
                                     // (videostrip.cpp:252-272), i.e. any hypothesis with >= 4 inliers
constexpr int MIN_INLIERS_STRICT = 6;   // UWIP_OVERLAP_MIN6 (uwip_overlap_match_ex): fewer inliers = no homography (-2.0): four chance
                                        // matches always fit one
constexpr float KC_REF = 0.5f;       // contrast factor at and above which the detector threshold is DTHRESH itself
constexpr int RANSAC_ITERS = 512;
constexpr int TW = 640, TH = 480;  // TARGET_WIDTH / TARGET_HEIGHT (videostrip.hpp:48-49)

const float H_SIGMA[NLEV] = {1.6f, 2.2627417f, 3.2f, 4.5254834f};
const int H_SSIZE[NLEV] = {2, 3, 5, 7};
__constant__ int D_SSIZE[NLEV] = {2, 3, 5, 7};

struct Keypoint {
    float x, y, response;
    int32_t level, xi, yi;
    float co, si;        // unit vector of the dominant orientation ((1, 0): upright)
};

struct ConvK {
    int ks;
    float k[16];
};

}  // namespace

// The opaque feature set: everything calcOverlap caches in `struct keyframe`
// (videostrip.hpp:62-68: keypoints + descriptors of a frame), for a batch of frames.
struct uwip_features {
    uwip_ctx *ctx = nullptr;
    int capacity = 0;      // frames
    int frames = 0;        // valid frames
    int w = 0, h = 0;      // working (640-wide) size
    Keypoint *d_kp = nullptr;      // [capacity][MAXKP]
    uint8_t *d_desc = nullptr;     // [capacity][MAXKP][64]   packed
    int8_t *d_bits = nullptr;      // [capacity][MAXKP][512]  0/1 bytes (i8 MFMA operand)
    uint32_t *d_nib = nullptr;     // [capacity][MAXKP][64]   0/1 as FP4 E2M1 nibbles, 0x0 / 0x2 = 0.0 / 1.0 (f8f6f4 MFMA operand)
    int32_t *d_pop = nullptr;      // [capacity][MAXKP]       popcounts
    int32_t *d_n = nullptr;        // [capacity]              keypoint counts
};

namespace {

// 8 descriptor bits -> 8 FP4 E2M1 nibbles (bit j -> nibble j): 0 -> 0b0000 = 0.0, 1 -> 0b0010 = 1.0
__device__ __forceinline__ uint32_t desc_byte_to_nibbles(uint32_t x)
{
    x = (x | (x << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return x << 1;
}

// ---- resize (INTER_LINEAR, 8UC3, fixed point) + BGR2GRAY + /255 ---------------------------
__global__ __launch_bounds__(256) void k_ov_resize_gray(const uint8_t *__restrict__ src, size_t step, size_t fs,
                                                       int rows, int cols, int oh, int ow,
                                                       const int *__restrict__ xo, const short *__restrict__ xa,
                                                       const short *__restrict__ xb, const int *__restrict__ yo,
                                                       const short *__restrict__ ya, const short *__restrict__ yb,
                                                       uint8_t *__restrict__ gray, float *__restrict__ L0)
{
    const int f = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= ow || y >= oh) return;
    const uint8_t *b = src + (size_t)f * fs;
    const int sy = yo[y], sy1 = sy + 1 < rows ? sy + 1 : sy;
    const uint8_t *r0 = b + (size_t)sy * step, *r1 = b + (size_t)sy1 * step;
    const int sx = xo[x], sx1 = sx + 1 < cols ? sx + 1 : sx;
    const int a0 = xa[x], a1 = xb[x], b0 = ya[y], b1 = yb[y];
    int px[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int S0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
        const int S1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
        const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        px[c] = min(max(v, 0), 255);
    }
    const int g = (px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + 8192) >> 14;
    const size_t o = ((size_t)f * oh + y) * ow + x;
    gray[o] = (uint8_t)g;
    L0[o] = (float)g / 255.0f;
}

// cv::resize(frame, res_frame, Size(), f, f) alone (main.cpp:242,287,311): the 8UC3 result the reference hands to
// calcOverlap and calcBlur.  Same fixed-point arithmetic as the fused kernel above.
__global__ __launch_bounds__(256) void k_ov_resize_bgr(const uint8_t *__restrict__ src, size_t step, size_t fs, int rows, int cols,
                                                      int oh, int ow, const int *__restrict__ xo, const short *__restrict__ xa,
                                                      const short *__restrict__ xb, const int *__restrict__ yo,
                                                      const short *__restrict__ ya, const short *__restrict__ yb,
                                                      uint8_t *__restrict__ dst, size_t dstep, size_t dfs)
{
    const int f = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= ow || y >= oh) return;
    const uint8_t *b = src + (size_t)f * fs;
    const int sy = yo[y], sy1 = sy + 1 < rows ? sy + 1 : sy;
    const uint8_t *r0 = b + (size_t)sy * step, *r1 = b + (size_t)sy1 * step;
    const int sx = xo[x], sx1 = sx + 1 < cols ? sx + 1 : sx;
    const int a0 = xa[x], a1 = xb[x], b0 = ya[y], b1 = yb[y];
    uint8_t *o = dst + (size_t)f * dfs + (size_t)y * dstep + (size_t)x * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int S0 = r0[sx * 3 + c] * a0 + r0[sx1 * 3 + c] * a1;
        const int S1 = r1[sx * 3 + c] * a0 + r1[sx1 * 3 + c] * a1;
        const int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        o[c] = (uint8_t)min(max(v, 0), 255);
    }
}

// gray u8 (already at working size) -> L0
__global__ void k_ov_gray_to_L0(const uint8_t *__restrict__ gray, float *__restrict__ L0, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) L0[i] = (float)gray[i] / 255.0f;
}

// ---- separable Gaussian, reflect-101 border ----------------------------------------------
// Both passes of the separable Gaussian in one launch: a block owns a 64 x 16 output tile, stages the tile + halo once
// (reflect-101 applied to the global indices), convolves along x into a second LDS plane (tile rows + halo rows) and
// along y out of it.  Every output goes through the same multiplies and adds in the same order as the two-pass form
// (the intermediate row of image row reflect101(y) is what the y pass of the two-pass form reads there).
constexpr int CV_TW = 64, CV_TH = 16, CV_RMAX = 7;
// KS = the kernel size when it is one of the usual ones (5 for sigma 1, 9 for sigma 1.6: loops unrolled, the staging index a
// constant division), 0 = any odd size up to 2 CV_RMAX + 1 at run time.
template <int KS>
__global__ __launch_bounds__(256) void k_ov_conv2(const float *__restrict__ in, float *__restrict__ out, int h, int w, ConvK K)
{
    __shared__ float s_in[(CV_TH + 2 * CV_RMAX) * (CV_TW + 2 * CV_RMAX)];
    __shared__ float s_tmp[(CV_TH + 2 * CV_RMAX) * CV_TW];
    const int f = blockIdx.z, x0 = blockIdx.x * CV_TW, y0 = blockIdx.y * CV_TH;
    const float *I = in + (size_t)f * h * w;
    const int ks = KS ? KS : K.ks;
    const int r = ks / 2, RW = CV_TW + 2 * r, RH = CV_TH + 2 * r;
    const bool inside = x0 - r >= 0 && y0 - r >= 0 && x0 - r + RW <= w && y0 - r + RH <= h;     // block-uniform
    if (inside) {
        const float *base = I + (size_t)(y0 - r) * w + (x0 - r);
        for (int i = threadIdx.x; i < RH * RW; i += 256) {
            const int ry = i / RW, rx = i - ry * RW;
            s_in[i] = base[(size_t)ry * w + rx];
        }
    } else {
        for (int i = threadIdx.x; i < RH * RW; i += 256) {
            const int ry = i / RW, rx = i - ry * RW;
            s_in[i] = I[(size_t)reflect101(y0 - r + ry, h) * w + reflect101(x0 - r + rx, w)];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < RH * CV_TW; i += 256) {
        const int ry = i / CV_TW, tx = i - ry * CV_TW;
        const float *row = s_in + ry * RW + tx;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < ks; ++k) acc = acc + K.k[k] * row[k];
        s_tmp[i] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CV_TH * CV_TW; i += 256) {
        const int ty = i / CV_TW, tx = i - ty * CV_TW;
        const float *col = s_tmp + ty * CV_TW + tx;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < ks; ++k) acc = acc + K.k[k] * col[k * CV_TW];
        const int x = x0 + tx, y = y0 + ty;
        if (x < w && y < h) out[((size_t)f * h + y) * w + x] = acc;
    }
}
template <class... A>
static void launch_conv2(int ks, dim3 grid, hipStream_t st, A... a)
{
    switch (ks) {
    case 5: k_ov_conv2<5><<<grid, 256, 0, st>>>(a...); break;
    case 7: k_ov_conv2<7><<<grid, 256, 0, st>>>(a...); break;
    case 9: k_ov_conv2<9><<<grid, 256, 0, st>>>(a...); break;
    case 11: k_ov_conv2<11><<<grid, 256, 0, st>>>(a...); break;
    default: k_ov_conv2<0><<<grid, 256, 0, st>>>(a...); break;
    }
}

// INSIDE: the caller knows that the 3 x 3 neighbourhood lies in the image (block-uniform test): no border rule
template <bool INSIDE = false>
__device__ __forceinline__ void scharr_at(const float *I, int h, int w, int y, int x, float &gx, float &gy)
{
    const int ym = INSIDE ? y - 1 : reflect101(y - 1, h), yp = INSIDE ? y + 1 : reflect101(y + 1, h),
              xm = INSIDE ? x - 1 : reflect101(x - 1, w), xp = INSIDE ? x + 1 : reflect101(x + 1, w);
    const float a0 = I[(size_t)ym * w + xm], a1 = I[(size_t)ym * w + x], a2 = I[(size_t)ym * w + xp];
    const float b0 = I[(size_t)y * w + xm], b2 = I[(size_t)y * w + xp];
    const float c0 = I[(size_t)yp * w + xm], c1 = I[(size_t)yp * w + x], c2 = I[(size_t)yp * w + xp];
    float t0 = 3.0f * (a2 - a0), t1 = 10.0f * (b2 - b0), t2 = 3.0f * (c2 - c0);
    gx = (t0 + t1) + t2;
    t0 = 3.0f * (c0 - a0); t1 = 10.0f * (c1 - a1); t2 = 3.0f * (c2 - a2);
    gy = (t0 + t1) + t2;
}

// ---- contrast factor: 70th percentile of the gradient-magnitude histogram ------------------
// PASS 0: per-frame max (float bits as uint, values >= 0).  PASS 1: 300-bin histogram.
constexpr int KC_ROWS = 8;      // groups of four rows per block of k_ov_kc
template <int PASS>
__global__ __launch_bounds__(256) void k_ov_kc(const float *__restrict__ Lsm, int h, int w, uint32_t *__restrict__ hmax_bits,
                                              uint32_t *__restrict__ hist /*[F][304]*/)
{
    // gradient magnitudes of a smooth frame crowd into a few low bins: KC_REP copies of the histogram keyed by the lane,
    // 304 + 1 words apart (equal bins of neighbouring copies in different banks), so that one atomic instruction rarely
    // sends many lanes to one word
    constexpr int KC_REP = 8, KC_STRIDE = 305;
    __shared__ uint32_t s_hist[PASS == 1 ? KC_REP * KC_STRIDE : 1];
    const int f = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const float *I = Lsm + (size_t)f * h * w;
    if (PASS == 1) {
        for (int i = threadIdx.x; i < KC_REP * KC_STRIDE; i += 256) s_hist[i] = 0;
        __syncthreads();
    }
    const float hmax = PASS == 1 ? __uint_as_float(hmax_bits[f]) : 0.0f;
    uint32_t bmax = 0;
    // a block walks KC_ROWS groups of four rows: its 300 global atomics (one partial maximum) are paid once per 64 x 32
    // pixels -- at one group per block 900 blocks of a frame queued on the same few hundred L2 words
#pragma unroll 2
    for (int ry = 0; ry < KC_ROWS; ++ry) {
        const int y = (blockIdx.y * KC_ROWS + ry) * 4 + (threadIdx.x >> 6);
        const bool in = x >= 1 && x < w - 1 && y >= 1 && y < h - 1;
        float m = 0.0f;
        if (in) {
            float gx, gy;
            scharr_at<true>(I, h, w, y, x, gx, gy);      // in: x +- 1, y +- 1 are pixels of the image
            m = sqrtf(gx * gx + gy * gy);
        }
        if (PASS == 0) bmax = max(bmax, __float_as_uint(m));
        else if (in && m != 0.0f && hmax != 0.0f) {
            int nbin = (int)floorf(300.0f * (m / hmax));
            if (nbin >= 300) nbin = 299;
            atomicAdd(&s_hist[(threadIdx.x & (KC_REP - 1)) * KC_STRIDE + nbin], 1u);
            // npoints = the sum of the bins (k_ov_kc_final): a counter word of its own would take every lane through one address
        }
    }
    if (PASS == 0) {
        uint32_t b = bmax;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) b = max(b, (uint32_t)__shfl_xor((int)b, d, 64));
        // one partial per block, reduced by k_ov_kc_max (hundreds of blocks polling one word of a frame serialise
        // on a single L2 channel; max is exact whatever the order)
        __shared__ uint32_t s_mx[4];
        if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = b;
        __syncthreads();
        if (threadIdx.x == 0)
            hist[((size_t)f * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
    } else {
        __syncthreads();
        for (int i = threadIdx.x; i < 300; i += 256) {
            uint32_t c = 0;
#pragma unroll
            for (int r = 0; r < KC_REP; ++r) c += s_hist[r * KC_STRIDE + i];
            if (c) atomicAdd(&hist[(size_t)f * 304 + i], c);
        }
    }
}

__global__ __launch_bounds__(256) void k_ov_kc_max(const uint32_t *__restrict__ part, int nb, uint32_t *__restrict__ hmax_bits)
{
    __shared__ uint32_t s_mx[4];
    const int f = blockIdx.x;
    uint32_t b = 0;
    for (int i = threadIdx.x; i < nb; i += 256) b = max(b, part[(size_t)f * nb + i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) b = max(b, (uint32_t)__shfl_xor((int)b, d, 64));
    if ((threadIdx.x & 63) == 0) s_mx[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) hmax_bits[f] = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
}

__global__ void k_ov_kc_final(const uint32_t *__restrict__ hmax_bits, const uint32_t *__restrict__ hist, float *__restrict__ kc, int F)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const float hmax = __uint_as_float(hmax_bits[f]);
    if (hmax == 0.0f) { kc[f] = 0.03f; return; }
    const uint32_t *hh = hist + (size_t)f * 304;
    int npoints = 0;
    for (int k = 0; k < 300; ++k) npoints += (int)hh[k];
    const int nthreshold = (int)((float)npoints * 0.7f);
    int k = 0, nelements = 0;
    for (k = 0; nelements < nthreshold && k < 300; k++) nelements += (int)hh[k];
    kc[f] = nelements < nthreshold ? 0.03f : hmax * ((float)k / 300.0f);
}

// (the Perona-Malik g2 conductivity is written by k_ov_deriv1, which reads the same smoothed plane)

// ---- one explicit FED diffusion step ----------------------------------------------------------
__global__ __launch_bounds__(256) void k_ov_fed(const float *__restrict__ Lin, const float *__restrict__ cin, float *__restrict__ out,
                                               int h, int w, float tau)
{
    const int f = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float *L = Lin + (size_t)f * h * w, *c = cin + (size_t)f * h * w;
    const float step = 0.5f * tau;
    const int ym = y > 0 ? y - 1 : 0, yp = y < h - 1 ? y + 1 : h - 1;
    const int xm = x > 0 ? x - 1 : 0, xp = x < w - 1 ? x + 1 : w - 1;
    const size_t i = (size_t)y * w + x;
    const float xpos = (c[i] + c[(size_t)y * w + xp]) * (L[(size_t)y * w + xp] - L[i]);
    const float xneg = (c[(size_t)y * w + xm] + c[i]) * (L[i] - L[(size_t)y * w + xm]);
    const float ypos = (c[i] + c[(size_t)yp * w + x]) * (L[(size_t)yp * w + x] - L[i]);
    const float yneg = (c[(size_t)ym * w + x] + c[i]) * (L[i] - L[(size_t)ym * w + x]);
    float d = xpos - xneg;
    d = d + ypos;
    d = d - yneg;
    out[(size_t)f * h * w + i] = L[i] + step * d;
}

// Two consecutive FED steps in one launch: a block stages L and the conductivity of its 64 x 16 tile + 2 halo pixels
// (loaded at edge-clamped coordinates, which is exactly the neighbour rule of k_ov_fed), takes step 1 on the tile + 1
// halo into LDS and step 2 out of it.  Every value goes through the same operations in the same order as two
// k_ov_fed launches.
constexpr int FD_TW = 64, FD_TH = 16;
__device__ __forceinline__ float fed_px(float Lc, float Lxm, float Lxp, float Lym, float Lyp, float cc, float cxm, float cxp,
                                        float cym, float cyp, float step)
{
    const float xpos = (cc + cxp) * (Lxp - Lc);
    const float xneg = (cxm + cc) * (Lc - Lxm);
    const float ypos = (cc + cyp) * (Lyp - Lc);
    const float yneg = (cym + cc) * (Lc - Lym);
    float d = xpos - xneg;
    d = d + ypos;
    d = d - yneg;
    return Lc + step * d;
}
__global__ __launch_bounds__(256) void k_ov_fed2(const float *__restrict__ Lin, const float *__restrict__ cin, float *__restrict__ out,
                                                int h, int w, float tau1, float tau2)
{
    constexpr int W2 = FD_TW + 4, H2 = FD_TH + 4, W1 = FD_TW + 2, H1 = FD_TH + 2;
    __shared__ float s_L[H2 * W2], s_c[H2 * W2], s_M[H1 * W1];
    const int f = blockIdx.z, x0 = blockIdx.x * FD_TW, y0 = blockIdx.y * FD_TH;
    const float *L = Lin + (size_t)f * h * w, *c = cin + (size_t)f * h * w;
    for (int i = threadIdx.x; i < H2 * W2; i += 256) {
        const int ry = i / W2, rx = i - ry * W2;
        const int y = min(max(y0 - 2 + ry, 0), h - 1), x = min(max(x0 - 2 + rx, 0), w - 1);
        s_L[i] = L[(size_t)y * w + x];
        s_c[i] = c[(size_t)y * w + x];
    }
    __syncthreads();
    const float step1 = 0.5f * tau1, step2 = 0.5f * tau2;
    // step 1 at tile + 1 halo: local (ly, lx) of s_M is local (ly + 1, lx + 1) of s_L
    for (int i = threadIdx.x; i < H1 * W1; i += 256) {
        const int ly = i / W1, lx = i - ly * W1;
        const int o = (ly + 1) * W2 + lx + 1;
        s_M[i] = fed_px(s_L[o], s_L[o - 1], s_L[o + 1], s_L[o - W2], s_L[o + W2], s_c[o], s_c[o - 1], s_c[o + 1], s_c[o - W2],
                        s_c[o + W2], step1);
    }
    __syncthreads();
    // step 2 on the tile: the neighbours of an image-border pixel are the pixel itself (indices clamped in the image)
    for (int i = threadIdx.x; i < FD_TH * FD_TW; i += 256) {
        const int ty = i / FD_TW, tx = i - ty * FD_TW;
        const int x = x0 + tx, y = y0 + ty;
        if (x >= w || y >= h) continue;
        const int xm = max(x - 1, 0) - (x0 - 1), xp = min(x + 1, w - 1) - (x0 - 1);
        const int ym = max(y - 1, 0) - (y0 - 1), yp = min(y + 1, h - 1) - (y0 - 1);
        const int mx = tx + 1, my = ty + 1;
        // conductivity from the clamped-load plane: local (ly, lx) of s_M is (ly + 1, lx + 1) of s_c
        out[((size_t)f * h + y) * w + x] =
            fed_px(s_M[my * W1 + mx], s_M[my * W1 + xm], s_M[my * W1 + xp], s_M[ym * W1 + mx], s_M[yp * W1 + mx],
                   s_c[(my + 1) * W2 + mx + 1], s_c[(my + 1) * W2 + xm + 1], s_c[(my + 1) * W2 + xp + 1],
                   s_c[(ym + 1) * W2 + mx + 1], s_c[(yp + 1) * W2 + mx + 1], step2);
    }
}

// NS consecutive FED steps in one launch (NS <= FDN_MAX): the tile + NS halo pixels of L and of the conductivity are staged
// once (edge-clamped loads), every step shrinks the valid region by one pixel, ping-ponging between two LDS planes.  A
// neighbour index is clamped INSIDE THE IMAGE before it is turned into a plane index -- k_ov_fed's border rule -- so values
// computed at out-of-image halo positions are never read.  Same operations in the same order as NS launches of k_ov_fed;
// the planes travel through L2 / HBM once instead of NS times.
constexpr int FDN_MAX = 4;
struct FedTaus { float t[FDN_MAX]; };
template <int NS>
__global__ __launch_bounds__(256) void k_ov_fedn(const float *__restrict__ Lin, const float *__restrict__ cin, float *__restrict__ out,
                                                int h, int w, FedTaus taus)
{
    constexpr int W2 = FD_TW + 2 * NS, H2 = FD_TH + 2 * NS;
    __shared__ float s_c[H2 * W2], s_A[H2 * W2], s_B[H2 * W2];
    const int f = blockIdx.z, x0 = blockIdx.x * FD_TW - NS, y0 = blockIdx.y * FD_TH - NS;   // image coordinates of plane (0, 0)
    const float *L = Lin + (size_t)f * h * w, *c = cin + (size_t)f * h * w;
    for (int i = threadIdx.x; i < H2 * W2; i += 256) {
        const int ry = i / W2, rx = i - ry * W2;
        const int y = min(max(y0 + ry, 0), h - 1), x = min(max(x0 + rx, 0), w - 1);
        s_A[i] = L[(size_t)y * w + x];
        s_c[i] = c[(size_t)y * w + x];
    }
    __syncthreads();
    float *src = s_A, *dst = s_B;
    // a plane that lies inside the image (three tiles in four at 640 x 360) needs no border rule: neighbours are +-1, +-W2
    const bool inside = x0 >= 0 && y0 >= 0 && x0 + W2 <= w && y0 + H2 <= h;      // block-uniform
#pragma unroll
    for (int k = 1; k <= NS; ++k) {
        const float step = 0.5f * taus.t[k - 1];
        const int RW = W2 - 2 * k, RH = H2 - 2 * k;          // region of this step: plane coordinates [k, W2 - k) x [k, H2 - k)
        if (inside) {
            for (int i = threadIdx.x; i < RH * RW; i += 256) {
                const int q = i / RW;
                const int o = (q + k) * W2 + (i - q * RW) + k;
                const float v = fed_px(src[o], src[o - 1], src[o + 1], src[o - W2], src[o + W2], s_c[o], s_c[o - 1], s_c[o + 1],
                                       s_c[o - W2], s_c[o + W2], step);
                if (k < NS) dst[o] = v;
                else out[((size_t)f * h + (y0 + q + k)) * w + x0 + (i - q * RW) + k] = v;
            }
        } else
        for (int i = threadIdx.x; i < RH * RW; i += 256) {
            const int ry = i / RW + k, rx = i - (i / RW) * RW + k;
            const int x = x0 + rx, y = y0 + ry;
            const int xm = min(max(x - 1, 0), w - 1) - x0, xp = min(max(x + 1, 0), w - 1) - x0;
            const int ym = min(max(y - 1, 0), h - 1) - y0, yp = min(max(y + 1, 0), h - 1) - y0;
            // an out-of-image position has its neighbour indices clamped onto in-plane positions as well (|delta| <= 1 from
            // a clamped coordinate): harmless, its value is never used
            const int o = ry * W2 + rx;
            const int oxm = ry * W2 + min(max(xm, 0), W2 - 1), oxp = ry * W2 + min(max(xp, 0), W2 - 1);
            const int oym = min(max(ym, 0), H2 - 1) * W2 + rx, oyp = min(max(yp, 0), H2 - 1) * W2 + rx;
            const float v = fed_px(src[o], src[oxm], src[oxp], src[oym], src[oyp], s_c[o], s_c[oxm], s_c[oxp], s_c[oym], s_c[oyp], step);
            if (k < NS) dst[o] = v;
            else if (x < w && y < h) out[((size_t)f * h + y) * w + x] = v;   // k == NS: the region is the tile itself
        }
        if (k < NS) {
            __syncthreads();
            float *t = src; src = dst; dst = t;
        }
    }
}

// ---- scale-s first derivative (taps at -s, 0, +s) ------------------------------------------------
template <bool INSIDE = false>
__device__ __forceinline__ float deriv_at(const float *I, int h, int w, int y, int x, int s, bool along_x)
{
    const float wgt = 10.0f / 3.0f;
    const float norm = 1.0f / (2.0f * (float)s * (wgt + 2.0f));
    const float wn = wgt * norm;
    const int ym = INSIDE ? y - s : reflect101(y - s, h), yp = INSIDE ? y + s : reflect101(y + s, h);
    const int xm = INSIDE ? x - s : reflect101(x - s, w), xp = INSIDE ? x + s : reflect101(x + s, w);
    float t0, t1, t2;
    if (along_x) {
        t0 = norm * (I[(size_t)ym * w + xp] - I[(size_t)ym * w + xm]);
        t1 = wn * (I[(size_t)y * w + xp] - I[(size_t)y * w + xm]);
        t2 = norm * (I[(size_t)yp * w + xp] - I[(size_t)yp * w + xm]);
    } else {
        t0 = norm * (I[(size_t)yp * w + xm] - I[(size_t)ym * w + xm]);
        t1 = wn * (I[(size_t)yp * w + x] - I[(size_t)ym * w + x]);
        t2 = norm * (I[(size_t)yp * w + xp] - I[(size_t)ym * w + xp]);
    }
    return (t0 + t1) + t2;
}

// also writes the Perona-Malik conductivity of the same smoothed plane when `flow` is given (k_ov_flow's arithmetic: the
// two kernels read the same plane, one launch and one read of it instead of two)
__global__ __launch_bounds__(256) void k_ov_deriv1(const float *__restrict__ Lsm, float2 *__restrict__ Lxy, int h, int w, int s,
                                                  const float *__restrict__ kc, float *__restrict__ flow)
{
    const int f = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float *I = Lsm + (size_t)f * h * w;
    const size_t o = ((size_t)f * h + y) * w + x;
    // block-uniform: every tap of the block's 64 x 4 pixels lies in the image (s >= 1 covers the Scharr taps too)
    const int bx0 = blockIdx.x * 64, by0 = blockIdx.y * 4;
    const bool inside = bx0 - s >= 0 && bx0 + 63 + s < w && by0 - s >= 0 && by0 + 3 + s < h;
    float gx, gy;
    // (Lx, Ly) as ONE float2 plane: k_ov_ldet and k_ov_describe want both at the same positions -- one gather instead of two
    if (inside) {
        Lxy[o] = make_float2(deriv_at<true>(I, h, w, y, x, s, true), deriv_at<true>(I, h, w, y, x, s, false));
        if (flow) scharr_at<true>(I, h, w, y, x, gx, gy);
    } else {
        Lxy[o] = make_float2(deriv_at(I, h, w, y, x, s, true), deriv_at(I, h, w, y, x, s, false));
        if (flow) scharr_at(I, h, w, y, x, gx, gy);
    }
    if (flow) {
        const float k = kc[f];
        const float inv_k = 1.0f / (k * k);
        flow[o] = 1.0f / (1.0f + (gx * gx + gy * gy) * inv_k);
    }
}

// second derivatives of one pixel from the (Lx, Ly) plane: Lxx = d/dx of Lx, Lyy = d/dy of Ly, Lxy = d/dy of Lx -- deriv_at's
// operations on eight float2 taps (the four corner taps serve all three, the two d/dy centre taps serve Lyy and Lxy)
template <bool INSIDE>
__device__ __forceinline__ void second_derivs(const float2 *P, int h, int w, int y, int x, int s, float &lxx, float &lyy, float &lxy)
{
    const float wgt = 10.0f / 3.0f;
    const float norm = 1.0f / (2.0f * (float)s * (wgt + 2.0f));
    const float wn = wgt * norm;
    const int ym = INSIDE ? y - s : reflect101(y - s, h), yp = INSIDE ? y + s : reflect101(y + s, h);
    const int xm = INSIDE ? x - s : reflect101(x - s, w), xp = INSIDE ? x + s : reflect101(x + s, w);
    const float2 mm = P[(size_t)ym * w + xm], m0 = P[(size_t)ym * w + x], mp = P[(size_t)ym * w + xp];
    const float2 zm = P[(size_t)y * w + xm], zp = P[(size_t)y * w + xp];
    const float2 pm = P[(size_t)yp * w + xm], p0 = P[(size_t)yp * w + x], pp = P[(size_t)yp * w + xp];
    float t0 = norm * (mp.x - mm.x), t1 = wn * (zp.x - zm.x), t2 = norm * (pp.x - pm.x);
    lxx = (t0 + t1) + t2;
    t0 = norm * (pm.y - mm.y); t1 = wn * (p0.y - m0.y); t2 = norm * (pp.y - mp.y);
    lyy = (t0 + t1) + t2;
    t0 = norm * (pm.x - mm.x); t1 = wn * (p0.x - m0.x); t2 = norm * (pp.x - mp.x);
    lxy = (t0 + t1) + t2;
}
__global__ __launch_bounds__(256) void k_ov_ldet(const float2 *__restrict__ Lxy, float *__restrict__ Ldet, int h, int w, int s)
{
    const int f = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float2 *P = Lxy + (size_t)f * h * w;
    const int bx0 = blockIdx.x * 64, by0 = blockIdx.y * 4;
    const bool inside = bx0 - s >= 0 && bx0 + 63 + s < w && by0 - s >= 0 && by0 + 3 + s < h;     // block-uniform
    float lxx, lyy, lxy;
    if (inside) second_derivs<true>(P, h, w, y, x, s, lxx, lyy, lxy);
    else second_derivs<false>(P, h, w, y, x, s, lxx, lyy, lxy);
    const float ss = (float)(s * s), s4 = ss * ss;
    Ldet[((size_t)f * h + y) * w + x] = (lxx * lyy - lxy * lxy) * s4;
}

// ---- extrema: candidate response map over all levels ------------------------------------------------
// Ldet: [NLEV][F][h][w] (each level a dense batch); cand: [F][NLEV][h][w], response or 0
// The detector threshold is relative to the frame's contrast factor k (det of the Hessian scales with contrast squared;
// raw frames of turbid water have no response above a fixed 1e-3): DTHRESH * min(1, (k / KC_REF)^2), in the oracle's
// operations (UWIP_OVERLAP_RELATIVE_THRESHOLD); `fixed` keeps DTHRESH (the default).
// One block = a 64 x 8 tile of ALL levels: the four level tiles + 1 halo pixel are staged in LDS once (coalesced rows), and
// every comparison of the 3 x 3 x 3 test and the sub-pixel check reads LDS -- the dense map of every level is read once
// (x 1.29 for the halo) instead of once plus 26 scattered neighbour loads wherever any lane of a wave passes the threshold.
constexpr int EX_TW = 64, EX_TH = 8, EX_PW = EX_TW + 2, EX_PH = EX_TH + 2;
__global__ __launch_bounds__(256) void k_ov_extrema(const float *__restrict__ Ldet, float *__restrict__ cand, int h, int w, int F,
                                                   const float *__restrict__ kc, int fixed, uint32_t *__restrict__ selhist)
{
    __shared__ float s_D[NLEV][EX_PH * EX_PW];
    const int f = blockIdx.z, x0 = blockIdx.x * EX_TW, y0 = blockIdx.y * EX_TH;
    const size_t n = (size_t)h * w;
    for (int i = threadIdx.x; i < EX_PH * EX_PW; i += 256) {
        const int ry = i / EX_PW, rx = i - ry * EX_PW;
        const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
        const bool in = gy >= 0 && gy < h && gx >= 0 && gx < w;       // positions outside are never compared (BORDER >= 1)
#pragma unroll
        for (int lv = 0; lv < NLEV; ++lv) s_D[lv][i] = in ? Ldet[((size_t)lv * F + f) * n + (size_t)gy * w + gx] : 0.0f;
    }
    __syncthreads();
    const float kr = kc[f] / KC_REF;
    float ks = kr * kr;
    if (!(ks < 1.0f)) ks = 1.0f;
    const float dthr = fixed ? DTHRESH : DTHRESH * ks;
    const int tx = threadIdx.x & 63, x = x0 + tx;
#pragma unroll
    for (int k = 0; k < EX_TH / 4; ++k) {
        const int ty = (threadIdx.x >> 6) + 4 * k, y = y0 + ty;
        if (x >= w || y >= h) continue;
        const bool inb = x >= BORDER && x < w - BORDER && y >= BORDER && y < h - BORDER;
        const int c = (ty + 1) * EX_PW + tx + 1;
#pragma unroll
        for (int lv = 0; lv < NLEV; ++lv) {
            const float *D = s_D[lv];
            float out = 0.0f;
            const float v = D[c];
            bool ok = inb && v > dthr;
            if (ok) {
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx)
                        if ((dx != 0 || dy != 0) && !(v > D[c + dy * EX_PW + dx])) ok = false;
            }
            if (ok) {
#pragma unroll
                for (int o = -1; o <= 1; o += 2) {
                    const int l2 = lv + o;
                    if (l2 < 0 || l2 >= NLEV) continue;
                    const float *E = s_D[l2];
#pragma unroll
                    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                        for (int dx = -1; dx <= 1; ++dx)
                            if (!(v > E[c + dy * EX_PW + dx])) ok = false;
                }
            }
            if (ok) {
                // candidates that the sub-pixel refinement would discard are dropped here, so that
                // the top-K selection sees exactly the oracle's candidate list
                const float vxp = D[c + 1], vxm = D[c - 1];
                const float vyp = D[c + EX_PW], vym = D[c - EX_PW];
                const float Dx = 0.5f * (vxp - vxm), Dy = 0.5f * (vyp - vym);
                const float Dxx = (vxp + vxm) - 2.0f * v, Dyy = (vyp + vym) - 2.0f * v;
                const float Dxy = 0.25f * (D[c + EX_PW + 1] + D[c - EX_PW - 1]) - 0.25f * (D[c + EX_PW - 1] + D[c - EX_PW + 1]);
                const float det = Dxx * Dyy - Dxy * Dxy;
                if (det == 0.0f) ok = false;
                else {
                    const float ox = -(Dyy * Dx - Dxy * Dy) / det, oy = -(Dxx * Dy - Dxy * Dx) / det;
                    if (!(fabsf(ox) <= 1.0f && fabsf(oy) <= 1.0f)) ok = false;
                }
                if (ok) {
                    out = v;
                    // first pass of the top-K radix select (k_ov_sel_hist<0>'s histogram of the high 16 response bits) counted
                    // here: candidates are a few thousand per frame, and the dense map is read once less
                    atomicAdd(&selhist[(size_t)f * 65536 + (__float_as_uint(v) >> 16)], 1u);
                }
            }
            cand[((size_t)f * NLEV + lv) * n + (size_t)y * w + x] = out;
        }
    }
}

// ---- top-K selection: 2-pass radix select on the float bit patterns ---------------------------------
// hist: [F][65536]; sel: [F][4] = {prefix, remaining, total, threshold_bits}
template <int PASS>
__global__ __launch_bounds__(256) void k_ov_sel_hist(const float *__restrict__ cand, size_t n4, uint32_t *__restrict__ hist,
                                                    const uint32_t *__restrict__ sel)
{
    const int f = blockIdx.y;
    const float *c = cand + (size_t)f * n4;
    const uint32_t prefix = PASS == 1 ? sel[(size_t)f * 4] : 0u;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const uint32_t b = __float_as_uint(c[i]);
        if (b == 0) continue;
        if (PASS == 1 && (b >> 16) != prefix) continue;
        atomicAdd(&hist[(size_t)f * 65536 + (PASS == 0 ? (b >> 16) : (b & 0xffffu))], 1u);
    }
}

// one block per frame: walk the 65536 bins from the top until `remaining` candidates are covered
template <int PASS>
__global__ __launch_bounds__(256) void k_ov_sel_pick(const uint32_t *__restrict__ hist, uint32_t *__restrict__ sel)
{
    __shared__ uint32_t s_sum[256];
    __shared__ uint32_t s_scratch[8];
    const int f = blockIdx.x, t = threadIdx.x;
    const uint32_t *hh = hist + (size_t)f * 65536;
    // thread t owns bins [65535 - 256 t - 255, 65535 - 256 t] (descending order across threads)
    uint32_t mine = 0;
    const int top = 65535 - 256 * t;
    for (int b = top; b > top - 256; --b) mine += hh[b];
    const uint32_t incl = block256_incl_scan_u32(mine, s_scratch);
    s_sum[t] = incl;
    __syncthreads();
    uint32_t *s = sel + (size_t)f * 4;
    const uint32_t total = s_sum[255];
    uint32_t remaining = PASS == 0 ? (uint32_t)MAXKP : s[1];
    if (PASS == 0 && t == 0) s[2] = total;
    if (PASS == 0 && total <= (uint32_t)MAXKP) {
        if (t == 0) { s[0] = 0; s[1] = 0; s[3] = 1u; }      // accept every candidate (bits >= 1)
        return;
    }
    if (PASS == 1 && s[3] == 1u && s[2] <= (uint32_t)MAXKP) return;
    const uint32_t before = incl - mine;
    if (before < remaining && incl >= remaining) {
        // the K-th strongest lies in this thread's 256 bins
        uint32_t acc = before;
        int b = top;
        for (; b > top - 256; --b) {
            if (acc + hh[b] >= remaining) break;
            acc += hh[b];
        }
        if (PASS == 0) { s[0] = (uint32_t)b; s[1] = remaining - acc; }
        else { s[3] = (s[0] << 16) | (uint32_t)b; }
    }
}

// ---- ordered compaction + sub-pixel refinement ---------------------------------------------------------
constexpr int CMP_CHUNK = 1024;
// One WAVE per chunk of CMP_CHUNK map entries (ballot + popcount: no LDS, no barrier)
__global__ __launch_bounds__(256) void k_ov_count(const float *__restrict__ cand, size_t n4, const uint32_t *__restrict__ sel,
                                                 uint32_t *__restrict__ counts, int nchunks)
{
    const int f = blockIdx.y, ch = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ch >= nchunks) return;
    const uint32_t thr = sel[(size_t)f * 4 + 3];
    const float *c = cand + (size_t)f * n4;
    uint32_t tot = 0;
#pragma unroll 4
    for (int k = 0; k < CMP_CHUNK / 64; ++k) {
        const size_t i = (size_t)ch * CMP_CHUNK + (size_t)k * 64 + lane;
        bool flag = false;
        if (i < n4) {
            const uint32_t b = __float_as_uint(c[i]);
            flag = b != 0 && b >= thr;
        }
        tot += (uint32_t)__popcll(__ballot(flag));
    }
    if (lane == 0) counts[(size_t)f * nchunks + ch] = tot;
}

__global__ __launch_bounds__(256) void k_ov_scan_chunks(uint32_t *__restrict__ counts, int nchunks, int32_t *__restrict__ nkp)
{
    __shared__ uint32_t scratch[8];
    __shared__ uint32_t carry;
    const int f = blockIdx.x;
    uint32_t *c = counts + (size_t)f * nchunks;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nchunks; base += 256) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < nchunks ? c[i] : 0u;
        const uint32_t incl = block256_incl_scan_u32(v, scratch);
        const uint32_t off = carry;
        if (i < nchunks) c[i] = off + incl - v;       // exclusive offset
        __syncthreads();
        if (threadIdx.x == 255) carry = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) nkp[f] = (int32_t)min(carry, (uint32_t)MAXKP);
}

__global__ __launch_bounds__(256) void k_ov_compact(const float *__restrict__ cand, const float *__restrict__ Ldet, int h, int w,
                                                   const uint32_t *__restrict__ sel, const uint32_t *__restrict__ offsets,
                                                   int nchunks, Keypoint *__restrict__ kps, int F)
{
    const int f = blockIdx.y, ch = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ch >= nchunks) return;
    const size_t n = (size_t)h * w, n4 = n * NLEV;
    const uint32_t thr = sel[(size_t)f * 4 + 3];
    const float *c = cand + (size_t)f * n4;
    uint32_t base = offsets[(size_t)f * nchunks + ch];          // wave-uniform: entries selected before this chunk
    for (int k = 0; k < CMP_CHUNK / 64 && base < (uint32_t)MAXKP; ++k) {
        const size_t i = (size_t)ch * CMP_CHUNK + (size_t)k * 64 + lane;
        bool flag = false;
        float v = 0.0f;
        if (i < n4) {
            v = c[i];
            const uint32_t b = __float_as_uint(v);
            flag = b != 0 && b >= thr;
        }
        const unsigned long long mask = __ballot(flag);
        const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        base += (uint32_t)__popcll(mask);
        if (flag && pos < (uint32_t)MAXKP) {
            const int lv = (int)(i / n);
            const size_t r = i - (size_t)lv * n;
            const int y = (int)(r / w), x = (int)(r - (size_t)y * w);
            const float *D = Ldet + ((size_t)lv * F + f) * n;
            const float vxp = D[(size_t)y * w + x + 1], vxm = D[(size_t)y * w + x - 1];
            const float vyp = D[(size_t)(y + 1) * w + x], vym = D[(size_t)(y - 1) * w + x];
            const float Dx = 0.5f * (vxp - vxm), Dy = 0.5f * (vyp - vym);
            const float Dxx = (vxp + vxm) - 2.0f * v, Dyy = (vyp + vym) - 2.0f * v;
            const float Dxy = 0.25f * (D[(size_t)(y + 1) * w + x + 1] + D[(size_t)(y - 1) * w + x - 1]) -
                              0.25f * (D[(size_t)(y + 1) * w + x - 1] + D[(size_t)(y - 1) * w + x + 1]);
            const float det = Dxx * Dyy - Dxy * Dxy;
            const float ox = -(Dyy * Dx - Dxy * Dy) / det, oy = -(Dxx * Dy - Dxy * Dx) / det;
            Keypoint kp;
            kp.x = (float)x + ox; kp.y = (float)y + oy; kp.response = v;
            kp.level = lv; kp.xi = x; kp.yi = y; kp.co = 1.0f; kp.si = 0.0f;
            kps[(size_t)f * MAXKP + pos] = kp;
        }
    }
}

// ---- oriented M-LDB: one wave per keypoint ----------------------------------------------------------------
// The reference's detector is oriented SURF (videostrip.cpp:206-208, upright = false).  Orientation: AKAZE's sliding
// pi/3 sector over the Gaussian-weighted scale-s derivatives of a radius-6 disc, stated without angles (sector
// membership = two cross products against tabulated boundary unit vectors; the result is the unit vector (co, si)):
// +, *, /, sqrt only, in the oracle's order, so it is bit-exact.  Lane s < 42 owns sector s and walks the 109 samples
// (LDS broadcast reads) sequentially = the oracle's summation order.
/* ORIENT-TABLES-BEGIN (generated by tools/gen_orient_tables.py, identical text in oracle/uwip_oracle_overlap.c) */
static __device__ const float D_GAUSS25[7][7] = {
    {1.0f, 0.923116326f, 0.726149023f, 0.486752242f, 0.27803731f, 0.135335281f, 0.0561347641f},
    {0.923116326f, 0.852143764f, 0.670320034f, 0.449328959f, 0.256660789f, 0.12493021f, 0.0518189184f},
    {0.726149023f, 0.670320034f, 0.52729243f, 0.353454679f, 0.201896518f, 0.0982735828f, 0.0407622047f},
    {0.486752242f, 0.449328959f, 0.353454679f, 0.236927763f, 0.135335281f, 0.0658747554f, 0.0273237228f},
    {0.27803731f, 0.256660789f, 0.201896518f, 0.135335281f, 0.0773047432f, 0.0376282558f, 0.0156075582f},
    {0.135335281f, 0.12493021f, 0.0982735828f, 0.0658747554f, 0.0376282558f, 0.0183156393f, 0.00759701384f},
    {0.0561347641f, 0.0518189184f, 0.0407622047f, 0.0273237228f, 0.0156075582f, 0.00759701384f, 0.00315111154f},
};
/* sector k: [a_k, a_k + pi/3), a_k = 0.15 k; {cos a_k, sin a_k, cos(a_k + pi/3), sin(a_k + pi/3)} */
static __device__ const float D_SECTOR[42][4] = {
    {1.0f, 0.0f, 0.5f, 0.866025388f},
    {0.988771081f, 0.149438128f, 0.36496833f, 0.931019962f},
    {0.955336511f, 0.295520216f, 0.221740231f, 0.975105762f},
    {0.90044713f, 0.434965521f, 0.0735323504f, 0.997292817f},
    {0.825335622f, 0.564642489f, -0.0763269216f, 0.997082829f},
    {0.731688857f, 0.681638777f, -0.224472046f, 0.97448051f},
    {0.621609986f, 0.783326924f, -0.367576033f, 0.929993451f},
    {0.497571051f, 0.867423236f, -0.502425015f, 0.864620805f},
    {0.362357765f, 0.932039082f, -0.625990629f, 0.779830575f},
    {0.219006687f, 0.975723386f, -0.735497892f, 0.67752701f},
    {0.070737198f, 0.997494996f, -0.828487396f, 0.560007691f},
    {-0.0791208893f, 0.996865034f, -0.902870893f, 0.429911822f},
    {-0.227202088f, 0.973847628f, -0.956977844f, 0.290161043f},
    {-0.370180845f, 0.928959727f, -0.989593148f, 0.143893853f},
    {-0.504846096f, 0.863209367f, -0.999984264f, -0.00560486829f},
    {-0.628173649f, 0.778073192f, -0.98791796f, -0.154977724f},
    {-0.737393737f, 0.6754632f, -0.953665137f, -0.300870091f},
    {-0.830053508f, 0.557683706f, -0.897995055f, -0.4400056f},
    {-0.904072165f, 0.427379876f, -0.822157919f, -0.569259524f},
    {-0.957787216f, 0.287478f, -0.727856874f, -0.685729086f},
    {-0.989992499f, 0.141120002f, -0.617209733f, -0.786798656f},
    {-0.999964654f, -0.00840724725f, -0.492701441f, -0.870198429f},
    {-0.987479746f, -0.157745689f, -0.357128114f, -0.934055388f},
    {-0.952818215f, -0.303541511f, -0.213534445f, -0.976935506f},
    {-0.896758437f, -0.44252044f, -0.0651452616f, -0.99787581f},
    {-0.820559382f, -0.571561337f, 0.0847069398f, -0.9964059f},
    {-0.7259323f, -0.687766135f, 0.232656807f, -0.972558916f},
    {-0.615002394f, -0.788525283f, 0.375381708f, -0.926870286f},
    {-0.49026081f, -0.871575773f, 0.509676337f, -0.860366225f},
    {-0.354509056f, -0.935052574f, 0.632524729f, -0.774540126f},
    {-0.210795805f, -0.977530122f, 0.741168022f, -0.671319604f},
    {-0.0623485148f, -0.998054445f, 0.833166242f, -0.553022623f},
    {0.0874989852f, -0.99616462f, 0.906453371f, -0.422305971f},
    {0.235381439f, -0.971903086f, 0.959383488f, -0.282105237f},
    {0.377977729f, -0.925814688f, 0.990767896f, -0.135569021f},
    {0.512085497f, -0.858934522f, 0.999901831f, 0.0140117854f},
    {0.634692848f, -0.772764504f, 0.986580133f, 0.163277909f},
    {0.743046463f, -0.669239879f, 0.951101959f, 0.30887717f},
    {0.834712803f, -0.550685525f, 0.894264042f, 0.447539717f},
    {0.907633305f, -0.419764012f, 0.817342937f, 0.57615149f},
    {0.960170269f, -0.279415488f, 0.722066045f, 0.691824138f},
    {0.991143942f, -0.132791907f, 0.610573113f, 0.791959882f},
};
/* the 109 lattice points of the radius-6 disc in the oracle's loop order (i = x offset outer, j = y offset inner) */
static __device__ const signed char D_DISC[109][2] = {
    {-5, -3}, {-5, -2}, {-5, -1}, {-5, 0}, {-5, 1}, {-5, 2}, {-5, 3}, {-4, -4}, {-4, -3}, {-4, -2}, {-4, -1}, {-4, 0},
    {-4, 1}, {-4, 2}, {-4, 3}, {-4, 4}, {-3, -5}, {-3, -4}, {-3, -3}, {-3, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-3, 2},
    {-3, 3}, {-3, 4}, {-3, 5}, {-2, -5}, {-2, -4}, {-2, -3}, {-2, -2}, {-2, -1}, {-2, 0}, {-2, 1}, {-2, 2}, {-2, 3},
    {-2, 4}, {-2, 5}, {-1, -5}, {-1, -4}, {-1, -3}, {-1, -2}, {-1, -1}, {-1, 0}, {-1, 1}, {-1, 2}, {-1, 3}, {-1, 4},
    {-1, 5}, {0, -5}, {0, -4}, {0, -3}, {0, -2}, {0, -1}, {0, 0}, {0, 1}, {0, 2}, {0, 3}, {0, 4}, {0, 5},
    {1, -5}, {1, -4}, {1, -3}, {1, -2}, {1, -1}, {1, 0}, {1, 1}, {1, 2}, {1, 3}, {1, 4}, {1, 5}, {2, -5},
    {2, -4}, {2, -3}, {2, -2}, {2, -1}, {2, 0}, {2, 1}, {2, 2}, {2, 3}, {2, 4}, {2, 5}, {3, -5}, {3, -4},
    {3, -3}, {3, -2}, {3, -1}, {3, 0}, {3, 1}, {3, 2}, {3, 3}, {3, 4}, {3, 5}, {4, -4}, {4, -3}, {4, -2},
    {4, -1}, {4, 0}, {4, 1}, {4, 2}, {4, 3}, {4, 4}, {5, -3}, {5, -2}, {5, -1}, {5, 0}, {5, 1}, {5, 2},
    {5, 3},
};
/* ORIENT-TABLES-END */
// bit b (0..485) of the descriptor compares cells ia > ib of channel c: the oracle's enumeration (grid z, channel, pair a < bb
// in lexicographic order) unrolled into a table at compile time -- entry = ia | ib << 5 | c << 10, cells numbered 0..28
struct PairTab { unsigned short v[486]; };
constexpr PairTab make_pair_tab()
{
    PairTab t{};
    int b = 0;
    const int ncells[3] = {4, 9, 16}, bases[3] = {0, 4, 13};
    for (int z = 0; z < 3; ++z)
        for (int c = 0; c < 3; ++c)
            for (int a = 0; a < ncells[z]; ++a)
                for (int bb = a + 1; bb < ncells[z]; ++bb)
                    t.v[b++] = (unsigned short)((bases[z] + a) | ((bases[z] + bb) << 5) | (c << 10));
    return t;
}
static __device__ const PairTab D_PAIRS = make_pair_tab();
// The 29 cells (4 + 9 + 16) x 3 channels are 87 sequential sums (fixed order = the oracle's): a lane owns one or two of them
// (the 100-sample sums of the 2 x 2 grid on lanes 0..11 set the length of the phase); then the 486 comparisons are spread
// over the 64 lanes.
__global__ __launch_bounds__(64) void k_ov_describe(const float *__restrict__ Lt, const float2 *__restrict__ Lxy,
                                                   int h, int w, Keypoint *__restrict__ kps, const int32_t *__restrict__ nkp,
                                                   uint8_t *__restrict__ desc, int8_t *__restrict__ bits, uint32_t *__restrict__ nib,
                                                   int32_t *__restrict__ pop, int F, int upright)
{
    __shared__ __attribute__((aligned(16))) float2 s_v[112];     // (vx, vy) of the 109 disc samples; 109..111 stay zero
    __shared__ float s_val[29][3];
    __shared__ uint32_t s_words[16];
    __shared__ float s_patch[3][21][22];   // [plane][x offset][y offset], padded
    // Workgroups are dealt round-robin over the 8 XCDs: give every XCD a contiguous range of (frame, keypoint) so that
    // the raster-ordered keypoints of a frame gather through ONE L2 instead of fetching their patches into all eight.
    const unsigned total = (unsigned)MAXKP * F, per = (total + 7u) / 8u;
    const unsigned m = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (m >= total) return;
    const int f = m / MAXKP, q = m % MAXKP, lane = threadIdx.x;
    const int n = nkp[f];
    uint8_t *d = desc + ((size_t)f * MAXKP + q) * DESC_BYTES;
    int8_t *bq = bits + ((size_t)f * MAXKP + q) * DESC_K;
    uint32_t *nq = nib + ((size_t)f * MAXKP + q) * DESC_NIBW;
    if (q >= n) {
        // unused slots: all-zero descriptor (keeps the MFMA operand defined)
        nq[lane] = 0u;
        if (lane < 16) reinterpret_cast<uint32_t *>(d)[lane] = 0;
        for (int i = lane; i < DESC_K / 4; i += 64) reinterpret_cast<uint32_t *>(bq)[i] = 0;
        if (lane == 0) pop[(size_t)f * MAXKP + q] = 0;
        return;
    }
    const Keypoint kp = kps[(size_t)f * MAXKP + q];
    const size_t npx = (size_t)h * w;
    const float *T = Lt + ((size_t)kp.level * F + f) * npx;
    const float2 *XY = Lxy + ((size_t)kp.level * F + f) * npx;
    const float sc = (float)D_SSIZE[kp.level];
    if (lane < 16) s_words[lane] = 0;
    float co = 1.0f, si = 0.0f;
    if (!upright) {
        for (int t = lane; t < 109; t += 64) {
            const int i = D_DISC[t][0], j = D_DISC[t][1];
            const int x1 = min(max((int)floorf(kp.x + (float)i * sc + 0.5f), 0), w - 1);
            const int y1 = min(max((int)floorf(kp.y + (float)j * sc + 0.5f), 0), h - 1);
            const float g = D_GAUSS25[abs(i)][abs(j)];
            const size_t o = (size_t)y1 * w + x1;
            const float2 d2 = XY[o];
            s_v[t] = make_float2(g * d2.x, g * d2.y);
        }
        if (lane < 3) s_v[109 + lane] = make_float2(0.0f, 0.0f);      // a zero vector is in no sector (c2 < 0 fails)
        __syncthreads();
        float sx = 0.0f, sy = 0.0f;
        if (lane < 42) {
            const float d0 = D_SECTOR[lane][0], d1 = D_SECTOR[lane][1], d2 = D_SECTOR[lane][2], d3 = D_SECTOR[lane][3];
            // two samples per 16-byte LDS read (all lanes read the same address: a broadcast), same order of additions
            for (int q2 = 0; q2 < 110; q2 += 2) {
                const float4 p = *reinterpret_cast<const float4 *>(&s_v[q2]);
                const float ca1 = d0 * p.y - d1 * p.x, ca2 = d2 * p.y - d3 * p.x;
                if (ca1 >= 0.0f && ca2 < 0.0f) { sx = sx + p.x; sy = sy + p.y; }
                const float cb1 = d0 * p.w - d1 * p.z, cb2 = d2 * p.w - d3 * p.z;
                if (cb1 >= 0.0f && cb2 < 0.0f) { sx = sx + p.z; sy = sy + p.w; }
            }
        }
        const float m0 = sx * sx + sy * sy;
        float m = (lane < 42 && m0 > 0.0f) ? m0 : 0.0f;     // NaN and empty sectors never win (the oracle's `m > best`)
        int idx = lane;
        for (int off = 32; off >= 1; off >>= 1) {             // arg-max, the lowest sector among equal maxima
            const float om = __shfl_xor(m, off);
            const int oi = __shfl_xor(idx, off);
            if (om > m || (om == m && oi < idx)) { m = om; idx = oi; }
        }
        const float bx = __shfl(sx, idx), by = __shfl(sy, idx);
        if (m > 0.0f) {
            const float nrm = sqrtf(m);
            co = bx / nrm; si = by / nrm;
        }
        if (lane == 0) { kps[(size_t)f * MAXKP + q].co = co; kps[(size_t)f * MAXKP + q].si = si; }
    }
    // The three grids (2x2 cells of 10, 3x3 of 7, 4x4 of 5 samples a side) draw from one 21 x 21 lattice of sample
    // positions (offsets -10..10 times the scale): all 64 lanes fetch it once into LDS, x fastest so that a wave's
    // loads run along image rows, and the cells then sum from LDS in the oracle's order.
    for (int idx = lane; idx < 21 * 21; idx += 64) {
        const int l = idx / 21, kk = idx - l * 21;       // l: y offset index, kk: x offset index
        // lattice and derivative pair rotated into the keypoint's frame; (co, si) = (1, 0) gives the unrotated values exactly
        const float u = (float)(kk - 10) * sc, v = (float)(l - 10) * sc;
        const float sy = kp.y + (u * si + v * co), sx = kp.x + (u * co - v * si);
        const int y1 = min(max((int)floorf(sy + 0.5f), 0), h - 1);
        const int x1 = min(max((int)floorf(sx + 0.5f), 0), w - 1);
        const size_t o = (size_t)y1 * w + x1;
        const float2 d2 = XY[o];
        const float rx = d2.x, ry = d2.y;
        s_patch[0][kk][l] = T[o]; s_patch[1][kk][l] = rx * co + ry * si; s_patch[2][kk][l] = ry * co - rx * si;
    }
    __syncthreads();
    {
        // tasks 0..86 = cell * 3 + channel; lanes 39..61 take a second one (64..86) after their first
        auto run_task = [&](int task) {
            const int cell = task / 3, ch = task - cell * 3;
            int z, ci;
            if (cell < 4) { z = 0; ci = cell; } else if (cell < 13) { z = 1; ci = cell - 4; } else { z = 2; ci = cell - 13; }
            const int st = z == 0 ? 10 : (z == 1 ? 7 : 5), nc = z + 2;
            const int i0 = (ci / nc) * st, j0 = (ci % nc) * st;       // i (x) major, j (y) minor; offsets already + 10
            const float *pp = &s_patch[ch][i0][j0];
            float acc = 0.0f;
            for (int kk = 0; kk < st; ++kk, pp += 22)
                for (int l = 0; l < st; ++l) acc = acc + pp[l];
            s_val[cell][ch] = acc / (float)(st * st);
        };
        run_task(lane);
        if (lane >= 39 && lane < 62) run_task(lane + 25);
    }
    __syncthreads();
    for (int b = lane; b < 486; b += 64) {
        const uint32_t e = D_PAIRS.v[b];
        const int bit = s_val[e & 31u][e >> 10] > s_val[(e >> 5) & 31u][e >> 10] ? 1 : 0;
        bq[b] = (int8_t)bit;
        if (bit) atomicOr(&s_words[b >> 5], 1u << (b & 31));
    }
    for (int b = 486 + lane; b < DESC_K; b += 64) bq[b] = 0;
    __syncthreads();
    if (lane < 16) reinterpret_cast<uint32_t *>(d)[lane] = s_words[lane];
    nq[lane] = desc_byte_to_nibbles((s_words[lane >> 2] >> (8 * (lane & 3))) & 255u);      // one coalesced 256-byte store
    if (lane == 0) {
        int pc = 0;
        for (int i = 0; i < 16; ++i) pc += __popc(s_words[i]);
        pop[(size_t)f * MAXKP + q] = pc;
    }
}

// ---- brute-force Hamming kNN(2): dense q x t dot products on i8 MFMA -------------------------------------
// popcount(a xor b) = |a| + |b| - 2 a.b with a, b in {0,1}^512 held as bytes.
// Block = 4 waves = 64 queries; train descriptors staged through LDS 64 at a time (row stride 528 B:
// the 16 lanes of a ds_read_b128 group then fall on 16 different 16-byte slots).
typedef int v4i __attribute__((ext_vector_type(4)));
// Row stride of a staged train tile: 512 + 32 bytes.  A ds_read_b128 is served in four groups of 16 lanes --
// {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- i.e. rows of two neighbouring
// 16-byte columns kb, kb + 1 in one group; with a stride of 8 dwords mod 64 the 16-byte slot of (row, kb) is (2 row + kb) mod 16:
// the even kb of a group takes the even slots, the odd one the odd slots -- conflict-free.  (Rounds 1-3 used 512 + 16:
// slot (row + kb) mod 16, where row 11 of column kb + 1 meets row 12 of column kb -- SQ_LDS_BANK_CONFLICT was 42 % of
// SQ_LDS_IDX_ACTIVE, profiles/r04_matcher_counters.txt before the change.)
constexpr int MT_ROW = DESC_K + 32;

// Top-2 of (distance, index) pairs under the order "smaller distance, then lower index" (BFMatcher::knnMatch k = 2 with
// ties to the lower train index) on PACKED keys: key = distance << 11 | index (distance <= 512, index < 2048), so
// the lexicographic order is the integer order and one candidate costs a max and two mins instead of two
// compares and four selects per slot.  b0 <= b1 always; empty slots hold MT_EMPTY.
constexpr uint32_t MT_EMPTY = 0xffffffffu;
__device__ __forceinline__ void top2_push(uint32_t &b0, uint32_t &b1, uint32_t k)
{
    b1 = min(b1, max(b0, k));     // the second smallest of three (b0 <= b1)
    b0 = min(b0, k);
}

// QT query tiles of 16 per wave: every train fragment read from LDS feeds QT MFMAs
template <int QT, int NW>
__global__ __launch_bounds__(64 * NW) void k_ov_match(const int8_t *__restrict__ qbits, const int32_t *__restrict__ qpop,
                                                 const int32_t *__restrict__ qn, const int8_t *__restrict__ tbits,
                                                 const int32_t *__restrict__ tpop, const int32_t *__restrict__ tn,
                                                 const int32_t *__restrict__ pair_q, const int32_t *__restrict__ pair_t,
                                                 int32_t *__restrict__ out_idx /*[P][MAXKP][2]*/, int32_t *__restrict__ out_dist)
{
    extern __shared__ __attribute__((aligned(16))) int8_t s_t[];      // [64][MT_ROW]
    const int p = blockIdx.y;
    const int fq = pair_q[p], ft = pair_t[p];
    const int nq = qn[fq], nt = tn[ft];
    const int q0 = blockIdx.x * (16 * NW * QT);
    if (q0 >= nq) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 15, kb = lane >> 4;
    const int8_t *Q = qbits + (size_t)fq * MAXKP * DESC_K;
    const int8_t *T = tbits + (size_t)ft * MAXKP * DESC_K;
    // A fragments: query (q0 + (wave * QT + u) * 16 + row), bytes [64*ks + 16*kb, +16)
    v4i a[QT][8];
    int cq[QT][4];                   // popcount of this lane's 4 accumulator rows
    uint32_t b0[QT][4], b1[QT][4];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const int qrow = q0 + (wave * QT + u) * 16 + row;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            a[u][ks] = *reinterpret_cast<const v4i *>(Q + (size_t)qrow * DESC_K + ks * 64 + kb * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cq[u][r] = qpop[(size_t)fq * MAXKP + q0 + (wave * QT + u) * 16 + kb * 4 + r];
            b0[u][r] = b1[u][r] = MT_EMPTY;
        }
    }
    // 64 train descriptors (32 KB) per tile: 2048 16-byte pieces, 8 per thread.  The NEXT tile's pieces are requested
    // into registers before this tile's MFMAs, so the global-memory latency (it was exposed twice per tile between the
    // barriers and held the kernel at 18 % of the matrix peak) hides behind them.
    constexpr int NP = 2048 / (64 * NW);      // 16-byte pieces per thread
    v4i stage[NP];
    auto fetch = [&](int t0) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = threadIdx.x + 64 * NW * j, tr = i >> 5, piece = i & 31;
            stage[j] = *reinterpret_cast<const v4i *>(T + (size_t)(t0 + tr) * DESC_K + piece * 16);
        }
    };
    // two LDS tiles: while tile t is multiplied, the registers holding tile t+1 (requested a whole tile earlier) are
    // parked in the other buffer and tile t+2 is requested -- ONE barrier per tile, no wait on global memory in the loop
    auto park = [&](int8_t *buf) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = threadIdx.x + 64 * NW * j, tr = i >> 5, piece = i & 31;
            *reinterpret_cast<v4i *>(buf + (size_t)tr * MT_ROW + piece * 16) = stage[j];
        }
    };
    int8_t *const bufA = s_t, *const bufB = s_t + (size_t)64 * MT_ROW;
    if (nt > 0) {
        fetch(0);
        park(bufA);
        if (64 < nt) fetch(64);                  // rows up to MAXKP exist for every slot (zero descriptors past the count)
    }
    __syncthreads();
    for (int t0 = 0, it = 0; t0 < nt; t0 += 64, ++it) {
        const int8_t *cur = (it & 1) ? bufB : bufA;
        // (|b| + 512) << 11 | t for this lane's four train columns of the tile (a dead column, t >= nt, keeps every key
        // above any live key); requested before the MFMAs so that the popcount loads hide behind them.  The query's own
        // popcount is the same for all candidates of a row, so it is added when the result is written, not per candidate.
        // A dead column's key must not depend on what its descriptor row holds (k_ov_describe and uwip_features_upload
        // zero the rows past the count, but nothing else guarantees it): its dot product is multiplied by 0 instead of
        // -4096, at no cost in the epilogue (the multiplier of the v_mad_i32_i24 is a register either way).
        uint32_t tb[4];
        int mf[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const int t = t0 + tt * 16 + row;
            tb[tt] = t < nt ? ((((uint32_t)tpop[(size_t)ft * MAXKP + t] + 512u) << 11) | (uint32_t)t) : 0x7ff00000u;
            mf[tt] = t < nt ? -4096 : 0;
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            v4i acc[QT];
#pragma unroll
            for (int u = 0; u < QT; ++u) acc[u] = v4i{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const v4i b = *reinterpret_cast<const v4i *>(cur + (size_t)(tt * 16 + row) * MT_ROW + ks * 64 + kb * 16);
#pragma unroll
                for (int u = 0; u < QT; ++u) acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[u][ks], b, acc[u], 0, 0, 0);
            }
            // C/D: col = lane & 15 (train t), row = (lane >> 4) * 4 + reg (query).  popcount(a xor b) = |a| + |b| - 2 a.b:
            // key = tb - (a.b << 12), one v_mad_i32_i24
#pragma unroll
            for (int u = 0; u < QT; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    top2_push(b0[u][r], b1[u][r], (uint32_t)(__mul24(acc[u][r], mf[tt]) + (int)tb[tt]));
        }
        if (t0 + 64 < nt) {
            park((it & 1) ? bufA : bufB);        // tile t+1: its buffer was last read in iteration t-1, before that barrier
            if (t0 + 128 < nt) fetch(t0 + 128);
        }
        __syncthreads();
    }
    // merge the 16 lanes (lane & 15) that hold different columns of the same query rows
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const uint32_t o0 = (uint32_t)__shfl_xor((int)b0[u][r], d, 64), o1 = (uint32_t)__shfl_xor((int)b1[u][r], d, 64);
                top2_push(b0[u][r], b1[u][r], o0);
                top2_push(b0[u][r], b1[u][r], o1);
            }
            const int q = q0 + (wave * QT + u) * 16 + kb * 4 + r;
            if (row == 0 && q < nq) {
                const size_t o = ((size_t)p * MAXKP + q) * 2;
                const bool h0 = b0[u][r] < 0x7ff00000u, h1 = b1[u][r] < 0x7ff00000u;
                out_idx[o] = h0 ? (int)(b0[u][r] & 2047u) : -1; out_idx[o + 1] = h1 ? (int)(b1[u][r] & 2047u) : -1;
                out_dist[o] = h0 ? (int)(b0[u][r] >> 11) - 512 + cq[u][r] : -1;
                out_dist[o + 1] = h1 ? (int)(b1[u][r] >> 11) - 512 + cq[u][r] : -1;
            }
        }
}

// The same matcher, software-pipelined (round 4).  What the ISA of the form above shows (llvm-objdump): (i) the four
// `tpop` loads of a tile sit behind four lane-mask branches, each followed by `s_waitcnt vmcnt(0)` -- four serialised
// global-memory round trips per 64 MFMAs; (ii) the whole top-2 epilogue of a tile (112 vector instructions) runs AFTER its
// 64 MFMAs, right before the barrier, so all eight waves of the block alternate between a matrix phase and a vector
// phase in step; (iii) the B fragments are read two at a time and waited for at once.  Here:
//   * the train keys ((|b| + 512) << 11 | t, or the dead-column key) travel with the tile: fetched by 64 threads a
//     tile ahead, parked in LDS beside the descriptors, read back with one ds_read_b32 per 16 columns;
//   * the B fragments of column group tt + 1 are requested before the MFMAs of group tt (two register sets);
//   * the epilogue of group tt - 1 (the last group's: of the previous tile) is issued between the MFMAs of group tt
//     -- an MFMA holds the SIMD's vector issue for 8 of its 16 cycles, two vector instructions fit in the rest
//     (MI355X_MICROARCH.md, "vector-instruction ISSUE cost") -- pinned with sched_group_barrier.
// Same results bit for bit (packed-key top-2 is order-independent).
template <int QT, int NW, int TG>      // TG: column groups of 16 per staged train tile (4 or 8): one barrier per TG * 16 columns
__global__ __launch_bounds__(64 * NW) void k_ov_match_sp(const int8_t *__restrict__ qbits, const int32_t *__restrict__ qpop,
                                                    const int32_t *__restrict__ qn, const int8_t *__restrict__ tbits,
                                                    const int32_t *__restrict__ tpop, const int32_t *__restrict__ tn,
                                                    const int32_t *__restrict__ pair_q, const int32_t *__restrict__ pair_t,
                                                    int32_t *__restrict__ out_idx /*[P][MAXKP][2]*/, int32_t *__restrict__ out_dist)
{
    extern __shared__ __attribute__((aligned(16))) int8_t s_t[];      // 2 x [64][MT_ROW] descriptors, then 2 x [64] keys
    const int p = blockIdx.y;
    const int fq = pair_q[p], ft = pair_t[p];
    const int nq = qn[fq], nt = tn[ft];
    const int q0 = blockIdx.x * (16 * NW * QT);
    if (q0 >= nq) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 15, kb = lane >> 4;
    const int8_t *Q = qbits + (size_t)fq * MAXKP * DESC_K;
    const int8_t *T = tbits + (size_t)ft * MAXKP * DESC_K;
    const int32_t *TP = tpop + (size_t)ft * MAXKP;
    v4i a[QT][8];
    int cq[QT][4];
    uint32_t b0[QT][4], b1[QT][4];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const int qrow = q0 + (wave * QT + u) * 16 + row;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            a[u][ks] = *reinterpret_cast<const v4i *>(Q + (size_t)qrow * DESC_K + ks * 64 + kb * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cq[u][r] = qpop[(size_t)fq * MAXKP + q0 + (wave * QT + u) * 16 + kb * 4 + r];
            b0[u][r] = b1[u][r] = MT_EMPTY;
        }
    }
    constexpr int TC = TG * 16;               // train columns per tile
    constexpr int NP = TC * 32 / (64 * NW);   // 16-byte pieces per thread
    constexpr int NK = (TC + 63) / 64;        // keys per lane
    constexpr uint32_t DEAD = 0x7ff00000u;
    v4i stage[NP];
    uint32_t stage_key[NK];
    auto fetch = [&](int t0) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = threadIdx.x + 64 * NW * j, tr = i >> 5, piece = i & 31;
            stage[j] = *reinterpret_cast<const v4i *>(T + (size_t)(t0 + tr) * DESC_K + piece * 16);
        }
        // the tile's 64 train keys (rows up to MAXKP exist; dead ones get the dead key): every wave loads and parks the
        // same 64 values -- no branch, so the loop body stays ONE basic block and the scheduler may interleave it
#pragma unroll
        for (int j = 0; j < NK; ++j) {
        const int t = t0 + lane + 64 * j;
        const uint32_t pc = (uint32_t)TP[min(t, MAXKP - 1)];
        // live keys are < 0x200800; a dead column ORs the dead key in (any key >= DEAD is dead).  Written as an OR, not
        // as a select between the two keys: a select whose one arm comes from a load is turned into a branch around the
        // load, with a vmcnt(0) wait inside it
        stage_key[j] = (((pc + 512u) << 11) | (uint32_t)t) | (t < nt ? 0u : DEAD);
        }
    };
    int8_t *const bufA = s_t, *const bufB = s_t + (size_t)TC * MT_ROW;
    uint32_t *const keyA = reinterpret_cast<uint32_t *>(s_t + (size_t)2 * TC * MT_ROW), *const keyB = keyA + TC;
    auto park = [&](int8_t *buf, uint32_t *kbuf) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = threadIdx.x + 64 * NW * j, tr = i >> 5, piece = i & 31;
            *reinterpret_cast<v4i *>(buf + (size_t)tr * MT_ROW + piece * 16) = stage[j];
        }
#pragma unroll
        for (int j = 0; j < NK; ++j) if (lane + 64 * j < TC) kbuf[lane + 64 * j] = stage_key[j];
    };
    if (nt > 0) {
        fetch(0);
        park(bufA, keyA);
        fetch(TC);                            // MAXKP >= 2 TC: the rows exist; keys past nt are dead
    }
    __syncthreads();
    // Two accumulator sets: group tt multiplies into acc[tt & 1] while the results of group tt - 1 in acc[(tt + 1) & 1] (for
    // tt = 0: the previous tile's last group) go through the top-2 insertion.  An even number of groups per tile, so the parity carries
    // over the tile loop without a register copy.
    v4i acc[2][QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) acc[1][u] = v4i{0, 0, 0, 0};
    uint32_t tb_last = DEAD;                  // key of the pending group of the previous tile (none yet: dead)
    auto epilogue_one = [&](const v4i (&ac)[QT], uint32_t tbk, int idx) {
        const int u = idx >> 2, r = idx & 3;
        const int mf = tbk >= DEAD ? 0 : -4096;
        top2_push(b0[u][r], b1[u][r], (uint32_t)(__mul24(ac[u][r], mf) + (int)tbk));
    };
    static_assert(TG % 2 == 0 && TC * 32 % (64 * NW) == 0, "tile shape");
    for (int t0 = 0, it = 0; t0 < nt; t0 += TC, ++it) {
        const int8_t *cur = (it & 1) ? bufB : bufA;
        const uint32_t *kcur = (it & 1) ? keyB : keyA;
        uint32_t tb[TG];
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) tb[tt] = kcur[tt * 16 + row];
        v4i bf[2][8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            bf[0][ks] = *reinterpret_cast<const v4i *>(cur + (size_t)row * MT_ROW + ks * 64 + kb * 16);
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) {
            if (tt < TG - 1) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    bf[(tt + 1) & 1][ks] = *reinterpret_cast<const v4i *>(cur + (size_t)((tt + 1) * 16 + row) * MT_ROW + ks * 64 + kb * 16);
            }
            const uint32_t tbk = tt == 0 ? tb_last : tb[(tt + TG - 1) % TG];
#pragma unroll
            for (int u = 0; u < QT; ++u) acc[tt & 1][u] = v4i{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
                for (int u = 0; u < QT; ++u)
                    acc[tt & 1][u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[u][ks], bf[tt & 1][ks], acc[tt & 1][u], 0, 0, 0);
                // QT * 4 pending results over the 8 steps
                if (QT * 4 >= 8) {
#pragma unroll
                    for (int e = 0; e < QT * 4 / 8; ++e) epilogue_one(acc[(tt + 1) & 1], tbk, ks * (QT * 4 / 8) + e);
                } else if ((ks & 1) == 0) {
                    epilogue_one(acc[(tt + 1) & 1], tbk, ks >> 1);
                }
            }
        }
        tb_last = tb[TG - 1];
        // unconditional (clamped) staging of the next tiles: a branch here would split the body and let the compiler sink
        // the epilogue behind it; the last two tiles park / fetch rows nobody reads
        park((it & 1) ? bufA : bufB, (it & 1) ? keyA : keyB);
        fetch(min(t0 + 2 * TC, MAXKP - TC));
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < QT * 4; ++i) epilogue_one(acc[1], tb_last, i);
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const uint32_t o0 = (uint32_t)__shfl_xor((int)b0[u][r], d, 64), o1 = (uint32_t)__shfl_xor((int)b1[u][r], d, 64);
                top2_push(b0[u][r], b1[u][r], o0);
                top2_push(b0[u][r], b1[u][r], o1);
            }
            const int q = q0 + (wave * QT + u) * 16 + kb * 4 + r;
            if (row == 0 && q < nq) {
                const size_t o = ((size_t)p * MAXKP + q) * 2;
                const bool h0 = b0[u][r] < DEAD, h1 = b1[u][r] < DEAD;
                out_idx[o] = h0 ? (int)(b0[u][r] & 2047u) : -1; out_idx[o + 1] = h1 ? (int)(b1[u][r] & 2047u) : -1;
                out_dist[o] = h0 ? (int)(b0[u][r] >> 11) - 512 + cq[u][r] : -1;
                out_dist[o + 1] = h1 ? (int)(b1[u][r] >> 11) - 512 + cq[u][r] : -1;
            }
        }
}

// ---- the same matcher on the FP4 form of the f8f6f4 MFMA (round 5) ---------------------------------------------------------
// v_mfma_scale_f32_16x16x128_f8f6f4 with both operands E2M1: a descriptor bit travels as a NIBBLE (0x0 = 0.0, 0x2 = 1.0), 256
// bytes per 512-bit descriptor instead of the 512 of the i8 form -- half the global and LDS bytes per MAC -- and one
// instruction covers K = 128: four MFMAs per 16 x 16 x 512 tile instead of eight, at the cycles of the i8 instruction (twice
// its MAC rate; MI355X_MICROARCH.md, Matrix cores).  Block scales are E8M0 bytes of 127 = 2^0.  The products are 0 or 1 and a
// sum is at most 512: exact in the float32 accumulator.  The packed key (distance, train index) is formed and compared as
// FLOAT -- (|t| + 512 - 2 a.b) * 2048 + t < 2^22 is exact in float32, one v_fma_f32 from the accumulator, v_min / v_max_f32
// for the top-2 insertion: the same four vector instructions per result as the integer form -- and converted once at the end.
// Both operands' lanes read their 32 nibbles of a K = 128 step from the same byte offsets of a descriptor, so whatever k
// order the hardware assigns inside a lane, the two sides agree: the sum is the dot product.
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int F4_DESC = DESC_NIBW * 4;                 // 256 bytes per descriptor
constexpr int F4_ROW = F4_DESC + 32;                   // LDS row stride (the i8 form's argument: 2 row + kb mod 16 slots)
constexpr float F4_DEAD = 8388608.0f;                  // any key >= 2^23 is dead (live keys are < 2^22)
constexpr float F4_EMPTY = 3.0e9f;
// b0 <= b1 always, so the new second-best min(b1, max(b0, k)) is the MEDIAN of (b0, b1, k): one v_med3_f32 instead of a
// max and a min -- three vector instructions per result (fma, med3, min) where the integer form has four
__device__ __forceinline__ void top2_push_f(float &b0, float &b1, float k)
{
    b1 = __builtin_amdgcn_fmed3f(b0, b1, k);
    b0 = fminf(b0, k);
}
__device__ __forceinline__ v4f mfma_f4(const v4i &a, const v4i &b, const v4f &c)
{
    const v8i A = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, B = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
    // cbsz = blgp = 4: FP4 E2M1 on both sides; scale operands: four E8M0 bytes of 127 (x 1.0), byte 0 selected
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}
template <int QT, int NW, int TG>
__global__ __launch_bounds__(64 * NW) void k_ov_match_f4(const uint32_t *__restrict__ qnib, const int32_t *__restrict__ qpop,
                                                    const int32_t *__restrict__ qn, const uint32_t *__restrict__ tnib,
                                                    const int32_t *__restrict__ tpop, const int32_t *__restrict__ tn,
                                                    const int32_t *__restrict__ pair_q, const int32_t *__restrict__ pair_t,
                                                    int32_t *__restrict__ out_idx /*[P][MAXKP][2]*/, int32_t *__restrict__ out_dist)
{
    extern __shared__ __attribute__((aligned(16))) int8_t s_t[];      // 2 x [TC][F4_ROW] descriptors, then 2 x [TC] keys
    const int p = blockIdx.y;
    const int fq = pair_q[p], ft = pair_t[p];
    const int nq = qn[fq], nt = tn[ft];
    const int q0 = blockIdx.x * (16 * NW * QT);
    if (q0 >= nq) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = lane & 15, kb = lane >> 4;
    const int8_t *Q = reinterpret_cast<const int8_t *>(qnib) + (size_t)fq * MAXKP * F4_DESC;
    const int8_t *T = reinterpret_cast<const int8_t *>(tnib) + (size_t)ft * MAXKP * F4_DESC;
    const int32_t *TP = tpop + (size_t)ft * MAXKP;
    v4i a[QT][4];
    int cq[QT][4];
    float b0[QT][4], b1[QT][4];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const int qrow = q0 + (wave * QT + u) * 16 + row;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            a[u][ks] = *reinterpret_cast<const v4i *>(Q + (size_t)qrow * F4_DESC + ks * 64 + kb * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            cq[u][r] = qpop[(size_t)fq * MAXKP + q0 + (wave * QT + u) * 16 + kb * 4 + r];
            b0[u][r] = b1[u][r] = F4_EMPTY;
        }
    }
    constexpr int TC = TG * 16;               // train columns per tile
    constexpr int NP = TC * 16 / (64 * NW);   // 16-byte pieces per thread
    constexpr int NK = (TC + 63) / 64;        // keys per lane
    static_assert(TG % 2 == 0 && TC * 16 % (64 * NW) == 0 && NP >= 1, "tile shape");
    v4i stage[NP];
    float stage_key[NK];
    auto fetch = [&](int t0) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = threadIdx.x + 64 * NW * j, tr = i >> 4, piece = i & 15;
            stage[j] = *reinterpret_cast<const v4i *>(T + (size_t)(t0 + tr) * F4_DESC + piece * 16);
        }
#pragma unroll
        for (int j = 0; j < NK; ++j) {
            const int t = t0 + lane + 64 * j;
            const int pc = TP[min(t, MAXKP - 1)];
            // live: (|t| + 512) * 2048 + t, exact in float32; a dead column adds 2^23 (no select on a loaded value: see the i8 form)
            stage_key[j] = (float)(((pc + 512) << 11) | t) + (t < nt ? 0.0f : F4_DEAD);
        }
    };
    int8_t *const bufA = s_t, *const bufB = s_t + (size_t)TC * F4_ROW;
    float *const keyA = reinterpret_cast<float *>(s_t + (size_t)2 * TC * F4_ROW), *const keyB = keyA + TC;
    auto park = [&](int8_t *buf, float *kbuf) {
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int i = threadIdx.x + 64 * NW * j, tr = i >> 4, piece = i & 15;
            *reinterpret_cast<v4i *>(buf + (size_t)tr * F4_ROW + piece * 16) = stage[j];
        }
#pragma unroll
        for (int j = 0; j < NK; ++j) if (lane + 64 * j < TC) kbuf[lane + 64 * j] = stage_key[j];
    };
    if (nt > 0) {
        fetch(0);
        park(bufA, keyA);
        fetch(TC);
    }
    __syncthreads();
    v4f acc[2][QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) acc[1][u] = v4f{0.f, 0.f, 0.f, 0.f};
    float tb_last = F4_DEAD;
    auto epilogue_one = [&](const v4f (&ac)[QT], float tbk, int idx) {
        const int u = idx >> 2, r = idx & 3;
        const float mf = tbk >= F4_DEAD ? 0.0f : -4096.0f;
        top2_push_f(b0[u][r], b1[u][r], fmaf(ac[u][r], mf, tbk));
    };
    for (int t0 = 0, it = 0; t0 < nt; t0 += TC, ++it) {
        const int8_t *cur = (it & 1) ? bufB : bufA;
        const float *kcur = (it & 1) ? keyB : keyA;
        float tb[TG];
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) tb[tt] = kcur[tt * 16 + row];
        v4i bf[2][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            bf[0][ks] = *reinterpret_cast<const v4i *>(cur + (size_t)row * F4_ROW + ks * 64 + kb * 16);
#pragma unroll
        for (int tt = 0; tt < TG; ++tt) {
            if (tt < TG - 1) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    bf[(tt + 1) & 1][ks] = *reinterpret_cast<const v4i *>(cur + (size_t)((tt + 1) * 16 + row) * F4_ROW + ks * 64 + kb * 16);
            }
            const float tbk = tt == 0 ? tb_last : tb[(tt + TG - 1) % TG];
#pragma unroll
            for (int u = 0; u < QT; ++u) acc[tt & 1][u] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int u = 0; u < QT; ++u) acc[tt & 1][u] = mfma_f4(a[u][ks], bf[tt & 1][ks], acc[tt & 1][u]);
                // the QT * 4 pending results of the previous group over the 4 steps of this one
#pragma unroll
                for (int e = 0; e < QT; ++e) epilogue_one(acc[(tt + 1) & 1], tbk, ks * QT + e);
            }
        }
        tb_last = tb[TG - 1];
        park((it & 1) ? bufA : bufB, (it & 1) ? keyA : keyB);
        fetch(min(t0 + 2 * TC, MAXKP - TC));
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < QT * 4; ++i) epilogue_one(acc[1], tb_last, i);
#pragma unroll
    for (int u = 0; u < QT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                const float o0 = __shfl_xor(b0[u][r], d, 64), o1 = __shfl_xor(b1[u][r], d, 64);
                top2_push_f(b0[u][r], b1[u][r], o0);
                top2_push_f(b0[u][r], b1[u][r], o1);
            }
            const int q = q0 + (wave * QT + u) * 16 + kb * 4 + r;
            if (row == 0 && q < nq) {
                const size_t o = ((size_t)p * MAXKP + q) * 2;
                const bool h0 = b0[u][r] < F4_DEAD, h1 = b1[u][r] < F4_DEAD;
                const int k0 = h0 ? (int)b0[u][r] : 0, k1 = h1 ? (int)b1[u][r] : 0;
                out_idx[o] = h0 ? (k0 & 2047) : -1; out_idx[o + 1] = h1 ? (k1 & 2047) : -1;
                out_dist[o] = h0 ? (k0 >> 11) - 512 + cq[u][r] : -1;
                out_dist[o + 1] = h1 ? (k1 >> 11) - 512 + cq[u][r] : -1;
            }
        }
}

// ---- ratio test + RANSAC homography + overlapArea, one block per pair ---------------------------------------
__device__ __forceinline__ uint32_t hash32(uint32_t a)
{
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}

// 8 x 8 Gaussian elimination with partial pivoting, the oracle's operations in the oracle's order -- but with every index
// a compile-time constant: the pivot row is swapped in by selects against each candidate row instead of A[p][k], so the
// 72 doubles live in registers.  (Indexed by the run-time pivot the array sat in scratch memory, and two of these solves
// per thread were half of k_ov_geometry's 0.4 ms.)  Columns left of the pivot column are never read again, so the swap
// and the elimination skip them.
__device__ __forceinline__ bool solve8(double (&A)[8][9])
{
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        int p = c;
        double best = fabs(A[c][c]);
#pragma unroll
        for (int r = c + 1; r < 8; ++r) {
            const double v = fabs(A[r][c]);
            if (v > best) { best = v; p = r; }          // strict: the first of equal maxima, as `p` walks in the oracle
        }
        if (!(best > 1e-12)) return false;
#pragma unroll
        for (int r = c + 1; r < 8; ++r) {
            const bool sw = p == r;
#pragma unroll
            for (int k = c; k < 9; ++k) {
                const double a = A[c][k], b = A[r][k];
                A[c][k] = sw ? b : a;
                A[r][k] = sw ? a : b;
            }
        }
#pragma unroll
        for (int r = c + 1; r < 8; ++r) {
            const double f = A[r][c] / A[c][c];
#pragma unroll
            for (int k = c; k < 9; ++k) A[r][k] = A[r][k] - f * A[c][k];
        }
    }
#pragma unroll
    for (int r = 7; r >= 0; --r) {
        double s = A[r][8];
#pragma unroll
        for (int k = r + 1; k < 8; ++k) s = s - A[r][k] * A[k][8];
        A[r][8] = s / A[r][r];
    }
    return true;
}

__device__ __forceinline__ bool is_inlier(const double *H, double x, double y, double X, double Y)
{
    const double wv = H[6] * x + H[7] * y + H[8];
    const double px = (H[0] * x + H[1] * y + H[2]) / wv, py = (H[3] * x + H[4] * y + H[5]) / wv;
    const double ex = px - X, ey = py - Y;
    return (ex * ex + ey * ey) <= 9.0;
}

__device__ bool clip_line(long long W, long long Hh, long long &x1, long long &y1, long long &x2, long long &y2)
{
    const long long right = W - 1, bottom = Hh - 1;
    int c1 = (x1 < 0) + (x1 > right) * 2 + (y1 < 0) * 4 + (y1 > bottom) * 8;
    int c2 = (x2 < 0) + (x2 > right) * 2 + (y2 < 0) * 4 + (y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            x1 += (long long)((double)(a - y1) * (double)(x2 - x1) / (double)(y2 - y1));
            y1 = a;
            c1 = (x1 < 0) + (x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            x2 += (long long)((double)(a - y2) * (double)(x2 - x1) / (double)(y2 - y1));
            y2 = a;
            c2 = (x2 < 0) + (x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                y1 += (long long)((double)(a - x1) * (double)(y2 - y1) / (double)(x2 - x1));
                x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                y2 += (long long)((double)(a - x2) * (double)(y2 - y1) / (double)(x2 - x1));
                x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

constexpr int MASK_WORDS = TW / 32;     // 20 words per row

__device__ void draw_line(uint32_t *mask, long long x1, long long y1, long long x2, long long y2)
{
    if (!clip_line(TW, TH, x1, y1, x2, y2)) return;
    if (x2 < x1) { long long t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
    int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
    const int sx = dx < 0 ? -1 : 1, sy = dy < 0 ? -1 : 1;
    dx = dx < 0 ? -dx : dx; dy = dy < 0 ? -dy : dy;
    int x = (int)x1, y = (int)y1;
    if (dy > dx) {
        int err = dy - (dx + dx);
        for (int i = 0; i <= dy; ++i) {
            mask[y * MASK_WORDS + (x >> 5)] |= 1u << (x & 31);
            const int m = err < 0;
            err += -(dx + dx) + (m ? dy + dy : 0);
            y += sy;
            if (m) x += sx;
        }
    } else {
        int err = dx - (dy + dy);
        for (int i = 0; i <= dx; ++i) {
            mask[y * MASK_WORDS + (x >> 5)] |= 1u << (x & 31);
            const int m = err < 0;
            err += -(dy + dy) + (m ? dx + dx : 0);
            x += sx;
            if (m) y += sy;
        }
    }
}

// scanline part of cv::fillConvexPoly (shift 0): per-row span ends into span[y] = (xx1, xx2) or (1, 0)
__device__ void fill_spans(const long long vx[4], const long long vy[4], short2 *span)
{
    const int XY_SHIFT = 16;
    const long long XY_ONE = 1 << XY_SHIFT;
    const int npts = 4;
    struct { int idx, di; long long x, dx; int ye; } edge[2];
    const int delta1 = (int)(XY_ONE >> 1), delta2 = (int)(XY_ONE >> 1);
    int imin = 0, edges = npts;
    long long xmin = vx[0], xmax = vx[0], ymin = vy[0], ymax = vy[0];
    for (int i = 0; i < npts; ++i) {
        if (vy[i] < ymin) { ymin = vy[i]; imin = i; }
        if (vy[i] > ymax) ymax = vy[i];
        if (vx[i] > xmax) xmax = vx[i];
        if (vx[i] < xmin) xmin = vx[i];
    }
    if ((int)xmax < 0 || (int)ymax < 0 || (int)xmin >= TW || (int)ymin >= TH) return;
    if (ymax > TH - 1) ymax = TH - 1;
    int y = (int)ymin;
    edge[0].idx = edge[1].idx = imin;
    edge[0].ye = edge[1].ye = y;
    edge[0].di = 1; edge[1].di = npts - 1;
    edge[0].x = edge[1].x = -XY_ONE;
    edge[0].dx = edge[1].dx = 0;
    do {
        for (int i = 0; i < 2; ++i) {
            if (y >= edge[i].ye) {
                int idx0 = edge[i].idx;
                const int di = edge[i].di;
                int idx = idx0 + di;
                if (idx >= npts) idx -= npts;
                for (; edges-- > 0;) {
                    const int ty = (int)vy[idx];
                    if (ty > y) {
                        const long long xs = vx[idx0] << XY_SHIFT, xe = vx[idx] << XY_SHIFT;
                        edge[i].ye = ty;
                        edge[i].dx = ((xe - xs) * 2 + (ty - y)) / (2 * (ty - y));
                        edge[i].x = xs;
                        edge[i].idx = idx;
                        break;
                    }
                    idx0 = idx;
                    idx += di;
                    if (idx >= npts) idx -= npts;
                }
            }
        }
        if (edges < 0) break;
        if (y >= 0) {
            int left = 0, right = 1;
            if (edge[0].x > edge[1].x) { left = 1; right = 0; }
            int xx1 = (int)((edge[left].x + delta1) >> XY_SHIFT);
            int xx2 = (int)((edge[right].x + delta2) >> XY_SHIFT);
            if (xx2 >= 0 && xx1 < TW) {
                if (xx1 < 0) xx1 = 0;
                if (xx2 >= TW) xx2 = TW - 1;
                span[y] = make_short2((short)xx1, (short)xx2);
            }
        }
        edge[0].x += edge[0].dx;
        edge[1].x += edge[1].dx;
    } while (++y <= (int)ymax);
}

// overlapArea(H), videostrip.cpp:291-319.  Must be called by the whole 256-thread block.
__device__ float overlap_area_block(const double *H, int videoW, int videoH, uint32_t *s_mask /*[TH*MASK_WORDS]*/,
                                    short2 *s_span /*[TH]*/, uint32_t *scratch, int *ov_out)
{
    __shared__ float s_f[8];
    for (int i = threadIdx.x; i < TH * MASK_WORDS; i += 256) s_mask[i] = 0;
    for (int i = threadIdx.x; i < TH; i += 256) s_span[i] = make_short2(1, 0);
    __syncthreads();
    if (threadIdx.x == 0) {
        const float px[4] = {0, (float)TW, (float)TW, 0}, py[4] = {0, 0, (float)TH, (float)TH};
        long long vx[4], vy[4];
        for (int i = 0; i < 4; ++i) {
            const double x = px[i], y = py[i];
            double wv = x * H[6] + y * H[7] + H[8];
            float fx = 0.0f, fy = 0.0f;
            if (fabs(wv) > 2.220446049250313e-16) {
                wv = 1.0 / wv;
                fx = (float)((x * H[0] + y * H[1] + H[2]) * wv);
                fy = (float)((x * H[3] + y * H[4] + H[5]) * wv);
            }
            s_f[i] = fx; s_f[4 + i] = fy;
            vx[i] = (long long)__float2int_rn(fx);
            vy[i] = (long long)__float2int_rn(fy);
        }
        for (int i = 0; i < 4; ++i) {
            const int p = (i + 3) % 4;
            draw_line(s_mask, vx[p], vy[p], vx[i], vy[i]);
        }
        fill_spans(vx, vy, s_span);
    }
    __syncthreads();
    uint32_t cnt = 0;
    for (int i = threadIdx.x; i < TH * MASK_WORDS; i += 256) {
        const int y = i / MASK_WORDS, wd = i - y * MASK_WORDS;
        uint32_t m = s_mask[i];
        const short2 sp = s_span[y];
        const int lo = max((int)sp.x, wd * 32), hi = min((int)sp.y, wd * 32 + 31);
        if (lo <= hi) {
            const int nb = hi - lo + 1;
            const uint32_t bitsm = nb == 32 ? 0xffffffffu : (((1u << nb) - 1u) << (lo & 31));
            m |= bitsm;
        }
        cnt += __popc(m);
    }
    const uint32_t ov = block256_sum_u32(cnt, scratch);
    if (ov_out) *ov_out = (int)ov;
    double a00 = 0;
    for (int i = 0; i < 4; ++i) {
        const int p = (i + 3) % 4;
        a00 += (double)s_f[p] * s_f[4 + i] - (double)s_f[4 + p] * s_f[i];
    }
    const float area1 = (float)(videoW * videoH), area2 = (float)fabs(a00 * 0.5), cur = (float)ov;
    return cur / (area1 + area2 - cur);
}

constexpr int NSUM = 44;

__global__ __launch_bounds__(256) void k_ov_geometry(const Keypoint *__restrict__ qkp, const Keypoint *__restrict__ tkp,
                                                    const int32_t *__restrict__ qn, const int32_t *__restrict__ tn,
                                                    const int32_t *__restrict__ pair_q, const int32_t *__restrict__ pair_t,
                                                    const int32_t *__restrict__ m_idx, const int32_t *__restrict__ m_dist,
                                                    int w, int h, int videoW, int videoH, uint32_t seed, int min_inliers,
                                                    float *__restrict__ ratio, int32_t *__restrict__ info /*[P][8]*/,
                                                    double *__restrict__ Hout /*[P][9]*/)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
    // carve: good points 4 x MAXKP floats (32 KB) | inlier flags MAXKP (2 KB) | mask (38.4 KB) | spans (1.9 KB)
    float *s_ox = reinterpret_cast<float *>(s_raw), *s_oy = s_ox + MAXKP, *s_sx = s_oy + MAXKP, *s_sy = s_sx + MAXKP;
    uint8_t *s_inl = reinterpret_cast<uint8_t *>(s_sy + MAXKP);
    uint32_t *s_mask = reinterpret_cast<uint32_t *>(s_inl + MAXKP);
    short2 *s_span = reinterpret_cast<short2 *>(s_mask + TH * MASK_WORDS);
    __shared__ uint32_t scratch[16];
    __shared__ int s_best_cnt[256], s_best_it[256];
    __shared__ double s_H[9];
    __shared__ double s_g[4][NSUM];
    __shared__ int s_ng;

    const int p = blockIdx.x, tid = threadIdx.x;
    const int fq = pair_q[p], ft = pair_t[p];
    const int nq = qn[fq], nt = tn[ft];
    const Keypoint *KQ = qkp + (size_t)fq * MAXKP, *KT = tkp + (size_t)ft * MAXKP;
    const int32_t *mi = m_idx + (size_t)p * MAXKP * 2, *md = m_dist + (size_t)p * MAXKP * 2;
    int32_t *inf = info + (size_t)p * 8;

    // ratio test (videostrip.cpp:233-242; last query skipped, B-12), order-preserving compaction
    if (tid == 0) s_ng = 0;
    __syncthreads();
    const int limit = (nt >= 2 && nq >= 1) ? nq - 1 : 0;
    for (int base = 0; base < limit; base += 256) {
        const int k = base + tid;
        uint32_t good = 0;
        if (k < limit) good = ((double)md[k * 2] < 0.8 * (double)md[k * 2 + 1]) ? 1u : 0u;
        const uint32_t incl = block256_incl_scan_u32(good, scratch);
        const int off = s_ng;
        if (good) {
            const int pos = off + (int)(incl - 1);
            const Keypoint a = KQ[k], b = KT[mi[k * 2]];
            s_ox[pos] = a.x; s_oy[pos] = a.y; s_sx[pos] = b.x; s_sy[pos] = b.y;
        }
        __syncthreads();
        if (tid == 255) s_ng = off + (int)incl;
        __syncthreads();
    }
    const int ng = s_ng;
    if (tid == 0) { inf[0] = nq; inf[1] = nt; inf[2] = ng; inf[3] = 0; inf[4] = 0; }
    if (ng < 4) {                                    // "Not enough good matches" -> -2.0 (videostrip.cpp:252-256)
        if (tid == 0) ratio[p] = -2.0f;
        return;
    }
    // 512 hypotheses, 2 per thread
    int my_cnt = 0, my_it = 0x7fffffff;
    for (int rep = 0; rep < RANSAC_ITERS / 256; ++rep) {
        const int it = rep * 256 + tid;
        int pick[4];
        for (int j = 0; j < 4; ++j) {
            uint32_t attempt = 0;
            for (;;) {
                const uint32_t r = hash32(seed ^ hash32((uint32_t)(it * 4 + j + 1) + attempt * 0x9e3779b9u));
                const int c = (int)(r % (uint32_t)ng);
                bool dup = false;
                for (int m = 0; m < j; ++m) dup = dup || (pick[m] == c);
                if (!dup || attempt >= 16) { pick[j] = c; break; }
                attempt++;
            }
        }
        double A[8][9];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double x = s_ox[pick[j]], y = s_oy[pick[j]], X = s_sx[pick[j]], Y = s_sy[pick[j]];
            double *r0 = A[2 * j], *r1 = A[2 * j + 1];
            r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -x * X; r0[7] = -y * X; r0[8] = X;
            r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -x * Y; r1[7] = -y * Y; r1[8] = Y;
        }
        if (!solve8(A)) continue;
        double Hc[9];
#pragma unroll
        for (int k = 0; k < 8; ++k) Hc[k] = A[k][8];
        Hc[8] = 1.0;
        int cnt = 0;
        for (int i = 0; i < ng; ++i) cnt += is_inlier(Hc, s_ox[i], s_oy[i], s_sx[i], s_sy[i]) ? 1 : 0;
        if (cnt > my_cnt) { my_cnt = cnt; my_it = it; }      // it increases: keeps the first maximum
    }
    s_best_cnt[tid] = my_cnt; s_best_it[tid] = my_it;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) {
            const int oc = s_best_cnt[tid + s], oi = s_best_it[tid + s];
            if (oc > s_best_cnt[tid] || (oc == s_best_cnt[tid] && oi < s_best_it[tid])) { s_best_cnt[tid] = oc; s_best_it[tid] = oi; }
        }
        __syncthreads();
    }
    const int best = s_best_cnt[0], best_it = s_best_it[0];
    if (best < min_inliers) {                         // H.empty() -> -2.0 (videostrip.cpp:272)
        if (tid == 0) ratio[p] = -2.0f;
        return;
    }
    if (tid == 0) {
        // rebuild the winning hypothesis
        int pick[4];
        for (int j = 0; j < 4; ++j) {
            uint32_t attempt = 0;
            for (;;) {
                const uint32_t r = hash32(seed ^ hash32((uint32_t)(best_it * 4 + j + 1) + attempt * 0x9e3779b9u));
                const int c = (int)(r % (uint32_t)ng);
                bool dup = false;
                for (int m = 0; m < j; ++m) dup = dup || (pick[m] == c);
                if (!dup || attempt >= 16) { pick[j] = c; break; }
                attempt++;
            }
        }
        double A[8][9];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double x = s_ox[pick[j]], y = s_oy[pick[j]], X = s_sx[pick[j]], Y = s_sy[pick[j]];
            double *r0 = A[2 * j], *r1 = A[2 * j + 1];
            r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -x * X; r0[7] = -y * X; r0[8] = X;
            r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -x * Y; r1[7] = -y * Y; r1[8] = Y;
        }
        solve8(A);
        for (int k = 0; k < 8; ++k) s_H[k] = A[k][8];
        s_H[8] = 1.0;
        inf[3] = best;
    }
    __syncthreads();
    for (int i = tid; i < ng; i += 256) s_inl[i] = is_inlier(s_H, s_ox[i], s_oy[i], s_sx[i], s_sy[i]) ? 1 : 0;
    __syncthreads();
    // least-squares refit in fixed-normalised coordinates; summation order = the oracle's
    const double cx = 0.5 * (double)w, cy = 0.5 * (double)h, sN = 0.5 * (double)w;
    double part[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) part[k] = 0.0;
    for (int i = tid; i < ng; i += 256) {
        if (!s_inl[i]) continue;
        const double x = ((double)s_ox[i] - cx) / sN, y = ((double)s_oy[i] - cy) / sN;
        const double X = ((double)s_sx[i] - cx) / sN, Y = ((double)s_sy[i] - cy) / sN;
        const double a[8] = {x, y, 1, 0, 0, 0, -x * X, -y * X}, b[8] = {0, 0, 0, x, y, 1, -x * Y, -y * Y};
        int k = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = r; c < 8; ++c) { part[k] = part[k] + (a[r] * a[c] + b[r] * b[c]); k++; }
#pragma unroll
        for (int r = 0; r < 8; ++r) { part[k] = part[k] + (a[r] * X + b[r] * Y); k++; }
    }
#pragma unroll
    for (int k = 0; k < NSUM; ++k) {
        double v = part[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v = v + __shfl_xor(v, d, 64);
        if ((tid & 63) == 0) s_g[tid >> 6][k] = v;
    }
    __syncthreads();
    if (tid == 0) {
        double A[8][9];
        int k = 0;
        for (int r = 0; r < 8; ++r)
            for (int c = r; c < 8; ++c) {
                const double t = ((s_g[0][k] + s_g[1][k]) + s_g[2][k]) + s_g[3][k];
                A[r][c] = t; A[c][r] = t; k++;
            }
        for (int r = 0; r < 8; ++r) { A[r][8] = ((s_g[0][k] + s_g[1][k]) + s_g[2][k]) + s_g[3][k]; k++; }
        int ok = solve8(A) ? 1 : 0;
        if (ok) {
            const double hn[9] = {A[0][8], A[1][8], A[2][8], A[3][8], A[4][8], A[5][8], A[6][8], A[7][8], 1.0};
            double M[9], R[9];
            for (int r = 0; r < 3; ++r) {
                M[r * 3 + 0] = hn[r * 3 + 0] / sN;
                M[r * 3 + 1] = hn[r * 3 + 1] / sN;
                M[r * 3 + 2] = (hn[r * 3 + 2] - hn[r * 3 + 0] * (cx / sN)) - hn[r * 3 + 1] * (cy / sN);
            }
            for (int c = 0; c < 3; ++c) {
                R[0 * 3 + c] = sN * M[0 * 3 + c] + cx * M[2 * 3 + c];
                R[1 * 3 + c] = sN * M[1 * 3 + c] + cy * M[2 * 3 + c];
                R[2 * 3 + c] = M[2 * 3 + c];
            }
            if (R[8] == 0.0 || R[8] != R[8]) ok = 0;
            else for (int i = 0; i < 9; ++i) s_H[i] = R[i] / R[8];
        }
        (void)ok;
        if (Hout) for (int i = 0; i < 9; ++i) Hout[(size_t)p * 9 + i] = s_H[i];
    }
    __syncthreads();
    int ov = 0;
    const float r = overlap_area_block(s_H, videoW, videoH, s_mask, s_span, scratch, &ov);
    if (tid == 0) { ratio[p] = r; inf[4] = ov; }
}

// standalone overlapArea on a list of homographies
__global__ __launch_bounds__(256) void k_ov_area_only(const double *__restrict__ Hs, int videoW, int videoH, float *__restrict__ ratio,
                                                     int32_t *__restrict__ ovc)
{
    __shared__ uint32_t s_mask[TH * MASK_WORDS];
    __shared__ short2 s_span[TH];
    __shared__ uint32_t scratch[16];
    __shared__ double s_H[9];
    if (threadIdx.x < 9) s_H[threadIdx.x] = Hs[(size_t)blockIdx.x * 9 + threadIdx.x];
    __syncthreads();
    int ov = 0;
    const float r = overlap_area_block(s_H, videoW, videoH, s_mask, s_span, scratch, &ov);
    if (threadIdx.x == 0) { ratio[blockIdx.x] = r; if (ovc) ovc[blockIdx.x] = ov; }
}

// ---- V5 calcBlur: gray -> Laplacian (aperture 3, saturated to u8) -> population stddev ---------------------------
__global__ __launch_bounds__(256) void k_ov_blur(const uint8_t *__restrict__ gray, int h, int w, double *__restrict__ part /*[F][nb][2]*/)
{
    __shared__ double scratch[8];
    const int f = blockIdx.y;
    const uint8_t *g = gray + (size_t)f * h * w;
    const size_t n = (size_t)h * w;
    double s = 0.0, s2 = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i - (size_t)y * w);
        const int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h), xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
        int v = 2 * (g[(size_t)ym * w + xm] + g[(size_t)ym * w + xp] + g[(size_t)yp * w + xm] + g[(size_t)yp * w + xp]) - 8 * g[(size_t)y * w + x];
        v = min(max(v, 0), 255);
        s += v; s2 += (double)v * v;
    }
    // integer-valued sums: exact in double, so the reduction order is immaterial
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    s = wave_sum_f64(s); s2 = wave_sum_f64(s2);
    if (lane == 0) { scratch[wave] = s; scratch[4 + wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *o = part + ((size_t)f * gridDim.x + blockIdx.x) * 2;
        o[0] = scratch[0] + scratch[1] + scratch[2] + scratch[3];
        o[1] = scratch[4] + scratch[5] + scratch[6] + scratch[7];
    }
}

__global__ void k_ov_blur_final(const double *__restrict__ part, int nb, double npix, float *__restrict__ out, int F)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    double s = 0, s2 = 0;
    for (int k = 0; k < nb; ++k) { s += part[((size_t)f * nb + k) * 2]; s2 += part[((size_t)f * nb + k) * 2 + 1]; }
    const double mean = s / npix;
    const double var = s2 / npix - mean * mean;
    out[f] = (float)sqrt(var < 0 ? 0 : var);
}

// --------------------------------------------------------------------------------------------------------------------
ConvK gauss_kernel(float sigma)
{
    ConvK K{};
    int ks = (int)std::ceil(2.0 * (1.0 + ((double)sigma - 0.8) / 0.3));
    if ((ks & 1) == 0) ks++;
    const int r = ks / 2;
    double sum = 0, tmp[32];
    for (int i = 0; i < ks; ++i) { tmp[i] = std::exp(-((double)(i - r) * (i - r)) / (2.0 * (double)sigma * (double)sigma)); sum += tmp[i]; }
    K.ks = ks;
    for (int i = 0; i < ks; ++i) K.k[i] = (float)(tmp[i] / sum);
    return K;
}

int fed_taus(float T, float *tau)
{
    const double tau_max = 0.25;
    int n = (int)(std::ceil(std::sqrt(3.0 * (double)T / tau_max + 0.25) - 0.5 - 1.0e-8) + 0.5);
    if (n < 1) n = 1;
    const double scale = 3.0 * (double)T / (tau_max * (double)(n * (n + 1)));
    const double c = 1.0 / (4.0 * (double)n + 2.0), d = scale * tau_max / 2.0;
    for (int k = 0; k < n; ++k) {
        const double hh = std::cos(3.14159265358979323846 * (2.0 * (double)k + 1.0) * c);
        tau[k] = (float)(d / (hh * hh));
    }
    return n;
}

void resize_dims(int rows, int cols, int target_w, int *orows, int *ocols)
{
    const float f = (float)target_w / (float)cols;                 // hResizeFactor (main.cpp:242)
    *ocols = (int)std::lrint((double)cols * (double)f);
    *orows = (int)std::lrint((double)rows * (double)f);
}

struct ResizeTab {
    std::vector<int> ofs;
    std::vector<short> c0, c1;
};

void resize_tab(int ssize, int dsize, ResizeTab &t)
{
    t.ofs.resize(dsize); t.c0.resize(dsize); t.c1.resize(dsize);
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    for (int d = 0; d < dsize; ++d) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        int sx = (int)std::floor(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= ssize - 1) { fx = 0; sx = ssize - 1; }
        t.ofs[d] = sx;
        const long r0 = std::lrintf((1.0f - fx) * 2048.0f), r1 = std::lrintf(fx * 2048.0f);
        t.c0[d] = (short)std::min<long>(r0, 32767);
        t.c1[d] = (short)std::min<long>(r1, 32767);
    }
}

// device table: [ofs int32 x d][c0 int16 x d][c1 int16 x d]
const void *resize_table(uwip_ctx *ctx, int ssize, int dsize)
{
    char key[64];
    snprintf(key, sizeof key, "resize:%d:%d", ssize, dsize);
    const void *d = uwip_table_find(ctx, key, nullptr);
    if (d) return d;
    ResizeTab t;
    resize_tab(ssize, dsize, t);
    std::vector<uint8_t> buf((size_t)dsize * 8);
    memcpy(buf.data(), t.ofs.data(), (size_t)dsize * 4);
    memcpy(buf.data() + (size_t)dsize * 4, t.c0.data(), (size_t)dsize * 2);
    memcpy(buf.data() + (size_t)dsize * 6, t.c1.data(), (size_t)dsize * 2);
    return uwip_table_put(ctx, key, buf.data(), buf.size());
}

dim3 grid2d(int w, int h, int z) { return dim3(uwip_cdiv(w, 64), uwip_cdiv(h, 4), (unsigned)z); }

struct OvWork {
    uint8_t *gray;
    float *L0, *Lsm, *flow, *ping, *Lt, *Ldet, *cand, *kc;
    float2 *Lxy;       // (Lx, Ly) interleaved, [NLEV][F][h][w]
    uint32_t *hmax, *khist, *selhist, *sel, *counts;
};

int alloc_work(uwip_ctx *ctx, int F, int h, int w, OvWork *W)
{
    const size_t n = (size_t)h * w;
    const int nchunks = (int)((n * NLEV + CMP_CHUNK - 1) / CMP_CHUNK);
    W->gray = (uint8_t *)uwip_ws(ctx, "ov.gray", n * F);
    W->L0 = (float *)uwip_ws(ctx, "ov.L0", n * F * 4);
    W->Lsm = (float *)uwip_ws(ctx, "ov.Lsm", n * F * 4);
    W->flow = (float *)uwip_ws(ctx, "ov.flow", n * F * 4);
    W->ping = (float *)uwip_ws(ctx, "ov.ping", n * F * 4);
    W->Lt = (float *)uwip_ws(ctx, "ov.Lt", n * F * 4 * NLEV);
    W->Lxy = (float2 *)uwip_ws(ctx, "ov.Lxy", n * F * 8 * NLEV);
    W->Ldet = (float *)uwip_ws(ctx, "ov.Ldet", n * F * 4 * NLEV);
    W->cand = (float *)uwip_ws(ctx, "ov.cand", n * F * 4 * NLEV);
    W->kc = (float *)uwip_ws(ctx, "ov.kc", sizeof(float) * F);
    W->hmax = (uint32_t *)uwip_ws(ctx, "ov.hmax", sizeof(uint32_t) * F);
    W->khist = (uint32_t *)uwip_ws(ctx, "ov.khist", sizeof(uint32_t) * 304 * F);
    W->selhist = (uint32_t *)uwip_ws(ctx, "ov.selhist", sizeof(uint32_t) * 65536 * F);
    W->sel = (uint32_t *)uwip_ws(ctx, "ov.sel", sizeof(uint32_t) * 4 * F);
    W->counts = (uint32_t *)uwip_ws(ctx, "ov.counts", sizeof(uint32_t) * nchunks * F);
    if (!W->gray || !W->L0 || !W->Lsm || !W->flow || !W->ping || !W->Lt || !W->Lxy || !W->Ldet || !W->cand ||
        !W->kc || !W->hmax || !W->khist || !W->selhist || !W->sel || !W->counts)
        return UWIP_ERR_NOMEM;
    return UWIP_OK;
}

// Level images are stored level-major, [NLEV][F][h][w]: every level is itself a dense batch, so the per-level
// kernels write their results in place (no staging copies).
// detect + describe every frame whose gray/L0 already sit in W (working size h x w)
int detect_describe(uwip_ctx *ctx, OvWork &W, int F, int h, int w, uwip_features *ft, int first_slot, int upright, int fixed_thr)
{
    const size_t n = (size_t)h * w, lvl = n * F;
    const dim3 g = grid2d(w, h, F);
    ctx->ov_last_frames = F;
    {
        uwip_kscope ks(ctx, "k_ov_scale_space");
        const ConvK K0 = gauss_kernel(H_SIGMA[0]), K1 = gauss_kernel(1.0f);
        UWIP_REQUIRE(ctx, K0.ks / 2 <= CV_RMAX && K1.ks / 2 <= CV_RMAX && (K0.ks & 1) && (K1.ks & 1), "Gaussian kernel too wide for k_ov_conv2");
        const dim3 gc(uwip_cdiv(w, CV_TW), uwip_cdiv(h, CV_TH), (unsigned)F);
        launch_conv2(K0.ks, gc, ctx->stream, (const float *)W.L0, W.Lt, h, w, K0);
        for (int lv = 0; lv < NLEV; ++lv) {
            float *Lt = W.Lt + lv * lvl;
            launch_conv2(K1.ks, gc, ctx->stream, (const float *)Lt, W.Lsm, h, w, K1);
            if (lv == 0) {
                const dim3 gk(g.x, uwip_cdiv(g.y, KC_ROWS), g.z);
                const int nbk = (int)(gk.x * gk.y);
                uint32_t *kpart = (uint32_t *)uwip_ws(ctx, "ov.kcpart", sizeof(uint32_t) * nbk * F);
                if (!kpart) return UWIP_ERR_NOMEM;
                k_ov_kc<0><<<gk, 256, 0, ctx->stream>>>(W.Lsm, h, w, W.hmax, kpart);
                k_ov_kc_max<<<F, 256, 0, ctx->stream>>>(kpart, nbk, W.hmax);
                UWIP_HIP(ctx, hipMemsetAsync(W.khist, 0, sizeof(uint32_t) * 304 * F, ctx->stream));
                k_ov_kc<1><<<gk, 256, 0, ctx->stream>>>(W.Lsm, h, w, W.hmax, W.khist);
                k_ov_kc_final<<<uwip_cdiv(F, 64), 64, 0, ctx->stream>>>(W.hmax, W.khist, W.kc, F);
            }
            const int s = H_SSIZE[lv];
            k_ov_deriv1<<<g, 256, 0, ctx->stream>>>(W.Lsm, W.Lxy + lv * lvl, h, w, s, W.kc, lv + 1 < NLEV ? W.flow : nullptr);
            k_ov_ldet<<<g, 256, 0, ctx->stream>>>(W.Lxy + lv * lvl, W.Ldet + lv * lvl, h, w, s);
            if (lv + 1 < NLEV) {
                const float e0 = 0.5f * H_SIGMA[lv] * H_SIGMA[lv], e1 = 0.5f * H_SIGMA[lv + 1] * H_SIGMA[lv + 1];
                float taus[32];
                const int nt = fed_taus(e1 - e0, taus);
                // up to FDN_MAX steps per launch, split as evenly as possible (8 -> 4 + 4, 6 -> 3 + 3, 4 -> 4); ping-pong
                // between Lt[lv+1] and a scratch plane so that the last launch lands in Lt[lv+1]
                float *next = W.Lt + (lv + 1) * lvl;
                const float *src = Lt;
                const int nl = (nt + FDN_MAX - 1) / FDN_MAX;
                const dim3 gf(uwip_cdiv(w, FD_TW), uwip_cdiv(h, FD_TH), (unsigned)F);
                for (int j = 0, k = 0; j < nl; ++j) {
                    float *dst = ((nl - j) & 1) ? next : W.ping;
                    const int ns = (nt - k + (nl - j) - 1) / (nl - j);
                    FedTaus tk;
                    for (int q = 0; q < FDN_MAX; ++q) tk.t[q] = q < ns ? taus[k + q] : 0.0f;
                    switch (ns) {
                    case 1: k_ov_fed<<<g, 256, 0, ctx->stream>>>(src, W.flow, dst, h, w, taus[k]); break;
                    case 2: k_ov_fed2<<<gf, 256, 0, ctx->stream>>>(src, W.flow, dst, h, w, taus[k], taus[k + 1]); break;
                    case 3: k_ov_fedn<3><<<gf, 256, 0, ctx->stream>>>(src, W.flow, dst, h, w, tk); break;
                    default: k_ov_fedn<4><<<gf, 256, 0, ctx->stream>>>(src, W.flow, dst, h, w, tk); break;
                    }
                    k += ns;
                    src = dst;
                }
            }
        }
        UWIP_HIP(ctx, hipGetLastError());
    }
    const size_t n4 = n * NLEV;
    const int nchunks = (int)((n4 + CMP_CHUNK - 1) / CMP_CHUNK);
    Keypoint *kps = ft->d_kp + (size_t)first_slot * MAXKP;
    int32_t *nkp = ft->d_n + first_slot;
    {
        uwip_kscope ks(ctx, "k_ov_detect");
        UWIP_HIP(ctx, hipMemsetAsync(W.selhist, 0, sizeof(uint32_t) * 65536 * F, ctx->stream));
        k_ov_extrema<<<dim3(uwip_cdiv(w, EX_TW), uwip_cdiv(h, EX_TH), (unsigned)F), 256, 0, ctx->stream>>>(W.Ldet, W.cand, h, w, F, W.kc, fixed_thr, W.selhist);
        k_ov_sel_pick<0><<<F, 256, 0, ctx->stream>>>(W.selhist, W.sel);
        UWIP_HIP(ctx, hipMemsetAsync(W.selhist, 0, sizeof(uint32_t) * 65536 * F, ctx->stream));
        k_ov_sel_hist<1><<<dim3(64, F), 256, 0, ctx->stream>>>(W.cand, n4, W.selhist, W.sel);
        k_ov_sel_pick<1><<<F, 256, 0, ctx->stream>>>(W.selhist, W.sel);
        k_ov_count<<<dim3(uwip_cdiv(nchunks, 4), F), 256, 0, ctx->stream>>>(W.cand, n4, W.sel, W.counts, nchunks);
        k_ov_scan_chunks<<<F, 256, 0, ctx->stream>>>(W.counts, nchunks, nkp);
        k_ov_compact<<<dim3(uwip_cdiv(nchunks, 4), F), 256, 0, ctx->stream>>>(W.cand, W.Ldet, h, w, W.sel, W.counts, nchunks, kps, F);
        UWIP_HIP(ctx, hipGetLastError());
    }
    {
        uwip_kscope ks(ctx, "k_ov_describe");
        k_ov_describe<<<8u * (((unsigned)MAXKP * F + 7u) / 8u), 64, 0, ctx->stream>>>(W.Lt, W.Lxy, h, w, kps, nkp,
                                                             ft->d_desc + (size_t)first_slot * MAXKP * DESC_BYTES,
                                                             ft->d_bits + (size_t)first_slot * MAXKP * DESC_K,
                                                             ft->d_nib + (size_t)first_slot * MAXKP * DESC_NIBW,
                                                             ft->d_pop + (size_t)first_slot * MAXKP, F, upright);
        UWIP_HIP(ctx, hipGetLastError());
    }
    return UWIP_OK;
}

}  // namespace

// ---- exported entry points -------------------------------------------------------------------------------------

UWIP_API int uwip_features_create(uwip_ctx *ctx, int max_frames, uwip_features **out)
{
    if (!ctx || !out) return UWIP_ERR_INVALID;
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    UWIP_REQUIRE(ctx, max_frames >= 1 && max_frames <= 4096, "max_frames must be in [1,4096]");
    uwip_features *f = new (std::nothrow) uwip_features();
    if (!f) return UWIP_ERR_NOMEM;
    f->ctx = ctx; f->capacity = max_frames;
    const size_t K = (size_t)max_frames * MAXKP;
    if (hipMalloc(&f->d_kp, K * sizeof(Keypoint)) != hipSuccess || hipMalloc(&f->d_desc, K * DESC_BYTES) != hipSuccess ||
        hipMalloc(&f->d_bits, K * DESC_K) != hipSuccess || hipMalloc(&f->d_nib, K * DESC_NIBW * 4) != hipSuccess ||
        hipMalloc(&f->d_pop, K * sizeof(int32_t)) != hipSuccess ||
        hipMalloc(&f->d_n, sizeof(int32_t) * max_frames) != hipSuccess) {
        (void)hipFree(f->d_kp); (void)hipFree(f->d_desc); (void)hipFree(f->d_bits); (void)hipFree(f->d_nib); (void)hipFree(f->d_pop); (void)hipFree(f->d_n);
        delete f;
        return ctx->fail(UWIP_ERR_NOMEM, "feature set hipMalloc");
    }
    uwip_trace_range(ctx, "device", "features.kp", f->d_kp, K * sizeof(Keypoint));
    uwip_trace_range(ctx, "device", "features.desc", f->d_desc, K * DESC_BYTES);
    uwip_trace_range(ctx, "device", "features.bits", f->d_bits, K * DESC_K);
    uwip_trace_range(ctx, "device", "features.pop", f->d_pop, K * sizeof(int32_t));
    (void)hipMemsetAsync(f->d_n, 0, sizeof(int32_t) * max_frames, ctx->stream);
    (void)hipMemsetAsync(f->d_bits, 0, K * DESC_K, ctx->stream);
    (void)hipMemsetAsync(f->d_nib, 0, K * DESC_NIBW * 4, ctx->stream);
    (void)hipMemsetAsync(f->d_pop, 0, K * sizeof(int32_t), ctx->stream);
    *out = f;
    return UWIP_OK;
}

UWIP_API int uwip_features_destroy(uwip_features *f)
{
    if (!f) return UWIP_OK;
    (void)hipSetDevice(f->ctx->device);
    (void)uwip_stream_wait(f->ctx);
    (void)hipFree(f->d_kp); (void)hipFree(f->d_desc); (void)hipFree(f->d_bits); (void)hipFree(f->d_nib); (void)hipFree(f->d_pop); (void)hipFree(f->d_n);
    delete f;
    return UWIP_OK;
}

UWIP_API int uwip_overlap_working_size(int rows, int cols, int *orows, int *ocols)
{
    if (!orows || !ocols || rows <= 0 || cols <= 0) return UWIP_ERR_INVALID;
    resize_dims(rows, cols, TW, orows, ocols);
    return UWIP_OK;
}

// frames: full-resolution BGR (resized to 640 wide inside, main.cpp:242,311) or, when `already_gray`
// is set, 8UC1 planes already at the working size.  Fills slots [first_slot, first_slot+frames).
UWIP_API int uwip_overlap_detect(uwip_ctx *ctx, const uwip_batch_u8 *frames, uwip_features *feats, int first_slot)
{
    return uwip_overlap_detect_ex(ctx, frames, feats, first_slot, 0u);
}

UWIP_API int uwip_overlap_detect_ex(uwip_ctx *ctx, const uwip_batch_u8 *frames, uwip_features *feats, int first_slot, unsigned flags)
{
    int rc = uwip_check_batch(ctx, frames, 0);
    if (rc) return rc;
    UWIP_REQUIRE(ctx, (flags & ~(unsigned)(UWIP_OVERLAP_UPRIGHT | UWIP_OVERLAP_FIXED_THRESHOLD | UWIP_OVERLAP_RELATIVE_THRESHOLD)) == 0, "unknown flag");
    UWIP_REQUIRE(ctx, (flags & (UWIP_OVERLAP_FIXED_THRESHOLD | UWIP_OVERLAP_RELATIVE_THRESHOLD)) != (UWIP_OVERLAP_FIXED_THRESHOLD | UWIP_OVERLAP_RELATIVE_THRESHOLD),
                 "UWIP_OVERLAP_FIXED_THRESHOLD and UWIP_OVERLAP_RELATIVE_THRESHOLD exclude each other");
    UWIP_REQUIRE(ctx, feats != nullptr && feats->ctx == ctx, "feature set belongs to another context");
    UWIP_REQUIRE(ctx, first_slot >= 0 && first_slot + frames->frames <= feats->capacity, "feature set too small");
    if (frames->frames == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, !uwip_batch_empty(frames), "empty image");           // calcOverlap returns -1 there
    const int F = frames->frames;
    int h, w;
    if (frames->channels == 3) resize_dims(frames->rows, frames->cols, TW, &h, &w);
    else { h = frames->rows; w = frames->cols; }
    UWIP_REQUIRE(ctx, h >= 2 * BORDER + 3 && w >= 2 * BORDER + 3, "working image too small");
    UWIP_REQUIRE(ctx, feats->w == 0 || (feats->w == w && feats->h == h), "feature set holds frames of another size");
    OvWork W;
    rc = alloc_work(ctx, F, h, w, &W);
    if (rc) return rc;
    const size_t n = (size_t)h * w;
    if (frames->channels == 3) {
        const uint8_t *tx = (const uint8_t *)resize_table(ctx, frames->cols, w);
        const uint8_t *ty = (const uint8_t *)resize_table(ctx, frames->rows, h);
        if (!tx || !ty) return UWIP_ERR_NOMEM;
        uwip_kscope ks(ctx, "k_ov_resize_gray");
        k_ov_resize_gray<<<grid2d(w, h, F), 256, 0, ctx->stream>>>(
            (const uint8_t *)frames->data, frames->step, frames->frame_stride, frames->rows, frames->cols, h, w,
            (const int *)tx, (const short *)(tx + (size_t)w * 4), (const short *)(tx + (size_t)w * 6),
            (const int *)ty, (const short *)(ty + (size_t)h * 4), (const short *)(ty + (size_t)h * 6), W.gray, W.L0);
        UWIP_HIP(ctx, hipGetLastError());
    } else {
        for (int f = 0; f < F; ++f)
            UWIP_HIP(ctx, hipMemcpy2DAsync(W.gray + (size_t)f * n, (size_t)w, (const uint8_t *)frames->data + (size_t)f * frames->frame_stride,
                                           frames->step, (size_t)w, (size_t)h, hipMemcpyDeviceToDevice, ctx->stream));
        k_ov_gray_to_L0<<<uwip_cdiv(n * F, 256), 256, 0, ctx->stream>>>(W.gray, W.L0, n * F);
        UWIP_HIP(ctx, hipGetLastError());
    }
    feats->w = w; feats->h = h;
    feats->frames = std::max(feats->frames, first_slot + F);
    return detect_describe(ctx, W, F, h, w, feats, first_slot, (flags & UWIP_OVERLAP_UPRIGHT) ? 1 : 0,
                           (flags & UWIP_OVERLAP_RELATIVE_THRESHOLD) ? 0 : 1);
}

// tap for tests: one slot's keypoints / packed descriptors to the host
UWIP_API int uwip_features_download(uwip_ctx *ctx, const uwip_features *feats, int slot, void *h_kps /*[2048] 32-byte records*/,
                                    uint8_t *h_desc /*[2048][64]*/, int32_t *h_count)
{
    if (!ctx || !feats) return UWIP_ERR_INVALID;
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    UWIP_REQUIRE(ctx, slot >= 0 && slot < feats->capacity, "slot out of range");
    UWIP_HIP(ctx, uwip_stream_wait(ctx));
    if (h_count) UWIP_HIP(ctx, hipMemcpy(h_count, feats->d_n + slot, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (h_kps) UWIP_HIP(ctx, hipMemcpy(h_kps, feats->d_kp + (size_t)slot * MAXKP, sizeof(Keypoint) * MAXKP, hipMemcpyDeviceToHost));
    if (h_desc) UWIP_HIP(ctx, hipMemcpy(h_desc, feats->d_desc + (size_t)slot * MAXKP * DESC_BYTES, (size_t)MAXKP * DESC_BYTES, hipMemcpyDeviceToHost));
    return UWIP_OK;
}

// The opposite direction: fill one slot from host keypoints / packed descriptors (the reference's `struct keyframe`
// has public keypoints / descriptors members that a caller may fill itself, videostrip.hpp:62-68); also how the
// matcher is measured on a full 2048 x 2048 descriptor set.  Rows >= count are zeroed.
namespace {
__global__ __launch_bounds__(256) void k_ov_unpack_desc(const uint8_t *__restrict__ desc, int count, int8_t *__restrict__ bits,
                                                       uint32_t *__restrict__ nib, int32_t *__restrict__ pop)
{
    const int k = blockIdx.x;                         // keypoint
    const int t = threadIdx.x;                        // 2 bits per thread -> 512
    const uint8_t *d = desc + (size_t)k * DESC_BYTES;
    const bool live = k < count;
    const int b0 = live ? (d[(2 * t) >> 3] >> ((2 * t) & 7)) & 1 : 0, b1 = live ? (d[(2 * t + 1) >> 3] >> ((2 * t + 1) & 7)) & 1 : 0;
    bits[(size_t)k * DESC_K + 2 * t] = (int8_t)b0;
    bits[(size_t)k * DESC_K + 2 * t + 1] = (int8_t)b1;
    if (t < DESC_NIBW) nib[(size_t)k * DESC_NIBW + t] = live ? desc_byte_to_nibbles(d[t]) : 0u;
    __shared__ int s_c[4];
    int c = b0 + b1;
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) c += __shfl_xor(c, sft, 64);
    if ((t & 63) == 0) s_c[t >> 6] = c;
    __syncthreads();
    if (t == 0) pop[k] = s_c[0] + s_c[1] + s_c[2] + s_c[3];
}
}  // namespace

UWIP_API int uwip_features_upload(uwip_ctx *ctx, uwip_features *feats, int slot, int rows, int cols, const void *h_kps,
                                  const uint8_t *h_desc, int32_t count)
{
    if (!ctx || !feats) return UWIP_ERR_INVALID;
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    UWIP_REQUIRE(ctx, feats->ctx == ctx, "feature set of another context");
    UWIP_REQUIRE(ctx, slot >= 0 && slot < feats->capacity, "slot out of range");
    UWIP_REQUIRE(ctx, count >= 0 && count <= MAXKP, "count out of range");
    UWIP_REQUIRE(ctx, rows > 0 && cols > 0 && (feats->w == 0 || (feats->w == cols && feats->h == rows)), "working size mismatch");
    UWIP_REQUIRE(ctx, count == 0 || (h_kps && h_desc), "null buffer");
    feats->w = cols; feats->h = rows;
    uint8_t *stage = (uint8_t *)uwip_ws(ctx, "ov.upload.desc", (size_t)MAXKP * DESC_BYTES);
    if (!stage) return UWIP_ERR_NOMEM;
    UWIP_HIP(ctx, uwip_stream_wait(ctx));
    if (count) {
        UWIP_HIP(ctx, hipMemcpy(stage, h_desc, (size_t)count * DESC_BYTES, hipMemcpyHostToDevice));
        UWIP_HIP(ctx, hipMemcpy(feats->d_kp + (size_t)slot * MAXKP, h_kps, sizeof(Keypoint) * (size_t)count, hipMemcpyHostToDevice));
        UWIP_HIP(ctx, hipMemcpy(feats->d_desc + (size_t)slot * MAXKP * DESC_BYTES, h_desc, (size_t)count * DESC_BYTES, hipMemcpyHostToDevice));
    }
    UWIP_HIP(ctx, hipMemcpy(feats->d_n + slot, &count, sizeof(int32_t), hipMemcpyHostToDevice));
    k_ov_unpack_desc<<<MAXKP, 256, 0, ctx->stream>>>(stage, count, feats->d_bits + (size_t)slot * MAXKP * DESC_K,
                                                     feats->d_nib + (size_t)slot * MAXKP * DESC_NIBW, feats->d_pop + (size_t)slot * MAXKP);
    UWIP_HIP(ctx, hipGetLastError());
    UWIP_HIP(ctx, uwip_stream_wait(ctx));
    return UWIP_OK;
}

// scale-space tap for tests: level images of slot-0 work buffers after the last detect call
UWIP_API int uwip_overlap_debug_level(uwip_ctx *ctx, int frame, int level, int rows, int cols, float *h_Lt, float *h_Lx,
                                      float *h_Ly, float *h_Ldet, float *h_kcontrast)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    UWIP_REQUIRE(ctx, level >= 0 && level < NLEV && frame >= 0, "bad level/frame");
    const size_t n = (size_t)rows * cols;
    auto get = [&](const char *name) -> float * {
        auto it = ctx->ws.find(name);
        return it == ctx->ws.end() ? nullptr : (float *)it->second.ptr;
    };
    float *Lt = get("ov.Lt"), *Lxy = get("ov.Lxy"), *Ld = get("ov.Ldet"), *kc = get("ov.kc");
    UWIP_REQUIRE(ctx, Lt && Lxy && Ld && kc, "no detect call yet");
    UWIP_HIP(ctx, uwip_stream_wait(ctx));
    UWIP_REQUIRE(ctx, frame < ctx->ov_last_frames, "frame beyond the last detect batch");
    const size_t off = ((size_t)level * ctx->ov_last_frames + frame) * n;
    if (h_Lt) UWIP_HIP(ctx, hipMemcpy(h_Lt, Lt + off, n * 4, hipMemcpyDeviceToHost));
    if (h_Lx || h_Ly) {        // the derivative pair is one interleaved plane on the device
        std::vector<float> xy(2 * n);
        UWIP_HIP(ctx, hipMemcpy(xy.data(), Lxy + 2 * off, n * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) {
            if (h_Lx) h_Lx[i] = xy[2 * i];
            if (h_Ly) h_Ly[i] = xy[2 * i + 1];
        }
    }
    if (h_Ldet) UWIP_HIP(ctx, hipMemcpy(h_Ldet, Ld + off, n * 4, hipMemcpyDeviceToHost));
    if (h_kcontrast) UWIP_HIP(ctx, hipMemcpy(h_kcontrast, kc + frame, 4, hipMemcpyDeviceToHost));
    return UWIP_OK;
}

// match query slots against train slots and turn each pair into an overlap ratio.
// h_pair_q / h_pair_t: host arrays of slot indices (query = object frame, train = key frame).
// d_ratio [npairs]: overlap ratio or -2.0 (videostrip.cpp:252-256,272).  d_info (may be NULL) [npairs][8]:
// nkp_obj, nkp_key, ngood, ninliers, overlap pixel count.  d_H (may be NULL) [npairs][9].
// d_match_idx / d_match_dist (may be NULL) [npairs][2048][2]: the kNN(2) result.
UWIP_API int uwip_overlap_match(uwip_ctx *ctx, const uwip_features *fq, const uwip_features *ft, const int32_t *h_pair_q,
                                const int32_t *h_pair_t, int npairs, int videoWidth, int videoHeight, uint32_t seed,
                                float *d_ratio, int32_t *d_info, double *d_H, int32_t *d_match_idx, int32_t *d_match_dist)
{
    return uwip_overlap_match_ex(ctx, fq, ft, h_pair_q, h_pair_t, npairs, videoWidth, videoHeight, seed, 0u, d_ratio, d_info, d_H,
                                 d_match_idx, d_match_dist);
}

UWIP_API int uwip_overlap_match_ex(uwip_ctx *ctx, const uwip_features *fq, const uwip_features *ft, const int32_t *h_pair_q,
                                   const int32_t *h_pair_t, int npairs, int videoWidth, int videoHeight, uint32_t seed, unsigned flags,
                                   float *d_ratio, int32_t *d_info, double *d_H, int32_t *d_match_idx, int32_t *d_match_dist)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    UWIP_REQUIRE(ctx, (flags & ~(unsigned)(UWIP_OVERLAP_MIN4 | UWIP_OVERLAP_MIN6)) == 0, "unknown flag");
    UWIP_REQUIRE(ctx, (flags & (UWIP_OVERLAP_MIN4 | UWIP_OVERLAP_MIN6)) != (UWIP_OVERLAP_MIN4 | UWIP_OVERLAP_MIN6), "UWIP_OVERLAP_MIN4 and UWIP_OVERLAP_MIN6 exclude each other");
    const int min_inliers = (flags & UWIP_OVERLAP_MIN6) ? MIN_INLIERS_STRICT : MIN_INLIERS;
    UWIP_REQUIRE(ctx, fq && ft && fq->ctx == ctx && ft->ctx == ctx, "bad feature sets");
    UWIP_REQUIRE(ctx, npairs >= 0 && npairs <= 65535, "npairs out of range");
    if (npairs == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, h_pair_q && h_pair_t && d_ratio, "null buffer");
    UWIP_REQUIRE(ctx, fq->w == ft->w && fq->h == ft->h && fq->w > 0, "feature sets of different working sizes");
    for (int p = 0; p < npairs; ++p)
        UWIP_REQUIRE(ctx, h_pair_q[p] >= 0 && h_pair_q[p] < fq->capacity && h_pair_t[p] >= 0 && h_pair_t[p] < ft->capacity, "pair slot out of range");
    int32_t *h_pairs = (int32_t *)uwip_host_ws(ctx, "ov.pairs", sizeof(int32_t) * 2 * (size_t)npairs);
    int32_t *d_pairs = (int32_t *)uwip_ws(ctx, "ov.pairs", sizeof(int32_t) * 2 * (size_t)npairs);
    int32_t *m_idx = d_match_idx ? d_match_idx : (int32_t *)uwip_ws(ctx, "ov.midx", sizeof(int32_t) * 2 * MAXKP * (size_t)npairs);
    int32_t *m_dist = d_match_dist ? d_match_dist : (int32_t *)uwip_ws(ctx, "ov.mdist", sizeof(int32_t) * 2 * MAXKP * (size_t)npairs);
    int32_t *info = d_info ? d_info : (int32_t *)uwip_ws(ctx, "ov.info", sizeof(int32_t) * 8 * (size_t)npairs);
    if (!h_pairs || !d_pairs || !m_idx || !m_dist || !info) return UWIP_ERR_NOMEM;
    // the same pair list as last time (every batch of a stream: frame i against frame i - 1) is already in d_pairs: nothing
    // to stage, and the host does not have to wait for the stream before reusing the pinned buffer
    const bool same = ctx->ov_pairs_dev == d_pairs && ctx->ov_pairs_host.size() == 2 * (size_t)npairs &&
                      memcmp(ctx->ov_pairs_host.data(), h_pair_q, sizeof(int32_t) * npairs) == 0 &&
                      memcmp(ctx->ov_pairs_host.data() + npairs, h_pair_t, sizeof(int32_t) * npairs) == 0;
    if (!same) {
        UWIP_HIP(ctx, uwip_stream_wait(ctx));          // pinned staging reuse
        memcpy(h_pairs, h_pair_q, sizeof(int32_t) * npairs);
        memcpy(h_pairs + npairs, h_pair_t, sizeof(int32_t) * npairs);
        UWIP_HIP(ctx, hipMemcpyAsync(d_pairs, h_pairs, sizeof(int32_t) * 2 * (size_t)npairs, hipMemcpyHostToDevice, ctx->stream));
        ctx->ov_pairs_host.assign(h_pairs, h_pairs + 2 * (size_t)npairs);
        ctx->ov_pairs_dev = d_pairs;
    }
    {
        uwip_kscope ks(ctx, "k_ov_match");
        constexpr int QT = 2;            // 2 query tiles of 16 per wave
        // UWIP_MATCH_FORM: 4 (default) FP4 operands; 3 the i8 form of round 4; 0 / 2 / 5 / 6 older and experimental shapes
        auto read_form = [] { const char *e = std::getenv("UWIP_MATCH_FORM"); return e && *e ? std::atoi(e) : 4; };
        static const int form_once = read_form();
        const int form = uwip_test_hooks() ? read_form() : form_once;      // tests switch forms inside one process
#define UWIP_LAUNCH_MATCH(KERNEL, NWV, LDSB)                                                                                   \
        do {                                                                                                                   \
            int rc_l = uwip_lds_optin(ctx, #KERNEL, (const void *)KERNEL, (LDSB));                                             \
            if (rc_l) return rc_l;                                                                                             \
            KERNEL<<<dim3(MAXKP / (16 * (NWV) * QT), npairs), 64 * (NWV), (LDSB), ctx->stream>>>(fq->d_bits, fq->d_pop, fq->d_n, ft->d_bits, \
                                                                                         ft->d_pop, ft->d_n, d_pairs, d_pairs + npairs, m_idx, m_dist); \
        } while (0)
        const size_t lds0 = (size_t)2 * 64 * MT_ROW, lds1 = lds0 + 2 * 64 * sizeof(uint32_t), lds2 = 2 * lds1;
#define UWIP_LAUNCH_MATCH_F4(KERNEL, NWV, LDSB)                                                                                \
        do {                                                                                                                   \
            int rc_l = uwip_lds_optin(ctx, #KERNEL, (const void *)KERNEL, (LDSB));                                             \
            if (rc_l) return rc_l;                                                                                             \
            KERNEL<<<dim3(MAXKP / (16 * (NWV) * QT), npairs), 64 * (NWV), (LDSB), ctx->stream>>>(fq->d_nib, fq->d_pop, fq->d_n, ft->d_nib, \
                                                                                         ft->d_pop, ft->d_n, d_pairs, d_pairs + npairs, m_idx, m_dist); \
        } while (0)
        const size_t ldsf4 = (size_t)2 * 64 * F4_ROW + 2 * 64 * sizeof(float), ldsf8 = (size_t)2 * 128 * F4_ROW + 2 * 128 * sizeof(float);
        switch (form) {                  // variants kept for A/B (tools/matcher_only.py): 0 = the round 2-3 kernel
        case 0: UWIP_LAUNCH_MATCH((k_ov_match<QT, 8>), 8, lds0); break;
        case 2: UWIP_LAUNCH_MATCH((k_ov_match_sp<QT, 8, 8>), 8, lds2); break;       // 128 train columns per barrier
        case 3: UWIP_LAUNCH_MATCH((k_ov_match_sp<QT, 8, 4>), 8, lds1); break;       // round 4's i8 form
        case 5: UWIP_LAUNCH_MATCH_F4((k_ov_match_f4<QT, 8, 8>), 8, ldsf8); break;   // FP4, 128 train columns per barrier
        case 6: UWIP_LAUNCH_MATCH_F4((k_ov_match_f4<QT, 4, 4>), 4, ldsf4); break;   // FP4, 4-wave blocks
        default: UWIP_LAUNCH_MATCH_F4((k_ov_match_f4<QT, 8, 4>), 8, ldsf4); break;  // FP4 operands (round 5)
        }
#undef UWIP_LAUNCH_MATCH_F4
#undef UWIP_LAUNCH_MATCH
        UWIP_HIP(ctx, hipGetLastError());
    }
    {
        uwip_kscope ks(ctx, "k_ov_geometry");
        const size_t lds = (size_t)MAXKP * 16 + MAXKP + (size_t)TH * MASK_WORDS * 4 + (size_t)TH * 4;
        int rc = uwip_lds_optin(ctx, "k_ov_geometry", (const void *)k_ov_geometry, lds);
        if (rc) return rc;
        k_ov_geometry<<<npairs, 256, lds, ctx->stream>>>(fq->d_kp, ft->d_kp, fq->d_n, ft->d_n, d_pairs, d_pairs + npairs, m_idx, m_dist,
                                                        fq->w, fq->h, videoWidth, videoHeight, seed, min_inliers, d_ratio, info, d_H);
        UWIP_HIP(ctx, hipGetLastError());
    }
    return UWIP_OK;
}

// overlapArea(Mat H), videostrip.cpp:291-319, for n homographies (device, row-major 3x3 doubles)
UWIP_API int uwip_overlapArea(uwip_ctx *ctx, const double *d_H, int n, int videoWidth, int videoHeight, float *d_ratio,
                              int32_t *d_count)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    UWIP_REQUIRE(ctx, n >= 0, "negative count");
    if (n == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, d_H && d_ratio, "null buffer");
    uwip_kscope ks(ctx, "k_ov_area_only");
    k_ov_area_only<<<n, 256, 0, ctx->stream>>>(d_H, videoWidth, videoHeight, d_ratio, d_count);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

// calcBlur(Mat frame), videostrip.cpp:170-184, per frame of a BGR batch ALREADY at the working size
// (the reference calls it on res_frame, main.cpp:338,355): d_blur [frames].
UWIP_API int uwip_calcBlur(uwip_ctx *ctx, const uwip_batch_u8 *frames, float *d_blur)
{
    int rc = uwip_check_batch(ctx, frames, 3);
    if (rc) return rc;
    if (frames->frames == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, !uwip_batch_empty(frames) && d_blur, "empty image or null output");
    const int F = frames->frames, h = frames->rows, w = frames->cols;
    const size_t n = (size_t)h * w;
    uint8_t *gray = (uint8_t *)uwip_ws(ctx, "blur.gray", n * F);
    float *L0 = (float *)uwip_ws(ctx, "blur.L0", n * F * 4);
    const int nb = 64;
    double *part = (double *)uwip_ws(ctx, "blur.part", sizeof(double) * 2 * nb * F);
    if (!gray || !L0 || !part) return UWIP_ERR_NOMEM;
    const uint8_t *tx = (const uint8_t *)resize_table(ctx, w, w), *ty = (const uint8_t *)resize_table(ctx, h, h);
    if (!tx || !ty) return UWIP_ERR_NOMEM;
    uwip_kscope ks(ctx, "k_ov_blur");
    // identity "resize" = the fused BGR2GRAY pass
    k_ov_resize_gray<<<grid2d(w, h, F), 256, 0, ctx->stream>>>((const uint8_t *)frames->data, frames->step, frames->frame_stride, h, w, h, w,
                                                              (const int *)tx, (const short *)(tx + (size_t)w * 4), (const short *)(tx + (size_t)w * 6),
                                                              (const int *)ty, (const short *)(ty + (size_t)h * 4), (const short *)(ty + (size_t)h * 6),
                                                              gray, L0);
    k_ov_blur<<<dim3(nb, F), 256, 0, ctx->stream>>>(gray, h, w, part);
    k_ov_blur_final<<<uwip_cdiv(F, 64), 64, 0, ctx->stream>>>(part, nb, (double)n, d_blur, F);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

// cv::resize(frame, res_frame, cv::Size(), hResizeFactor, hResizeFactor), main.cpp:242,287,311 (INTER_LINEAR, 8UC3):
// dst must have the size uwip_overlap_working_size gives for src.
UWIP_API int uwip_resize_bgr(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst)
{
    int rc = uwip_check_batch(ctx, src, 3);
    if (rc) return rc;
    rc = uwip_check_batch(ctx, dst, 3);
    if (rc) return rc;
    UWIP_REQUIRE(ctx, src->frames == dst->frames, "frame count mismatch");
    if (src->frames == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, !uwip_batch_empty(src), "empty image");
    int oh = 0, ow = 0;
    resize_dims(src->rows, src->cols, TW, &oh, &ow);
    UWIP_REQUIRE(ctx, dst->rows == oh && dst->cols == ow, "dst is not the working size of src (uwip_overlap_working_size)");
    const uint8_t *tx = (const uint8_t *)resize_table(ctx, src->cols, ow), *ty = (const uint8_t *)resize_table(ctx, src->rows, oh);
    if (!tx || !ty) return UWIP_ERR_NOMEM;
    uwip_kscope ks(ctx, "k_ov_resize_bgr");
    k_ov_resize_bgr<<<grid2d(ow, oh, src->frames), 256, 0, ctx->stream>>>((const uint8_t *)src->data, src->step, src->frame_stride, src->rows, src->cols, oh, ow,
                                                                        (const int *)tx, (const short *)(tx + (size_t)ow * 4), (const short *)(tx + (size_t)ow * 6),
                                                                        (const int *)ty, (const short *)(ty + (size_t)oh * 4), (const short *)(ty + (size_t)oh * 6),
                                                                        (uint8_t *)dst->data, dst->step, dst->frame_stride);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

// keep a frame's cached keypoints/descriptors (what `struct keyframe` holds) in another slot
UWIP_API int uwip_features_copy(uwip_ctx *ctx, const uwip_features *src, int src_slot, uwip_features *dst, int dst_slot)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    UWIP_REQUIRE(ctx, src && dst && src->ctx == ctx && dst->ctx == ctx, "bad feature sets");
    UWIP_REQUIRE(ctx, src_slot >= 0 && src_slot < src->capacity && dst_slot >= 0 && dst_slot < dst->capacity, "slot out of range");
    UWIP_REQUIRE(ctx, dst->w == 0 || (dst->w == src->w && dst->h == src->h), "feature sets of different working sizes");
    if (src == dst && src_slot == dst_slot) return UWIP_OK;
    dst->w = src->w; dst->h = src->h;
    const size_t s = (size_t)src_slot * MAXKP, d = (size_t)dst_slot * MAXKP;
    UWIP_HIP(ctx, hipMemcpyAsync(dst->d_kp + d, src->d_kp + s, sizeof(Keypoint) * MAXKP, hipMemcpyDeviceToDevice, ctx->stream));
    UWIP_HIP(ctx, hipMemcpyAsync(dst->d_desc + d * DESC_BYTES, src->d_desc + s * DESC_BYTES, (size_t)MAXKP * DESC_BYTES, hipMemcpyDeviceToDevice, ctx->stream));
    UWIP_HIP(ctx, hipMemcpyAsync(dst->d_bits + d * DESC_K, src->d_bits + s * DESC_K, (size_t)MAXKP * DESC_K, hipMemcpyDeviceToDevice, ctx->stream));
    UWIP_HIP(ctx, hipMemcpyAsync(dst->d_nib + d * DESC_NIBW, src->d_nib + s * DESC_NIBW, (size_t)MAXKP * DESC_NIBW * 4, hipMemcpyDeviceToDevice, ctx->stream));
    UWIP_HIP(ctx, hipMemcpyAsync(dst->d_pop + d, src->d_pop + s, sizeof(int32_t) * MAXKP, hipMemcpyDeviceToDevice, ctx->stream));
    UWIP_HIP(ctx, hipMemcpyAsync(dst->d_n + dst_slot, src->d_n + src_slot, sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
    return UWIP_OK;
}
