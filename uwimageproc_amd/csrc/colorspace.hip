// 8-bit colour-space conversions behind histretch's non-BGR letters (modules/histretch/src/histretch.cpp:155-156,
// 230-241: cvtColor(BGR2xxx) ... cvtColor(xxx2BGR) for HSV, HLS, Lab, YCrCb) for gfx950.
//
// The arithmetic lives in OpenCV's imgproc/src/color.cpp, which is not in the reference tree and not in this image:
// every conversion below is a restatement of the 8-bit code path of OpenCV 3.x from its published sources
// (RGB2HSV_b / HSV2RGB_b, RGB2HLS_b / HLS2RGB_b, RGB2Lab_b / Lab2RGB_b with the sRGB gamma tables, RGB2YCrCb_i /
// YCrCb2RGB_i) -- "parity unpinned" (SURVEY.md 8c): checked against the CPU oracle's independent restatement and
// against hand-computed known answers (tests/test_oracle_integer.py, tests/test_histretch_gpu.py), not against a real
// OpenCV.  Lab follows the OpenCV 3.2 form the reference's READMEs name (integer forward with gamma / cube-root tables,
// float inverse with the spline-interpolated inverse gamma); OpenCV 3.4.x replaced the inverse by an integer LUT form
// that can differ by one level.
#include "uwip_internal.hpp"
#include "device_utils.hpp"
#include <cmath>
#include <cstring>
#include <vector>

namespace {

// ---- HSV (hrange 180): integer forward tables (hsv_shift 12), float inverse ---------------------------------
__device__ __forceinline__ void bgr2hsv_u8(int b, int g, int r, const int *__restrict__ sdiv, const int *__restrict__ hdiv, int &H, int &S, int &V)
{
    const int v = max(b, max(g, r)), vmin = min(b, min(g, r));
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    S = (diff * sdiv[v] + (1 << 11)) >> 12;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * hdiv[diff] + (1 << 11)) >> 12;
    h += h < 0 ? 180 : 0;
    H = (int)(uint8_t)h; S = (int)(uint8_t)S; V = v;
}
__device__ __forceinline__ void hsv2bgr_u8(int H, int S, int V, int &B, int &G, int &R)
{
    float hf = (float)H;
    const float sf = (float)S * (1.f / 255.f), vf = (float)V * (1.f / 255.f);
    float ob, og, orr;
    if (sf == 0.0f) {
        ob = og = orr = vf;
    } else {
        hf *= (6.f / 180.f);
        if (hf < 0) do hf += 6; while (hf < 0);
        else if (hf >= 6) do hf -= 6; while (hf >= 6);
        int sector = (int)floorf(hf);
        hf -= (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; hf = 0.f; }
        const float t0 = vf, t1 = vf * (1.f - sf), t2 = vf * (1.f - sf * hf), t3 = vf * (1.f - sf * (1.f - hf));
        switch (sector) {
            case 0: ob = t1; og = t3; orr = t0; break;
            case 1: ob = t1; og = t0; orr = t2; break;
            case 2: ob = t3; og = t0; orr = t1; break;
            case 3: ob = t0; og = t2; orr = t1; break;
            case 4: ob = t0; og = t1; orr = t3; break;
            default: ob = t2; og = t1; orr = t0; break;
        }
    }
    B = (int)sat_u8_rne(ob * 255.f); G = (int)sat_u8_rne(og * 255.f); R = (int)sat_u8_rne(orr * 255.f);
}

// ---- HLS (RGB2HLS_b / HLS2RGB_b: the float kernels on x/255, hrange 180) ------------------------------------
__device__ __forceinline__ void bgr2hls_u8(int bi, int gi, int ri, int &H, int &L, int &S)
{
    const float b = (float)bi * (1.f / 255.f), g = (float)gi * (1.f / 255.f), r = (float)ri * (1.f / 255.f);
    float h = 0.f, s = 0.f, l;
    float vmax = r, vmin = r;
    if (vmax < g) vmax = g;
    if (vmax < b) vmax = b;
    if (vmin > g) vmin = g;
    if (vmin > b) vmin = b;
    float diff = vmax - vmin;
    l = (vmax + vmin) * 0.5f;
    if (diff > 1.1920929e-07f) {
        s = l < 0.5f ? diff / (vmax + vmin) : diff / (2 - vmax - vmin);
        diff = 60.f / diff;
        if (vmax == r) h = (g - b) * diff;
        else if (vmax == g) h = (b - r) * diff + 120.f;
        else h = (r - g) * diff + 240.f;
        if (h < 0.f) h += 360.f;
    }
    H = (int)sat_u8_rne(h * 0.5f); L = (int)sat_u8_rne(l * 255.f); S = (int)sat_u8_rne(s * 255.f);
}
__device__ __forceinline__ void hls2bgr_u8(int H, int Li, int Si, int &B, int &G, int &R)
{
    float h = (float)H;
    const float l = (float)Li * (1.f / 255.f), s = (float)Si * (1.f / 255.f);
    float b, g, r;
    if (s == 0) {
        b = g = r = l;
    } else {
        const float p2 = l <= 0.5f ? l * (1 + s) : l + s - l * s;
        const float p1 = 2 * l - p2;
        h *= (6.f / 180.f);
        if (h < 0) do h += 6; while (h < 0);
        else if (h >= 6) do h -= 6; while (h >= 6);
        int sector = (int)floorf(h);
        h -= (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
        const float t0 = p2, t1 = p1, t2 = p1 + (p2 - p1) * (1 - h), t3 = p1 + (p2 - p1) * h;
        switch (sector) {                 // sector_data {1,3,0} {1,0,2} {3,0,1} {0,2,1} {0,1,3} {2,1,0}
            case 0: b = t1; g = t3; r = t0; break;
            case 1: b = t1; g = t0; r = t2; break;
            case 2: b = t3; g = t0; r = t1; break;
            case 3: b = t0; g = t2; r = t1; break;
            case 4: b = t0; g = t1; r = t3; break;
            default: b = t2; g = t1; r = t0; break;
        }
    }
    B = (int)sat_u8_rne(b * 255.f); G = (int)sat_u8_rne(g * 255.f); R = (int)sat_u8_rne(r * 255.f);
}

// ---- YCrCb (RGB2YCrCb_i / YCrCb2RGB_i, shift 14, delta 128) ---------------------------------------------------
__device__ __forceinline__ int descale14(int x) { return (x + (1 << 13)) >> 14; }
__device__ __forceinline__ int sat8(int x) { return min(max(x, 0), 255); }
__device__ __forceinline__ void bgr2ycc_u8(int b, int g, int r, int &Y, int &Cr, int &Cb)
{
    const int y = descale14(b * 1868 + g * 9617 + r * 4899);
    Cr = sat8(descale14((r - y) * 11682 + (128 << 14)));
    Cb = sat8(descale14((b - y) * 9241 + (128 << 14)));
    Y = sat8(y);
}
__device__ __forceinline__ void ycc2bgr_u8(int Y, int Cr, int Cb, int &B, int &G, int &R)
{
    B = sat8(Y + descale14((Cb - 128) * 29049));
    G = sat8(Y + descale14((Cb - 128) * -5636 + (Cr - 128) * -11698));
    R = sat8(Y + descale14((Cr - 128) * 22987));
}

// ---- Lab (RGB2Lab_b: gamma / cube-root tables, lab_shift 12, gamma_shift 3; Lab2RGB_b -> Lab2RGB_f + inverse gamma) ----
struct LabTabs {
    const uint16_t *gamma;      // sRGBGammaTab_b[256]
    const uint16_t *cbrt;       // LabCbrtTab_b[3072]
    const float *invgamma;      // sRGBInvGammaTab[1024 * 4] (cubic spline coefficients)
    int C[9];                   // forward coefficients for B, G, R order
    float K[9];                 // inverse coefficients: rows give R, G, B from (x, y, z)
    // OpenCV 3.4.x: Lab2RGBinteger (bit-exact integer inverse, the default of cv::cvtColor for 8-bit Lab since 3.4.0)
    const uint16_t *yf;         // LabToYF_b[256 * 2] = (y, ify) per L, base 2^14
    const uint16_t *invgamma_b; // sRGBInvGammaTab_b[4096]
    int Ki[9];                  // cvRound(2^12 * XYZ2sRGB_D65 * D65): rows give R, G, B
    int rule;                   // 0 = OpenCV 3.4.x (integer inverse), 1 = OpenCV 3.2 (float inverse through the spline)
};
__device__ __forceinline__ int descale_n(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ void bgr2lab_u8(int b, int g, int r, const LabTabs &T, int &L, int &A, int &Bq)
{
    const int Bl = T.gamma[b], Gl = T.gamma[g], Rl = T.gamma[r];
    const int fX = T.cbrt[descale_n(Bl * T.C[0] + Gl * T.C[1] + Rl * T.C[2], 12)];
    const int fY = T.cbrt[descale_n(Bl * T.C[3] + Gl * T.C[4] + Rl * T.C[5], 12)];
    const int fZ = T.cbrt[descale_n(Bl * T.C[6] + Gl * T.C[7] + Rl * T.C[8], 12)];
    const int Lscale = (116 * 255 + 50) / 100;
    const int Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    L = sat8(descale_n(Lscale * fY + Lshift, 15));
    A = sat8(descale_n(500 * (fX - fY) + 128 * (1 << 15), 15));
    Bq = sat8(descale_n(200 * (fY - fZ) + 128 * (1 << 15), 15));
}
__device__ __forceinline__ float spline1024(float x, const float *__restrict__ tab)
{
    int ix = min(max((int)x, 0), 1023);
    x -= (float)ix;
    const float *t = tab + ix * 4;
    return ((t[3] * x + t[2]) * x + t[1]) * x + t[0];
}
__device__ __forceinline__ float clip01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }
__device__ __forceinline__ void lab2bgr_u8(int Li, int Ai, int Bi, const LabTabs &T, int &B, int &G, int &R)
{
    const float li = (float)Li * (100.f / 255.f), ai = (float)(Ai - 128), bi = (float)(Bi - 128);
    const float lThresh = 0.008856f * 903.3f, fThresh = 7.787f * 0.008856f + 16.0f / 116.0f;
    float y, fy;
    if (li <= lThresh) { y = li / 903.3f; fy = 7.787f * y + 16.0f / 116.0f; }
    else { fy = (li + 16.0f) / 116.0f; y = fy * fy * fy; }
    float fxz[2] = {ai / 500.0f + fy, fy - bi / 200.0f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (fxz[j] <= fThresh) fxz[j] = (fxz[j] - 16.0f / 116.0f) / 7.787f;
        else fxz[j] = fxz[j] * fxz[j] * fxz[j];
    }
    const float x = fxz[0], z = fxz[1];
    float ro = clip01(T.K[0] * x + T.K[1] * y + T.K[2] * z);
    float go = clip01(T.K[3] * x + T.K[4] * y + T.K[5] * z);
    float bo = clip01(T.K[6] * x + T.K[7] * y + T.K[8] * z);
    ro = spline1024(ro * 1024.f, T.invgamma);
    go = spline1024(go * 1024.f, T.invgamma);
    bo = spline1024(bo * 1024.f, T.invgamma);
    B = (int)sat_u8_rne(bo * 255.f); G = (int)sat_u8_rne(go * 255.f); R = (int)sat_u8_rne(ro * 255.f);
}

// Lab2RGBinteger::process of OpenCV 3.4.x (imgproc/src/color_lab.cpp; restated, parity unpinned): y and fy from a table of
// L, fx = fy + a / 500 and fz = fy - b / 200 in base 2^14 through the multiply-shift forms OpenCV uses, the cube (or the
// linear branch below 6/29) in integer arithmetic -- abToXZ_b is a table there, computed here: its entries are two integer
// divisions -- the 3x3 matrix in 2^12 fixed point, descale by 14, the inverse sRGB gamma from a 4096-entry table.
__device__ __forceinline__ int lab_ab_to_xz(int i)
{
    constexpr int BASE = 1 << 14;
    // C division (truncation towards zero) as in initLabTabs: i may be negative on the linear branch
    return i <= 3390 ? i * 108 / 841 - BASE * 16 / 116 * 108 / 841 : i * i / BASE * i / BASE;
}
__device__ __forceinline__ void lab2bgr_int_u8(int LL, int aa, int bb, const LabTabs &T, int &B, int &G, int &R)
{
    constexpr int BASE = 1 << 14;
    const int y = T.yf[LL * 2], ify = T.yf[LL * 2 + 1];
    const int adiv = ((5 * aa * 53687 + (1 << 7)) >> 13) - 128 * BASE / 500;
    const int bdiv = ((bb * 41943 + (1 << 4)) >> 9) - 128 * BASE / 200 + 1;
    const int x = lab_ab_to_xz(ify + adiv), z = lab_ab_to_xz(ify - bdiv);
    int ro = descale_n(T.Ki[0] * x + T.Ki[1] * y + T.Ki[2] * z, 14);
    int go = descale_n(T.Ki[3] * x + T.Ki[4] * y + T.Ki[5] * z, 14);
    int bo = descale_n(T.Ki[6] * x + T.Ki[7] * y + T.Ki[8] * z, 14);
    ro = max(0, min(4095, ro)); go = max(0, min(4095, go)); bo = max(0, min(4095, bo));
    R = sat8(T.invgamma_b[ro]); G = sat8(T.invgamma_b[go]); B = sat8(T.invgamma_b[bo]);
}

// space: 1 HSV, 2 HLS, 3 Lab, 4 YCrCb (numSpace, preprocessing.cpp:155-160); dir: 0 BGR -> space, 1 space -> BGR,
// 2 the 8-bit round trip in one pass (the as-written letters)
template <int SPACE>
__global__ __launch_bounds__(256) void k_cvt_space(const uint8_t *__restrict__ src, size_t sstep, size_t sfs, uint8_t *__restrict__ dst,
                                                  size_t dstep, size_t dfs, int rows, int cols, int dir, const int *__restrict__ hsvtab,
                                                  LabTabs T)
{
    const int f = blockIdx.z, y = blockIdx.y;
    const uint8_t *s = src + (size_t)f * sfs + (size_t)y * sstep;
    uint8_t *d = dst + (size_t)f * dfs + (size_t)y * dstep;
    (void)rows;
    for (int x = blockIdx.x * 256 + threadIdx.x; x < cols; x += gridDim.x * 256) {
        int a = s[3 * x], b = s[3 * x + 1], c = s[3 * x + 2], p = 0, q = 0, r = 0;
        if (dir != 1) {           // forward
            if (SPACE == 1) bgr2hsv_u8(a, b, c, hsvtab, hsvtab + 256, p, q, r);
            if (SPACE == 2) bgr2hls_u8(a, b, c, p, q, r);
            if (SPACE == 3) bgr2lab_u8(a, b, c, T, p, q, r);
            if (SPACE == 4) bgr2ycc_u8(a, b, c, p, q, r);
            a = p; b = q; c = r;
        }
        if (dir != 0) {           // inverse
            if (SPACE == 1) hsv2bgr_u8(a, b, c, p, q, r);
            if (SPACE == 2) hls2bgr_u8(a, b, c, p, q, r);
            if (SPACE == 3) { if (T.rule == 0) lab2bgr_int_u8(a, b, c, T, p, q, r); else lab2bgr_u8(a, b, c, T, p, q, r); }
            if (SPACE == 4) ycc2bgr_u8(a, b, c, p, q, r);
            a = p; b = q; c = r;
        }
        d[3 * x] = (uint8_t)a; d[3 * x + 1] = (uint8_t)b; d[3 * x + 2] = (uint8_t)c;
    }
}

// OpenCV's splineBuild (natural cubic spline through n+1 samples -> n x 4 coefficients)
void spline_build(const std::vector<float> &f, int n, std::vector<float> &tab)
{
    tab.assign((size_t)n * 4, 0.f);
    float cn = 0;
    tab[0] = tab[1] = 0.f;
    for (int i = 1; i < n - 1; ++i) {
        const float t = 3 * (f[i + 1] - 2 * f[i] + f[i - 1]);
        const float l = 1 / (4 - tab[(size_t)(i - 1) * 4]);
        tab[(size_t)i * 4] = l;
        tab[(size_t)i * 4 + 1] = (t - tab[(size_t)(i - 1) * 4 + 1]) * l;
    }
    for (int i = n - 1; i >= 0; --i) {
        const float c = tab[(size_t)i * 4 + 1] - tab[(size_t)i * 4] * cn;
        const float b = f[i + 1] - f[i] - (cn + c * 2) * 0.3333333333333333f;
        const float d = (cn - c) * 0.3333333333333333f;
        tab[(size_t)i * 4] = f[i]; tab[(size_t)i * 4 + 1] = b; tab[(size_t)i * 4 + 2] = c; tab[(size_t)i * 4 + 3] = d;
        cn = c;
    }
}

// The two tables of Lab2RGBinteger (OpenCV 3.4.x initLabTabs).  OpenCV builds them with its softfloat class -- IEEE float32
// operations, each correctly rounded -- which plain float arithmetic is, this file being compiled without contraction or
// fast-math; cvRound = round half to even = lrintf; pow is libm's powf (softfloat's own pow may differ from it in the last
// place, which shows only where 255 * x lands within an ulp of a rounding tie).
void lab_int_tables(uint16_t *yf, uint16_t *ig)
{
    const int BASE = 1 << 14;
    for (int i = 0; i < 256; ++i) {
        long y, ify;
        if (i <= 20) {                                       // 8 * 255 / 100 = 20.4
            y = std::lrintf((float)(i * BASE * 20 * 9) / (float)(17 * 29 * 29 * 29));
            ify = std::lrintf((float)BASE * ((float)16 / (float)116 + (float)(i * 5) / (float)(3 * 17 * 29)));
        } else {
            const float fy = (float)(i * 100 * BASE) / (float)(255 * 116) + (float)(16 * BASE) / (float)116;
            ify = std::lrintf(fy);
            y = std::lrintf(fy * fy * fy / (float)(BASE * BASE));
        }
        yf[i * 2] = (uint16_t)y;
        yf[i * 2 + 1] = (uint16_t)ify;
    }
    const float thr = (float)7827 / (float)2500000, lowScale = (float)323 / (float)25, power = (float)12 / (float)5,
                xshift = (float)11 / (float)200;
    for (int i = 0; i < 4096; ++i) {
        const float x = (1.0f / 4096.0f) * (float)i;
        const float g = x <= thr ? x * lowScale : std::pow(x, 1.0f / power) * (1.0f + xshift) - xshift;
        ig[i] = (uint16_t)std::lrintf(255.0f * g);
    }
}

int lab_tables(uwip_ctx *ctx, LabTabs *T)
{
    static const float sRGB2XYZ_D65[9] = {0.412453f, 0.357580f, 0.180423f, 0.212671f, 0.715160f, 0.072169f, 0.019334f, 0.119193f, 0.950227f};
    static const float XYZ2sRGB_D65[9] = {3.240479f, -1.53715f, -0.498535f, -0.969256f, 1.875991f, 0.041556f, 0.055648f, -0.204043f, 1.057311f};
    static const float D65[3] = {0.950456f, 1.f, 1.088754f};
    const void *d = uwip_table_find(ctx, "lab.tables", nullptr);
    if (!d) {
        std::vector<uint8_t> buf(512 + 6144 + 16384 + 1024 + 8192);
        lab_int_tables((uint16_t *)(buf.data() + 512 + 6144 + 16384), (uint16_t *)(buf.data() + 512 + 6144 + 16384 + 1024));
        uint16_t *g = (uint16_t *)buf.data(), *cb = (uint16_t *)(buf.data() + 512);
        for (int i = 0; i < 256; ++i) {
            const float x = (float)i * (1.f / 255.f);
            const float v = 255.f * 8.f * (x <= 0.04045f ? x * (1.f / 12.92f) : (float)std::pow((double)(x + 0.055) * (1. / 1.055), 2.4));
            g[i] = (uint16_t)std::min(std::max(std::lrintf(v), 0L), 65535L);
        }
        for (int i = 0; i < 3072; ++i) {
            const float x = (float)i * (1.f / (255.f * 8.f));
            const float v = 32768.f * (x < 0.008856f ? x * 7.787f + 0.13793103448275862f : std::cbrt(x));   // cvCbrt is exact to 2^-24
            cb[i] = (uint16_t)std::min(std::max(std::lrintf(v), 0L), 65535L);
        }
        std::vector<float> ig(1025), tab;
        for (int i = 0; i <= 1024; ++i) {
            const float x = (float)i * (1.f / 1024.f);
            ig[i] = x <= 0.0031308f ? x * 12.92f : (float)(1.055 * std::pow((double)x, 1. / 2.4) - 0.055);
        }
        spline_build(ig, 1024, tab);
        std::memcpy(buf.data() + 512 + 6144, tab.data(), 16384);
        d = uwip_table_put(ctx, "lab.tables", buf.data(), buf.size());
        if (!d) return UWIP_ERR_NOMEM;
    }
    T->gamma = (const uint16_t *)d;
    T->cbrt = (const uint16_t *)((const uint8_t *)d + 512);
    T->invgamma = (const float *)((const uint8_t *)d + 512 + 6144);
    T->yf = (const uint16_t *)((const uint8_t *)d + 512 + 6144 + 16384);
    T->invgamma_b = (const uint16_t *)((const uint8_t *)d + 512 + 6144 + 16384 + 1024);
    {
        // cvRound(lshift * c * whitePt) in softdouble = double arithmetic here
        static const double X2R[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};
        static const double W[3] = {0.950456, 1., 1.088754};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) T->Ki[j * 3 + i] = (int)std::lrint(4096.0 * X2R[j * 3 + i] * W[i]);
    }
    const float scale[3] = {(float)(1 << 12) / D65[0], (float)(1 << 12), (float)(1 << 12) / D65[2]};
    for (int i = 0; i < 3; ++i) {           // pixel order B, G, R: blueIdx = 0
        T->C[i * 3 + 2] = (int)std::lrintf(sRGB2XYZ_D65[i * 3] * scale[i]);
        T->C[i * 3 + 1] = (int)std::lrintf(sRGB2XYZ_D65[i * 3 + 1] * scale[i]);
        T->C[i * 3 + 0] = (int)std::lrintf(sRGB2XYZ_D65[i * 3 + 2] * scale[i]);
    }
    for (int i = 0; i < 3; ++i) {           // K rows: R, G, B from (x, y, z)
        T->K[i] = XYZ2sRGB_D65[i] * D65[i];
        T->K[i + 3] = XYZ2sRGB_D65[i + 3] * D65[i];
        T->K[i + 6] = XYZ2sRGB_D65[i + 6] * D65[i];
    }
    return UWIP_OK;
}

const int *hsv_tables2(uwip_ctx *ctx)
{
    const void *d = uwip_table_find(ctx, "hsv.tables", nullptr);
    if (d) return (const int *)d;
    std::vector<int> t(512, 0);
    for (int i = 1; i < 256; ++i) {
        t[i] = (int)std::lrint((255 << 12) / (1. * i));          // sdiv_table
        t[256 + i] = (int)std::lrint((180 << 12) / (6. * i));    // hdiv_table180
    }
    return (const int *)uwip_table_put(ctx, "hsv.tables", t.data(), t.size() * sizeof(int));
}

}  // namespace

// cv::cvtColor(src, dst, COLOR_BGR2{HSV,HLS,Lab,YCrCb}) (dir 0), the matching ...2BGR (dir 1) or both in one pass
// (dir 2), 8UC3; space = numSpace's index 1..4.  dst may alias src.
int uwip_cvt_space_internal(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst, int space, int dir, int opencv_rule)
{
    UWIP_REQUIRE(ctx, opencv_rule == 0 || opencv_rule == 1, "opencv_rule must be 0 (OpenCV 3.4.x) or 1 (OpenCV 3.2)");
    UWIP_REQUIRE(ctx, space >= 1 && space <= 4 && dir >= 0 && dir <= 2, "bad colour space / direction");
    UWIP_REQUIRE(ctx, src->rows == dst->rows && src->cols == dst->cols && src->frames == dst->frames, "src/dst shape mismatch");
    if (uwip_batch_empty(src)) return UWIP_OK;
    UWIP_REQUIRE(ctx, src->rows <= 65535 && src->frames <= 65535, "too many rows/frames for one launch");
    LabTabs T{};
    const int *hsv = nullptr;
    if (space == 3) { int rc = lab_tables(ctx, &T); if (rc) return rc; T.rule = opencv_rule; }
    if (space == 1) { hsv = hsv_tables2(ctx); if (!hsv) return UWIP_ERR_NOMEM; }
    const dim3 grid(std::min(uwip_cdiv(src->cols, 256), 32u), (unsigned)src->rows, (unsigned)src->frames);
    uwip_kscope ks(ctx, "k_cvt_space");
#define UWIP_CVT(SP) k_cvt_space<SP><<<grid, 256, 0, ctx->stream>>>((const uint8_t *)src->data, src->step, src->frame_stride, (uint8_t *)dst->data, \
                                                                     dst->step, dst->frame_stride, src->rows, src->cols, dir, hsv, T)
    if (space == 1) UWIP_CVT(1);
    else if (space == 2) UWIP_CVT(2);
    else if (space == 3) UWIP_CVT(3);
    else UWIP_CVT(4);
#undef UWIP_CVT
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

UWIP_API int uwip_cvtColor_ex(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst, int space, int to_bgr, int opencv_rule)
{
    int rc = uwip_check_batch(ctx, src, 3);
    if (rc) return rc;
    rc = uwip_check_batch(ctx, dst, 3);
    if (rc) return rc;
    return uwip_cvt_space_internal(ctx, src, dst, space, to_bgr ? 1 : 0, opencv_rule);
}

UWIP_API int uwip_cvtColor(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst, int space, int to_bgr)
{
    return uwip_cvtColor_ex(ctx, src, dst, space, to_bgr, 0);
}
