/*
 * uwip_oracle_color.c -- CPU restatement of the 8-bit colour conversions behind histretch's non-BGR letters
 * (modules/histretch/src/histretch.cpp:155-156,230-241: cv::cvtColor BGR2{HSV,HLS,Lab,YCrCb} and back).
 *
 * TEST INFRASTRUCTURE ONLY.  The arithmetic is OpenCV's imgproc/src/color.cpp (not in the reference tree, not in this
 * image): restated from the published OpenCV 3.x sources -- RGB2HLS_b / HLS2RGB_b (float kernels on x/255, hrange 180),
 * RGB2Lab_b (sRGB gamma + cube-root tables, lab_shift 12, gamma_shift 3) and Lab2RGB_b -> Lab2RGB_f with the
 * spline-interpolated inverse gamma table (the OpenCV 3.2 form; 3.4.x uses an integer inverse that can differ by one
 * level).  PARITY UNPINNED: no OpenCV and no reference fixture exists to check these against; the tests pin known
 * colours (tests/test_oracle_integer.py) and the device against this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

void orc_bgr_to_hsv_px(int b, int g, int r, int *h, int *s, int *v);
void orc_hsv_to_bgr_px(int h, int s, int v, uint8_t out[3]);
int orc_numChannel(char c);
int orc_numSpace(char c);
void orc_imgChannelStretch(uint8_t *data, int rows, int cols, size_t step, int pix, int lo, int hi);

static uint8_t sat_rne(float v)
{
    if (!(v < 2147483648.0f)) return 0;
    long r = lrintf(v);
    return (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
}
static int sat8i(int x) { return x < 0 ? 0 : (x > 255 ? 255 : x); }
static int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

static void bgr2hls(const uint8_t *p, uint8_t *o)
{
    const float b = p[0] * (1.f / 255.f), g = p[1] * (1.f / 255.f), r = p[2] * (1.f / 255.f);
    float h = 0.f, s = 0.f, l, vmax = r, vmin = r, diff;
    if (vmax < g) vmax = g;
    if (vmax < b) vmax = b;
    if (vmin > g) vmin = g;
    if (vmin > b) vmin = b;
    diff = vmax - vmin;
    l = (vmax + vmin) * 0.5f;
    if (diff > 1.1920929e-07f) {
        s = l < 0.5f ? diff / (vmax + vmin) : diff / (2 - vmax - vmin);
        diff = 60.f / diff;
        if (vmax == r) h = (g - b) * diff;
        else if (vmax == g) h = (b - r) * diff + 120.f;
        else h = (r - g) * diff + 240.f;
        if (h < 0.f) h += 360.f;
    }
    o[0] = sat_rne(h * 0.5f); o[1] = sat_rne(l * 255.f); o[2] = sat_rne(s * 255.f);
}

static void hls2bgr(const uint8_t *p, uint8_t *o)
{
    static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    float h = (float)p[0];
    const float l = p[1] * (1.f / 255.f), s = p[2] * (1.f / 255.f);
    float b, g, r;
    if (s == 0) {
        b = g = r = l;
    } else {
        float tab[4];
        const float p2 = l <= 0.5f ? l * (1 + s) : l + s - l * s;
        const float p1 = 2 * l - p2;
        int sector;
        h *= (6.f / 180.f);
        if (h < 0) do h += 6; while (h < 0);
        else if (h >= 6) do h -= 6; while (h >= 6);
        sector = (int)floorf(h);
        h -= (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
        tab[0] = p2; tab[1] = p1; tab[2] = p1 + (p2 - p1) * (1 - h); tab[3] = p1 + (p2 - p1) * h;
        b = tab[sector_data[sector][0]]; g = tab[sector_data[sector][1]]; r = tab[sector_data[sector][2]];
    }
    o[0] = sat_rne(b * 255.f); o[1] = sat_rne(g * 255.f); o[2] = sat_rne(r * 255.f);
}

static void bgr2ycc(const uint8_t *p, uint8_t *o)
{
    const int b = p[0], g = p[1], r = p[2];
    const int Y = descale(b * 1868 + g * 9617 + r * 4899, 14);
    o[1] = (uint8_t)sat8i(descale((r - Y) * 11682 + (128 << 14), 14));
    o[2] = (uint8_t)sat8i(descale((b - Y) * 9241 + (128 << 14), 14));
    o[0] = (uint8_t)sat8i(Y);
}
static void ycc2bgr(const uint8_t *p, uint8_t *o)
{
    const int Y = p[0], Cr = p[1], Cb = p[2];
    o[0] = (uint8_t)sat8i(Y + descale((Cb - 128) * 29049, 14));
    o[1] = (uint8_t)sat8i(Y + descale((Cb - 128) * -5636 + (Cr - 128) * -11698, 14));
    o[2] = (uint8_t)sat8i(Y + descale((Cr - 128) * 22987, 14));
}

/* ---- Lab tables (initLabTabs) ---- */
static uint16_t g_gamma[256], g_cbrt[3072];
static float g_invgamma[1024 * 4];
static int g_C[9];
static float g_K[9];
static int g_lab_ready = 0;
static void lab_init(void)
{
    static const float sRGB2XYZ_D65[9] = {0.412453f, 0.357580f, 0.180423f, 0.212671f, 0.715160f, 0.072169f, 0.019334f, 0.119193f, 0.950227f};
    static const float XYZ2sRGB_D65[9] = {3.240479f, -1.53715f, -0.498535f, -0.969256f, 1.875991f, 0.041556f, 0.055648f, -0.204043f, 1.057311f};
    static const float D65[3] = {0.950456f, 1.f, 1.088754f};
    if (g_lab_ready) return;
    for (int i = 0; i < 256; ++i) {
        const float x = i * (1.f / 255.f);
        const float v = 255.f * 8.f * (x <= 0.04045f ? x * (1.f / 12.92f) : (float)pow((double)(x + 0.055) * (1. / 1.055), 2.4));
        long q = lrintf(v);
        g_gamma[i] = (uint16_t)(q < 0 ? 0 : (q > 65535 ? 65535 : q));
    }
    for (int i = 0; i < 3072; ++i) {
        const float x = i * (1.f / (255.f * 8.f));
        const float v = 32768.f * (x < 0.008856f ? x * 7.787f + 0.13793103448275862f : cbrtf(x));
        long q = lrintf(v);
        g_cbrt[i] = (uint16_t)(q < 0 ? 0 : (q > 65535 ? 65535 : q));
    }
    {   /* splineBuild over x = i / 1024 of the inverse sRGB gamma */
        float f[1025], cn = 0;
        float *tab = g_invgamma;
        const int n = 1024;
        for (int i = 0; i <= n; ++i) {
            const float x = i * (1.f / 1024.f);
            f[i] = x <= 0.0031308f ? x * 12.92f : (float)(1.055 * pow((double)x, 1. / 2.4) - 0.055);
        }
        memset(tab, 0, sizeof g_invgamma);
        for (int i = 1; i < n - 1; ++i) {
            const float t = 3 * (f[i + 1] - 2 * f[i] + f[i - 1]);
            const float l = 1 / (4 - tab[(i - 1) * 4]);
            tab[i * 4] = l; tab[i * 4 + 1] = (t - tab[(i - 1) * 4 + 1]) * l;
        }
        for (int i = n - 1; i >= 0; --i) {
            const float c = tab[i * 4 + 1] - tab[i * 4] * cn;
            const float b = f[i + 1] - f[i] - (cn + c * 2) * 0.3333333333333333f;
            const float d = (cn - c) * 0.3333333333333333f;
            tab[i * 4] = f[i]; tab[i * 4 + 1] = b; tab[i * 4 + 2] = c; tab[i * 4 + 3] = d;
            cn = c;
        }
    }
    {
        const float scale[3] = {(float)(1 << 12) / D65[0], (float)(1 << 12), (float)(1 << 12) / D65[2]};
        for (int i = 0; i < 3; ++i) {
            g_C[i * 3 + 2] = (int)lrintf(sRGB2XYZ_D65[i * 3] * scale[i]);
            g_C[i * 3 + 1] = (int)lrintf(sRGB2XYZ_D65[i * 3 + 1] * scale[i]);
            g_C[i * 3 + 0] = (int)lrintf(sRGB2XYZ_D65[i * 3 + 2] * scale[i]);
            g_K[i] = XYZ2sRGB_D65[i] * D65[i];
            g_K[i + 3] = XYZ2sRGB_D65[i + 3] * D65[i];
            g_K[i + 6] = XYZ2sRGB_D65[i + 6] * D65[i];
        }
    }
    g_lab_ready = 1;
}
static void bgr2lab(const uint8_t *p, uint8_t *o)
{
    const int B = g_gamma[p[0]], G = g_gamma[p[1]], R = g_gamma[p[2]];
    const int fX = g_cbrt[descale(B * g_C[0] + G * g_C[1] + R * g_C[2], 12)];
    const int fY = g_cbrt[descale(B * g_C[3] + G * g_C[4] + R * g_C[5], 12)];
    const int fZ = g_cbrt[descale(B * g_C[6] + G * g_C[7] + R * g_C[8], 12)];
    const int Lscale = (116 * 255 + 50) / 100, Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    o[0] = (uint8_t)sat8i(descale(Lscale * fY + Lshift, 15));
    o[1] = (uint8_t)sat8i(descale(500 * (fX - fY) + 128 * (1 << 15), 15));
    o[2] = (uint8_t)sat8i(descale(200 * (fY - fZ) + 128 * (1 << 15), 15));
}
static float spline1024(float x)
{
    int ix = (int)x;
    ix = ix < 0 ? 0 : (ix > 1023 ? 1023 : ix);
    x -= (float)ix;
    const float *t = g_invgamma + ix * 4;
    return ((t[3] * x + t[2]) * x + t[1]) * x + t[0];
}
static float clip01(float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); }
static void lab2bgr(const uint8_t *p, uint8_t *o)
{
    const float li = p[0] * (100.f / 255.f), ai = (float)(p[1] - 128), bi = (float)(p[2] - 128);
    const float lThresh = 0.008856f * 903.3f, fThresh = 7.787f * 0.008856f + 16.0f / 116.0f;
    float y, fy, fxz[2];
    if (li <= lThresh) { y = li / 903.3f; fy = 7.787f * y + 16.0f / 116.0f; }
    else { fy = (li + 16.0f) / 116.0f; y = fy * fy * fy; }
    fxz[0] = ai / 500.0f + fy; fxz[1] = fy - bi / 200.0f;
    for (int j = 0; j < 2; ++j) {
        if (fxz[j] <= fThresh) fxz[j] = (fxz[j] - 16.0f / 116.0f) / 7.787f;
        else fxz[j] = fxz[j] * fxz[j] * fxz[j];
    }
    {
        const float x = fxz[0], z = fxz[1];
        float ro = clip01(g_K[0] * x + g_K[1] * y + g_K[2] * z);
        float go = clip01(g_K[3] * x + g_K[4] * y + g_K[5] * z);
        float bo = clip01(g_K[6] * x + g_K[7] * y + g_K[8] * z);
        ro = spline1024(ro * 1024.f); go = spline1024(go * 1024.f); bo = spline1024(bo * 1024.f);
        o[0] = sat_rne(bo * 255.f); o[1] = sat_rne(go * 255.f); o[2] = sat_rne(ro * 255.f);
    }
}

/* ---- Lab -> BGR, OpenCV 3.4.x: Lab2RGBinteger (imgproc/src/color_lab.cpp of 3.4.x, restated from memory -- PARITY UNPINNED:
 * the reference holds no fixture and OpenCV is not importable here).  INSTALL.md:47-63 pins 3.4.6, whose cvtColor takes this
 * integer path for 8-bit Lab2BGR; the float form above is what OpenCV 3.2 (the READMEs' version) does.
 *   y, ify      = LabToYF_b[L]                      (base 2^14: y = Y/Yn, ify = f(Y/Yn))
 *   adiv, bdiv  = (a - 128) / 500, (b - 128) / 200  in base 2^14 by multiply-shift
 *   x, z        = abToXZ_b[ify + adiv], abToXZ_b[ify - bdiv]   (cube, or the linear branch below 6/29)
 *   r, g, b     = descale(C . (x, y, z), 14) clamped to [0, 4095] -> sRGBInvGammaTab_b
 * Tables as initLabTabs builds them with softfloat (= correctly rounded float32 operations; cvRound = lrintf). */
static uint16_t g_yf[512], g_invgamma_b[4096];
static int g_Ki[9];
static int g_labi_ready = 0;
static void labi_init(void)
{
    static const double X2R[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};
    static const double W[3] = {0.950456, 1., 1.088754};
    const int BASE = 1 << 14;
    if (g_labi_ready) return;
    for (int i = 0; i < 256; ++i) {
        long y, ify;
        if (i <= 20) {
            y = lrintf((float)(i * BASE * 20 * 9) / (float)(17 * 29 * 29 * 29));
            ify = lrintf((float)BASE * ((float)16 / (float)116 + (float)(i * 5) / (float)(3 * 17 * 29)));
        } else {
            const float fy = (float)(i * 100 * BASE) / (float)(255 * 116) + (float)(16 * BASE) / (float)116;
            ify = lrintf(fy);
            y = lrintf(fy * fy * fy / (float)(BASE * BASE));
        }
        g_yf[i * 2] = (uint16_t)y; g_yf[i * 2 + 1] = (uint16_t)ify;
    }
    {
        const float thr = (float)7827 / (float)2500000, lowScale = (float)323 / (float)25, power = (float)12 / (float)5, xshift = (float)11 / (float)200;
        for (int i = 0; i < 4096; ++i) {
            const float x = (1.0f / 4096.0f) * (float)i;
            const float g = x <= thr ? x * lowScale : powf(x, 1.0f / power) * (1.0f + xshift) - xshift;
            g_invgamma_b[i] = (uint16_t)lrintf(255.0f * g);
        }
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) g_Ki[j * 3 + i] = (int)lrint(4096.0 * X2R[j * 3 + i] * W[i]);
    g_labi_ready = 1;
}
static int ab_to_xz(int i)
{
    const int BASE = 1 << 14;
    return i <= 3390 ? i * 108 / 841 - BASE * 16 / 116 * 108 / 841 : i * i / BASE * i / BASE;
}
static void lab2bgr_int(const uint8_t *p, uint8_t *o)
{
    const int BASE = 1 << 14;
    const int y = g_yf[p[0] * 2], ify = g_yf[p[0] * 2 + 1];
    const int adiv = ((5 * p[1] * 53687 + (1 << 7)) >> 13) - 128 * BASE / 500;
    const int bdiv = ((p[2] * 41943 + (1 << 4)) >> 9) - 128 * BASE / 200 + 1;
    const int x = ab_to_xz(ify + adiv), z = ab_to_xz(ify - bdiv);
    int ro = descale(g_Ki[0] * x + g_Ki[1] * y + g_Ki[2] * z, 14);
    int go = descale(g_Ki[3] * x + g_Ki[4] * y + g_Ki[5] * z, 14);
    int bo = descale(g_Ki[6] * x + g_Ki[7] * y + g_Ki[8] * z, 14);
    ro = ro < 0 ? 0 : (ro > 4095 ? 4095 : ro); go = go < 0 ? 0 : (go > 4095 ? 4095 : go); bo = bo < 0 ? 0 : (bo > 4095 ? 4095 : bo);
    o[0] = (uint8_t)sat8i(g_invgamma_b[bo]); o[1] = (uint8_t)sat8i(g_invgamma_b[go]); o[2] = (uint8_t)sat8i(g_invgamma_b[ro]);
}

/* cvtColor on an 8UC3 image: space 1 HSV, 2 HLS, 3 Lab, 4 YCrCb; dir 0 from BGR, 1 to BGR.  out may alias img.
 * rule: 0 = OpenCV 3.4.x (Lab -> BGR by the integer form), 1 = OpenCV 3.2 (the float form); only Lab's inverse differs. */
ORC_API int orc_cvt_space_ex(const uint8_t *img, int rows, int cols, size_t step, uint8_t *out, size_t ostep, int space, int dir, int rule);
ORC_API int orc_cvt_space(const uint8_t *img, int rows, int cols, size_t step, uint8_t *out, size_t ostep, int space, int dir)
{
    return orc_cvt_space_ex(img, rows, cols, step, out, ostep, space, dir, 0);
}
ORC_API int orc_cvt_space_ex(const uint8_t *img, int rows, int cols, size_t step, uint8_t *out, size_t ostep, int space, int dir, int rule)
{
    if (space < 1 || space > 4 || dir < 0 || dir > 1 || rule < 0 || rule > 1) return -1;
    if (space == 3) { lab_init(); labi_init(); }
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const uint8_t *p = img + (size_t)y * step + (size_t)x * 3;
            uint8_t o[3];
            if (space == 1) {
                if (dir == 0) { int h, s, v; orc_bgr_to_hsv_px(p[0], p[1], p[2], &h, &s, &v); o[0] = (uint8_t)h; o[1] = (uint8_t)s; o[2] = (uint8_t)v; }
                else orc_hsv_to_bgr_px(p[0], p[1], p[2], o);
            } else if (space == 2) { if (dir == 0) bgr2hls(p, o); else hls2bgr(p, o); }
            else if (space == 3) { if (dir == 0) bgr2lab(p, o); else if (rule == 0) lab2bgr_int(p, o); else lab2bgr(p, o); }
            else { if (dir == 0) bgr2ycc(p, o); else ycc2bgr(p, o); }
            uint8_t *q = out + (size_t)y * ostep + (size_t)x * 3;
            q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
        }
    return 0;
}

/* the per-letter loop of histretch.cpp:217-254 with every colour space; flags bit 0 (fixed order) = 0: as written (B-3, the
 * non-BGR letters leave the 8-bit colour round trip), 1: convert, stretch the letter's plane, merge, convert back; bit 1 =
 * OpenCV 3.2's float Lab inverse instead of 3.4.x's integer one (uwip.h UWIP_HISTRETCH_OPENCV32) */
ORC_API int orc_histretch_bgr_ex(uint8_t *img, int rows, int cols, size_t step, const char *letters, int lo, int hi, int flags)
{
    const int fixed_order = flags & 1, rule = (flags >> 1) & 1;
    uint8_t *tmp = (uint8_t *)malloc((size_t)rows * cols * 3 + 1);
    if (!tmp) return -1;
    for (const char *c = letters; *c; ++c) {
        const int ch = orc_numChannel(*c), sp = orc_numSpace(*c);
        if (sp == -1) continue;
        if (sp == 0) { orc_imgChannelStretch(img + ch, rows, cols, step, 3, lo, hi); continue; }
        orc_cvt_space_ex(img, rows, cols, step, tmp, (size_t)cols * 3, sp, 0, rule);
        if (fixed_order) orc_imgChannelStretch(tmp + ch, rows, cols, (size_t)cols * 3, 3, lo, hi);
        orc_cvt_space_ex(tmp, rows, cols, (size_t)cols * 3, img, step, sp, 1, rule);
    }
    free(tmp);
    return 0;
}
