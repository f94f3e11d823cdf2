/*
 * uwip_oracle_overlap.c -- CPU restatement of the videostrip overlap path
 * (SURVEY.md section 8a rows V1-V5).  TEST INFRASTRUCTURE ONLY (see the header
 * of uwip_oracle.c).
 *
 * What follows the reference and what does not:
 *   - calcOverlap's control flow, the ratio test with its skip-last quirk, the
 *     -1 / -2.0 sentinels and overlapArea follow
 *     modules/videostrip/src/videostrip.cpp:192-319.
 *   - The detector/descriptor/matcher are NOT the reference's (it calls
 *     OpenCV-contrib SURF + L2 BFMatcher, which are not in its tree).
 *     BASELINE.json's north_star asks for an AKAZE-style detector with binary
 *     descriptors and a brute-force Hamming matcher; this file is the
 *     specification of that design (DESIGN.md "overlap stage"): a 4-level
 *     nonlinear (Perona-Malik g2, FED-stepped) scale space, determinant-of-
 *     Hessian extrema with sub-pixel refinement, upright M-LDB 486-bit
 *     descriptors (Alcantarilla et al., BMVC 2013), Hamming top-2 match,
 *     a deterministic 512-hypothesis RANSAC homography.  There is no reference
 *     output to pin it to: "parity unpinned" by construction; the product is
 *     compared with THIS restatement (keypoints, descriptors and matches are
 *     exact-integer / bit-exact float comparisons, the ratio is compared
 *     within 0.01).
 *   - cv::resize(INTER_LINEAR, 8U), cvtColor(BGR2GRAY), perspectiveTransform,
 *     fillConvexPoly, contourArea restate OpenCV 3.4.x (absent): unpinned.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

#define OV_NLEVELS 4
#define OV_MAXKP 2048
#define OV_DESC_BYTES 64
#define OV_BORDER 8
#define OV_DTHRESH 0.001f
#define OV_KC_REF 0.5f      /* contrast factor at and above which the detector threshold is OV_DTHRESH itself */
#define OV_RANSAC_ITERS 512
/* A homography needs this many inliers to count.  The reference accepts whatever findHomography(RANSAC) returns for >= 4
 * matches (videostrip.cpp:252-272) -- but four matches always fit SOME homography exactly, and with the contrast-relative
 * detector threshold pure sensor noise yields a few dozen keypoints of which 4 can pass the ratio test by chance: such a
 * fit is reported as -2.0 ("no homography"), the sentinel main.cpp:321-326 turns into "do not trigger". */
#define OV_MIN_INLIERS 4          /* the reference's rule (videostrip.cpp:252-272); 6 = the opt-in strict rule */

typedef struct {
    float x, y;          /* refined position (pixels of the 640-wide image) */
    float response;      /* scale-normalised det of Hessian */
    int32_t level;       /* evolution level 0..3 */
    int32_t xi, yi;      /* integer extremum position */
    float co, si;        /* unit vector of the dominant orientation ((1, 0) for an upright descriptor) */
} orc_keypoint;

static const float OV_SIGMA[OV_NLEVELS] = {1.6f, 2.2627417f, 3.2f, 4.5254834f};
static const int OV_SSIZE[OV_NLEVELS] = {2, 3, 5, 7};          /* round(1.5 sigma) */

static inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ---------------- resize (INTER_LINEAR, 8UC3) + BGR2GRAY ------------------- */
/* OpenCV 3.4 resize for 8U: 11-bit fixed-point coefficients, horizontal pass in
 * int, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2. */
ORC_API void orc_resize_dims(int rows, int cols, int target_w, int *orows, int *ocols)
{
    float f = (float)target_w / (float)cols;                 /* hResizeFactor, main.cpp:242 */
    *ocols = (int)lrint((double)cols * (double)f);           /* Size() + fx: cvRound(cols*fx) */
    *orows = (int)lrint((double)rows * (double)f);
}

static void resize_tab(int ssize, int dsize, int *ofs, short *c0, short *c1)
{
    double scale = 1.0 / ((double)dsize / (double)ssize);
    for (int d = 0; d < dsize; ++d) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= ssize - 1) { fx = 0; sx = ssize - 1; }
        ofs[d] = sx;
        float a1 = fx * 2048.0f, a0 = (1.0f - fx) * 2048.0f;
        long r0 = lrintf(a0), r1 = lrintf(a1);
        c0[d] = (short)(r0 > 32767 ? 32767 : r0);
        c1[d] = (short)(r1 > 32767 ? 32767 : r1);
    }
}

ORC_API void orc_resize_gray(const uint8_t *bgr, int rows, int cols, size_t step, int orows, int ocols,
                             uint8_t *gray /* orows*ocols */, uint8_t *small_bgr /* optional orows*ocols*3 */)
{
    int *xo = (int *)malloc(sizeof(int) * ocols), *yo = (int *)malloc(sizeof(int) * orows);
    short *xa = (short *)malloc(sizeof(short) * ocols), *xb = (short *)malloc(sizeof(short) * ocols);
    short *ya = (short *)malloc(sizeof(short) * orows), *yb = (short *)malloc(sizeof(short) * orows);
    resize_tab(cols, ocols, xo, xa, xb);
    resize_tab(rows, orows, yo, ya, yb);
    for (int y = 0; y < orows; ++y) {
        const uint8_t *r0 = bgr + (size_t)yo[y] * step;
        const uint8_t *r1 = bgr + (size_t)(yo[y] + 1 < rows ? yo[y] + 1 : yo[y]) * step;
        for (int x = 0; x < ocols; ++x) {
            int sx = xo[x], sx1 = sx + 1 < cols ? sx + 1 : sx;
            int px[3];
            for (int c = 0; c < 3; ++c) {
                int S0 = r0[sx * 3 + c] * xa[x] + r0[sx1 * 3 + c] * xb[x];
                int S1 = r1[sx * 3 + c] * xa[x] + r1[sx1 * 3 + c] * xb[x];
                int v = (((ya[y] * (S0 >> 4)) >> 16) + ((yb[y] * (S1 >> 4)) >> 16) + 2) >> 2;
                px[c] = clampi(v, 0, 255);
            }
            if (small_bgr) {
                small_bgr[((size_t)y * ocols + x) * 3 + 0] = (uint8_t)px[0];
                small_bgr[((size_t)y * ocols + x) * 3 + 1] = (uint8_t)px[1];
                small_bgr[((size_t)y * ocols + x) * 3 + 2] = (uint8_t)px[2];
            }
            gray[(size_t)y * ocols + x] = (uint8_t)((px[0] * 1868 + px[1] * 9617 + px[2] * 4899 + 8192) >> 14);
        }
    }
    free(xo); free(yo); free(xa); free(xb); free(ya); free(yb);
}

/* ---------------- scale space ------------------------------------------------ */
static int gauss_kernel(float sigma, float *k /* >= 16 */)
{
    int ks = (int)ceil(2.0 * (1.0 + ((double)sigma - 0.8) / 0.3));
    if ((ks & 1) == 0) ks++;
    int r = ks / 2;
    double sum = 0, tmp[32];
    for (int i = 0; i < ks; ++i) { tmp[i] = exp(-((double)(i - r) * (i - r)) / (2.0 * (double)sigma * (double)sigma)); sum += tmp[i]; }
    for (int i = 0; i < ks; ++i) k[i] = (float)(tmp[i] / sum);
    return ks;
}

static void gauss(const float *in, float *out, float *tmp, int h, int w, float sigma)
{
    float k[32];
    int ks = gauss_kernel(sigma, k), r = ks / 2;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float acc = 0.0f;
            for (int i = 0; i < ks; ++i) acc = acc + k[i] * in[(size_t)y * w + reflect101(x + i - r, w)];
            tmp[(size_t)y * w + x] = acc;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float acc = 0.0f;
            for (int i = 0; i < ks; ++i) acc = acc + k[i] * tmp[(size_t)reflect101(y + i - r, h) * w + x];
            out[(size_t)y * w + x] = acc;
        }
}

static inline void scharr_at(const float *I, int h, int w, int y, int x, float *gx, float *gy)
{
    int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h), xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
    float a0 = I[(size_t)ym * w + xm], a1 = I[(size_t)ym * w + x], a2 = I[(size_t)ym * w + xp];
    float b0 = I[(size_t)y * w + xm], b2 = I[(size_t)y * w + xp];
    float c0 = I[(size_t)yp * w + xm], c1 = I[(size_t)yp * w + x], c2 = I[(size_t)yp * w + xp];
    float t0 = 3.0f * (a2 - a0), t1 = 10.0f * (b2 - b0), t2 = 3.0f * (c2 - c0);
    *gx = (t0 + t1) + t2;
    t0 = 3.0f * (c0 - a0); t1 = 10.0f * (c1 - a1); t2 = 3.0f * (c2 - a2);
    *gy = (t0 + t1) + t2;
}

/* 70th percentile of the gradient magnitude histogram (AKAZE compute_k_percentile) */
static float k_contrast(const float *Lsm, int h, int w)
{
    float hmax = 0.0f;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            float gx, gy;
            scharr_at(Lsm, h, w, y, x, &gx, &gy);
            float m = sqrtf(gx * gx + gy * gy);
            if (m > hmax) hmax = m;
        }
    if (hmax == 0.0f) return 0.03f;
    int hist[300];
    memset(hist, 0, sizeof hist);
    int npoints = 0;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            float gx, gy;
            scharr_at(Lsm, h, w, y, x, &gx, &gy);
            float m = sqrtf(gx * gx + gy * gy);
            if (m != 0.0f) {
                int nbin = (int)floorf(300.0f * (m / hmax));
                if (nbin >= 300) nbin = 299;
                hist[nbin]++;
                npoints++;
            }
        }
    int nthreshold = (int)((float)npoints * 0.7f);
    int k = 0, nelements = 0;
    for (k = 0; nelements < nthreshold && k < 300; k++) nelements += hist[k];
    if (nelements < nthreshold) return 0.03f;
    return hmax * ((float)k / 300.0f);
}

/* FED step sizes for process time T (tau_max = 0.25), natural order */
static int fed_taus(float T, float *tau /* >= 32 */)
{
    const double tau_max = 0.25;
    int n = (int)(ceil(sqrt(3.0 * (double)T / tau_max + 0.25) - 0.5 - 1.0e-8) + 0.5);
    if (n < 1) n = 1;
    double scale = 3.0 * (double)T / (tau_max * (double)(n * (n + 1)));
    double c = 1.0 / (4.0 * (double)n + 2.0), d = scale * tau_max / 2.0;
    for (int k = 0; k < n; ++k) {
        double hh = cos(3.14159265358979323846 * (2.0 * (double)k + 1.0) * c);
        tau[k] = (float)(d / (hh * hh));
    }
    return n;
}

static void fed_step(const float *L, const float *c, float *out, int h, int w, float tau)
{
    const float step = 0.5f * tau;
    for (int y = 0; y < h; ++y) {
        int ym = y > 0 ? y - 1 : 0, yp = y < h - 1 ? y + 1 : h - 1;
        for (int x = 0; x < w; ++x) {
            int xm = x > 0 ? x - 1 : 0, xp = x < w - 1 ? x + 1 : w - 1;
            size_t i = (size_t)y * w + x;
            float xpos = (c[i] + c[(size_t)y * w + xp]) * (L[(size_t)y * w + xp] - L[i]);
            float xneg = (c[(size_t)y * w + xm] + c[i]) * (L[i] - L[(size_t)y * w + xm]);
            float ypos = (c[i] + c[(size_t)yp * w + x]) * (L[(size_t)yp * w + x] - L[i]);
            float yneg = (c[(size_t)ym * w + x] + c[i]) * (L[i] - L[(size_t)ym * w + x]);
            float d = xpos - xneg;
            d = d + ypos;
            d = d - yneg;
            out[i] = L[i] + step * d;
        }
    }
}

/* scale-s Scharr-like first derivative (taps at -s, 0, +s) */
static void deriv(const float *I, float *out, int h, int w, int s, int along_x)
{
    const float wgt = 10.0f / 3.0f;
    const float norm = 1.0f / (2.0f * (float)s * (wgt + 2.0f));
    const float wn = wgt * norm;
    for (int y = 0; y < h; ++y) {
        int ym = reflect101(y - s, h), yp = reflect101(y + s, h);
        for (int x = 0; x < w; ++x) {
            int xm = reflect101(x - s, w), xp = reflect101(x + s, w);
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
            out[(size_t)y * w + x] = (t0 + t1) + t2;
        }
    }
}

typedef struct {
    int h, w;
    float *Lt[OV_NLEVELS], *Lx[OV_NLEVELS], *Ly[OV_NLEVELS], *Ldet[OV_NLEVELS];
    float kcontrast;
} scale_space;

static void ss_free(scale_space *s)
{
    for (int i = 0; i < OV_NLEVELS; ++i) { free(s->Lt[i]); free(s->Lx[i]); free(s->Ly[i]); free(s->Ldet[i]); }
}

static void build_scale_space(const uint8_t *gray, int h, int w, scale_space *S)
{
    size_t n = (size_t)h * w;
    S->h = h; S->w = w;
    for (int i = 0; i < OV_NLEVELS; ++i) {
        S->Lt[i] = (float *)malloc(n * 4); S->Lx[i] = (float *)malloc(n * 4);
        S->Ly[i] = (float *)malloc(n * 4); S->Ldet[i] = (float *)malloc(n * 4);
    }
    float *L0 = (float *)malloc(n * 4), *tmp = (float *)malloc(n * 4), *Lsm = (float *)malloc(n * 4);
    float *flow = (float *)malloc(n * 4), *ping = (float *)malloc(n * 4);
    float *Lxx = (float *)malloc(n * 4), *Lyy = (float *)malloc(n * 4), *Lxy = (float *)malloc(n * 4);
    for (size_t i = 0; i < n; ++i) L0[i] = (float)gray[i] / 255.0f;
    gauss(L0, S->Lt[0], tmp, h, w, OV_SIGMA[0]);
    for (int lv = 0; lv < OV_NLEVELS; ++lv) {
        /* smoothed copy of this level: feeds the Hessian of this level and the flow of the next */
        gauss(S->Lt[lv], Lsm, tmp, h, w, 1.0f);
        if (lv == 0) S->kcontrast = k_contrast(Lsm, h, w);
        const int s = OV_SSIZE[lv];
        deriv(Lsm, S->Lx[lv], h, w, s, 1);
        deriv(Lsm, S->Ly[lv], h, w, s, 0);
        deriv(S->Lx[lv], Lxx, h, w, s, 1);
        deriv(S->Ly[lv], Lyy, h, w, s, 0);
        deriv(S->Lx[lv], Lxy, h, w, s, 0);
        const float ss = (float)(s * s), s4 = ss * ss;
        for (size_t i = 0; i < n; ++i) S->Ldet[lv][i] = (Lxx[i] * Lyy[i] - Lxy[i] * Lxy[i]) * s4;
        if (lv + 1 < OV_NLEVELS) {
            const float inv_k = 1.0f / (S->kcontrast * S->kcontrast);
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    float gx, gy;
                    scharr_at(Lsm, h, w, y, x, &gx, &gy);
                    flow[(size_t)y * w + x] = 1.0f / (1.0f + (gx * gx + gy * gy) * inv_k);
                }
            float e0 = 0.5f * OV_SIGMA[lv] * OV_SIGMA[lv], e1 = 0.5f * OV_SIGMA[lv + 1] * OV_SIGMA[lv + 1];
            float taus[32];
            int nt = fed_taus(e1 - e0, taus);
            const float *cur = S->Lt[lv];
            float *bufs[2] = {S->Lt[lv + 1], ping};
            /* arrange so that the last step lands in Lt[lv+1] */
            int dst = (nt & 1) ? 0 : 1;
            for (int k = 0; k < nt; ++k) {
                fed_step(cur, flow, bufs[dst], h, w, taus[k]);
                cur = bufs[dst];
                dst ^= 1;
            }
        }
    }
    free(L0); free(tmp); free(Lsm); free(flow); free(ping); free(Lxx); free(Lyy); free(Lxy);
}

/* ---------------- detector --------------------------------------------------- */
/* The determinant-of-Hessian response scales with the SQUARE of the image contrast, and raw frames of turbid water have
 * little of it (the reference's photograph PIS_T1A_259: grey levels 81..146, strongest response 4.7e-5 -- no keypoint at
 * all above a fixed 1e-3, so calcOverlap answers -2.0 for every frame and the selector never moves on).  The threshold
 * is therefore taken relative to the frame's own contrast factor k (the 70th percentile of the gradient magnitude, already
 * computed for the diffusion): OV_DTHRESH * min(1, (k / OV_KC_REF)^2).  A frame as contrasted as the reference's
 * BUL_T1A_0028 (k = 0.51) keeps 1e-3; PIS_T1A_259 (k = 0.041) gets 6.7e-6 and 130 keypoints, and its overlap under yaw and
 * zoom is found within 0.004 (tests).  The default is the fixed threshold (SURF's hessianThreshold is fixed too); flags bit 4
 * (16, uwip.h UWIP_OVERLAP_RELATIVE_THRESHOLD) selects the relative one, bit 1 is accepted and names the default. */
static int detect(const scale_space *S, orc_keypoint *kps /* OV_MAXKP */, int flags)
{
    const int h = S->h, w = S->w;
    const float kr = S->kcontrast / OV_KC_REF;
    float ks = kr * kr;
    if (!(ks < 1.0f)) ks = 1.0f;
    const float dthr = (flags & 16) ? OV_DTHRESH * ks : OV_DTHRESH;
    size_t cap = 65536, cnt = 0;
    orc_keypoint *all = (orc_keypoint *)malloc(cap * sizeof(orc_keypoint));
    for (int lv = 0; lv < OV_NLEVELS; ++lv) {
        const float *D = S->Ldet[lv];
        for (int y = OV_BORDER; y < h - OV_BORDER; ++y)
            for (int x = OV_BORDER; x < w - OV_BORDER; ++x) {
                const float v = D[(size_t)y * w + x];
                if (!(v > dthr)) continue;
                int ok = 1;
                for (int dy = -1; dy <= 1 && ok; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        if (dx == 0 && dy == 0) continue;
                        if (!(v > D[(size_t)(y + dy) * w + x + dx])) { ok = 0; break; }
                    }
                for (int o = -1; o <= 1 && ok; o += 2) {
                    int l2 = lv + o;
                    if (l2 < 0 || l2 >= OV_NLEVELS) continue;
                    const float *E = S->Ldet[l2];
                    for (int dy = -1; dy <= 1 && ok; ++dy)
                        for (int dx = -1; dx <= 1; ++dx)
                            if (!(v > E[(size_t)(y + dy) * w + x + dx])) { ok = 0; break; }
                }
                if (!ok) continue;
                /* 2-D quadratic refinement */
                const float vxp = D[(size_t)y * w + x + 1], vxm = D[(size_t)y * w + x - 1];
                const float vyp = D[(size_t)(y + 1) * w + x], vym = D[(size_t)(y - 1) * w + x];
                const float Dx = 0.5f * (vxp - vxm), Dy = 0.5f * (vyp - vym);
                const float Dxx = (vxp + vxm) - 2.0f * v, Dyy = (vyp + vym) - 2.0f * v;
                const float Dxy = 0.25f * (D[(size_t)(y + 1) * w + x + 1] + D[(size_t)(y - 1) * w + x - 1]) -
                                  0.25f * (D[(size_t)(y + 1) * w + x - 1] + D[(size_t)(y - 1) * w + x + 1]);
                const float det = Dxx * Dyy - Dxy * Dxy;
                if (det == 0.0f) continue;
                const float ox = -(Dyy * Dx - Dxy * Dy) / det, oy = -(Dxx * Dy - Dxy * Dx) / det;
                if (!(fabsf(ox) <= 1.0f && fabsf(oy) <= 1.0f)) continue;
                if (cnt == cap) { cap *= 2; all = (orc_keypoint *)realloc(all, cap * sizeof(orc_keypoint)); }
                orc_keypoint k;
                memset(&k, 0, sizeof k);
                k.x = (float)x + ox; k.y = (float)y + oy; k.response = v; k.level = lv; k.xi = x; k.yi = y;
                all[cnt++] = k;
            }
    }
    /* keep the OV_MAXKP strongest; output stays in (level, y, x) raster order */
    size_t nout = 0;
    if (cnt <= OV_MAXKP) {
        memcpy(kps, all, cnt * sizeof(orc_keypoint));
        nout = cnt;
    } else {
        uint32_t *bits = (uint32_t *)malloc(cnt * 4);
        for (size_t i = 0; i < cnt; ++i) memcpy(&bits[i], &all[i].response, 4);
        /* K-th largest by 2-pass radix select on the (positive) float bit patterns */
        uint32_t prefix = 0;
        size_t remaining = OV_MAXKP;
        for (int pass = 0; pass < 2; ++pass) {
            static uint32_t hist[65536];
            memset(hist, 0, sizeof hist);
            for (size_t i = 0; i < cnt; ++i) {
                if (pass == 1 && (bits[i] >> 16) != prefix) continue;
                hist[pass == 0 ? (bits[i] >> 16) : (bits[i] & 0xffff)]++;
            }
            int b = 65535;
            size_t acc = 0;
            for (; b >= 0; --b) {
                if (acc + hist[b] >= remaining) break;
                acc += hist[b];
            }
            remaining -= acc;
            if (pass == 0) prefix = (uint32_t)b; else prefix = (prefix << 16) | (uint32_t)b;
        }
        const uint32_t thr = prefix;           /* response bits of the K-th strongest */
        for (size_t i = 0; i < cnt && nout < OV_MAXKP; ++i)
            if (bits[i] >= thr) kps[nout++] = all[i];
        free(bits);
    }
    free(all);
    return (int)nout;
}

/* ---------------- dominant orientation ------------------------------------------------
 * The reference's detector is oriented SURF (SURF::create(400), upright = false,
 * modules/videostrip/src/videostrip.cpp:206-208): an ROV yaws, so the replacement must not lose rotation invariance.
 * AKAZE's estimate (Alcantarilla et al.): the scale-s first derivatives at the 109 lattice points of a radius-6 disc
 * around the keypoint, Gaussian weighted (sigma 2.5), are summed inside a pi/3 sector that slides in steps of 0.15 rad;
 * the sector with the longest sum gives the direction.  Stated here WITHOUT any angle: membership of a vector in a
 * sector is two cross products against the sector's boundary unit vectors (tables below), and the result is the unit
 * vector (co, si) itself -- only +, *, /, sqrt, all correctly rounded on both sides, so the device matches bit for bit.
 * Sums run over the samples in raster order (i = x offset outer, j = y offset inner). */
/* ORIENT-TABLES-BEGIN (generated by tools/gen_orient_tables.py, identical text in csrc/overlap.hip) */
static const float OV_GAUSS25[7][7] = {
    {1.0f, 0.923116326f, 0.726149023f, 0.486752242f, 0.27803731f, 0.135335281f, 0.0561347641f},
    {0.923116326f, 0.852143764f, 0.670320034f, 0.449328959f, 0.256660789f, 0.12493021f, 0.0518189184f},
    {0.726149023f, 0.670320034f, 0.52729243f, 0.353454679f, 0.201896518f, 0.0982735828f, 0.0407622047f},
    {0.486752242f, 0.449328959f, 0.353454679f, 0.236927763f, 0.135335281f, 0.0658747554f, 0.0273237228f},
    {0.27803731f, 0.256660789f, 0.201896518f, 0.135335281f, 0.0773047432f, 0.0376282558f, 0.0156075582f},
    {0.135335281f, 0.12493021f, 0.0982735828f, 0.0658747554f, 0.0376282558f, 0.0183156393f, 0.00759701384f},
    {0.0561347641f, 0.0518189184f, 0.0407622047f, 0.0273237228f, 0.0156075582f, 0.00759701384f, 0.00315111154f},
};
/* sector k: [a_k, a_k + pi/3), a_k = 0.15 k; {cos a_k, sin a_k, cos(a_k + pi/3), sin(a_k + pi/3)} */
static const float OV_SECTOR[42][4] = {
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
/* ORIENT-TABLES-END */

static void orient(const scale_space *S, orc_keypoint *k)
{
    const int h = S->h, w = S->w;
    const float *Lx = S->Lx[k->level], *Ly = S->Ly[k->level];
    const float sc = (float)OV_SSIZE[k->level];
    float vx[109], vy[109];
    int t = 0;
    for (int i = -6; i <= 6; ++i)
        for (int j = -6; j <= 6; ++j) {
            if (i * i + j * j >= 36) continue;
            const int x1 = clampi((int)floorf(k->x + (float)i * sc + 0.5f), 0, w - 1);
            const int y1 = clampi((int)floorf(k->y + (float)j * sc + 0.5f), 0, h - 1);
            const float g = OV_GAUSS25[abs(i)][abs(j)];
            vx[t] = g * Lx[(size_t)y1 * w + x1];
            vy[t] = g * Ly[(size_t)y1 * w + x1];
            t++;
        }
    float best = 0.0f, bx = 0.0f, by = 0.0f;
    for (int s = 0; s < 42; ++s) {
        const float *d = OV_SECTOR[s];
        float sx = 0.0f, sy = 0.0f;
        for (int q = 0; q < 109; ++q) {
            const float c1 = d[0] * vy[q] - d[1] * vx[q], c2 = d[2] * vy[q] - d[3] * vx[q];
            if (c1 >= 0.0f && c2 < 0.0f) { sx = sx + vx[q]; sy = sy + vy[q]; }
        }
        const float m = sx * sx + sy * sy;
        if (m > best) { best = m; bx = sx; by = sy; }       /* first of equal maxima */
    }
    if (best > 0.0f) {
        const float nrm = sqrtf(best);
        k->co = bx / nrm; k->si = by / nrm;
    } else {
        k->co = 1.0f; k->si = 0.0f;
    }
}

/* ---------------- M-LDB descriptor (486 bits, zero padded to 512) ------------------------
 * The 21 x 21 sample lattice and the two derivative channels are rotated by the keypoint's (co, si); with (1, 0) --
 * upright -- every expression below reduces exactly to the unrotated one. */
static void describe(const scale_space *S, const orc_keypoint *kps, int n, uint8_t *desc /* n*64 */)
{
    static const int steps[3] = {10, 7, 5}, ncell1[3] = {2, 3, 4};
    const int h = S->h, w = S->w;
    memset(desc, 0, (size_t)n * OV_DESC_BYTES);
    for (int q = 0; q < n; ++q) {
        const orc_keypoint *k = &kps[q];
        const float *Lt = S->Lt[k->level], *Lx = S->Lx[k->level], *Ly = S->Ly[k->level];
        const float sc = (float)OV_SSIZE[k->level];
        int bit = 0;
        uint8_t *d = desc + (size_t)q * OV_DESC_BYTES;
        for (int z = 0; z < 3; ++z) {
            const int st = steps[z], nc = ncell1[z];
            float val[16][3];
            int ci = 0;
            for (int i = -10; i < 10; i += st)
                for (int j = -10; j < 10; j += st) {
                    float di = 0.0f, dx = 0.0f, dy = 0.0f;
                    int ns = 0;
                    for (int kk = i; kk < i + st; ++kk)
                        for (int l = j; l < j + st; ++l) {
                            const float u = (float)kk * sc, v = (float)l * sc;
                            const float sy = k->y + (u * k->si + v * k->co), sx = k->x + (u * k->co - v * k->si);
                            const int y1 = clampi((int)floorf(sy + 0.5f), 0, h - 1);
                            const int x1 = clampi((int)floorf(sx + 0.5f), 0, w - 1);
                            const float rx = Lx[(size_t)y1 * w + x1], ry = Ly[(size_t)y1 * w + x1];
                            di = di + Lt[(size_t)y1 * w + x1];
                            dx = dx + (rx * k->co + ry * k->si);          /* derivative along the keypoint's own x axis */
                            dy = dy + (ry * k->co - rx * k->si);          /* ... and y axis */
                            ns++;
                        }
                    val[ci][0] = di / (float)ns; val[ci][1] = dx / (float)ns; val[ci][2] = dy / (float)ns;
                    ci++;
                }
            const int ncell = nc * nc;
            for (int c = 0; c < 3; ++c)
                for (int a = 0; a < ncell; ++a)
                    for (int b = a + 1; b < ncell; ++b) {
                        if (val[a][c] > val[b][c]) d[bit >> 3] |= (uint8_t)(1u << (bit & 7));
                        bit++;
                    }
        }
    }
}

/* flags: bit 0 = upright (no orientation estimate: SURF's `upright` parameter); bit 4 (16) = contrast-relative detector threshold */
ORC_API int orc_detect_describe_ex(const uint8_t *gray, int h, int w, orc_keypoint *kps, uint8_t *desc, float *kcontrast, int flags)
{
    scale_space S;
    build_scale_space(gray, h, w, &S);
    int n = detect(&S, kps, flags);
    for (int q = 0; q < n; ++q) {
        if (flags & 1) { kps[q].co = 1.0f; kps[q].si = 0.0f; }
        else orient(&S, &kps[q]);
    }
    describe(&S, kps, n, desc);
    if (kcontrast) *kcontrast = S.kcontrast;
    ss_free(&S);
    return n;
}

ORC_API int orc_detect_describe(const uint8_t *gray, int h, int w, orc_keypoint *kps, uint8_t *desc, float *kcontrast)
{
    return orc_detect_describe_ex(gray, h, w, kps, desc, kcontrast, 0);
}

/* stage tap for parity tests: level images (each h*w floats; pointers may be NULL) */
ORC_API void orc_scale_space_level(const uint8_t *gray, int h, int w, int level, float *Lt, float *Lx, float *Ly, float *Ldet)
{
    scale_space S;
    build_scale_space(gray, h, w, &S);
    size_t n = (size_t)h * w * 4;
    if (Lt) memcpy(Lt, S.Lt[level], n);
    if (Lx) memcpy(Lx, S.Lx[level], n);
    if (Ly) memcpy(Ly, S.Ly[level], n);
    if (Ldet) memcpy(Ldet, S.Ldet[level], n);
    ss_free(&S);
}

/* ---------------- brute-force Hamming kNN (k = 2) --------------------------------- */
static inline int popcount64(uint64_t v) { return __builtin_popcountll(v); }

ORC_API void orc_match_knn2(const uint8_t *dq, int nq, const uint8_t *dt, int nt, int32_t *idx /* nq*2 */,
                            int32_t *dist /* nq*2 */)
{
    for (int q = 0; q < nq; ++q) {
        int b0 = 1 << 30, b1 = 1 << 30, i0 = -1, i1 = -1;
        uint64_t a[8];
        memcpy(a, dq + (size_t)q * 64, 64);
        for (int t = 0; t < nt; ++t) {
            uint64_t b[8];
            memcpy(b, dt + (size_t)t * 64, 64);
            int d = 0;
            for (int k = 0; k < 8; ++k) d += popcount64(a[k] ^ b[k]);
            if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = t; }          /* ties keep the lower index */
            else if (d < b1) { b1 = d; i1 = t; }
        }
        idx[q * 2] = i0; idx[q * 2 + 1] = i1; dist[q * 2] = i0 < 0 ? -1 : b0; dist[q * 2 + 1] = i1 < 0 ? -1 : b1;
    }
}

/* ratio test of videostrip.cpp:233-242 (0.8, last query skipped: B-12) */
ORC_API int orc_ratio_test(const int32_t *idx, const int32_t *dist, int nq, int nt, int32_t *good_q, int32_t *good_t)
{
    int n = 0;
    if (nt < 2 || nq < 1) return 0;
    for (int k = 0; k < nq - 1; ++k)
        if ((double)dist[k * 2] < 0.8 * (double)dist[k * 2 + 1]) { good_q[n] = k; good_t[n] = idx[k * 2]; n++; }
    return n;
}

/* ---------------- deterministic RANSAC homography ---------------------------------- */
static inline uint32_t hash32(uint32_t a)
{
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}

static int solve8(double A[8][9])
{
    for (int c = 0; c < 8; ++c) {
        int p = c;
        for (int r = c + 1; r < 8; ++r) if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
        if (!(fabs(A[p][c]) > 1e-12)) return 0;
        if (p != c) for (int k = 0; k < 9; ++k) { double t = A[c][k]; A[c][k] = A[p][k]; A[p][k] = t; }
        for (int r = c + 1; r < 8; ++r) {
            double f = A[r][c] / A[c][c];
            for (int k = c; k < 9; ++k) A[r][k] = A[r][k] - f * A[c][k];
        }
    }
    for (int r = 7; r >= 0; --r) {
        double s = A[r][8];
        for (int k = r + 1; k < 8; ++k) s = s - A[r][k] * A[k][8];
        A[r][8] = s / A[r][r];
    }
    return 1;
}

static void sample4(uint32_t seed, int it, int n, int pick[4])
{
    for (int j = 0; j < 4; ++j) {
        uint32_t attempt = 0;
        for (;;) {
            uint32_t r = hash32(seed ^ hash32((uint32_t)(it * 4 + j + 1) + attempt * 0x9e3779b9u));
            int c = (int)(r % (uint32_t)n);
            int dup = 0;
            for (int m = 0; m < j; ++m) if (pick[m] == c) dup = 1;
            if (!dup || attempt >= 16) { pick[j] = c; break; }
            attempt++;
        }
    }
}

static int hyp_from4(const float *ox, const float *oy, const float *sx, const float *sy, const int pick[4], double H[9])
{
    double A[8][9];
    for (int j = 0; j < 4; ++j) {
        const double x = ox[pick[j]], y = oy[pick[j]], X = sx[pick[j]], Y = sy[pick[j]];
        double *r0 = A[2 * j], *r1 = A[2 * j + 1];
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -x * X; r0[7] = -y * X; r0[8] = X;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -x * Y; r1[7] = -y * Y; r1[8] = Y;
    }
    if (!solve8(A)) return 0;
    for (int k = 0; k < 8; ++k) H[k] = A[k][8];
    H[8] = 1.0;
    return 1;
}

static inline int is_inlier(const double H[9], double x, double y, double X, double Y)
{
    const double wv = H[6] * x + H[7] * y + H[8];
    const double px = (H[0] * x + H[1] * y + H[2]) / wv, py = (H[3] * x + H[4] * y + H[5]) / wv;
    const double ex = px - X, ey = py - Y;
    return (ex * ex + ey * ey) <= 9.0;
}

/* Least squares refit on the inliers in coordinates normalised by the fixed map
 * u = (x - cx)/s, v = (y - cy)/s.  The 44 sums are accumulated in the order the
 * device uses: 256 strided partial sums, a butterfly inside each group of 64,
 * then the 4 group totals in order. */
static int refit(const float *ox, const float *oy, const float *sx, const float *sy, int n, const uint8_t *inl,
                 double cx, double cy, double s, double H[9])
{
    enum { NS = 44 };
    static double part[256][NS];
    memset(part, 0, sizeof part);
    for (int t = 0; t < 256; ++t)
        for (int i = t; i < n; i += 256) {
            if (!inl[i]) continue;
            const double x = ((double)ox[i] - cx) / s, y = ((double)oy[i] - cy) / s;
            const double X = ((double)sx[i] - cx) / s, Y = ((double)sy[i] - cy) / s;
            const double a[8] = {x, y, 1, 0, 0, 0, -x * X, -y * X}, b[8] = {0, 0, 0, x, y, 1, -x * Y, -y * Y};
            int k = 0;
            for (int r = 0; r < 8; ++r)
                for (int c = r; c < 8; ++c) { part[t][k] = part[t][k] + (a[r] * a[c] + b[r] * b[c]); k++; }
            for (int r = 0; r < 8; ++r) { part[t][k] = part[t][k] + (a[r] * X + b[r] * Y); k++; }
        }
    double tot[NS];
    for (int k = 0; k < NS; ++k) {
        double g[4];
        for (int grp = 0; grp < 4; ++grp) {
            double v[64];
            for (int l = 0; l < 64; ++l) v[l] = part[grp * 64 + l][k];
            for (int d = 32; d >= 1; d >>= 1) {
                double nv[64];
                for (int l = 0; l < 64; ++l) nv[l] = v[l] + v[l ^ d];
                memcpy(v, nv, sizeof v);
            }
            g[grp] = v[0];
        }
        tot[k] = ((g[0] + g[1]) + g[2]) + g[3];
    }
    double A[8][9];
    int k = 0;
    for (int r = 0; r < 8; ++r)
        for (int c = r; c < 8; ++c) { A[r][c] = tot[k]; A[c][r] = tot[k]; k++; }
    for (int r = 0; r < 8; ++r) A[r][8] = tot[k++];
    if (!solve8(A)) return 0;
    /* H = T^-1 * Hn * T with T = [[1/s,0,-cx/s],[0,1/s,-cy/s],[0,0,1]] */
    const double hn[9] = {A[0][8], A[1][8], A[2][8], A[3][8], A[4][8], A[5][8], A[6][8], A[7][8], 1.0};
    double M[9];                         /* Hn * T */
    for (int r = 0; r < 3; ++r) {
        M[r * 3 + 0] = hn[r * 3 + 0] / s;
        M[r * 3 + 1] = hn[r * 3 + 1] / s;
        M[r * 3 + 2] = (hn[r * 3 + 2] - hn[r * 3 + 0] * (cx / s)) - hn[r * 3 + 1] * (cy / s);
    }
    double R[9];                         /* T^-1 * M,  T^-1 = [[s,0,cx],[0,s,cy],[0,0,1]] */
    for (int c = 0; c < 3; ++c) {
        R[0 * 3 + c] = s * M[0 * 3 + c] + cx * M[2 * 3 + c];
        R[1 * 3 + c] = s * M[1 * 3 + c] + cy * M[2 * 3 + c];
        R[2 * 3 + c] = M[2 * 3 + c];
    }
    if (R[8] == 0.0 || R[8] != R[8]) return 0;
    for (int i = 0; i < 9; ++i) H[i] = R[i] / R[8];
    return 1;
}

/* findHomography(obj, scene, RANSAC) stand-in; returns the inlier count of the chosen
 * hypothesis (0 = no model: the reference's "H.empty()").  (w, h) = image size used for
 * the fixed normalisation of the refit. */
ORC_API int orc_find_homography_ex(const float *ox, const float *oy, const float *sx, const float *sy, int n, int w, int h,
                                   uint32_t seed, double H[9], int min_inliers);
ORC_API int orc_find_homography(const float *ox, const float *oy, const float *sx, const float *sy, int n, int w, int h,
                                uint32_t seed, double H[9])
{
    return orc_find_homography_ex(ox, oy, sx, sy, n, w, h, seed, H, OV_MIN_INLIERS);
}

/* min_inliers: OV_MIN_INLIERS (4) by default = the reference's rule (any homography findHomography returns for >= 4
 * good matches is used, videostrip.cpp:252-272); 6 = the strict opt-in rule (uwip.h UWIP_OVERLAP_MIN6) */
ORC_API int orc_find_homography_ex(const float *ox, const float *oy, const float *sx, const float *sy, int n, int w, int h,
                                   uint32_t seed, double H[9], int min_inliers)
{
    if (n < 4) return 0;
    int best = 0, best_it = -1;
    double Hb[9] = {0};
    for (int it = 0; it < OV_RANSAC_ITERS; ++it) {
        int pick[4];
        sample4(seed, it, n, pick);
        double Hc[9];
        if (!hyp_from4(ox, oy, sx, sy, pick, Hc)) continue;
        int cnt = 0;
        for (int i = 0; i < n; ++i) cnt += is_inlier(Hc, ox[i], oy[i], sx[i], sy[i]);
        if (cnt > best) { best = cnt; best_it = it; memcpy(Hb, Hc, sizeof Hb); }
    }
    if (best < min_inliers || best_it < 0) return 0;
    uint8_t *inl = (uint8_t *)malloc((size_t)n);
    for (int i = 0; i < n; ++i) inl[i] = (uint8_t)is_inlier(Hb, ox[i], oy[i], sx[i], sy[i]);
    double Hr[9];
    if (refit(ox, oy, sx, sy, n, inl, 0.5 * (double)w, 0.5 * (double)h, 0.5 * (double)w, Hr)) memcpy(H, Hr, sizeof Hr);
    else memcpy(H, Hb, sizeof Hb);
    free(inl);
    return best;
}

/* ---------------- overlapArea   videostrip.cpp:291-319 --------------------------------- */
/* cv::clipLine / cv::line (8-connected Bresenham) / cv::fillConvexPoly (shift = 0) restated from
 * OpenCV 3.4 drawing.cpp; the mask is TARGET_HEIGHT x TARGET_WIDTH = 480 x 640. */
#define TW 640
#define TH 480

static int clip_line(int64_t W, int64_t Hh, int64_t *x1, int64_t *y1, int64_t *x2, int64_t *y2)
{
    int64_t right = W - 1, bottom = Hh - 1;
    int c1 = (*x1 < 0) + (*x1 > right) * 2 + (*y1 < 0) * 4 + (*y1 > bottom) * 8;
    int c2 = (*x2 < 0) + (*x2 > right) * 2 + (*y2 < 0) * 4 + (*y2 > bottom) * 8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        int64_t a;
        if (c1 & 12) {
            a = c1 < 8 ? 0 : bottom;
            *x1 += (int64_t)((double)(a - *y1) * (double)(*x2 - *x1) / (double)(*y2 - *y1));
            *y1 = a;
            c1 = (*x1 < 0) + (*x1 > right) * 2;
        }
        if (c2 & 12) {
            a = c2 < 8 ? 0 : bottom;
            *x2 += (int64_t)((double)(a - *y2) * (double)(*x2 - *x1) / (double)(*y2 - *y1));
            *y2 = a;
            c2 = (*x2 < 0) + (*x2 > right) * 2;
        }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) {
                a = c1 == 1 ? 0 : right;
                *y1 += (int64_t)((double)(a - *x1) * (double)(*y2 - *y1) / (double)(*x2 - *x1));
                *x1 = a;
                c1 = 0;
            }
            if (c2) {
                a = c2 == 1 ? 0 : right;
                *y2 += (int64_t)((double)(a - *x2) * (double)(*y2 - *y1) / (double)(*x2 - *x1));
                *x2 = a;
                c2 = 0;
            }
        }
    }
    return (c1 | c2) == 0;
}

static void draw_line(uint8_t *mask, int64_t x1, int64_t y1, int64_t x2, int64_t y2)
{
    if (!clip_line(TW, TH, &x1, &y1, &x2, &y2)) return;
    if (x2 < x1) {                                   /* Line() iterates left to right */
        int64_t t = x1; x1 = x2; x2 = t;
        t = y1; y1 = y2; y2 = t;
    }
    int dx = (int)(x2 - x1), dy = (int)(y2 - y1);
    int sx = dx < 0 ? -1 : 1, sy = dy < 0 ? -1 : 1;
    dx = dx < 0 ? -dx : dx; dy = dy < 0 ? -dy : dy;
    int x = (int)x1, y = (int)y1;
    if (dy > dx) {                                   /* steep: step in y */
        int err = dy - (dx + dx);
        for (int i = 0; i <= dy; ++i) {
            mask[(size_t)y * TW + x] = 255;
            int m = err < 0;
            err += -(dx + dx) + (m ? dy + dy : 0);
            y += sy;
            if (m) x += sx;
        }
    } else {
        int err = dx - (dy + dy);
        for (int i = 0; i <= dx; ++i) {
            mask[(size_t)y * TW + x] = 255;
            int m = err < 0;
            err += -(dy + dy) + (m ? dx + dx : 0);
            x += sx;
            if (m) y += sy;
        }
    }
}

static void fill_convex_poly(uint8_t *mask, const int64_t vx[4], const int64_t vy[4])
{
    enum { XY_SHIFT = 16 };
    const int64_t XY_ONE = 1 << XY_SHIFT;
    const int npts = 4;
    struct { int idx, di; int64_t x, dx; int ye; } edge[2];
    const int delta1 = (int)(XY_ONE >> 1), delta2 = (int)(XY_ONE >> 1);
    int imin = 0, edges = npts;
    int64_t xmin = vx[0], xmax = vx[0], ymin = vy[0], ymax = vy[0];
    for (int i = 0; i < npts; ++i) {
        if (vy[i] < ymin) { ymin = vy[i]; imin = i; }
        if (vy[i] > ymax) ymax = vy[i];
        if (vx[i] > xmax) xmax = vx[i];
        if (vx[i] < xmin) xmin = vx[i];
        const int p = (i + npts - 1) % npts;
        draw_line(mask, vx[p], vy[p], vx[i], vy[i]);
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
                int idx0 = edge[i].idx, di = edge[i].di;
                int idx = idx0 + di;
                if (idx >= npts) idx -= npts;
                int ty = 0;
                for (; edges-- > 0;) {
                    ty = (int)vy[idx];
                    if (ty > y) {
                        const int64_t xs = vx[idx0] << XY_SHIFT, xe = vx[idx] << XY_SHIFT;
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
                for (int x = xx1; x <= xx2; ++x) mask[(size_t)y * TW + x] = 255;
            }
        }
        edge[0].x += edge[0].dx;
        edge[1].x += edge[1].dx;
    } while (++y <= (int)ymax);
}

ORC_API float orc_overlapArea(const double H[9], int videoWidth, int videoHeight, int32_t *ov_count /* optional */)
{
    const float px[4] = {0, TW, TW, 0}, py[4] = {0, 0, TH, TH};
    float fx[4], fy[4];
    int64_t vx[4], vy[4];
    for (int i = 0; i < 4; ++i) {
        /* perspectiveTransform: double arithmetic, results stored as float */
        const double x = px[i], y = py[i];
        double w = x * H[6] + y * H[7] + H[8];
        if (fabs(w) > 2.220446049250313e-16) {
            w = 1.0 / w;
            fx[i] = (float)((x * H[0] + y * H[1] + H[2]) * w);
            fy[i] = (float)((x * H[3] + y * H[4] + H[5]) * w);
        } else {
            fx[i] = fy[i] = 0.0f;
        }
        /* Point(Point2f): saturate_cast<int> = cvRound */
        vx[i] = (int64_t)lrintf(fx[i]);
        vy[i] = (int64_t)lrintf(fy[i]);
    }
    uint8_t *mask = (uint8_t *)calloc((size_t)TW * TH, 1);
    fill_convex_poly(mask, vx, vy);
    int cnt = 0;
    for (size_t i = 0; i < (size_t)TW * TH; ++i) cnt += mask[i] != 0;
    free(mask);
    if (ov_count) *ov_count = cnt;
    double a00 = 0;                                   /* contourArea on the float points */
    for (int i = 0; i < 4; ++i) {
        const int p = (i + 3) % 4;
        a00 += (double)fx[p] * fy[i] - (double)fy[p] * fx[i];
    }
    const float area_img1 = (float)(videoWidth * videoHeight);
    const float area_img2 = (float)fabs(a00 * 0.5);
    const float area_cur = (float)cnt;
    return area_cur / (area_img1 + area_img2 - area_cur);
}

/* ---------------- calcOverlap   videostrip.cpp:192-289 ----------------------------------- */
/* key / obj: full-resolution BGR frames; returns the overlap ratio, -1 or -2.0 as the reference does.
 * `info` (optional, 8 ints): nkp_obj, nkp_key, ngood, ninliers, ov_count. */
ORC_API float orc_calcOverlap_ex(const uint8_t *key, const uint8_t *obj, int rows, int cols, size_t step, int videoWidth,
                                 int videoHeight, uint32_t seed, int32_t *info, double *Hout, int flags);
ORC_API float orc_calcOverlap(const uint8_t *key, const uint8_t *obj, int rows, int cols, size_t step, int videoWidth,
                              int videoHeight, uint32_t seed, int32_t *info, double *Hout)
{
    return orc_calcOverlap_ex(key, obj, rows, cols, step, videoWidth, videoHeight, seed, info, Hout, 0);
}

/* flags: bit 0 = upright descriptors, bit 4 (16) = contrast-relative detector threshold, bit 2 = accepted, names the default (the reference's ">= 4 matches" rule), bit 3 = >= 6 inliers (UWIP_OVERLAP_MIN6) */
ORC_API float orc_calcOverlap_ex(const uint8_t *key, const uint8_t *obj, int rows, int cols, size_t step, int videoWidth,
                                 int videoHeight, uint32_t seed, int32_t *info, double *Hout, int flags)
{
    if (!key || !obj || rows <= 0 || cols <= 0) return -1.0f;
    int oh, ow;
    orc_resize_dims(rows, cols, 640, &oh, &ow);
    uint8_t *gk = (uint8_t *)malloc((size_t)oh * ow), *go = (uint8_t *)malloc((size_t)oh * ow);
    orc_resize_gray(key, rows, cols, step, oh, ow, gk, NULL);
    orc_resize_gray(obj, rows, cols, step, oh, ow, go, NULL);
    orc_keypoint *kk = (orc_keypoint *)malloc(sizeof(orc_keypoint) * OV_MAXKP), *ko = (orc_keypoint *)malloc(sizeof(orc_keypoint) * OV_MAXKP);
    uint8_t *dk = (uint8_t *)malloc((size_t)OV_MAXKP * 64), *dob = (uint8_t *)malloc((size_t)OV_MAXKP * 64);
    int nk = orc_detect_describe_ex(gk, oh, ow, kk, dk, NULL, flags);
    int no = orc_detect_describe_ex(go, oh, ow, ko, dob, NULL, flags);
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * 2 * (no > 0 ? no : 1)), *dist = (int32_t *)malloc(sizeof(int32_t) * 2 * (no > 0 ? no : 1));
    int32_t *gq = (int32_t *)malloc(sizeof(int32_t) * (no > 0 ? no : 1)), *gt = (int32_t *)malloc(sizeof(int32_t) * (no > 0 ? no : 1));
    orc_match_knn2(dob, no, dk, nk, idx, dist);
    int ng = orc_ratio_test(idx, dist, no, nk, gq, gt);
    float result = -2.0f;
    int ninl = 0, ovc = 0;
    if (ng >= 4) {
        float *ox = (float *)malloc(4 * ng), *oy = (float *)malloc(4 * ng), *sx = (float *)malloc(4 * ng), *sy = (float *)malloc(4 * ng);
        for (int i = 0; i < ng; ++i) { ox[i] = ko[gq[i]].x; oy[i] = ko[gq[i]].y; sx[i] = kk[gt[i]].x; sy[i] = kk[gt[i]].y; }
        double H[9];
        ninl = orc_find_homography_ex(ox, oy, sx, sy, ng, ow, oh, seed, H, (flags & 8) ? 6 : OV_MIN_INLIERS);
        if (ninl > 0) {
            result = orc_overlapArea(H, videoWidth, videoHeight, &ovc);
            if (Hout) memcpy(Hout, H, sizeof H);
        }
        free(ox); free(oy); free(sx); free(sy);
    }
    if (info) { info[0] = no; info[1] = nk; info[2] = ng; info[3] = ninl; info[4] = ovc; }
    free(gk); free(go); free(kk); free(ko); free(dk); free(dob); free(idx); free(dist); free(gq); free(gt);
    return result;
}
