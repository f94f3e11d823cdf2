/*
 * dehaze_oracle.c -- CPU restatement (plain C, float64) of the bgdehaze module.
 *
 * TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product path never does.
 *
 * Same algorithm, stage by stage, as oracle/dehaze_oracle.py (the numpy restatement that is
 * pinned by golden vectors produced from the reference's own Python, tests/golden/dehaze_*.npz);
 * this C form exists because the numpy one needs ~4 s per 1080p frame: it is the checker of the
 * full-size (1080p / 4K) parity tests and the `-O3 -march=native` CPU baseline bench.py times.
 * It is itself checked against the numpy oracle and the goldens in tests/test_oracle_dehaze.py.
 *
 * Follows (file:line in /root/reference/modules/bgdehaze/):
 *   main.py:16-17         normalisation by the global min / max of the uint8 image
 *   BGDehaze.py:14-26     Background_light (ties: first row-major index)
 *   BGDehaze.py:28-37     transmission_map (w = 15 always, B-10)
 *   BGDehaze.py:39-48     refined_t
 *   guidedfilter.py:23-51 boxfilter (cumulative sums along axis 0, then axis 1: same order, same rounding)
 *   guidedfilter.py:54-103 guided_filter
 *   BGDehaze.py:50-69     dehazed_BG, RC_correction (np.average = numpy's pairwise summation, restated)
 *   BGDehaze.py:71-89     adaptiveExp_map (cv2 BGR2YCrCb restated in fixed point: parity unpinned)
 *   main.py:19            imwrite(restored * 255): round half to even + saturate
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* numpy's pairwise summation of a contiguous double array (numpy/core/src/umath/loops_utils.h.src,
 * DOUBLE_pairwise_sum): blocks of <= 128 with 8 partial sums, recursive halves above. */
static double pairwise_sum(const double *a, size_t n)
{
    if (n < 8) {
        double res = 0.;
        for (size_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        size_t i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        size_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
    }
}

/* zero-padded w x w window max (is_max) or min of one plane, separable (BGDehaze.py:16-21, :32-36) */
static void window_reduce(const double *a, int M, int N, int w, int is_max, double *out, double *tmp)
{
    const int pad = w / 2;
    /* rows: tmp[y][x] over columns x-pad .. x-pad+w-1 (outside = 0) */
    for (int y = 0; y < M; y++) {
        for (int x = 0; x < N; x++) {
            double m = 0.0;
            int first = 1;
            for (int k = 0; k < w; k++) {
                const int xx = x - pad + k;
                const double v = (xx >= 0 && xx < N) ? a[(size_t)y * N + xx] : 0.0;
                if (first) { m = v; first = 0; }
                else m = is_max ? (v > m ? v : m) : (v < m ? v : m);
            }
            tmp[(size_t)y * N + x] = m;
        }
    }
    for (int y = 0; y < M; y++) {
        for (int x = 0; x < N; x++) {
            double m = 0.0;
            int first = 1;
            for (int k = 0; k < w; k++) {
                const int yy = y - pad + k;
                const double v = (yy >= 0 && yy < M) ? tmp[(size_t)yy * N + x] : 0.0;
                if (first) { m = v; first = 0; }
                else m = is_max ? (v > m ? v : m) : (v < m ? v : m);
            }
            out[(size_t)y * N + x] = m;
        }
    }
}

/* guidedfilter.py:23-51: out[y][x] = sum over the in-image part of rows y-r..y+r, cols x-r..x+r, by a cumulative
 * sum down the columns and a difference, then along the rows and a difference (S needs (M+1)*N, S2 M*(N+1)) */
static void boxfilter(const double *I, int M, int N, int r, double *out, double *S, double *S2)
{
    for (int x = 0; x < N; x++) S[x] = 0.0;
    for (int y = 0; y < M; y++)
        for (int x = 0; x < N; x++) S[(size_t)(y + 1) * N + x] = S[(size_t)y * N + x] + I[(size_t)y * N + x];
    for (int y = 0; y < M; y++) {
        const int hi = (y + r < M - 1 ? y + r : M - 1) + 1, lo = y - r > 0 ? y - r : 0;
        double *row = S2 + (size_t)y * (N + 1);
        row[0] = 0.0;
        for (int x = 0; x < N; x++) row[x + 1] = row[x] + (S[(size_t)hi * N + x] - S[(size_t)lo * N + x]);
    }
    for (int y = 0; y < M; y++) {
        const double *row = S2 + (size_t)y * (N + 1);
        for (int x = 0; x < N; x++) {
            const int hi = (x + r < N - 1 ? x + r : N - 1) + 1, lo = x - r > 0 ? x - r : 0;
            out[(size_t)y * N + x] = row[hi] - row[lo];
        }
    }
}

typedef struct {
    int M, N, r;
    double eps;
    double *base;        /* boxfilter(ones) */
    double *mean[3];     /* means of the guide channels */
    double *inv[6];      /* inverse of Sigma + eps I, upper triangle 00 01 02 11 12 22 */
    const double *I[3];  /* guide planes */
    double *S, *S2, *t0, *t1;
} gf_guide;

static void gf_free(gf_guide *g)
{
    free(g->base);
    for (int i = 0; i < 3; i++) free(g->mean[i]);
    for (int i = 0; i < 6; i++) free(g->inv[i]);
    free(g->S); free(g->S2); free(g->t0); free(g->t1);
}

/* the part of guided_filter (guidedfilter.py:60-93) that depends on the guide only */
static int gf_prepare(gf_guide *g, const double *I0, const double *I1, const double *I2, int M, int N, int r, double eps)
{
    const size_t n = (size_t)M * N;
    memset(g, 0, sizeof *g);
    g->M = M; g->N = N; g->r = r; g->eps = eps;
    g->I[0] = I0; g->I[1] = I1; g->I[2] = I2;
    g->base = malloc(n * sizeof(double));
    g->S = malloc((size_t)(M + 1) * N * sizeof(double));
    g->S2 = malloc((size_t)M * (N + 1) * sizeof(double));
    g->t0 = malloc(n * sizeof(double));
    g->t1 = malloc(n * sizeof(double));
    for (int i = 0; i < 3; i++) g->mean[i] = malloc(n * sizeof(double));
    for (int i = 0; i < 6; i++) g->inv[i] = malloc(n * sizeof(double));
    if (!g->base || !g->S || !g->S2 || !g->t0 || !g->t1) return -1;
    for (int i = 0; i < 3; i++) if (!g->mean[i]) return -1;
    for (int i = 0; i < 6; i++) if (!g->inv[i]) return -1;
    for (size_t i = 0; i < n; i++) g->t0[i] = 1.0;
    boxfilter(g->t0, M, N, r, g->base, g->S, g->S2);
    for (int c = 0; c < 3; c++) {
        boxfilter(g->I[c], M, N, r, g->mean[c], g->S, g->S2);
        for (size_t i = 0; i < n; i++) g->mean[c][i] /= g->base[i];
    }
    /* var_ij = box(I_i I_j) / base - mean_i mean_j, stored in inv[] and inverted in place below */
    int k = 0;
    for (int a = 0; a < 3; a++)
        for (int b = a; b < 3; b++, k++) {
            for (size_t i = 0; i < n; i++) g->t0[i] = g->I[a][i] * g->I[b][i];
            boxfilter(g->t0, M, N, r, g->inv[k], g->S, g->S2);
            for (size_t i = 0; i < n; i++) g->inv[k][i] = g->inv[k][i] / g->base[i] - g->mean[a][i] * g->mean[b][i];
        }
    for (size_t i = 0; i < n; i++) {
        const double s00 = g->inv[0][i] + eps, s01 = g->inv[1][i], s02 = g->inv[2][i], s11 = g->inv[3][i] + eps,
                     s12 = g->inv[4][i], s22 = g->inv[5][i] + eps;
        const double k00 = s11 * s22 - s12 * s12, k01 = s02 * s12 - s01 * s22, k02 = s01 * s12 - s02 * s11;
        const double k11 = s00 * s22 - s02 * s02, k12 = s01 * s02 - s00 * s12, k22 = s00 * s11 - s01 * s01;
        const double det = s00 * k00 + s01 * k01 + s02 * k02;
        g->inv[0][i] = k00 / det; g->inv[1][i] = k01 / det; g->inv[2][i] = k02 / det;
        g->inv[3][i] = k11 / det; g->inv[4][i] = k12 / det; g->inv[5][i] = k22 / det;
    }
    return 0;
}

/* the p-dependent part (guidedfilter.py:69-101): q = (box(a).I + box(b)) / base */
static int gf_apply(gf_guide *g, const double *p, double *q)
{
    const int M = g->M, N = g->N, r = g->r;
    const size_t n = (size_t)M * N;
    double *mean_p = malloc(n * sizeof(double)), *a[3], *b = malloc(n * sizeof(double)), *cov[3];
    int ok = mean_p && b;
    for (int c = 0; c < 3; c++) { a[c] = malloc(n * sizeof(double)); cov[c] = malloc(n * sizeof(double)); ok = ok && a[c] && cov[c]; }
    if (ok) {
        boxfilter(p, M, N, r, mean_p, g->S, g->S2);
        for (size_t i = 0; i < n; i++) mean_p[i] /= g->base[i];
        for (int c = 0; c < 3; c++) {
            for (size_t i = 0; i < n; i++) g->t0[i] = g->I[c][i] * p[i];
            boxfilter(g->t0, M, N, r, cov[c], g->S, g->S2);
            for (size_t i = 0; i < n; i++) cov[c][i] = cov[c][i] / g->base[i] - g->mean[c][i] * mean_p[i];
        }
        for (size_t i = 0; i < n; i++) {
            const double c0 = cov[0][i], c1 = cov[1][i], c2 = cov[2][i];
            const double a0 = c0 * g->inv[0][i] + c1 * g->inv[1][i] + c2 * g->inv[2][i];
            const double a1 = c0 * g->inv[1][i] + c1 * g->inv[3][i] + c2 * g->inv[4][i];
            const double a2 = c0 * g->inv[2][i] + c1 * g->inv[4][i] + c2 * g->inv[5][i];
            a[0][i] = a0; a[1][i] = a1; a[2][i] = a2;
            b[i] = mean_p[i] - a0 * g->mean[0][i] - a1 * g->mean[1][i] - a2 * g->mean[2][i];
        }
        for (int c = 0; c < 3; c++) {
            boxfilter(a[c], M, N, r, g->t1, g->S, g->S2);
            if (c == 0) for (size_t i = 0; i < n; i++) q[i] = g->t1[i] * g->I[0][i];
            else for (size_t i = 0; i < n; i++) q[i] = q[i] + g->t1[i] * g->I[c][i];
        }
        boxfilter(b, M, N, r, g->t1, g->S, g->S2);
        for (size_t i = 0; i < n; i++) q[i] = (q[i] + g->t1[i]) / g->base[i];
    }
    free(mean_p); free(b);
    for (int c = 0; c < 3; c++) { free(a[c]); free(cov[c]); }
    return ok ? 0 : -1;
}

/* guided_filter(I, p, r, eps) for an interleaved float64 guide [M][N][3] (test tap) */
ORC_API int orc_guided_filter(const double *guide, const double *p, int M, int N, int r, double eps, double *q)
{
    if (M < 2 * r + 1 || N < 2 * r + 1) return -2;
    const size_t n = (size_t)M * N;
    double *pl[3];
    for (int c = 0; c < 3; c++) {
        pl[c] = malloc(n * sizeof(double));
        if (!pl[c]) return -1;
        for (size_t i = 0; i < n; i++) pl[c][i] = guide[i * 3 + c];
    }
    gf_guide g;
    int rc = gf_prepare(&g, pl[0], pl[1], pl[2], M, N, r, eps);
    if (!rc) rc = gf_apply(&g, p, q);
    gf_free(&g);
    for (int c = 0; c < 3; c++) free(pl[c]);
    return rc;
}

static void minmax_plane(const double *a, size_t n, double *mn, double *mx)
{
    /* numpy's min / max propagate NaN */
    double lo = a[0], hi = a[0];
    int nan = 0;
    for (size_t i = 0; i < n; i++) {
        if (a[i] != a[i]) nan = 1;
        if (a[i] < lo) lo = a[i];
        if (a[i] > hi) hi = a[i];
    }
    if (nan) lo = hi = NAN;
    *mn = lo; *mx = hi;
}

static void bgr2ycrcb(int b, int g, int r, int *Y, int *Cr, int *Cb)
{
    int y = (b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14;
    int cr = ((r - y) * 11682 + (128 << 14) + (1 << 13)) >> 14;
    int cb = ((b - y) * 9241 + (128 << 14) + (1 << 13)) >> 14;
    *Y = y < 0 ? 0 : (y > 255 ? 255 : y);
    *Cr = cr < 0 ? 0 : (cr > 255 ? 255 : cr);
    *Cb = cb < 0 ? 0 : (cb > 255 ? 255 : cb);
}

static uint8_t to_u8_rne(double v)
{
    if (v != v) return 0;
    const double r = nearbyint(v);      /* default rounding mode: half to even */
    return (uint8_t)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
}

#define ORC_DEHAZE_FULL 1
#define ORC_DEHAZE_GUARD_S 2

/* generate_results(), main.py:14-20, for one uint8 BGR frame.  Optional outputs (NULL to skip):
 *   out [rows][ostep] uint8 BGR; tap_B[3]; tap_idx[2]; tap_traw [2][rows][cols] (transmission before the 0.2 clamp);
 *   tap_refined [2][rows][cols]; tap_restored [rows][cols][3] (RC_correction); tap_final [rows][cols][3] (before * 255).
 * B_inject[3] (or NULL) replaces Background_light's result.  Returns 0, -1 out of memory, -2 image smaller than 81. */
ORC_API int orc_dehaze_u8(const uint8_t *img, int rows, int cols, size_t step, int w, int flags, const double *B_inject,
                          uint8_t *out, size_t ostep, double *tap_B, int32_t *tap_idx, double *tap_traw,
                          double *tap_refined, double *tap_restored, double *tap_final)
{
    const int M = rows, N = cols, r = 40;
    const double eps = 1e-3, tmin = 0.2;
    if (M < 2 * r + 1 || N < 2 * r + 1) return -2;
    const size_t n = (size_t)M * N;
    int rc = -1;
    double *pl[3] = {0, 0, 0}, *f0 = 0, *f1 = 0, *tmp = 0, *p0 = 0, *p1 = 0, *q0 = 0, *q1 = 0, *red = 0, *S = 0, *RS = 0, *yi[3] = {0, 0, 0};
    uint8_t *YI = 0, *YJ = 0;
    gf_guide g;
    int have_g = 0;
    for (int c = 0; c < 3; c++) if (!(pl[c] = malloc(n * sizeof(double)))) goto done;
    if (!(f0 = malloc(n * sizeof(double))) || !(f1 = malloc(n * sizeof(double))) || !(tmp = malloc(n * sizeof(double))) ||
        !(p0 = malloc(n * sizeof(double))) || !(p1 = malloc(n * sizeof(double))) || !(q0 = malloc(n * sizeof(double))) ||
        !(q1 = malloc(n * sizeof(double))) || !(red = malloc(n * sizeof(double))))
        goto done;
    /* main.py:16-17 */
    int mn = 255, mx = 0;
    for (int y = 0; y < M; y++)
        for (int x = 0; x < 3 * N; x++) {
            const int v = img[(size_t)y * step + x];
            if (v < mn) mn = v;
            if (v > mx) mx = v;
        }
    for (int y = 0; y < M; y++)
        for (int x = 0; x < N; x++)
            for (int c = 0; c < 3; c++)
                pl[c][(size_t)y * N + x] = (double)(uint8_t)(img[(size_t)y * step + 3 * x + c] - mn) / (double)(uint8_t)(mx - mn);
    /* Background_light (BGDehaze.py:14-26) */
    double B[3];
    int i0 = 0, i1 = 0;
    {
        window_reduce(pl[2], M, N, w, 1, f0, tmp);              /* max R */
        window_reduce(pl[0], M, N, w, 1, f1, tmp);              /* max B */
        double best = 0; int have = 0;
        for (size_t i = 0; i < n; i++) { const double d = f0[i] - f1[i]; if (!have || d < best) { if (d == d) { best = d; i0 = (int)i; have = 1; } } }
        window_reduce(pl[1], M, N, w, 1, f1, tmp);              /* max G */
        have = 0;
        for (size_t i = 0; i < n; i++) { const double d = f0[i] - f1[i]; if (!have || d < best) { if (d == d) { best = d; i1 = (int)i; have = 1; } } }
        for (int c = 0; c < 3; c++) B[c] = (pl[c][i0] + pl[c][i1]) / 2.0;
    }
    if (B_inject) for (int c = 0; c < 3; c++) B[c] = B_inject[c];
    if (tap_B) for (int c = 0; c < 3; c++) tap_B[c] = B[c];
    if (tap_idx) { tap_idx[0] = i0; tap_idx[1] = i1; }
    /* transmission_map with w = 15 (BGDehaze.py:28-37, :52) and the 0.2 floor (:43-44) */
    for (int c = 0; c < 2; c++) {
        double *p = c ? p1 : p0;
        for (size_t i = 0; i < n; i++) f0[i] = pl[c][i] / B[c];
        window_reduce(f0, M, N, 15, 0, f1, tmp);
        for (size_t i = 0; i < n; i++) {
            const double t = 1 - f1[i];
            if (tap_traw) tap_traw[(size_t)c * n + i] = t;
            p[i] = t > tmin ? t : tmin;          /* np.maximum(t, 0.2): NaN propagates */
            if (t != t) p[i] = t;
        }
    }
    /* refined_t: two guided filters on the same guide */
    if (gf_prepare(&g, pl[0], pl[1], pl[2], M, N, r, eps)) { have_g = 1; goto done; }
    have_g = 1;
    if (gf_apply(&g, p0, q0) || gf_apply(&g, p1, q1)) goto done;
    if (tap_refined) { memcpy(tap_refined, q0, n * sizeof(double)); memcpy(tap_refined + n, q1, n * sizeof(double)); }
    /* dehazed_BG (BGDehaze.py:50-57) */
    double a, b;
    for (size_t i = 0; i < n; i++) q0[i] = (pl[0][i] - B[0]) / q0[i] + B[0];
    minmax_plane(q0, n, &a, &b);
    for (size_t i = 0; i < n; i++) q0[i] = (q0[i] - a) / (b - a);
    for (size_t i = 0; i < n; i++) q1[i] = (pl[1][i] - B[1]) / q1[i] + B[1];
    minmax_plane(q1, n, &a, &b);
    for (size_t i = 0; i < n; i++) q1[i] = (q1[i] - a) / (b - a);
    /* RC_correction (BGDehaze.py:59-69) */
    {
        const double avgRr = 1.5 - pairwise_sum(q0, n) / (double)n - pairwise_sum(q1, n) / (double)n;
        const double coeff = avgRr / (pairwise_sum(pl[2], n) / (double)n);
        for (size_t i = 0; i < n; i++) red[i] = pl[2][i] * coeff;
        minmax_plane(red, n, &a, &b);
        for (size_t i = 0; i < n; i++) red[i] = (red[i] - a) / (b - a);
    }
    if (tap_restored)
        for (size_t i = 0; i < n; i++) { tap_restored[i * 3] = q0[i]; tap_restored[i * 3 + 1] = q1[i]; tap_restored[i * 3 + 2] = red[i]; }
    const double *res[3] = {q0, q1, red};
    if (!(flags & ORC_DEHAZE_FULL)) {
        if (tap_final) for (size_t i = 0; i < n; i++) for (int c = 0; c < 3; c++) tap_final[i * 3 + c] = res[c][i];
        if (out)
            for (int y = 0; y < M; y++)
                for (int x = 0; x < N; x++)
                    for (int c = 0; c < 3; c++) out[(size_t)y * ostep + 3 * x + c] = to_u8_rne(res[c][(size_t)y * N + x] * 255);
        rc = 0;
        goto done;
    }
    /* adaptiveExp_map tail (BGDehaze.py:75-89) */
    if (!(YI = malloc(n * 3)) || !(YJ = malloc(n * 3)) || !(S = malloc(n * sizeof(double))) || !(RS = malloc(n * sizeof(double)))) goto done;
    for (int c = 0; c < 3; c++) if (!(yi[c] = malloc(n * sizeof(double)))) goto done;
    {
        int jmn = 255, jmx = 0, imn = 255, imx = 0;
        for (size_t i = 0; i < n; i++) {
            int v[3], u[3];
            for (int c = 0; c < 3; c++) {
                v[c] = (int)(uint8_t)(long long)(res[c][i] * 255);     /* astype(uint8): C truncation */
                u[c] = (int)(uint8_t)(long long)(pl[c][i] * 255);
            }
            int Y, Cr, Cb;
            bgr2ycrcb(v[0], v[1], v[2], &Y, &Cr, &Cb);
            YJ[i * 3] = (uint8_t)Y; YJ[i * 3 + 1] = (uint8_t)Cr; YJ[i * 3 + 2] = (uint8_t)Cb;
            { const int lo3 = Y < Cr ? (Y < Cb ? Y : Cb) : (Cr < Cb ? Cr : Cb), hi3 = Y > Cr ? (Y > Cb ? Y : Cb) : (Cr > Cb ? Cr : Cb);
              if (lo3 < jmn) jmn = lo3;
              if (hi3 > jmx) jmx = hi3; }
            bgr2ycrcb(u[0], u[1], u[2], &Y, &Cr, &Cb);
            YI[i * 3] = (uint8_t)Y; YI[i * 3 + 1] = (uint8_t)Cr; YI[i * 3 + 2] = (uint8_t)Cb;
            { const int lo3 = Y < Cr ? (Y < Cb ? Y : Cb) : (Cr < Cb ? Cr : Cb), hi3 = Y > Cr ? (Y > Cb ? Y : Cb) : (Cr > Cb ? Cr : Cb);
              if (lo3 < imn) imn = lo3;
              if (hi3 > imx) imx = hi3; }
        }
        for (size_t i = 0; i < n; i++) {
            for (int c = 0; c < 3; c++) yi[c][i] = (double)(uint8_t)(YI[i * 3 + c] - imn) / (double)(uint8_t)(imx - imn);
            const double Yi = yi[0][i], Yj = (double)(uint8_t)(YJ[i * 3] - jmn) / (double)(uint8_t)(jmx - jmn);
            const double num = Yj * Yi + 0.3 * (Yi * Yi), den = Yj * Yj + 0.3 * (Yi * Yi);
            S[i] = ((flags & ORC_DEHAZE_GUARD_S) && den == 0.0) ? 1.0 : num / den;
        }
    }
    gf_free(&g);
    have_g = 0;
    if (gf_prepare(&g, yi[0], yi[1], yi[2], M, N, r, eps)) { have_g = 1; goto done; }
    have_g = 1;
    if (gf_apply(&g, S, RS)) goto done;
    {
        double lo = 0, hi = 0;
        int first = 1, nan = 0;
        for (size_t i = 0; i < n; i++)
            for (int c = 0; c < 3; c++) {
                const double v = res[c][i] * RS[i];
                if (v != v) nan = 1;
                if (first) { lo = hi = v; first = 0; }
                if (v < lo) lo = v;
                if (v > hi) hi = v;
            }
        if (nan) lo = hi = NAN;
        for (int y = 0; y < M; y++)
            for (int x = 0; x < N; x++) {
                const size_t i = (size_t)y * N + x;
                for (int c = 0; c < 3; c++) {
                    const double v = (res[c][i] * RS[i] - lo) / (hi - lo);
                    if (tap_final) tap_final[i * 3 + c] = v;
                    if (out) out[(size_t)y * ostep + 3 * x + c] = to_u8_rne(v * 255);
                }
            }
    }
    rc = 0;
done:
    if (have_g) gf_free(&g);
    for (int c = 0; c < 3; c++) { free(pl[c]); free(yi[c]); }
    free(f0); free(f1); free(tmp); free(p0); free(p1); free(q0); free(q1); free(red); free(S); free(RS); free(YI); free(YJ);
    return rc;
}
