// The ACLAHE knee stage (modules/aclahe/python/functions.py:49-93: curve_fit -> MINPACK lmdif, splrep / splev, curvature
// arg-max) written ONCE for two back ends:
//   HostLanes  a vector of 64 sample slots held in an array, element loops (the host form, uwip_aclahe_select);
//   WaveLanes  one sample slot per lane of a wavefront (the device form, k_aclahe_knee in aclahe_device.hip).
// Everything that runs over the m = 49 samples (residuals, the forward-difference Jacobian, Householder QR, Q^T f, norms)
// is element-wise work plus REDUCTIONS, and every reduction keeps MINPACK's own order: a serial sum over the samples in
// ascending index (on the device the wave walks its lanes with v_readlane: ~600 cycles per sum, where a butterfly would
// take ~120 -- but a butterfly changes the rounding, and on ill-conditioned curves, where the fit runs to scipy's maxfev
// = 1000 evaluations, the last bits decide whether curve_fit "converges": a butterfly form built first reproduced the
// reference's own functions.py on 94 of the 95 golden curves, this order on 95 of 95).  So the two forms perform the
// same IEEE double operations in the same order and agree BIT FOR BIT: exp is uwip_exp (plain operations), division and
// sqrt are correctly rounded on both sides, contraction is off (-ffp-contract=off).  The n = 4 parameter-space part
// (lmpar, qrsolv, the 4-vector norms) is MINPACK's serial code, run redundantly by every lane.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LM_HD __host__ __device__
#else
#define LM_HD
#endif

namespace uwip_lm {

constexpr double EPSMCH = 2.220446049250313e-16;
constexpr double DWARF = 2.2250738585072014e-308;
constexpr int NP = 4;       // parameters of p0 e^(-p1 x) + p2 e^(-p3 x)
constexpr int MS = 49;      // samples of a curve
constexpr int NS25 = 25;    // spline knots

// exp(x) in plain IEEE double operations: x = k ln2 + r, |r| <= ln2 / 2 (Cody-Waite), Taylor polynomial to r^13 (truncation
// < 4e-18), 2^k in two exponent steps (results in the subnormal range round once).  <= 1 ulp from libm on 4 M samples.
LM_HD inline double uwip_exp(double x)
{
    const double xc = x < -746.0 ? -746.0 : (x > 710.0 ? 710.0 : x);      // beyond: 0 and +inf (NaN passes through)
    const double shifted = xc * 1.4426950408889634 + 6755399441055744.0;  // round-to-nearest-even by the 1.5 * 2^52 shift
    const double kf = shifted - 6755399441055744.0;
    const double r = (xc - kf * 6.93147180369123816490e-01) - kf * 1.90821492927058770002e-10;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    uint64_t sb;
    memcpy(&sb, &shifted, 8);
    const int64_t k = (int64_t)(int32_t)(uint32_t)sb;  // in [-1077, 1025]
    const int64_t k1 = k >> 1, k2 = k - k1;            // each within the normal exponent range
    const uint64_t b1 = (uint64_t)(k1 + 1023) << 52, b2 = (uint64_t)(k2 + 1023) << 52;
    double s1, s2;
    memcpy(&s1, &b1, 8);
    memcpy(&s2, &b2, 8);
    const double y = (p * s1) * s2;
    return x != x ? x : y;
}

// ---- back ends ------------------------------------------------------------------------------------------------------
// A `vec` holds one double per sample slot: W slots per execution context; slot e of a context is sample index(e).
struct HostLanes {
    static constexpr int W = 52;          // >= the 49 samples, a multiple of 4 (the element loops vectorise)
    struct vec { double v[W]; };
    static inline int index(int e) { return e; }
    // element loops over the samples [from, m): the host visits exactly those slots
    static inline int begin(int from) { return from; }
    static inline int end(int m) { return m; }
    static inline bool active(int, int, int) { return true; }
    static inline void set(vec &a, int i, double x) { a.v[i] = x; }
    static inline double pick(const double *M, int i) { return M[i]; }
    // serial sum of the slots from .. m-1 in ascending order, starting from 0.0 (MINPACK's loops)
    static inline double sum(const vec &a, int from, int m)
    {
        double s = 0.0;
        for (int i = from; i < m; ++i) s += a.v[i];
        return s;
    }
    static inline bool any(const bool (&f)[W])
    {
        for (int i = 0; i < W; ++i) if (f[i]) return true;
        return false;
    }
    static inline double max(const vec &a)
    {
        double m = a.v[0];
        for (int i = 1; i < W; ++i) m = a.v[i] > m ? a.v[i] : m;      // max is exact: any order
        return m;
    }
    static inline double at(const vec &a, int i) { return a.v[i]; }
    // gather: slot e of the result = a[idx(e)]
    template <class F> static inline vec gather(const vec &a, F idx)
    {
        vec r;
        for (int e = 0; e < W; ++e) r.v[e] = a.v[idx(e)];
        return r;
    }
    // lowest sample index whose flag is set, or -1
    template <class F> static inline int first(F flag)
    {
        for (int e = 0; e < W; ++e) if (flag(e)) return e;
        return -1;
    }
};

#if defined(__HIPCC__)
struct WaveLanes {
    static constexpr int W = 1;
    struct vec { double v[1]; };
    static __device__ inline int index(int) { return (int)(threadIdx.x & 63u); }
    // element loops over the samples [from, m): one slot per lane, predicated
    static __device__ inline int begin(int) { return 0; }
    static __device__ inline int end(int) { return 1; }
    static __device__ inline bool active(int, int from, int m) { const int i = (int)(threadIdx.x & 63u); return i >= from && i < m; }
    static __device__ inline void set(vec &a, int i, double x) { if ((int)(threadIdx.x & 63u) == i) a.v[0] = x; }
    static __device__ inline double pick(const double *M, int i)      // a select chain: no dynamic register indexing
    {
        double r = 0.0;
        for (int q = 0; q < NS25; ++q) r = q == i ? M[q] : r;
        return r;
    }
    static __device__ inline double shfl_xor(double x, int s)
    {
        const long long b = __double_as_longlong(x);
        const int lo = __shfl_xor((int)(b & 0xffffffffll), s, 64), hi = __shfl_xor((int)(b >> 32), s, 64);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    static __device__ inline double lane_value(double x, int i)      // i wave-uniform
    {
        const long long b = __double_as_longlong(x);
        const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), i), hi = __builtin_amdgcn_readlane((int)(b >> 32), i);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    // the same serial sum: the wave walks its lanes from .. m-1 (the result is wave-uniform)
    static __device__ inline double sum(const vec &a, int from, int m)
    {
        double s = 0.0;
        for (int i = from; i < m; ++i) s += lane_value(a.v[0], i);
        return s;
    }
    static __device__ inline bool any(const bool (&f)[1]) { return __ballot(f[0]) != 0ull; }
    static __device__ inline double max(const vec &a)
    {
        double t = a.v[0];
        for (int s = 1; s < 64; s <<= 1) { const double o = shfl_xor(t, s); t = o > t ? o : t; }
        return t;
    }
    static __device__ inline double at(const vec &a, int i)
    {
        const long long b = __double_as_longlong(a.v[0]);
        const int lo = __shfl((int)(b & 0xffffffffll), i, 64), hi = __shfl((int)(b >> 32), i, 64);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    template <class F> static __device__ inline vec gather(const vec &a, F idx)
    {
        vec r;
        r.v[0] = at(a, idx(0));
        return r;
    }
    template <class F> static __device__ inline int first(F flag)
    {
        const unsigned long long m = __ballot(flag(0));
        return m ? (int)__ffsll((long long)m) - 1 : -1;
    }
};
#endif

// ---- norms ----------------------------------------------------------------------------------------------------------
// MINPACK enorm for the short parameter-space vectors (serial, as published)
LM_HD inline double enorm_n(int n, const double *x)
{
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    const double agiant = rgiant / (double)n;
    for (int i = 0; i < n; ++i) {
        const double xabs = fabs(x[i]);
        if (xabs > rdwarf && xabs < agiant) {
            s2 += xabs * xabs;
        } else if (xabs <= rdwarf) {
            if (xabs > x3max) {
                const double t = x3max / xabs;
                s3 = 1.0 + s3 * (t * t);
                x3max = xabs;
            } else if (xabs != 0.0) {
                const double t = xabs / x3max;
                s3 += t * t;
            }
        } else {
            if (xabs > x1max) {
                const double t = x1max / xabs;
                s1 = 1.0 + s1 * (t * t);
                x1max = xabs;
            } else {
                const double t = xabs / x1max;
                s1 += t * t;
            }
        }
    }
    if (s1 != 0.0) return x1max * sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0.0) {
        if (s2 >= x3max) return sqrt(s2 * (1.0 + (x3max / s2) * (x3max * s3)));
        return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * sqrt(s3);
}

// MINPACK enorm over the sample slots from .. m-1 of a vec, in MINPACK's order.  Common case -- every non-zero component in
// the mid range -- is the serial sum of squares (zeros fall into enorm's "small" class and add nothing, s1 = s3 = 0, and
// the published final expression sqrt(s2 * (1 + (0 / s2) * (0 * 0))) is sqrt(s2)); any tiny or huge component sends all
// contexts through the published element-by-element code.
template <class B> LM_HD inline double enorm_m(const typename B::vec &x, int from, int m)
{
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    const int n = m - from;
    if (n <= 0) return 0.0;
    const double agiant = rgiant / (double)n;
    typename B::vec sq;
    bool odd[B::W];
    for (int e = 0; e < B::W; ++e) odd[e] = false;
    for (int e = B::begin(from); e < B::end(m); ++e) {
        const bool in = B::active(e, from, m);
        const double xabs = in ? fabs(x.v[e]) : 0.0;
        sq.v[e] = xabs * xabs;
        odd[e] = in && xabs != 0.0 && !(xabs > rdwarf && xabs < agiant);       // NaN lands here too
    }
    if (!B::any(odd)) {
        const double s2 = B::sum(sq, from, m);
        return s2 != 0.0 ? sqrt(s2 * (1.0 + (0.0 / s2) * (0.0 * 0.0))) : 0.0 * sqrt(0.0);
    }
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    for (int i = from; i < m; ++i) {
        const double xabs = fabs(B::at(x, i));
        if (xabs > rdwarf && xabs < agiant) {
            s2 += xabs * xabs;
        } else if (xabs <= rdwarf) {
            if (xabs > x3max) {
                const double t = x3max / xabs;
                s3 = 1.0 + s3 * (t * t);
                x3max = xabs;
            } else if (xabs != 0.0) {
                const double t = xabs / x3max;
                s3 += t * t;
            }
        } else {
            if (xabs > x1max) {
                const double t = x1max / xabs;
                s1 = 1.0 + s1 * (t * t);
                x1max = xabs;
            } else {
                const double t = xabs / x1max;
                s1 += t * t;
            }
        }
    }
    if (s1 != 0.0) return x1max * sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0.0) {
        if (s2 >= x3max) return sqrt(s2 * (1.0 + (x3max / s2) * (x3max * s3)));
        return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * sqrt(s3);
}

// 4-element arrays indexed by a run-time permutation entry: select chains, so that (with the n-loops unrolled) every
// array of the parameter-space code lives in registers on the device -- a run-time index sends it to scratch memory, and
// the solver is one long dependent chain through those loads (the first device build spent 2.25 ms per curve there)
LM_HD inline double get4(const double *a, int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3])); }
LM_HD inline int geti4(const int *a, int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : (i == 2 ? a[2] : a[3])); }
LM_HD inline void seti4(int *a, int i, int x)
{
    a[0] = i == 0 ? x : a[0]; a[1] = i == 1 ? x : a[1]; a[2] = i == 2 ? x : a[2]; a[3] = i == 3 ? x : a[3];
}
LM_HD inline void set4(double *a, int i, double x)
{
    a[0] = i == 0 ? x : a[0]; a[1] = i == 1 ? x : a[1]; a[2] = i == 2 ? x : a[2]; a[3] = i == 3 ? x : a[3];
}
#if defined(__HIP_DEVICE_COMPILE__)
#define LM_UNROLL _Pragma("unroll")
#else
#define LM_UNROLL
#endif

// ---- parameter-space solvers: MINPACK qrsolv / lmpar on the n x n triangle r (column-major, leading dimension NP) ------
LM_HD inline void qrsolv(double *r, const int *ipvt, const double *diag, const double *qtb, double *x, double *sdiag, double *wa)
{
    constexpr int n = NP, ldr = NP;
    LM_UNROLL for (int j = 0; j < n; ++j) {
        LM_UNROLL for (int i = j; i < n; ++i) r[j * ldr + i] = r[i * ldr + j];
        x[j] = r[j * ldr + j];
        wa[j] = qtb[j];
    }
    LM_UNROLL for (int j = 0; j < n; ++j) {
        const int l = ipvt[j];
        const double diag_l = get4(diag, l);
        if (diag_l != 0.0) {
            LM_UNROLL for (int k = j; k < n; ++k) sdiag[k] = 0.0;
            sdiag[j] = diag_l;
            double qtbpj = 0.0;
            LM_UNROLL for (int k = j; k < n; ++k) {
                if (sdiag[k] == 0.0) continue;
                double c, s;
                const double rkk = r[k * ldr + k];
                if (fabs(rkk) < fabs(sdiag[k])) {
                    const double cotan = rkk / sdiag[k];
                    s = 0.5 / sqrt(0.25 + 0.25 * (cotan * cotan));
                    c = s * cotan;
                } else {
                    const double tn = sdiag[k] / rkk;
                    c = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
                    s = c * tn;
                }
                r[k * ldr + k] = c * rkk + s * sdiag[k];
                const double temp = c * wa[k] + s * qtbpj;
                qtbpj = -s * wa[k] + c * qtbpj;
                wa[k] = temp;
                LM_UNROLL for (int i = k + 1; i < n; ++i) {
                    const double t2 = c * r[k * ldr + i] + s * sdiag[i];
                    sdiag[i] = -s * r[k * ldr + i] + c * sdiag[i];
                    r[k * ldr + i] = t2;
                }
            }
        }
        sdiag[j] = r[j * ldr + j];
        r[j * ldr + j] = x[j];
    }
    int nsing = n;
    LM_UNROLL for (int j = 0; j < n; ++j) {
        if (sdiag[j] == 0.0 && nsing == n) nsing = j;
        if (nsing < n) wa[j] = 0.0;
    }
    LM_UNROLL for (int j = n - 1; j >= 0; --j) {          // j = nsing - 1 .. 0
        if (j >= nsing) continue;
        double sum = 0.0;
        LM_UNROLL for (int i = j + 1; i < n; ++i) if (i < nsing) sum += r[j * ldr + i] * wa[i];
        wa[j] = (wa[j] - sum) / sdiag[j];
    }
    LM_UNROLL for (int j = 0; j < n; ++j) set4(x, ipvt[j], wa[j]);
}

LM_HD inline void lmpar(double *r, const int *ipvt, const double *diag, const double *qtb, double delta, double *par, double *x,
                        double *sdiag, double *wa1, double *wa2)
{
    constexpr int n = NP, ldr = NP;
    int nsing = n;
    LM_UNROLL for (int j = 0; j < n; ++j) {
        wa1[j] = qtb[j];
        if (r[j * ldr + j] == 0.0 && nsing == n) nsing = j;
        if (nsing < n) wa1[j] = 0.0;
    }
    LM_UNROLL for (int j = n - 1; j >= 0; --j) {          // j = nsing - 1 .. 0
        if (j >= nsing) continue;
        wa1[j] /= r[j * ldr + j];
        const double temp = wa1[j];
        LM_UNROLL for (int i = 0; i < j; ++i) wa1[i] -= r[j * ldr + i] * temp;
    }
    LM_UNROLL for (int j = 0; j < n; ++j) set4(x, ipvt[j], wa1[j]);
    int iter = 0;
    LM_UNROLL for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
    double dxnorm = enorm_n(n, wa2);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) { *par = 0.0; return; }
    double parl = 0.0;
    if (nsing >= n) {
        LM_UNROLL for (int j = 0; j < n; ++j) {
            const int l = ipvt[j];
            wa1[j] = get4(diag, l) * (get4(wa2, l) / dxnorm);
        }
        LM_UNROLL for (int j = 0; j < n; ++j) {
            double sum = 0.0;
            LM_UNROLL for (int i = 0; i < j; ++i) sum += r[j * ldr + i] * wa1[i];
            wa1[j] = (wa1[j] - sum) / r[j * ldr + j];
        }
        const double temp = enorm_n(n, wa1);
        parl = ((fp / delta) / temp) / temp;
    }
    LM_UNROLL for (int j = 0; j < n; ++j) {
        double sum = 0.0;
        LM_UNROLL for (int i = 0; i <= j; ++i) sum += r[j * ldr + i] * qtb[i];
        wa1[j] = sum / get4(diag, ipvt[j]);
    }
    const double gnorm = enorm_n(n, wa1);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = DWARF / fmin(delta, 0.1);
    *par = fmax(*par, parl);
    *par = fmin(*par, paru);
    if (*par == 0.0) *par = gnorm / dxnorm;
    for (;;) {
        ++iter;
        if (*par == 0.0) *par = fmax(DWARF, 0.001 * paru);
        double temp = sqrt(*par);
        LM_UNROLL for (int j = 0; j < n; ++j) wa1[j] = temp * diag[j];
        qrsolv(r, ipvt, wa1, qtb, x, sdiag, wa2);
        LM_UNROLL for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
        dxnorm = enorm_n(n, wa2);
        temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
        LM_UNROLL for (int j = 0; j < n; ++j) {
            const int l = ipvt[j];
            wa1[j] = get4(diag, l) * (get4(wa2, l) / dxnorm);
        }
        LM_UNROLL for (int j = 0; j < n; ++j) {
            wa1[j] /= sdiag[j];
            const double t = wa1[j];
            LM_UNROLL for (int i = j + 1; i < n; ++i) wa1[i] -= r[j * ldr + i] * t;
        }
        temp = enorm_n(n, wa1);
        const double parc = ((fp / delta) / temp) / temp;
        if (fp > 0.0) parl = fmax(parl, *par);
        if (fp < 0.0) paru = fmin(paru, *par);
        *par = fmax(parl, *par + parc);
    }
}

// ---- sample-space pieces --------------------------------------------------------------------------------------------
// residual of the model against y at the abscissae u = 1 .. m (slots >= m hold 0)
template <class B> LM_HD inline void residual(const double *p, const typename B::vec &y, int m, typename B::vec &f)
{
    for (int e = B::begin(0); e < B::end(m); ++e) {
        const double u = 1.0 + (double)B::index(e);
        const double r = (p[0] * uwip_exp(-p[1] * u) + p[2] * uwip_exp(-p[3] * u)) - y.v[e];
        f.v[e] = B::active(e, 0, m) ? r : 0.0;
    }
}

template <class B> LM_HD inline double dot_from(const typename B::vec &a, const typename B::vec &b, int from, int m)
{
    typename B::vec t;
    for (int e = B::begin(from); e < B::end(m); ++e) t.v[e] = a.v[e] * b.v[e];
    return B::sum(t, from, m);
}

template <class B> LM_HD inline void swap_cols(typename B::vec *a, int j, int k)
{
    if (j == k) return;
    for (int e = 0; e < B::W; ++e) { const double t = a[j].v[e]; a[j].v[e] = a[k].v[e]; a[k].v[e] = t; }
}

// MINPACK qrfac with column pivoting on the m x 4 Jacobian held as four vecs (column j = a[j], row i = sample slot i)
template <class B> LM_HD inline void qrfac(int m, typename B::vec *a, int *ipvt, double *rdiag, double *acnorm, double *wa)
{
    constexpr int n = NP;
    LM_UNROLL for (int j = 0; j < n; ++j) {
        acnorm[j] = enorm_m<B>(a[j], 0, m);
        rdiag[j] = acnorm[j];
        wa[j] = rdiag[j];
        ipvt[j] = j;
    }
    LM_UNROLL for (int j = 0; j < n; ++j) {
        int kmax = j;
        double rmax = rdiag[j];
        LM_UNROLL for (int k = j; k < n; ++k)
            if (rdiag[k] > rmax) { kmax = k; rmax = rdiag[k]; }
        if (kmax != j) {
            // constant indices on the device: the column vecs live in registers
            LM_UNROLL for (int k = 0; k < n; ++k) if (k == kmax) swap_cols<B>(a, j, k);
            set4(rdiag, kmax, rdiag[j]);
            set4(wa, kmax, wa[j]);
            const int t = ipvt[j]; ipvt[j] = geti4(ipvt, kmax); seti4(ipvt, kmax, t);
        }
        double ajnorm = enorm_m<B>(a[j], j, m);
        if (ajnorm != 0.0) {
            if (B::at(a[j], j) < 0.0) ajnorm = -ajnorm;
            for (int e = B::begin(j); e < B::end(m); ++e)
                if (B::active(e, j, m)) a[j].v[e] /= ajnorm;
            const double ajj = B::at(a[j], j) + 1.0;
            B::set(a[j], j, ajj);
            LM_UNROLL for (int k = j + 1; k < n; ++k) {
                const double sum = dot_from<B>(a[j], a[k], j, m);
                const double temp = sum / ajj;
                for (int e = B::begin(j); e < B::end(m); ++e)
                    if (B::active(e, j, m)) a[k].v[e] -= temp * a[j].v[e];
                if (rdiag[k] != 0.0) {
                    double t = B::at(a[k], j) / rdiag[k];
                    rdiag[k] *= sqrt(fmax(0.0, 1.0 - t * t));
                    t = rdiag[k] / wa[k];
                    if (0.05 * (t * t) <= EPSMCH) {
                        rdiag[k] = enorm_m<B>(a[k], j + 1, m);
                        wa[k] = rdiag[k];
                    }
                }
            }
        }
        rdiag[j] = -ajnorm;
    }
}

// lmdif with scipy.optimize.leastsq's defaults (ftol = xtol = 1.49012e-8, gtol = 0, maxfev = 200 (n + 1), factor 100,
// automatic scaling, epsfcn = machine epsilon).  Returns MINPACK's info code; x is updated in place.
template <class B> LM_HD inline int lmdif(const typename B::vec &y, int m, double *x)
{
    constexpr int n = NP;
    const double ftol = 1.49012e-8, xtol = 1.49012e-8, gtol = 0.0, factor = 100.0;
    const int maxfev = 200 * (n + 1);
    typename B::vec fvec{}, wa4{}, fjac[NP] = {};        // zeroed: slots >= m are swapped along with the columns
    double diag[NP], qtf[NP], wa1[NP], wa2[NP], wa3[NP], rmat[NP * NP];
    int ipvt[NP];
    int info = 0, nfev = 0;
    residual<B>(x, y, m, fvec);
    nfev = 1;
    double fnorm = enorm_m<B>(fvec, 0, m);
    double par = 0.0, delta = 0.0, xnorm = 0.0, gnorm = 0.0;
    int iter = 1;
    const double eps = sqrt(EPSMCH);
    for (;;) {
        // forward-difference Jacobian (fdjac2)
        LM_UNROLL for (int j = 0; j < n; ++j) {
            const double temp = x[j];
            double h = eps * fabs(temp);
            if (h == 0.0) h = eps;
            x[j] = temp + h;
            residual<B>(x, y, m, wa4);
            x[j] = temp;
            for (int e = B::begin(0); e < B::end(m); ++e) fjac[j].v[e] = B::active(e, 0, m) ? (wa4.v[e] - fvec.v[e]) / h : 0.0;
        }
        nfev += n;
        qrfac<B>(m, fjac, ipvt, wa1, wa2, wa3);
        if (iter == 1) {
            LM_UNROLL for (int j = 0; j < n; ++j) {
                diag[j] = wa2[j];
                if (wa2[j] == 0.0) diag[j] = 1.0;
            }
            LM_UNROLL for (int j = 0; j < n; ++j) wa3[j] = diag[j] * x[j];
            xnorm = enorm_n(n, wa3);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        for (int e = B::begin(0); e < B::end(m); ++e) wa4.v[e] = fvec.v[e];
        LM_UNROLL for (int j = 0; j < n; ++j) {
            const double fjj = B::at(fjac[j], j);
            if (fjj != 0.0) {
                const double sum = dot_from<B>(fjac[j], wa4, j, m);
                const double temp = -sum / fjj;
                for (int e = B::begin(j); e < B::end(m); ++e)
                    if (B::active(e, j, m)) wa4.v[e] += fjac[j].v[e] * temp;
            }
            B::set(fjac[j], j, wa1[j]);
            qtf[j] = B::at(wa4, j);
        }
        // the n x n upper triangle, column-major: r[j * n + i] = fjac[j][i], i <= j (lmpar overwrites its strict lower part)
        LM_UNROLL for (int j = 0; j < n; ++j)
            LM_UNROLL for (int i = 0; i < n; ++i) rmat[j * NP + i] = B::at(fjac[j], i);
        gnorm = 0.0;
        if (fnorm != 0.0) {
            LM_UNROLL for (int j = 0; j < n; ++j) {
                const double wa2l = get4(wa2, ipvt[j]);
                if (wa2l != 0.0) {
                    double sum = 0.0;
                    LM_UNROLL for (int i = 0; i <= j; ++i) sum += rmat[j * NP + i] * (qtf[i] / fnorm);
                    gnorm = fmax(gnorm, fabs(sum / wa2l));
                }
            }
        }
        if (gnorm <= gtol) { info = 4; break; }
        LM_UNROLL for (int j = 0; j < n; ++j) diag[j] = fmax(diag[j], wa2[j]);
        double ratio = 0.0;
        do {
            double sdiag[NP], scr[NP];
            lmpar(rmat, ipvt, diag, qtf, delta, &par, wa1, sdiag, wa3, scr);
            LM_UNROLL for (int j = 0; j < n; ++j) {
                wa1[j] = -wa1[j];
                wa2[j] = x[j] + wa1[j];
                wa3[j] = diag[j] * wa1[j];
            }
            const double pnorm = enorm_n(n, wa3);
            if (iter == 1) delta = fmin(delta, pnorm);
            residual<B>(wa2, y, m, wa4);
            ++nfev;
            const double fnorm1 = enorm_m<B>(wa4, 0, m);
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) {
                const double t = fnorm1 / fnorm;
                actred = 1.0 - t * t;
            }
            LM_UNROLL for (int j = 0; j < n; ++j) {
                wa3[j] = 0.0;
                const double temp = get4(wa1, ipvt[j]);
                LM_UNROLL for (int i = 0; i <= j; ++i) wa3[i] += rmat[j * NP + i] * temp;
            }
            const double temp1 = enorm_n(n, wa3) / fnorm;
            const double temp2 = (sqrt(par) * pnorm) / fnorm;
            const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
            const double dirder = -(temp1 * temp1 + temp2 * temp2);
            ratio = 0.0;
            if (prered != 0.0) ratio = actred / prered;
            if (ratio <= 0.25) {
                double temp = 0.5;
                if (actred < 0.0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                delta = temp * fmin(delta, pnorm / 0.1);
                par /= temp;
            } else if (par == 0.0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par *= 0.5;
            }
            if (ratio >= 1e-4) {
                LM_UNROLL for (int j = 0; j < n; ++j) {
                    x[j] = wa2[j];
                    wa2[j] = diag[j] * x[j];
                }
                for (int e = B::begin(0); e < B::end(m); ++e) fvec.v[e] = wa4.v[e];
                xnorm = enorm_n(n, wa2);
                fnorm = fnorm1;
                ++iter;
            }
            if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0 && info == 2) info = 3;
            if (info != 0) return info;
            if (nfev >= maxfev) info = 5;
            if (fabs(actred) <= EPSMCH && prered <= EPSMCH && 0.5 * ratio <= 1.0) info = 6;
            if (delta <= EPSMCH * xnorm) info = 7;
            if (gnorm <= EPSMCH) info = 8;
            if (info != 0) return info;
        } while (ratio < 1e-4);
    }
    return info;
}

// ---- not-a-knot cubic spline on the knots 1 .. 25 (what splrep(x22, y22), k = 3, s = 0 represents) ---------------------------
// The 25 x 25 system does not depend on the data: its elimination (pivot rows, multipliers, upper triangle) is made once
// on the host (make_spline_elim) and replayed on the right-hand side.
struct SplineElim {
    int piv[NS25];
    double f[NS25 * NS25];      // multiplier of row r at column c: f[c * 25 + r]
    double U[NS25 * NS25];      // the eliminated matrix (row-major; upper triangle used)
};

inline void make_spline_elim(SplineElim &E)
{
    constexpr int N = NS25;
    double *A = E.U;
    for (int i = 0; i < N * N; ++i) { A[i] = 0.0; E.f[i] = 0.0; }
    A[0] = 1; A[1] = -2; A[2] = 1;
    for (int i = 1; i < N - 1; ++i) { A[i * N + i - 1] = 1; A[i * N + i] = 4; A[i * N + i + 1] = 1; }
    A[(N - 1) * N + N - 3] = 1; A[(N - 1) * N + N - 2] = -2; A[(N - 1) * N + N - 1] = 1;
    for (int c = 0; c < N; ++c) {
        int piv = c;
        for (int r = c + 1; r < N; ++r)
            if (fabs(A[r * N + c]) > fabs(A[piv * N + c])) piv = r;
        E.piv[c] = piv;
        if (piv != c)
            for (int k = 0; k < N; ++k) { const double t = A[c * N + k]; A[c * N + k] = A[piv * N + k]; A[piv * N + k] = t; }
        for (int r = c + 1; r < N; ++r) {
            const double f = A[r * N + c] / A[c * N + c];
            E.f[c * N + r] = f;
            if (f == 0.0) continue;
            for (int k = c; k < N; ++k) A[r * N + k] -= f * A[c * N + k];
        }
    }
}

// y: the model at the knots 1 .. 25 (slots >= 25 ignored).  d1, d2: first / second derivative at 1, 1.5, ..., 25 (49 slots).
template <class B> LM_HD inline void spline_derivs(const SplineElim &E, const typename B::vec &y, typename B::vec &d1, typename B::vec &d2)
{
    constexpr int N = NS25;
    // right-hand side b_i = 6 (y_{i-1} - 2 y_i + y_{i+1}), i = 1 .. N-2; b_0 = b_{N-1} = 0
    const typename B::vec ym = B::gather(y, [](int e) { const int i = B::index(e); return (i >= 1 && i < MS) ? i - 1 : 0; });
    const typename B::vec yp = B::gather(y, [](int e) { const int i = B::index(e); return i + 1 < MS ? i + 1 : MS - 1; });
    typename B::vec b;
    for (int e = 0; e < B::W; ++e) {
        const int i = B::index(e);
        b.v[e] = (i >= 1 && i < N - 1) ? 6.0 * (ym.v[e] - 2.0 * y.v[e] + yp.v[e]) : 0.0;
    }
    // forward elimination replayed on b: rows are slots
    for (int c = 0; c < N; ++c) {
        const int p = E.piv[c];
        if (p != c) {
            const double bc = B::at(b, c), bp = B::at(b, p);
            for (int e = 0; e < B::W; ++e) {
                const int i = B::index(e);
                if (i == c) b.v[e] = bp;
                if (i == p) b.v[e] = bc;
            }
        }
        const double bc = B::at(b, c);
        for (int e = 0; e < B::W; ++e) {
            const int i = B::index(e);
            if (i > c && i < N) {
                const double f = E.f[c * N + i];
                if (f != 0.0) b.v[e] -= f * bc;
            }
        }
    }
    // back substitution: M_r = (b_r - sum_{k > r} U[r][k] M_k) / U[r][r]; the sum runs serially over k ascending, as the
    // host form always did (each slot would need a different subset: a butterfly buys nothing here)
    double M[N];
    for (int r = N - 1; r >= 0; --r) {
        double s = B::at(b, r);
        for (int k = r + 1; k < N; ++k) s -= E.U[r * N + k] * M[k];
        M[r] = s / E.U[r * N + r];
    }
    const typename B::vec y0v = B::gather(y, [](int e) { int i = B::index(e) / 2; if (i >= NS25 - 1) i = NS25 - 2; return i; });
    const typename B::vec y1v = B::gather(y, [](int e) { int i = B::index(e) / 2; if (i >= NS25 - 1) i = NS25 - 2; return i + 1; });
    for (int e = 0; e < B::W; ++e) {
        const int k = B::index(e);
        int i = k / 2;
        double t = (k % 2) ? 0.5 : 0.0;
        if (i >= N - 1) { i = N - 2; t = 1.0; }
        const double M0 = B::pick(M, i), M1 = B::pick(M, i + 1);
        // S(x) on [i, i+1], h = 1:  S' = (y1 - y0) - (2 M0 + M1) / 6 + M0 t + (M1 - M0) t^2 / 2
        d1.v[e] = (y1v.v[e] - y0v.v[e]) - (2.0 * M0 + M1) / 6.0 + M0 * t + (M1 - M0) * t * t / 2.0;
        d2.v[e] = M0 + (M1 - M0) * t;
    }
}

// the fitted model at the knots 1 .. 25
template <class B> LM_HD inline void model_at_knots(const double *p, typename B::vec &y22)
{
    for (int e = 0; e < B::W; ++e) {
        const double x = 1.0 + (double)B::index(e);
        y22.v[e] = p[0] * uwip_exp(-p[1] * x) + p[2] * uwip_exp(-p[3] * x);
    }
}

// DerivadaY + Curvatura for one entropy curve against the (precomputed) derivatives of the clip-limit axis fit:
// arg-max of |x' y'' - y' x''| / (x'^2 + y'^2)^1.5 over the 49 samples (the first NaN wins, as np.argmax), or -1 where
// curve_fit would raise ("Optimal parameters not found").
template <class B> LM_HD inline int knee_from_curve(const SplineElim &E, const typename B::vec &ys, const typename B::vec &xd1,
                                                   const typename B::vec &xd2, int x_info)
{
    double py[4] = {7, 0.4, 0.9, 5};
    const int iy = lmdif<B>(ys, MS, py);
    if (iy < 1 || iy > 4) return -1;
    if (x_info < 1 || x_info > 4) return -1;
    typename B::vec y22, y220, y221;
    model_at_knots<B>(py, y22);
    spline_derivs<B>(E, y22, y220, y221);
    typename B::vec kv;
    for (int e = 0; e < B::W; ++e) {
        const double k3 = xd1.v[e] * y221.v[e] - y220.v[e] * xd2.v[e];
        const double k4 = sqrt(k3 * k3);
        const double k6 = xd1.v[e] * xd1.v[e] + y220.v[e] * y220.v[e];
        const double k = k4 / sqrt(k6 * k6 * k6);
        kv.v[e] = B::index(e) < MS ? k : -1.0;          // curvatures are >= 0
    }
    const int nan_at = B::first([&](int e) { return B::index(e) < MS && kv.v[e] != kv.v[e]; });
    if (nan_at >= 0) return nan_at;
    const double best = B::max(kv);
    return B::first([&](int e) { return B::index(e) < MS && kv.v[e] == best; });
}

}  // namespace uwip_lm
