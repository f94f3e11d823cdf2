// ACLAHE parameter selection, host C++.
//
// The reference does this stage in Python with scipy
// (modules/aclahe/python/ACLAHE.py:66-129, functions.py:49-93):
//   curve_fit(f, u, Y, p0=(7,0.4,0.9,5))        -> MINPACK lmdif (leastsq defaults)
//   splrep(x22, f(x22)), splev(.., der=1|2)     -> FITPACK interpolating cubic
//   curvature |x'y'' - y'x''| / (x'^2+y'^2)^1.5 -> arg-max index
// and TODO:13 of the reference says the C++ port stalled for want of a curve
// fitting library.  This file restates the two published algorithms it needs:
// MINPACK-1 lmdif (More, Garbow, Hillstrom 1980: fdjac2, qrfac, lmpar, qrsolv,
// enorm) with scipy.optimize.leastsq's defaults (ftol = xtol = 1.49012e-8,
// gtol = 0, maxfev = 200*(n+1), factor = 100, automatic scaling), and the
// not-a-knot cubic interpolant that FITPACK's curfit returns for s = 0.
// Pinned by tests/golden/aclahe_knee.npz (indices produced by the reference's
// own functions.py under scipy 1.15.3).
#include "uwip_internal.hpp"
#include <cmath>
#include <cstring>
#include <sched.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr double EPSMCH = 2.220446049250313e-16;
constexpr double DWARF = 2.2250738585072014e-308;

// MINPACK enorm: euclidean norm guarding against over/underflow
double enorm(int n, const double *x)
{
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    double s1 = 0, s2 = 0, s3 = 0, x1max = 0, x3max = 0;
    const double agiant = rgiant / (double)n;
    for (int i = 0; i < n; ++i) {
        const double xabs = std::fabs(x[i]);
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
    if (s1 != 0.0) return x1max * std::sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0.0) {
        if (s2 >= x3max) return std::sqrt(s2 * (1.0 + (x3max / s2) * (x3max * s3)));
        return std::sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * std::sqrt(s3);
}

// exp(x) in plain IEEE double operations (no libm call, no FMA contraction): x = k ln2 + r with |r| <= ln2 / 2 (Cody-Waite
// split of ln2), exp(r) by its Taylor polynomial to r^13 (truncation < 4e-18), 2^k in two exponent steps so that results in
// the subnormal range round once.  A couple of ulp at worst.  Why not std::exp: the fit evaluates ~20 000 exponentials per
// curve and they were ~70 % of uwip_aclahe_select (0.46 ms per frame); this form is branch-free, the 49-sample loops
// vectorise (uwip_residual below is cloned for AVX2), and the same operations in the same order give the same bits on
// any host -- and on the device, should the choice move there.
static inline double uwip_exp(double x)
{
    const double xc = x < -746.0 ? -746.0 : (x > 710.0 ? 710.0 : x);      // beyond: 0 and +inf (NaN passes through)
    // k = round-to-nearest-even(x / ln2) by the 1.5 * 2^52 shift (the integer sits in the low mantissa bits): no libm
    // call and no scalar convert, so the sample loop vectorises
    const double shifted = xc * 1.4426950408889634 + 6755399441055744.0;
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
    std::memcpy(&sb, &shifted, 8);
    const int64_t k = (int64_t)(int32_t)(uint32_t)sb;  // in [-1077, 1025]
    const int64_t k1 = k >> 1, k2 = k - k1;            // each within the normal exponent range
    const uint64_t b1 = (uint64_t)(k1 + 1023) << 52, b2 = (uint64_t)(k2 + 1023) << 52;
    double s1, s2;
    std::memcpy(&s1, &b1, 8);
    std::memcpy(&s2, &b2, 8);
    const double y = (p * s1) * s2;
    return x != x ? x : y;
}

// residual of the double-exponential model: cloned for AVX2 where the host has it (the loop then runs 4 samples per
// operation; identical results: the same IEEE operations per sample, no contraction)
#if defined(__HIP_DEVICE_COMPILE__)
#define UWIP_HOST_CLONES
#else
#define UWIP_HOST_CLONES __attribute__((target_clones("avx2", "default")))
#endif
UWIP_HOST_CLONES static void uwip_residual(int m, const double *u, const double *y, const double *p, double *f)
{
    for (int i = 0; i < m; ++i)
        f[i] = (p[0] * uwip_exp(-p[1] * u[i]) + p[2] * uwip_exp(-p[3] * u[i])) - y[i];
}

// The model of functions.py:52-53 / :70-71 and its residual against the data.
struct Problem {
    int m;
    const double *u;     // abscissae 1..49
    const double *y;     // data
    void residual(const double *p, double *f) const { uwip_residual(m, u, y, p, f); }
};

constexpr int NP = 4;      // parameters
constexpr int MMAX = 64;   // data points (49 used)

// column-major fjac[j*m + i]
void qrfac(int m, int n, double *a, int *ipvt, double *rdiag, double *acnorm, double *wa)
{
    for (int j = 0; j < n; ++j) {
        acnorm[j] = enorm(m, a + (size_t)j * m);
        rdiag[j] = acnorm[j];
        wa[j] = rdiag[j];
        ipvt[j] = j;
    }
    const int minmn = m < n ? m : n;
    for (int j = 0; j < minmn; ++j) {
        int kmax = j;
        for (int k = j; k < n; ++k)
            if (rdiag[k] > rdiag[kmax]) kmax = k;
        if (kmax != j) {
            for (int i = 0; i < m; ++i) std::swap(a[(size_t)j * m + i], a[(size_t)kmax * m + i]);
            rdiag[kmax] = rdiag[j];
            wa[kmax] = wa[j];
            std::swap(ipvt[j], ipvt[kmax]);
        }
        double ajnorm = enorm(m - j, a + (size_t)j * m + j);
        if (ajnorm != 0.0) {
            if (a[(size_t)j * m + j] < 0.0) ajnorm = -ajnorm;
            for (int i = j; i < m; ++i) a[(size_t)j * m + i] /= ajnorm;
            a[(size_t)j * m + j] += 1.0;
            for (int k = j + 1; k < n; ++k) {
                double sum = 0.0;
                for (int i = j; i < m; ++i) sum += a[(size_t)j * m + i] * a[(size_t)k * m + i];
                const double temp = sum / a[(size_t)j * m + j];
                for (int i = j; i < m; ++i) a[(size_t)k * m + i] -= temp * a[(size_t)j * m + i];
                if (rdiag[k] != 0.0) {
                    double t = a[(size_t)k * m + j] / rdiag[k];
                    rdiag[k] *= std::sqrt(std::fmax(0.0, 1.0 - t * t));
                    t = rdiag[k] / wa[k];
                    if (0.05 * (t * t) <= EPSMCH) {
                        rdiag[k] = enorm(m - j - 1, a + (size_t)k * m + j + 1);
                        wa[k] = rdiag[k];
                    }
                }
            }
        }
        rdiag[j] = -ajnorm;
    }
}

// r is the n x n upper triangle stored column-major with leading dimension ldr
void qrsolv(int n, double *r, int ldr, const int *ipvt, const double *diag, const double *qtb, double *x,
            double *sdiag, double *wa)
{
    for (int j = 0; j < n; ++j) {
        for (int i = j; i < n; ++i) r[(size_t)j * ldr + i] = r[(size_t)i * ldr + j];
        x[j] = r[(size_t)j * ldr + j];
        wa[j] = qtb[j];
    }
    for (int j = 0; j < n; ++j) {
        const int l = ipvt[j];
        if (diag[l] != 0.0) {
            for (int k = j; k < n; ++k) sdiag[k] = 0.0;
            sdiag[j] = diag[l];
            double qtbpj = 0.0;
            for (int k = j; k < n; ++k) {
                if (sdiag[k] == 0.0) continue;
                double c, s;
                const double rkk = r[(size_t)k * ldr + k];
                if (std::fabs(rkk) < std::fabs(sdiag[k])) {
                    const double cotan = rkk / sdiag[k];
                    s = 0.5 / std::sqrt(0.25 + 0.25 * (cotan * cotan));
                    c = s * cotan;
                } else {
                    const double tn = sdiag[k] / rkk;
                    c = 0.5 / std::sqrt(0.25 + 0.25 * (tn * tn));
                    s = c * tn;
                }
                r[(size_t)k * ldr + k] = c * rkk + s * sdiag[k];
                const double temp = c * wa[k] + s * qtbpj;
                qtbpj = -s * wa[k] + c * qtbpj;
                wa[k] = temp;
                for (int i = k + 1; i < n; ++i) {
                    const double t2 = c * r[(size_t)k * ldr + i] + s * sdiag[i];
                    sdiag[i] = -s * r[(size_t)k * ldr + i] + c * sdiag[i];
                    r[(size_t)k * ldr + i] = t2;
                }
            }
        }
        sdiag[j] = r[(size_t)j * ldr + j];
        r[(size_t)j * ldr + j] = x[j];
    }
    int nsing = n;
    for (int j = 0; j < n; ++j) {
        if (sdiag[j] == 0.0 && nsing == n) nsing = j;
        if (nsing < n) wa[j] = 0.0;
    }
    for (int k = 1; k <= nsing; ++k) {
        const int j = nsing - k;
        double sum = 0.0;
        for (int i = j + 1; i < nsing; ++i) sum += r[(size_t)j * ldr + i] * wa[i];
        wa[j] = (wa[j] - sum) / sdiag[j];
    }
    for (int j = 0; j < n; ++j) x[ipvt[j]] = wa[j];
}

void lmpar(int n, double *r, int ldr, const int *ipvt, const double *diag, const double *qtb, double delta,
           double *par, double *x, double *sdiag, double *wa1, double *wa2)
{
    int nsing = n;
    for (int j = 0; j < n; ++j) {
        wa1[j] = qtb[j];
        if (r[(size_t)j * ldr + j] == 0.0 && nsing == n) nsing = j;
        if (nsing < n) wa1[j] = 0.0;
    }
    for (int k = 1; k <= nsing; ++k) {
        const int j = nsing - k;
        wa1[j] /= r[(size_t)j * ldr + j];
        const double temp = wa1[j];
        for (int i = 0; i < j; ++i) wa1[i] -= r[(size_t)j * ldr + i] * temp;
    }
    for (int j = 0; j < n; ++j) x[ipvt[j]] = wa1[j];
    int iter = 0;
    for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
    double dxnorm = enorm(n, wa2);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) { *par = 0.0; return; }
    double parl = 0.0;
    if (nsing >= n) {
        for (int j = 0; j < n; ++j) {
            const int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int j = 0; j < n; ++j) {
            double sum = 0.0;
            for (int i = 0; i < j; ++i) sum += r[(size_t)j * ldr + i] * wa1[i];
            wa1[j] = (wa1[j] - sum) / r[(size_t)j * ldr + j];
        }
        const double temp = enorm(n, wa1);
        parl = ((fp / delta) / temp) / temp;
    }
    for (int j = 0; j < n; ++j) {
        double sum = 0.0;
        for (int i = 0; i <= j; ++i) sum += r[(size_t)j * ldr + i] * qtb[i];
        wa1[j] = sum / diag[ipvt[j]];
    }
    const double gnorm = enorm(n, wa1);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = DWARF / std::fmin(delta, 0.1);
    *par = std::fmax(*par, parl);
    *par = std::fmin(*par, paru);
    if (*par == 0.0) *par = gnorm / dxnorm;
    for (;;) {
        ++iter;
        if (*par == 0.0) *par = std::fmax(DWARF, 0.001 * paru);
        double temp = std::sqrt(*par);
        for (int j = 0; j < n; ++j) wa1[j] = temp * diag[j];
        qrsolv(n, r, ldr, ipvt, wa1, qtb, x, sdiag, wa2);
        for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
        dxnorm = enorm(n, wa2);
        temp = fp;
        fp = dxnorm - delta;
        if (std::fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
        for (int j = 0; j < n; ++j) {
            const int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int j = 0; j < n; ++j) {
            wa1[j] /= sdiag[j];
            const double t = wa1[j];
            for (int i = j + 1; i < n; ++i) wa1[i] -= r[(size_t)j * ldr + i] * t;
        }
        temp = enorm(n, wa1);
        const double parc = ((fp / delta) / temp) / temp;
        if (fp > 0.0) parl = std::fmax(parl, *par);
        if (fp < 0.0) paru = std::fmin(paru, *par);
        *par = std::fmax(parl, *par + parc);
    }
}

// returns MINPACK's info code; x is updated in place
int lmdif(const Problem &pb, double *x)
{
    const int m = pb.m, n = NP;
    const double ftol = 1.49012e-8, xtol = 1.49012e-8, gtol = 0.0, factor = 100.0;
    const int maxfev = 200 * (n + 1);
    double fvec[MMAX], wa4[MMAX], fjac[NP * MMAX];
    double diag[NP], qtf[NP], wa1[NP], wa2[NP], wa3[NP];
    int ipvt[NP];
    int info = 0, nfev = 0;
    pb.residual(x, fvec);
    nfev = 1;
    double fnorm = enorm(m, fvec);
    double par = 0.0, delta = 0.0, xnorm = 0.0, gnorm = 0.0;
    int iter = 1;
    const double eps = std::sqrt(EPSMCH);   // epsfcn = None -> machine epsilon
    for (;;) {
        // forward-difference Jacobian (fdjac2)
        for (int j = 0; j < n; ++j) {
            const double temp = x[j];
            double h = eps * std::fabs(temp);
            if (h == 0.0) h = eps;
            x[j] = temp + h;
            pb.residual(x, wa4);
            x[j] = temp;
            for (int i = 0; i < m; ++i) fjac[(size_t)j * m + i] = (wa4[i] - fvec[i]) / h;
        }
        nfev += n;
        qrfac(m, n, fjac, ipvt, wa1, wa2, wa3);
        if (iter == 1) {
            for (int j = 0; j < n; ++j) {
                diag[j] = wa2[j];
                if (wa2[j] == 0.0) diag[j] = 1.0;
            }
            for (int j = 0; j < n; ++j) wa3[j] = diag[j] * x[j];
            xnorm = enorm(n, wa3);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        for (int i = 0; i < m; ++i) wa4[i] = fvec[i];
        for (int j = 0; j < n; ++j) {
            if (fjac[(size_t)j * m + j] != 0.0) {
                double sum = 0.0;
                for (int i = j; i < m; ++i) sum += fjac[(size_t)j * m + i] * wa4[i];
                const double temp = -sum / fjac[(size_t)j * m + j];
                for (int i = j; i < m; ++i) wa4[i] += fjac[(size_t)j * m + i] * temp;
            }
            fjac[(size_t)j * m + j] = wa1[j];
            qtf[j] = wa4[j];
        }
        gnorm = 0.0;
        if (fnorm != 0.0) {
            for (int j = 0; j < n; ++j) {
                const int l = ipvt[j];
                if (wa2[l] != 0.0) {
                    double sum = 0.0;
                    for (int i = 0; i <= j; ++i) sum += fjac[(size_t)j * m + i] * (qtf[i] / fnorm);
                    gnorm = std::fmax(gnorm, std::fabs(sum / wa2[l]));
                }
            }
        }
        if (gnorm <= gtol) { info = 4; break; }
        for (int j = 0; j < n; ++j) diag[j] = std::fmax(diag[j], wa2[j]);
        double ratio = 0.0;
        do {
            double sdiag[NP];
            lmpar(n, fjac, m, ipvt, diag, qtf, delta, &par, wa1, sdiag, wa3, wa4 /*scratch n*/);
            for (int j = 0; j < n; ++j) {
                wa1[j] = -wa1[j];
                wa2[j] = x[j] + wa1[j];
                wa3[j] = diag[j] * wa1[j];
            }
            const double pnorm = enorm(n, wa3);
            if (iter == 1) delta = std::fmin(delta, pnorm);
            pb.residual(wa2, wa4);
            ++nfev;
            const double fnorm1 = enorm(m, wa4);
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) {
                const double t = fnorm1 / fnorm;
                actred = 1.0 - t * t;
            }
            for (int j = 0; j < n; ++j) {
                wa3[j] = 0.0;
                const double temp = wa1[ipvt[j]];
                for (int i = 0; i <= j; ++i) wa3[i] += fjac[(size_t)j * m + i] * temp;
            }
            const double temp1 = enorm(n, wa3) / fnorm;
            const double temp2 = (std::sqrt(par) * pnorm) / fnorm;
            const double prered = temp1 * temp1 + temp2 * temp2 / 0.5;
            const double dirder = -(temp1 * temp1 + temp2 * temp2);
            ratio = 0.0;
            if (prered != 0.0) ratio = actred / prered;
            if (ratio <= 0.25) {
                double temp = 0.5;
                if (actred < 0.0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                delta = temp * std::fmin(delta, pnorm / 0.1);
                par /= temp;
            } else if (par == 0.0 || ratio >= 0.75) {
                delta = pnorm / 0.5;
                par *= 0.5;
            }
            if (ratio >= 1e-4) {
                for (int j = 0; j < n; ++j) {
                    x[j] = wa2[j];
                    wa2[j] = diag[j] * x[j];
                }
                for (int i = 0; i < m; ++i) fvec[i] = wa4[i];
                xnorm = enorm(n, wa2);
                fnorm = fnorm1;
                ++iter;
            }
            if (std::fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0) info = 1;
            if (delta <= xtol * xnorm) info = 2;
            if (std::fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0 && info == 2) info = 3;
            if (info != 0) return info;
            if (nfev >= maxfev) info = 5;
            if (std::fabs(actred) <= EPSMCH && prered <= EPSMCH && 0.5 * ratio <= 1.0) info = 6;
            if (delta <= EPSMCH * xnorm) info = 7;
            if (gnorm <= EPSMCH) info = 8;
            if (info != 0) return info;
        } while (ratio < 1e-4);
    }
    return info;
}

// Not-a-knot cubic interpolant on the uniform grid x = 1..N (h = 1): what
// splrep(x22, y22) (k = 3, s = 0) represents.  Returns first and second
// derivatives at x = 1, 1.5, ..., N  (splev(x222, tck, der=1|2)).
void spline_derivs(const double *y, int N, double *d1, double *d2)
{
    // second derivatives M_i: M_{i-1} + 4 M_i + M_{i+1} = 6 (y_{i-1} - 2 y_i + y_{i+1}), i = 1..N-2,
    // not-a-knot: M_0 - 2 M_1 + M_2 = 0 and M_{N-3} - 2 M_{N-2} + M_{N-1} = 0.
    // Dense Gaussian elimination with partial pivoting (N = 25).  The matrix does not depend on the data, so its
    // elimination -- pivot rows, multipliers, the upper triangle -- is done once per N and replayed on the right-hand
    // side: the same operations on b in the same order as eliminating [A | b] every time (bit-identical), N^2 instead
    // of N^3 work and no allocation per curve.
    struct Elim {
        int N = 0;
        std::vector<double> U;          // the eliminated matrix (upper triangle used)
        std::vector<int> piv;           // row swapped into position c
        std::vector<double> f;          // multiplier of row r at column c: f[c * N + r]
    };
    static thread_local Elim cache;
    if (cache.N != N) {
        Elim e;
        e.N = N;
        std::vector<double> &A = e.U;
        A.assign((size_t)N * N, 0.0);
        e.piv.assign(N, 0);
        e.f.assign((size_t)N * N, 0.0);
        A[0] = 1; A[1] = -2; A[2] = 1;
        for (int i = 1; i < N - 1; ++i) { A[(size_t)i * N + i - 1] = 1; A[(size_t)i * N + i] = 4; A[(size_t)i * N + i + 1] = 1; }
        A[(size_t)(N - 1) * N + N - 3] = 1; A[(size_t)(N - 1) * N + N - 2] = -2; A[(size_t)(N - 1) * N + N - 1] = 1;
        for (int c = 0; c < N; ++c) {
            int piv = c;
            for (int r = c + 1; r < N; ++r)
                if (std::fabs(A[(size_t)r * N + c]) > std::fabs(A[(size_t)piv * N + c])) piv = r;
            e.piv[c] = piv;
            if (piv != c)
                for (int k = 0; k < N; ++k) std::swap(A[(size_t)c * N + k], A[(size_t)piv * N + k]);
            for (int r = c + 1; r < N; ++r) {
                const double f = A[(size_t)r * N + c] / A[(size_t)c * N + c];
                e.f[(size_t)c * N + r] = f;
                if (f == 0.0) continue;
                for (int k = c; k < N; ++k) A[(size_t)r * N + k] -= f * A[(size_t)c * N + k];
            }
        }
        cache = std::move(e);
    }
    const std::vector<double> &A = cache.U;
    double b[64], M[64];
    if (N > 64) return;
    b[0] = 0.0; b[N - 1] = 0.0;
    for (int i = 1; i < N - 1; ++i) b[i] = 6.0 * (y[i - 1] - 2.0 * y[i] + y[i + 1]);
    for (int c = 0; c < N; ++c) {
        if (cache.piv[c] != c) std::swap(b[c], b[cache.piv[c]]);
        for (int r = c + 1; r < N; ++r) {
            const double f = cache.f[(size_t)c * N + r];
            if (f == 0.0) continue;
            b[r] -= f * b[c];
        }
    }
    for (int r = N - 1; r >= 0; --r) {
        double s = b[r];
        for (int k = r + 1; k < N; ++k) s -= A[(size_t)r * N + k] * M[k];
        M[r] = s / A[(size_t)r * N + r];
    }
    const int NS = 2 * N - 1;
    for (int k = 0; k < NS; ++k) {
        int i = k / 2;
        double t = (k % 2) ? 0.5 : 0.0;
        if (i == N - 1) { i = N - 2; t = 1.0; }
        // S(x) on [i, i+1], h = 1:  S' = (y1 - y0) - (2 M0 + M1)/6 + M0 t + (M1 - M0) t^2 / 2
        const double y0 = y[i], y1 = y[i + 1], M0 = M[i], M1 = M[i + 1];
        d1[k] = (y1 - y0) - (2.0 * M0 + M1) / 6.0 + M0 * t + (M1 - M0) * t * t / 2.0;
        d2[k] = M0 + (M1 - M0) * t;
    }
}

inline double model(const double *p, double x) { return p[0] * uwip_exp(-p[1] * x) + p[2] * uwip_exp(-p[3] * x); }

// DerivadaY + DerivadaX + Curvatura (functions.py:49-93); -1 when curve_fit would raise
// DerivadaX (functions.py:67-80): the clip-limit axis fit does not depend on the image, so for the
// standard grid 0.5 .. 24.5 it is computed once per process.
struct XFit {
    int info;
    double d1[49], d2[49];
};

XFit fit_x(const float *xs)
{
    XFit r{};
    double u[49], xd[49];
    for (int i = 0; i < 49; ++i) { u[i] = 1.0 + i; xd[i] = (double)xs[i]; }
    double px[4] = {7, 0.4, 0.9, 5};
    const Problem PX{49, u, xd};
    r.info = lmdif(PX, px);
    double x22v[25];
    for (int i = 0; i < 25; ++i) x22v[i] = model(px, 1.0 + i);
    spline_derivs(x22v, 25, r.d1, r.d2);
    return r;
}

const XFit &standard_xfit()
{
    static const XFit cached = [] {
        float xs[49];
        for (int i = 0; i < 49; ++i) xs[i] = 0.5f * (float)(i + 1);
        return fit_x(xs);
    }();
    return cached;
}

int knee_index(const float *xs, const float *ys)
{
    double u[49], yd[49];
    bool standard = true;
    for (int i = 0; i < 49; ++i) {
        u[i] = 1.0 + i;
        yd[i] = (double)ys[i];
        standard = standard && (xs[i] == 0.5f * (float)(i + 1));
    }
    double py[4] = {7, 0.4, 0.9, 5};
    const Problem PY{49, u, yd};
    const int iy = lmdif(PY, py);
    if (iy < 1 || iy > 4) return -1;          // curve_fit: "Optimal parameters not found"
    XFit local;
    const XFit *xf = &local;
    if (standard) xf = &standard_xfit();
    else local = fit_x(xs);
    if (xf->info < 1 || xf->info > 4) return -1;
    double y22[25];
    for (int i = 0; i < 25; ++i) y22[i] = model(py, 1.0 + i);
    double y220[49], y221[49];
    spline_derivs(y22, 25, y220, y221);
    const double *y223 = xf->d1, *y225 = xf->d2;
    int best = 0;
    double bestk = 0.0;
    bool have = false;
    for (int i = 0; i < 49; ++i) {
        const double k3 = y223[i] * y221[i] - y220[i] * y225[i];
        const double k4 = std::sqrt(k3 * k3);
        const double k6 = y223[i] * y223[i] + y220[i] * y220[i];
        const double k = k4 / std::sqrt(k6 * k6 * k6);
        if (k != k) return i;                 // np.argmax returns the first NaN
        if (!have || k > bestk) { bestk = k; best = i; have = true; }
    }
    return best;
}

// float32 -> float16 (round to nearest even) -> float, for the BS comparison of ACLAHE.py:102
float through_half(float f)
{
    _Float16 h = (_Float16)f;
    return (float)h;
}

// ---- persistent host worker pool -----------------------------------------------------------------------------------------
// One per process (= per rank), created on first use and shared by all contexts: min(16, CPUs in the process's affinity
// mask - 1) workers (UWIP_HOST_THREADS overrides), so eight ranks that each pinned themselves to their GPU's NUMA node
// share the host without oversubscribing it, and a step creates no threads.  parallel_for splits [0, n) into chunks that
// the workers AND the calling thread pull from a shared counter; several callers (one per sub-batch stream) may be inside
// at once.
class HostPool {
public:
    static HostPool &get()
    {
        static HostPool pool;
        return pool;
    }
    void parallel_for(int n, const std::function<void(int, int)> &fn)
    {
        if (n <= 0) return;
        const int workers = (int)th_.size();
        if (n < 4 || workers == 0) { fn(0, n); return; }
        auto job = std::make_shared<Job>();
        job->fn = &fn;
        job->n = n;
        job->grain = std::max(1, n / (4 * (workers + 1)));
        job->chunks = (n + job->grain - 1) / job->grain;
        job->left.store(job->chunks);
        const int helpers = std::min(workers, job->chunks - 1);
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (int i = 0; i < helpers; ++i) q_.push_back(job);
        }
        if (helpers == 1) cv_.notify_one(); else cv_.notify_all();
        run(*job);                                           // the caller works too
        std::unique_lock<std::mutex> lk(job->mu);
        job->cv.wait(lk, [&] { return job->left.load() == 0; });
    }

private:
    struct Job {
        const std::function<void(int, int)> *fn = nullptr;
        int n = 0, grain = 1, chunks = 0;
        std::atomic<int> next{0}, left{0};
        std::mutex mu;
        std::condition_variable cv;
    };
    static void run(Job &j)
    {
        for (;;) {
            const int c = j.next.fetch_add(1);
            if (c >= j.chunks) return;
            const int a = c * j.grain, b = std::min(j.n, a + j.grain);
            (*j.fn)(a, b);
            if (j.left.fetch_sub(1) == 1) {
                std::lock_guard<std::mutex> lk(j.mu);
                j.cv.notify_all();
            }
        }
    }
    HostPool()
    {
        int want = -1;
        if (const char *e = std::getenv("UWIP_HOST_THREADS")) want = std::atoi(e);
        budget_ = cpu_budget(&ranks_);
        if (want < 0) {
            // the caller works too, so budget - 1 helpers; never more than 16 (a sub-batch has 64 frames: the selection
            // is ~0.3 ms each, more threads only add wake-ups)
            want = std::min(16, std::max(0, (int)std::floor(budget_ + 1e-9) - 1));
        }
        for (int i = 0; i < want; ++i) th_.emplace_back([this] { loop(); });
    }
    // CPUs this rank may really use: min(affinity mask, cgroup CPU quota) / ranks on the node.  The affinity mask alone
    // overstates it: the GPU boxes show 256 CPUs in the mask under a cgroup quota of 16, and eight ranks of a node share it.
    static double cpu_budget(int *ranks_out)
    {
        double cpus = 0;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = CPU_COUNT(&set);
        if (cpus <= 0) cpus = (double)std::thread::hardware_concurrency();
        const char *root = std::getenv("UWIP_CGROUP_ROOT");           // tests point this at a fake tree
        const std::string base = root && *root ? root : "/sys/fs/cgroup";
        double quota = 0;
        if (FILE *f = std::fopen((base + "/cpu.max").c_str(), "r")) {          // cgroup v2: "<quota|max> <period>"
            char q[64] = {0};
            double per = 0;
            if (std::fscanf(f, "%63s %lf", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) quota = std::atof(q) / per;
            std::fclose(f);
        } else {
            double q = 0, per = 0;
            if (FILE *a = std::fopen((base + "/cpu/cpu.cfs_quota_us").c_str(), "r")) { if (std::fscanf(a, "%lf", &q) != 1) q = 0; std::fclose(a); }
            if (FILE *b = std::fopen((base + "/cpu/cpu.cfs_period_us").c_str(), "r")) { if (std::fscanf(b, "%lf", &per) != 1) per = 0; std::fclose(b); }
            if (q > 0 && per > 0) quota = q / per;
        }
        if (quota > 0 && quota < cpus) cpus = quota;
        int ranks = 1;
        for (const char *name : {"UWIP_RANKS_ON_NODE", "LOCAL_WORLD_SIZE"})
            if (const char *e = std::getenv(name)) { const int v = std::atoi(e); if (v > 0) { ranks = v; break; } }
        if (ranks_out) *ranks_out = ranks;
        return std::max(1.0, cpus / ranks);
    }
public:
    int workers() const { return (int)th_.size(); }
    double budget() const { return budget_; }
    int ranks() const { return ranks_; }
private:
    double budget_ = 1.0;
    int ranks_ = 1;
    ~HostPool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void loop()
    {
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                job = q_.front();
                q_.pop_front();
            }
            run(*job);
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<Job>> q_;
    bool stop_ = false;
};

}  // namespace

UWIP_API int uwip_host_pool_info(int *workers, double *cpu_budget, int *ranks_on_node)
{
    HostPool &p = HostPool::get();
    if (workers) *workers = p.workers();
    if (cpu_budget) *cpu_budget = p.budget();
    if (ranks_on_node) *ranks_on_node = p.ranks();
    return UWIP_OK;
}

UWIP_API int uwip_aclahe_knee(const float *h_xs49, const float *h_ys49, int32_t *index)
{
    if (!h_xs49 || !h_ys49 || !index) return UWIP_ERR_INVALID;
    *index = knee_index(h_xs49, h_ys49);
    return UWIP_OK;
}

// h_entropy [frames][5][51] -> per frame BS (block size), CL (clip limit, an index as the reference
// uses it).  h_knee (optional) [frames][5].  h_need_eval (optional) [frames]: 1 when 2*CL > 50, i.e.
// the BS choice needs entropies at a clip limit outside the swept grid (the caller supplies them
// through h_extra [frames][5], used when h_extra_valid[frame] != 0).
int uwip_aclahe_select_internal(const float *h_entropy, int frames, int32_t *h_bs, int32_t *h_cl, int32_t *h_knee,
                                int32_t *h_need_eval, const float *h_extra, const int32_t *h_extra_valid);

UWIP_API int uwip_aclahe_select(const float *h_entropy, int frames, int32_t *h_bs, int32_t *h_cl, int32_t *h_knee)
{
    return uwip_aclahe_select_internal(h_entropy, frames, h_bs, h_cl, h_knee, nullptr, nullptr, nullptr);
}

int uwip_aclahe_select_internal(const float *h_entropy, int frames, int32_t *h_bs, int32_t *h_cl, int32_t *h_knee,
                                int32_t *h_need_eval, const float *h_extra, const int32_t *h_extra_valid)
{
    static const int BlockSize[5] = {2, 4, 8, 16, 32};
    if (frames < 0 || (frames > 0 && (!h_entropy || !h_bs || !h_cl))) return UWIP_ERR_INVALID;
    const std::function<void(int, int)> work = [&](int f0, int f1) {
        for (int f = f0; f < f1; ++f) {
            const float *tab = h_entropy + (size_t)f * 5 * 51;
            float xs[49];
            for (int i = 0; i < 49; ++i) xs[i] = 0.5f * (float)(i + 1);
            int d = -1;
            for (int g = 0; g < 5; ++g) {
                const int k = knee_index(xs, tab + (size_t)g * 51 + 1);
                if (h_knee) h_knee[(size_t)f * 5 + g] = k;
                if (k > d) d = k;
            }
            if (d < 0) d = 0;
            const bool outside = 2 * d > 50;
            const bool have_extra = outside && h_extra && h_extra_valid && h_extra_valid[f];
            if (h_need_eval) h_need_eval[f] = (outside && !have_extra) ? 1 : 0;
            int w = 0;
            float best = 0.f;
            for (int g = 0; g < 5; ++g) {
                const float e = have_extra ? h_extra[(size_t)f * 5 + g] : tab[(size_t)g * 51 + (outside ? 50 : 2 * d)];
                const float h = through_half(e);
                if (g == 0 || h >= best) { best = h; w = g; }      // last maximum wins (ACLAHE.py:118-124)
            }
            h_bs[f] = BlockSize[w];
            h_cl[f] = d;
        }
    };
    // ~0.3 ms of Levenberg-Marquardt per frame: spread over the process's persistent host pool (no thread is created
    // per call; the pool is sized from the CPUs this rank may run on, see HostPool)
    HostPool::get().parallel_for(frames, work);
    return UWIP_OK;
}
