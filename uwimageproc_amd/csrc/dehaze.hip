// bgdehaze hot path for gfx950, float64.
//
// modules/bgdehaze/BGDehaze.py + guidedfilter.py in the reference are
// per-pixel Python loops over float64 arrays.  Here every stage is a kernel
// over a batch of frames; all real-valued state is float64 (MI355X has the
// FP64 rate and the HBM to afford it, and it keeps the result within ~1e-12 of
// the reference instead of the ~1e-4 an fp32 pipeline would give).
//
// Because the input is 8-bit, normI = (I - min)/(max - min) takes at most 256
// distinct values, and the window max/min filters of D1/D2 commute with that
// monotone map: they run on the uint8 planes and the float64 value is looked up
// afterwards (exact, not an approximation).
//
// The guided filter (radius 40 box filters) is in guided_filter_ws.hip: two
// streaming kernels, vertical sums in registers, horizontal sums by a wave
// prefix scan; every mean divides by the analytic in-image window size, as
// guidedfilter.py:67 does.
#include "uwip_internal.hpp"
#include "device_utils.hpp"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {

// per-frame scalar slots (double)
enum {
    SC_MN = 0, SC_MX = 1, SC_B0 = 2, SC_B1 = 3, SC_B2 = 4,
    SC_JMIN0 = 5, SC_JMAX0 = 6, SC_JMIN1 = 7, SC_JMAX1 = 8,
    SC_MEANJ0 = 9, SC_MEANJ1 = 10, SC_MEANR = 11, SC_COEFF = 12, SC_RMIN = 13, SC_RMAX = 14,
    SC_YJMN = 16, SC_YJMX = 17, SC_YIMN = 18, SC_YIMX = 19, SC_OMN = 20, SC_OMX = 21,
    SC_COUNT = 32
};
// per-frame integer slots
enum { SI_MN = 0, SI_MX = 1, SI_RMN = 2, SI_RMX = 3, SI_YJMN = 4, SI_YJMX = 5, SI_YIMN = 6, SI_YIMX = 7,
       SI_IDX0 = 8, SI_IDX1 = 9, SI_COUNT = 16 };

constexpr int RED_BLOCKS = 128;   // blocks per frame for the streaming reduction kernels

__device__ __forceinline__ double normv(int v, int mn, int mx) { return (double)(v - mn) / (double)(mx - mn); }
// An 8-bit sample takes 256 values, so every per-sample expression of the frame scalars is a 256-entry table: the
// block's 256 threads evaluate it once (same operations, same roundings) and the pixel loop looks it up in LDS
// instead of running float64 divisions per pixel.  Call with all 256 threads; ends with a barrier.
// (row, column) of the pixels a thread visits in the loop  for (i = block * 256 + thread; i < n; i += grid * 256):
// one 32-bit division up front, then a step of (stride / W, stride % W) -- not a 64-bit division per pixel.
struct RowCol {
    int x, y, dq, dr, W;
    __device__ __forceinline__ explicit RowCol(int W_) : W(W_)
    {
        const uint32_t S = gridDim.x * 256u, i0 = blockIdx.x * 256u + threadIdx.x;
        dq = (int)(S / (uint32_t)W);
        dr = (int)(S - (uint32_t)dq * (uint32_t)W);
        y = (int)(i0 / (uint32_t)W);
        x = (int)(i0 - (uint32_t)y * (uint32_t)W);
    }
    __device__ __forceinline__ void step()
    {
        x += dr; y += dq;
        if (x >= W) { x -= W; ++y; }
    }
};

template <class F>
__device__ __forceinline__ void fill_table256(double *tab, F f)
{
    tab[threadIdx.x] = f((int)threadIdx.x);
    __syncthreads();
}

__device__ __forceinline__ double block_reduce_f64(double v, int op /*0 sum 1 min 2 max*/, double *scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double o = __shfl_xor(v, d, 64);
        v = op == 0 ? v + o : (op == 1 ? fmin(v, o) : fmax(v, o));
    }
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    double r = scratch[0];
    for (int w = 1; w < nw; ++w) r = op == 0 ? r + scratch[w] : (op == 1 ? fmin(r, scratch[w]) : fmax(r, scratch[w]));
    return r;
}

// ---- D0: global u8 min/max (all channels) and red-channel min/max -------------
__global__ __launch_bounds__(256) void k_dz_minmax(const uint8_t *__restrict__ img, size_t step, size_t fs,
                                                   int H, int W, int *__restrict__ si)
{
    const int f = blockIdx.y;
    const uint8_t *b = img + (size_t)f * fs;
    int mn = 255, mx = 0, rmn = 255, rmx = 0;
    const long long n = (long long)H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i - (long long)y * W);
        const uint8_t *p = b + (size_t)y * step + (size_t)x * 3;
        const int B = p[0], G = p[1], R = p[2];
        mn = min(mn, min(B, min(G, R)));
        mx = max(mx, max(B, max(G, R)));
        rmn = min(rmn, R);
        rmx = max(rmx, R);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mn = min(mn, __shfl_xor(mn, d, 64)); mx = max(mx, __shfl_xor(mx, d, 64));
        rmn = min(rmn, __shfl_xor(rmn, d, 64)); rmx = max(rmx, __shfl_xor(rmx, d, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        int *s = si + (size_t)f * SI_COUNT;
        atomicMin(&s[SI_MN], mn); atomicMax(&s[SI_MX], mx);
        atomicMin(&s[SI_RMN], rmn); atomicMax(&s[SI_RMX], rmx);
    }
}

__global__ void k_dz_init_scalars(int *si, int F)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    int *s = si + (size_t)f * SI_COUNT;
    s[SI_MN] = 255; s[SI_MX] = 0; s[SI_RMN] = 255; s[SI_RMX] = 0;
    s[SI_YJMN] = 255; s[SI_YJMX] = 0; s[SI_YIMN] = 255; s[SI_YIMX] = 0;
    s[SI_IDX0] = 0; s[SI_IDX1] = 0;
}

// ---- window max / min on the uint8 planes (D1, D2) ------------------------------
// out[f][c][y][x] = max/min over the IN-IMAGE part of rows [y-pad, y-pad+w) x cols [x-pad, x-pad+w)
constexpr int WF_TW = 64, WF_TH = 16;
template <bool IS_MAX>
__global__ __launch_bounds__(256) void k_winfilter(const uint8_t *__restrict__ img, size_t step, size_t fs,
                                                   int H, int W, int w, int pad,
                                                   uint8_t *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mem[];
    const int RW = WF_TW + w - 1, RH = WF_TH + w - 1;
    uint8_t *s_in = s_mem;                        // [3][RH][RW]
    uint8_t *s_h = s_mem + (size_t)3 * RH * RW;   // [3][RH][WF_TW]
    const int f = blockIdx.z;
    const int x0 = blockIdx.x * WF_TW, y0 = blockIdx.y * WF_TH;
    const uint8_t *b = img + (size_t)f * fs;
    const uint8_t ident = IS_MAX ? 0 : 255;
    for (int i = threadIdx.x; i < RH * RW; i += 256) {
        const int ry = i / RW, rx = i - ry * RW;
        const int y = y0 - pad + ry, x = x0 - pad + rx;
        uint8_t v0 = ident, v1 = ident, v2 = ident;
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const uint8_t *p = b + (size_t)y * step + (size_t)x * 3;
            v0 = p[0]; v1 = p[1]; v2 = p[2];
        }
        s_in[i] = v0; s_in[RH * RW + i] = v1; s_in[2 * RH * RW + i] = v2;
    }
    __syncthreads();
    // Each thread produces 4 adjacent outputs: the w-4+1 samples common to their windows are reduced once and
    // every output only adds its few private samples (for w = 15: 23 ops and 18 LDS bytes per 4 outputs
    // instead of 56 and 60).  Falls back to the plain loop for w < 4.
    const bool blocked = w >= 4;
    for (int i = threadIdx.x; i < 3 * RH * (WF_TW / 4); i += 256) {
        const int c = i / (RH * (WF_TW / 4)), rem = i - c * RH * (WF_TW / 4);
        const int ry = rem / (WF_TW / 4), tx = (rem - ry * (WF_TW / 4)) * 4;
        const uint8_t *row = s_in + ((size_t)c * RH + ry) * RW + tx;
        uint8_t o[4];
        if (blocked) {
            uint8_t m = ident;
            for (int k = 3; k < w; ++k) m = IS_MAX ? max(m, row[k]) : min(m, row[k]);      // shared by all four windows
            const uint8_t a0 = row[0], a1 = row[1], a2 = row[2], b0 = row[w], b1 = row[w + 1], b2 = row[w + 2];
            if (IS_MAX) {
                o[0] = max(max(m, a0), max(a1, a2)); o[1] = max(max(m, a1), max(a2, b0));
                o[2] = max(max(m, a2), max(b0, b1)); o[3] = max(max(m, b0), max(b1, b2));
            } else {
                o[0] = min(min(m, a0), min(a1, a2)); o[1] = min(min(m, a1), min(a2, b0));
                o[2] = min(min(m, a2), min(b0, b1)); o[3] = min(min(m, b0), min(b1, b2));
            }
        } else {
            for (int j = 0; j < 4; ++j) {
                uint8_t m = ident;
                for (int k = 0; k < w; ++k) m = IS_MAX ? max(m, row[j + k]) : min(m, row[j + k]);
                o[j] = m;
            }
        }
        uint8_t *dst = s_h + ((size_t)c * RH + ry) * WF_TW + tx;
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * (WF_TH / 4) * WF_TW; i += 256) {
        const int c = i / ((WF_TH / 4) * WF_TW), rem = i - c * (WF_TH / 4) * WF_TW;
        const int ty = (rem / WF_TW) * 4, tx = rem - (rem / WF_TW) * WF_TW;
        const uint8_t *col = s_h + ((size_t)c * RH + ty) * WF_TW + tx;
        uint8_t o[4];
        if (blocked) {
            uint8_t m = ident;
            for (int k = 3; k < w; ++k) m = IS_MAX ? max(m, col[(size_t)k * WF_TW]) : min(m, col[(size_t)k * WF_TW]);
            const uint8_t a0 = col[0], a1 = col[WF_TW], a2 = col[2 * WF_TW];
            const uint8_t b0 = col[(size_t)w * WF_TW], b1 = col[(size_t)(w + 1) * WF_TW], b2 = col[(size_t)(w + 2) * WF_TW];
            if (IS_MAX) {
                o[0] = max(max(m, a0), max(a1, a2)); o[1] = max(max(m, a1), max(a2, b0));
                o[2] = max(max(m, a2), max(b0, b1)); o[3] = max(max(m, b0), max(b1, b2));
            } else {
                o[0] = min(min(m, a0), min(a1, a2)); o[1] = min(min(m, a1), min(a2, b0));
                o[2] = min(min(m, a2), min(b0, b1)); o[3] = min(min(m, b0), min(b1, b2));
            }
        } else {
            for (int j = 0; j < 4; ++j) {
                uint8_t m = ident;
                for (int k = 0; k < w; ++k) m = IS_MAX ? max(m, col[(size_t)(j + k) * WF_TW]) : min(m, col[(size_t)(j + k) * WF_TW]);
                o[j] = m;
            }
        }
        const int x = x0 + tx;
        if (x >= W) continue;
        for (int j = 0; j < 4; ++j) {
            const int y = y0 + ty + j;
            if (y < H) out[(((size_t)f * 3 + c) * H + y) * W + x] = o[j];
        }
    }
}

// ---- D1: background light -----------------------------------------------------
// D0 = mxR - mxB, D1 = mxR - mxG (BGDehaze.py:19-21); arg-min with first-index ties.
__global__ __launch_bounds__(256) void k_bglight_partial(const uint8_t *__restrict__ mx, int H, int W,
                                                         const int *__restrict__ si,
                                                         double *__restrict__ pval, int *__restrict__ pidx)
{
    __shared__ double s_v[2][256];
    __shared__ int s_i[2][256];
    const int f = blockIdx.y;
    const int mn = si[(size_t)f * SI_COUNT + SI_MN], mxv = si[(size_t)f * SI_COUNT + SI_MX];
    const size_t n = (size_t)H * W;
    const uint8_t *pB = mx + (size_t)f * 3 * n, *pG = pB + n, *pR = pG + n;
    double best0 = 1e300, best1 = 1e300;
    int i0 = 0x7fffffff, i1 = 0x7fffffff;
    __shared__ double s_T[256];
    fill_table256(s_T, [&](int v) { return normv(v, mn, mxv); });
    if ((n & 3) == 0 && (((uintptr_t)mx) & 3) == 0) {
        // four pixels per thread and load (the planes are dense: n bytes each), visited in index order
        const uint32_t *qB = reinterpret_cast<const uint32_t *>(pB), *qG = reinterpret_cast<const uint32_t *>(pG),
                       *qR = reinterpret_cast<const uint32_t *>(pR);
        for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < n / 4; g += (size_t)gridDim.x * 256) {
            const uint32_t wb = qB[g], wg = qG[g], wr = qR[g];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double r = s_T[(wr >> (8 * k)) & 255u];
                const double d0 = r - s_T[(wb >> (8 * k)) & 255u];
                const double d1 = r - s_T[(wg >> (8 * k)) & 255u];
                if (d0 < best0) { best0 = d0; i0 = (int)(4 * g) + k; }     // indices increase per thread: keeps the first
                if (d1 < best1) { best1 = d1; i1 = (int)(4 * g) + k; }
            }
        }
    } else
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double r = s_T[pR[i]];
        const double d0 = r - s_T[pB[i]];
        const double d1 = r - s_T[pG[i]];
        if (d0 < best0) { best0 = d0; i0 = (int)i; }     // i increases per thread: keeps the first
        if (d1 < best1) { best1 = d1; i1 = (int)i; }
    }
    s_v[0][threadIdx.x] = best0; s_i[0][threadIdx.x] = i0;
    s_v[1][threadIdx.x] = best1; s_i[1][threadIdx.x] = i1;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const double ov = s_v[k][threadIdx.x + s];
                const int oi = s_i[k][threadIdx.x + s];
                if (ov < s_v[k][threadIdx.x] || (ov == s_v[k][threadIdx.x] && oi < s_i[k][threadIdx.x])) {
                    s_v[k][threadIdx.x] = ov; s_i[k][threadIdx.x] = oi;
                }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const size_t o = ((size_t)f * gridDim.x + blockIdx.x) * 2;
        pval[o] = s_v[0][0]; pidx[o] = s_i[0][0];
        pval[o + 1] = s_v[1][0]; pidx[o + 1] = s_i[1][0];
    }
}

__global__ void k_bglight_final(const double *__restrict__ pval, const int *__restrict__ pidx, int nb,
                                const uint8_t *__restrict__ img, size_t step, size_t fs, int W,
                                int *__restrict__ si, double *__restrict__ sc)
{
    const int f = blockIdx.x;
    double b0 = 1e300, b1 = 1e300;
    int i0 = 0x7fffffff, i1 = 0x7fffffff;
    for (int b = (int)threadIdx.x; b < nb; b += 64) {        // launched with one wave; (value, index) order = first-index ties
        const size_t o = ((size_t)f * nb + b) * 2;
        if (pval[o] < b0 || (pval[o] == b0 && pidx[o] < i0)) { b0 = pval[o]; i0 = pidx[o]; }
        if (pval[o + 1] < b1 || (pval[o + 1] == b1 && pidx[o + 1] < i1)) { b1 = pval[o + 1]; i1 = pidx[o + 1]; }
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        const double v0 = __shfl_xor(b0, sft, 64), v1 = __shfl_xor(b1, sft, 64);
        const int j0 = __shfl_xor(i0, sft, 64), j1 = __shfl_xor(i1, sft, 64);
        if (v0 < b0 || (v0 == b0 && j0 < i0)) { b0 = v0; i0 = j0; }
        if (v1 < b1 || (v1 == b1 && j1 < i1)) { b1 = v1; i1 = j1; }
    }
    if (threadIdx.x != 0) return;
    int *s = si + (size_t)f * SI_COUNT;
    double *d = sc + (size_t)f * SC_COUNT;
    // A constant frame (max == min) normalises to 0/0 = NaN everywhere: no difference ever compares below the running
    // minimum and the indices keep their initial value.  numpy's argmin returns the first NaN, i.e. pixel 0.
    if (i0 == 0x7fffffff) i0 = 0;
    if (i1 == 0x7fffffff) i1 = 0;
    s[SI_IDX0] = i0; s[SI_IDX1] = i1;
    const int mn = s[SI_MN], mx = s[SI_MX];
    d[SC_MN] = mn; d[SC_MX] = mx;
    const uint8_t *p0 = img + (size_t)f * fs + (size_t)(i0 / W) * step + (size_t)(i0 % W) * 3;
    const uint8_t *p1 = img + (size_t)f * fs + (size_t)(i1 / W) * step + (size_t)(i1 % W) * 3;
    for (int c = 0; c < 3; ++c)                                   // np.average of the two pixels (:26)
        d[SC_B0 + c] = (normv(p0[c], mn, mx) + normv(p1[c], mn, mx)) / 2.0;
}

// ---- D2 + clamp: p_c = max(1 - min_window(I_c/B_c), tmin), c = blue, green -----------
__global__ __launch_bounds__(256) void k_transmission(const uint8_t *__restrict__ mnp, int H, int W, int w,
                                                      int pad, const int *__restrict__ si,
                                                      const double *__restrict__ sc, double tmin,
                                                      double *__restrict__ P, double *__restrict__ traw)
{
    const int f = blockIdx.y;
    const int mn = si[(size_t)f * SI_COUNT + SI_MN], mx = si[(size_t)f * SI_COUNT + SI_MX];
    const double B0 = sc[(size_t)f * SC_COUNT + SC_B0], B1 = sc[(size_t)f * SC_COUNT + SC_B1];
    const size_t n = (size_t)H * W;
    const uint8_t *m0 = mnp + (size_t)f * 3 * n, *m1 = m0 + n;
    __shared__ double s_Q0[256], s_Q1[256];
    s_Q0[threadIdx.x] = normv((int)threadIdx.x, mn, mx) / B0;
    fill_table256(s_Q1, [&](int v) { return normv(v, mn, mx) / B1; });
    RowCol rcw(W);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256, rcw.step()) {
        const int y = rcw.y, x = rcw.x;
        // zero padding (BGDehaze.py:32): a window that leaves the image contains a 0
        const bool inside = (y - pad >= 0) && (y - pad + w <= H) && (x - pad >= 0) && (x - pad + w <= W);
        const double q0 = inside ? s_Q0[m0[i]] : 0.0;
        const double q1 = inside ? s_Q1[m1[i]] : 0.0;
        const double t0 = 1.0 - q0, t1 = 1.0 - q1;
        if (traw) { traw[(size_t)f * 2 * n + i] = t0; traw[(size_t)f * 2 * n + n + i] = t1; }
        P[(size_t)f * 2 * n + i] = fmax(t0, tmin);
        P[(size_t)f * 2 * n + n + i] = fmax(t1, tmin);
    }
}

// ---- D5: scene recovery J_c = (I_c - B_c)/t_c + B_c, in place over Q; partial min/max, sum of I_r ----
__global__ __launch_bounds__(256) void k_recover(const uint8_t *__restrict__ img, size_t step, size_t fs,
                                                 const int *__restrict__ si, const double *__restrict__ sc,
                                                 double *__restrict__ Q, int H, int W,
                                                 double *__restrict__ part /*[F][nb][5]*/)
{
    __shared__ double scratch[4];
    const int f = blockIdx.y;
    const size_t n = (size_t)H * W;
    const int mn = si[(size_t)f * SI_COUNT + SI_MN], mx = si[(size_t)f * SI_COUNT + SI_MX];
    const double B0 = sc[(size_t)f * SC_COUNT + SC_B0], B1 = sc[(size_t)f * SC_COUNT + SC_B1];
    double *q0 = Q + (size_t)f * 2 * n, *q1 = q0 + n;
    const uint8_t *b = img + (size_t)f * fs;
    double mn0 = 1e300, mx0 = -1e300, mn1 = 1e300, mx1 = -1e300, sr = 0.0;
    __shared__ double s_T[256];
    fill_table256(s_T, [&](int v) { return normv(v, mn, mx); });
    RowCol rcw(W);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256, rcw.step()) {
        const int y = rcw.y, x = rcw.x;
        const uint8_t *p = b + (size_t)y * step + (size_t)x * 3;
        const double j0 = (s_T[p[0]] - B0) / q0[i] + B0;
        const double j1 = (s_T[p[1]] - B1) / q1[i] + B1;
        q0[i] = j0; q1[i] = j1;
        mn0 = fmin(mn0, j0); mx0 = fmax(mx0, j0); mn1 = fmin(mn1, j1); mx1 = fmax(mx1, j1);
        sr += s_T[p[2]];
    }
    double *o = part + ((size_t)f * gridDim.x + blockIdx.x) * 5;
    const double a = block_reduce_f64(mn0, 1, scratch), bb = block_reduce_f64(mx0, 2, scratch);
    const double c = block_reduce_f64(mn1, 1, scratch), d = block_reduce_f64(mx1, 2, scratch);
    const double e = block_reduce_f64(sr, 0, scratch);
    if (threadIdx.x == 0) { o[0] = a; o[1] = bb; o[2] = c; o[3] = d; o[4] = e; }
}

__global__ void k_recover_final(const double *__restrict__ part, int nb, double *__restrict__ sc, double npix)
{
    const int f = blockIdx.x;
    if (threadIdx.x != 0) return;
    double a = 1e300, b = -1e300, c = 1e300, d = -1e300, e = 0.0;
    for (int k = 0; k < nb; ++k) {
        const double *p = part + ((size_t)f * nb + k) * 5;
        a = fmin(a, p[0]); b = fmax(b, p[1]); c = fmin(c, p[2]); d = fmax(d, p[3]); e += p[4];
    }
    double *s = sc + (size_t)f * SC_COUNT;
    s[SC_JMIN0] = a; s[SC_JMAX0] = b; s[SC_JMIN1] = c; s[SC_JMAX1] = d; s[SC_MEANR] = e / npix;
}

// Scene recovery fused into the guided filter (uwip_gf_recover): per-block (min, max, sum) triples of J for both planes.
// The frame scalars that k_recover / k_normJ and their finalisers produced come from them and from the exact integer
// sum of the red bytes, each mean with ONE rounding of an exactly known total instead of n roundings:
//   mean((v - mn) / (mx - mn))      = (sum(v) - n mn) / (mx - mn) / n
//   mean((J - Jmin) / (Jmax - Jmin)) = (sum(J) - n Jmin) / (Jmax - Jmin) / n     (BGDehaze.py:54-64)
__global__ void k_recover_final2(const double *__restrict__ jpart, int nb, const unsigned long long *__restrict__ redsum,
                                 const int *__restrict__ si, double *__restrict__ sc, double npix)
{
    const int f = blockIdx.x;
    // the wave (launched with 64 threads) shares the nb partials of the frame's two planes: a single thread walking them is a
    // chain of dependent loads (a few hundred partials at 4K)
    double a = 1e300, b = -1e300, c = 1e300, d = -1e300, s0 = 0.0, s1 = 0.0;
    for (int k = (int)threadIdx.x; k < nb; k += 64) {
        const double *p0 = jpart + ((size_t)(2 * f) * nb + k) * 3, *p1 = jpart + ((size_t)(2 * f + 1) * nb + k) * 3;
        a = fmin(a, p0[0]); b = fmax(b, p0[1]); s0 += p0[2];
        c = fmin(c, p1[0]); d = fmax(d, p1[1]); s1 += p1[2];
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        a = fmin(a, __shfl_xor(a, sft, 64)); b = fmax(b, __shfl_xor(b, sft, 64)); s0 += __shfl_xor(s0, sft, 64);
        c = fmin(c, __shfl_xor(c, sft, 64)); d = fmax(d, __shfl_xor(d, sft, 64)); s1 += __shfl_xor(s1, sft, 64);
    }
    if (threadIdx.x != 0) return;
    const int *ii = si + (size_t)f * SI_COUNT;
    const int mn = ii[SI_MN], mx = ii[SI_MX];
    const double sumr = (double)((long long)redsum[f] - (long long)npix * mn);
    double *s = sc + (size_t)f * SC_COUNT;
    s[SC_JMIN0] = a; s[SC_JMAX0] = b; s[SC_JMIN1] = c; s[SC_JMAX1] = d;
    s[SC_MEANR] = sumr / (double)(mx - mn) / npix;
    s[SC_MEANJ0] = (s0 - npix * a) / (b - a) / npix;
    s[SC_MEANJ1] = (s1 - npix * c) / (d - c) / npix;
    // red-channel compensation coefficient and its min-max, as k_normJ_final
    const double avgRr = 1.5 - s[SC_MEANJ0] - s[SC_MEANJ1];
    const double coeff = avgRr / s[SC_MEANR];
    s[SC_COEFF] = coeff;
    const double lo = normv(ii[SI_RMN], mn, mx) * coeff, hi = normv(ii[SI_RMX], mn, mx) * coeff;
    s[SC_RMIN] = fmin(lo, hi); s[SC_RMAX] = fmax(lo, hi);
}

// partial sums for the means of the min-max normalised J (BGDehaze.py:54,56,61).  J stays as it is in Q: every consumer
// applies (J - min) / (max - min) itself (restored_px), which saves rewriting two float64 planes.
__global__ __launch_bounds__(256) void k_normJ(const double *__restrict__ Q, const double *__restrict__ sc, int H, int W,
                                               double *__restrict__ part /*[F][nb][2]*/)
{
    __shared__ double scratch[4];
    const int f = blockIdx.y;
    const size_t n = (size_t)H * W;
    const double *s = sc + (size_t)f * SC_COUNT;
    const double a0 = s[SC_JMIN0], d0 = s[SC_JMAX0] - s[SC_JMIN0], a1 = s[SC_JMIN1], d1 = s[SC_JMAX1] - s[SC_JMIN1];
    const double *q0 = Q + (size_t)f * 2 * n, *q1 = q0 + n;
    double s0 = 0.0, s1 = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double v0 = (q0[i] - a0) / d0, v1 = (q1[i] - a1) / d1;
        s0 += v0; s1 += v1;
    }
    const double r0 = block_reduce_f64(s0, 0, scratch), r1 = block_reduce_f64(s1, 0, scratch);
    if (threadIdx.x == 0) {
        double *o = part + ((size_t)f * gridDim.x + blockIdx.x) * 2;
        o[0] = r0; o[1] = r1;
    }
}

// red-channel compensation coefficient and its min-max (BGDehaze.py:61-64)
__global__ void k_normJ_final(const double *__restrict__ part, int nb, const int *__restrict__ si,
                              double *__restrict__ sc, double npix)
{
    const int f = blockIdx.x;
    if (threadIdx.x != 0) return;
    double s0 = 0.0, s1 = 0.0;
    for (int k = 0; k < nb; ++k) { s0 += part[((size_t)f * nb + k) * 2]; s1 += part[((size_t)f * nb + k) * 2 + 1]; }
    double *s = sc + (size_t)f * SC_COUNT;
    const int *ii = si + (size_t)f * SI_COUNT;
    s[SC_MEANJ0] = s0 / npix; s[SC_MEANJ1] = s1 / npix;
    const double avgRr = 1.5 - s[SC_MEANJ0] - s[SC_MEANJ1];
    const double coeff = avgRr / s[SC_MEANR];
    s[SC_COEFF] = coeff;
    // Rrec = I_r * coeff is monotone in the 8-bit red value: its extrema sit at the red min / max
    const double lo = normv(ii[SI_RMN], ii[SI_MN], ii[SI_MX]) * coeff;
    const double hi = normv(ii[SI_RMX], ii[SI_MN], ii[SI_MX]) * coeff;
    s[SC_RMIN] = fmin(lo, hi); s[SC_RMAX] = fmax(lo, hi);
}

// the compensated, min-max normalised red channel as a function of the 8-bit red value (BGDehaze.py:61-64)
__device__ __forceinline__ void fill_red_table(double *tab, const double *s, int mn, int mx)
{
    fill_table256(tab, [&](int v) {
        const double rrec = normv(v, mn, mx) * s[SC_COEFF];
        return (rrec - s[SC_RMIN]) / (s[SC_RMAX] - s[SC_RMIN]);
    });
}
__device__ __forceinline__ void restored_px(const uint8_t *p, const double *nJ0, const double *nJ1, size_t i,
                                            const double *s, const double *red_tab, double out[3])
{
    out[0] = (nJ0[i] - s[SC_JMIN0]) / (s[SC_JMAX0] - s[SC_JMIN0]);   // min-max normalised J (BGDehaze.py:54,56)
    out[1] = (nJ1[i] - s[SC_JMIN1]) / (s[SC_JMAX1] - s[SC_JMIN1]);
    out[2] = red_tab[p[2]];
}

__device__ __forceinline__ uint8_t f64_to_u8_rne(double v)
{
    // cv::saturate_cast<uchar>(double): cvRound (RNE) then saturate; NaN -> 0
    if (!(v == v)) return 0;
    const double r = rint(v);
    return (uint8_t)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));
}

// RC_correction output -> uint8 (imwrite(restored*255), main.py:19) and/or float64 tap
__global__ __launch_bounds__(256) void k_rc_out(const uint8_t *__restrict__ img, size_t step, size_t fs,
                                                const int *__restrict__ si, const double *__restrict__ sc,
                                                const double *__restrict__ Q, int H, int W,
                                                uint8_t *__restrict__ out, size_t ostep, size_t ofs,
                                                double *__restrict__ tap /*[F][H][W][3] or null*/)
{
    const int f = blockIdx.y;
    const size_t n = (size_t)H * W;
    const int mn = si[(size_t)f * SI_COUNT + SI_MN], mx = si[(size_t)f * SI_COUNT + SI_MX];
    const double *s = sc + (size_t)f * SC_COUNT;
    const double *q0 = Q + (size_t)f * 2 * n, *q1 = q0 + n;
    __shared__ double s_red[256];
    fill_red_table(s_red, s, mn, mx);
    RowCol rcw(W);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256, rcw.step()) {
        const int y = rcw.y, x = rcw.x;
        const uint8_t *p = img + (size_t)f * fs + (size_t)y * step + (size_t)x * 3;
        double v[3];
        restored_px(p, q0, q1, i, s, s_red, v);
        if (tap) { double *t = tap + ((size_t)f * n + i) * 3; t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; }
        if (out) {
            uint8_t *o = out + (size_t)f * ofs + (size_t)y * ostep + (size_t)x * 3;
            o[0] = f64_to_u8_rne(v[0] * 255); o[1] = f64_to_u8_rne(v[1] * 255); o[2] = f64_to_u8_rne(v[2] * 255);
        }
    }
}

// ---- D6: adaptive exposure map (BGDehaze.py:71-89) ------------------------------------------
__device__ __forceinline__ void bgr2ycrcb(int b, int g, int r, int &Y, int &Cr, int &Cb)
{
    Y = (b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14;
    Cr = ((r - Y) * 11682 + (128 << 14) + (1 << 13)) >> 14;
    Cb = ((b - Y) * 9241 + (128 << 14) + (1 << 13)) >> 14;
    Y = min(max(Y, 0), 255); Cr = min(max(Cr, 0), 255); Cb = min(max(Cb, 0), 255);
}

// (x*255).astype(uint8): C truncation of a non-negative double
__device__ __forceinline__ int trunc_u8(double v) { return (int)(unsigned char)(long long)(v * 255); }

__global__ __launch_bounds__(256) void k_exp_prep(const uint8_t *__restrict__ img, size_t step, size_t fs,
                                                  int *__restrict__ si, const double *__restrict__ sc,
                                                  const double *__restrict__ Q, int H, int W,
                                                  uint8_t *__restrict__ YI /*[F][H][W][3]*/,
                                                  uint8_t *__restrict__ YJ /*[F][H][W]*/)
{
    const int f = blockIdx.y;
    const size_t n = (size_t)H * W;
    int *ii = si + (size_t)f * SI_COUNT;
    const int mn = ii[SI_MN], mx = ii[SI_MX];
    const double *s = sc + (size_t)f * SC_COUNT;
    const double *q0 = Q + (size_t)f * 2 * n, *q1 = q0 + n;
    int jmn = 255, jmx = 0, imn = 255, imx = 0;
    __shared__ double s_red[256];
    __shared__ int s_u8[256];     // (normalised input * 255).astype(uint8) per 8-bit value
    s_u8[threadIdx.x] = trunc_u8(normv((int)threadIdx.x, mn, mx));
    fill_red_table(s_red, s, mn, mx);
    RowCol rcw(W);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256, rcw.step()) {
        const int y = rcw.y, x = rcw.x;
        const uint8_t *p = img + (size_t)f * fs + (size_t)y * step + (size_t)x * 3;
        double v[3];
        restored_px(p, q0, q1, i, s, s_red, v);
        int Yj, Crj, Cbj, Yi, Cri, Cbi;
        bgr2ycrcb(trunc_u8(v[0]), trunc_u8(v[1]), trunc_u8(v[2]), Yj, Crj, Cbj);
        bgr2ycrcb(s_u8[p[0]], s_u8[p[1]], s_u8[p[2]], Yi, Cri, Cbi);
        uint8_t *o = YI + ((size_t)f * n + i) * 3;
        o[0] = (uint8_t)Yi; o[1] = (uint8_t)Cri; o[2] = (uint8_t)Cbi;
        YJ[(size_t)f * n + i] = (uint8_t)Yj;
        jmn = min(jmn, min(Yj, min(Crj, Cbj))); jmx = max(jmx, max(Yj, max(Crj, Cbj)));
        imn = min(imn, min(Yi, min(Cri, Cbi))); imx = max(imx, max(Yi, max(Cri, Cbi)));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        jmn = min(jmn, __shfl_xor(jmn, d, 64)); jmx = max(jmx, __shfl_xor(jmx, d, 64));
        imn = min(imn, __shfl_xor(imn, d, 64)); imx = max(imx, __shfl_xor(imx, d, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&ii[SI_YJMN], jmn); atomicMax(&ii[SI_YJMX], jmx);
        atomicMin(&ii[SI_YIMN], imn); atomicMax(&ii[SI_YIMX], imx);
    }
}

// S = (Yj*Yi + 0.3*Yi^2) / (Yj^2 + 0.3*Yi^2)   (BGDehaze.py:83)
__global__ __launch_bounds__(256) void k_exp_S(const uint8_t *__restrict__ YI, const uint8_t *__restrict__ YJ,
                                               const int *__restrict__ si, size_t n, double *__restrict__ S,
                                               int guard)
{
    const int f = blockIdx.y;
    const int *ii = si + (size_t)f * SI_COUNT;
    const int jmn = ii[SI_YJMN], jmx = ii[SI_YJMX], imn = ii[SI_YIMN], imx = ii[SI_YIMX];
    __shared__ double s_Ti[256], s_Tj[256];
    s_Ti[threadIdx.x] = normv((int)threadIdx.x, imn, imx);
    fill_table256(s_Tj, [&](int v) { return normv(v, jmn, jmx); });
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double Yi = s_Ti[YI[((size_t)f * n + i) * 3]];
        const double Yj = s_Tj[YJ[(size_t)f * n + i]];
        const double num = Yj * Yi + 0.3 * (Yi * Yi), den = Yj * Yj + 0.3 * (Yi * Yi);
        // as written, 0/0 = NaN poisons the whole frame; the guard is an opt-in deviation
        S[(size_t)f * n + i] = (guard && den == 0.0) ? 1.0 : num / den;
    }
}

// OutputExp = restored * refinedS : PASS 0 = partial min/max, PASS 1 = normalise + write
// HIST (PASS 1 only): also count the bytes it writes into hist[f][channel][256] -- the histogram histretch needs of this
// very image -- in per-wave LDS bins that drain behind the f64 stream, so the chained stage skips its own read pass.
template <int PASS, bool HIST>
__global__ __launch_bounds__(256) void k_exp_out(const uint8_t *__restrict__ img, size_t step, size_t fs,
                                                 const int *__restrict__ si, double *__restrict__ sc,
                                                 const double *__restrict__ Q, const double *__restrict__ RS,
                                                 int H, int W, double *__restrict__ part,
                                                 uint8_t *__restrict__ out, size_t ostep, size_t ofs,
                                                 double *__restrict__ tap, uint32_t *__restrict__ hist)
{
    __shared__ double scratch[4];
    __shared__ uint32_t s_hist[HIST ? 4 * 768 : 1];
    uint32_t *my_hist = s_hist + (HIST ? (threadIdx.x >> 6) * 768 : 0);
    if (HIST) for (int i = threadIdx.x; i < 4 * 768; i += 256) s_hist[i] = 0;   // fill_red_table's barrier orders this
    const int f = blockIdx.y;
    const size_t n = (size_t)H * W;
    const int mn = si[(size_t)f * SI_COUNT + SI_MN], mx = si[(size_t)f * SI_COUNT + SI_MX];
    const double *s = sc + (size_t)f * SC_COUNT;
    const double *q0 = Q + (size_t)f * 2 * n, *q1 = q0 + n;
    const double omn = s[SC_OMN], od = s[SC_OMX] - s[SC_OMN];
    double lo = 1e300, hi = -1e300, nanflag = 0.0;
    __shared__ double s_red[256];
    fill_red_table(s_red, s, mn, mx);
    RowCol rcw(W);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256, rcw.step()) {
        const int y = rcw.y, x = rcw.x;
        const uint8_t *p = img + (size_t)f * fs + (size_t)y * step + (size_t)x * 3;
        double v[3];
        restored_px(p, q0, q1, i, s, s_red, v);
        const double rs = RS[(size_t)f * n + i];
        v[0] *= rs; v[1] *= rs; v[2] *= rs;
        if (PASS == 0) {
            if (!(v[0] == v[0]) || !(v[1] == v[1]) || !(v[2] == v[2])) nanflag = 1.0;   // numpy min/max propagate NaN
            lo = fmin(lo, fmin(v[0], fmin(v[1], v[2])));
            hi = fmax(hi, fmax(v[0], fmax(v[1], v[2])));
        } else {
            v[0] = (v[0] - omn) / od; v[1] = (v[1] - omn) / od; v[2] = (v[2] - omn) / od;
            if (tap) { double *t = tap + ((size_t)f * n + i) * 3; t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; }
            if (out) {
                uint8_t *o = out + (size_t)f * ofs + (size_t)y * ostep + (size_t)x * 3;
                const uint8_t b0 = f64_to_u8_rne(v[0] * 255), b1 = f64_to_u8_rne(v[1] * 255), b2 = f64_to_u8_rne(v[2] * 255);
                o[0] = b0; o[1] = b1; o[2] = b2;
                if (HIST) { atomicAdd(&my_hist[b0], 1u); atomicAdd(&my_hist[256 + b1], 1u); atomicAdd(&my_hist[512 + b2], 1u); }
            }
        }
    }
    if (HIST) {
        __syncthreads();
        uint32_t *oh = hist + (size_t)f * 768;
        for (int i = threadIdx.x; i < 768; i += 256) {
            const uint32_t c = s_hist[i] + s_hist[768 + i] + s_hist[2 * 768 + i] + s_hist[3 * 768 + i];
            if (c) atomicAdd(&oh[i], c);
        }
    }
    if (PASS == 0) {
        const double a = block_reduce_f64(lo, 1, scratch), b = block_reduce_f64(hi, 2, scratch);
        const double c = block_reduce_f64(nanflag, 2, scratch);
        if (threadIdx.x == 0) {
            double *o = part + ((size_t)f * gridDim.x + blockIdx.x) * 3;
            o[0] = a; o[1] = b; o[2] = c;
        }
    }
}

__global__ void k_exp_out_final(const double *__restrict__ part, int nb, double *__restrict__ sc)
{
    const int f = blockIdx.x;
    double a = 1e300, b = -1e300, c = 0.0;
    for (int k = (int)threadIdx.x; k < nb; k += 64) {        // launched with one wave
        const double *p = part + ((size_t)f * nb + k) * 3;
        a = fmin(a, p[0]); b = fmax(b, p[1]); c = fmax(c, p[2]);
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        a = fmin(a, __shfl_xor(a, sft, 64)); b = fmax(b, __shfl_xor(b, sft, 64)); c = fmax(c, __shfl_xor(c, sft, 64));
    }
    if (threadIdx.x != 0) return;
    if (c != 0.0) a = b = __longlong_as_double(0x7ff8000000000000ll);
    sc[(size_t)f * SC_COUNT + SC_OMN] = a;
    sc[(size_t)f * SC_COUNT + SC_OMX] = b;
}

__global__ void k_set_B(double *sc, const double *B, int F)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    for (int c = 0; c < 3; ++c) sc[(size_t)f * SC_COUNT + SC_B0 + c] = B[(size_t)f * 3 + c];
}

__global__ void k_get_B(const double *sc, const int *si, double *B, int *idx, int F)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    for (int c = 0; c < 3; ++c) B[(size_t)f * 3 + c] = sc[(size_t)f * SC_COUNT + SC_B0 + c];
    if (idx) { idx[(size_t)f * 2] = si[(size_t)f * SI_COUNT + SI_IDX0]; idx[(size_t)f * 2 + 1] = si[(size_t)f * SI_COUNT + SI_IDX1]; }
}

// ------------------------------------------------------------------------------------------
struct DzBufs {
    int *si; double *sc; double *part; int *pidx;
    uint8_t *u8planes;       // [F][3][H][W] window max/min
    double *P, *AB, *Q;
    uint8_t *u8min = nullptr;   // window-min planes when the fused 15x15 kernel produced them already
    unsigned long long *redsum = nullptr;   // ... and the exact per-frame sum of the red bytes
};

// guided filter with a normalised-u8 guide: P [F][np] planes -> Q [F][np] planes
// The guided filter itself lives in guided_filter_ws.hip (wave-strip kernels).
int guided_filter_u8(uwip_ctx *ctx, const uint8_t *guide, size_t step, size_t fs, const int *gnorm, int gstride,
                     const double *P, double *Q, double *AB, int F, int np, int H, int W, int r, double eps,
                     const uwip_gf_pu8 *pu8 = nullptr, uwip_gf_recover *rec = nullptr)
{
    UWIP_REQUIRE(ctx, H >= 2 * r + 1 && W >= 2 * r + 1, "guided filter needs rows, cols >= 2r+1 (guidedfilter.py:39-41)");
    UWIP_REQUIRE(ctx, r <= 96, "guided filter radius > 96 is not supported");
    return uwip_gf_wave_strip(ctx, guide, step, fs, gnorm, gstride, P, Q, AB, F, np, H, W, r, eps, pu8, rec);
}

int alloc_bufs(uwip_ctx *ctx, int F, int H, int W, DzBufs *b)
{
    const size_t n = (size_t)H * W;
    b->si = (int *)uwip_ws(ctx, "dz.si", sizeof(int) * SI_COUNT * F);
    b->sc = (double *)uwip_ws(ctx, "dz.sc", sizeof(double) * SC_COUNT * F);
    b->part = (double *)uwip_ws(ctx, "dz.part", sizeof(double) * 8 * RED_BLOCKS * F);
    b->pidx = (int *)uwip_ws(ctx, "dz.pidx", sizeof(int) * 2 * RED_BLOCKS * F);
    b->u8planes = (uint8_t *)uwip_ws(ctx, "dz.u8", (size_t)4 * n * F);
    b->P = (double *)uwip_ws(ctx, "dz.P", sizeof(double) * 2 * n * F);
    b->AB = (double *)uwip_ws(ctx, "dz.AB", sizeof(double) * 8 * n * F);
    b->Q = (double *)uwip_ws(ctx, "dz.Q", sizeof(double) * 2 * n * F);
    if (!b->si || !b->sc || !b->part || !b->pidx || !b->u8planes || !b->P || !b->AB || !b->Q)
        return UWIP_ERR_NOMEM;
    return UWIP_OK;
}

int check_in(uwip_ctx *ctx, const uwip_batch_u8 *in, int w)
{
    int rc = uwip_check_batch(ctx, in, 3);
    if (rc) return rc;
    UWIP_REQUIRE(ctx, w >= 1 && w <= 63, "window must be in [1,63]");
    UWIP_REQUIRE(ctx, in->frames <= 16384, "too many frames for one launch");
    return UWIP_OK;
}

// stages D0-D1: scalars + background light
int run_bglight(uwip_ctx *ctx, const uwip_batch_u8 *in, int w, DzBufs &b, const double *d_B_inject, bool also_min = false)
{
    const int F = in->frames, H = in->rows, W = in->cols;
    const uint8_t *img = (const uint8_t *)in->data;
    k_dz_init_scalars<<<uwip_cdiv(F, 64), 64, 0, ctx->stream>>>(b.si, F);
    const bool fast15 = uwip_winfilter15_ok(img, in->step, in->frame_stride, H, W, w);
    static_assert(SI_MN == 0 && SI_MX == 1 && SI_RMN == 2 && SI_RMX == 3, "k_winfilter15 writes the four scalars in this order");
    if (!(fast15 && also_min)) {   // otherwise k_winfilter15 gathers the frame min / max while it filters
        uwip_kscope ks(ctx, "k_dz_minmax");
        k_dz_minmax<<<dim3(RED_BLOCKS, F), 256, 0, ctx->stream>>>(img, in->step, in->frame_stride, H, W, b.si);
    }
    const int pad = w / 2;
    b.u8min = nullptr;
    b.redsum = nullptr;
    if (fast15) {
        // one pass for the window maximum and (when the transmission follows) the window minimum
        uint8_t *mn = nullptr;
        if (also_min) {
            mn = (uint8_t *)uwip_ws(ctx, "dz.u8min", (size_t)3 * H * W * F);
            if (!mn) return UWIP_ERR_NOMEM;
        }
        unsigned long long *rs = nullptr;
        if (mn) {
            rs = (unsigned long long *)uwip_ws(ctx, "dz.redsum", sizeof(unsigned long long) * F);
            if (!rs) return UWIP_ERR_NOMEM;
        }
        const int rc = uwip_winfilter15(ctx, img, in->step, in->frame_stride, F, H, W, b.u8planes, mn, mn ? b.si : nullptr, SI_COUNT, rs);
        if (rc) return rc;
        b.u8min = mn;
        b.redsum = rs;
    } else {
        const size_t lds = (size_t)3 * (WF_TH + w - 1) * (WF_TW + w - 1) + (size_t)3 * (WF_TH + w - 1) * WF_TW;
        uwip_kscope ks(ctx, "k_winfilter<max>");
        k_winfilter<true><<<dim3(uwip_cdiv(W, WF_TW), uwip_cdiv(H, WF_TH), F), 256, lds, ctx->stream>>>(
            img, in->step, in->frame_stride, H, W, w, pad, b.u8planes);
    }
    {
        uwip_kscope ks(ctx, "k_bglight");
        k_bglight_partial<<<dim3(RED_BLOCKS, F), 256, 0, ctx->stream>>>(b.u8planes, H, W, b.si, b.part, b.pidx);
        k_bglight_final<<<F, 64, 0, ctx->stream>>>(b.part, b.pidx, RED_BLOCKS, img, in->step, in->frame_stride, W, b.si, b.sc);
    }
    if (d_B_inject) k_set_B<<<uwip_cdiv(F, 64), 64, 0, ctx->stream>>>(b.sc, d_B_inject, F);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

int run_transmission(uwip_ctx *ctx, const uwip_batch_u8 *in, DzBufs &b, double tmin, double *d_traw)
{
    const int F = in->frames, H = in->rows, W = in->cols;
    const uint8_t *img = (const uint8_t *)in->data;
    const int w = 15, pad = 7;                                   // refined_t drops w (BGDehaze.py:52, B-10)
    const uint8_t *mnp = b.u8min;
    if (!mnp) {
        mnp = b.u8planes;
        if (uwip_winfilter15_ok(img, in->step, in->frame_stride, H, W, w)) {
            const int rc = uwip_winfilter15(ctx, img, in->step, in->frame_stride, F, H, W, nullptr, b.u8planes);
            if (rc) return rc;
        } else {
            const size_t lds = (size_t)3 * (WF_TH + w - 1) * (WF_TW + w - 1) + (size_t)3 * (WF_TH + w - 1) * WF_TW;
            uwip_kscope ks(ctx, "k_winfilter<min>");
            k_winfilter<false><<<dim3(uwip_cdiv(W, WF_TW), uwip_cdiv(H, WF_TH), F), 256, lds, ctx->stream>>>(
                img, in->step, in->frame_stride, H, W, w, pad, b.u8planes);
        }
    }
    {
        uwip_kscope ks(ctx, "k_transmission");
        k_transmission<<<dim3(512, F), 256, 0, ctx->stream>>>(mnp, H, W, w, pad, b.si, b.sc, tmin, b.P, d_traw);
    }
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

}  // namespace

// ---- exported entry points ---------------------------------------------------------------

UWIP_API int uwip_dehaze_background_light(uwip_ctx *ctx, const uwip_batch_u8 *in, int w, double *d_B, int32_t *d_idx)
{
    int rc = check_in(ctx, in, w);
    if (rc) return rc;
    if (in->frames == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, !uwip_batch_empty(in) && d_B, "empty image or null output");
    DzBufs b;
    rc = alloc_bufs(ctx, in->frames, in->rows, in->cols, &b);
    if (rc) return rc;
    rc = run_bglight(ctx, in, w, b, nullptr);
    if (rc) return rc;
    k_get_B<<<uwip_cdiv(in->frames, 64), 64, 0, ctx->stream>>>(b.sc, b.si, d_B, d_idx, in->frames);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

UWIP_API int uwip_dehaze_transmission(uwip_ctx *ctx, const uwip_batch_u8 *in, const double *d_B, double *d_t)
{
    int rc = check_in(ctx, in, 15);
    if (rc) return rc;
    if (in->frames == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, !uwip_batch_empty(in) && d_B && d_t, "empty image or null buffer");
    DzBufs b;
    rc = alloc_bufs(ctx, in->frames, in->rows, in->cols, &b);
    if (rc) return rc;
    rc = run_bglight(ctx, in, 15, b, d_B, true);
    if (rc) return rc;
    return run_transmission(ctx, in, b, 0.2, d_t);
}

UWIP_API int uwip_guided_filter(uwip_ctx *ctx, const uwip_batch_u8 *guide, const double *d_p, int r, double eps,
                                double *d_q)
{
    int rc = check_in(ctx, guide, 15);
    if (rc) return rc;
    if (guide->frames == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, !uwip_batch_empty(guide) && d_p && d_q, "empty image or null buffer");
    UWIP_REQUIRE(ctx, r >= 1, "radius must be >= 1");
    DzBufs b;
    const int F = guide->frames, H = guide->rows, W = guide->cols;
    rc = alloc_bufs(ctx, F, H, W, &b);
    if (rc) return rc;
    k_dz_init_scalars<<<uwip_cdiv(F, 64), 64, 0, ctx->stream>>>(b.si, F);
    {
        uwip_kscope ks(ctx, "k_dz_minmax");
        k_dz_minmax<<<dim3(RED_BLOCKS, F), 256, 0, ctx->stream>>>((const uint8_t *)guide->data, guide->step,
                                                               guide->frame_stride, H, W, b.si);
    }
    return guided_filter_u8(ctx, (const uint8_t *)guide->data, guide->step, guide->frame_stride, b.si, SI_COUNT, d_p,
                            d_q, b.AB, F, 1, H, W, r, eps);
}

// d_hist_out (internal): [F][3][256] counts of the bytes written to `out`, zeroed by the caller; FULL form only
int uwip_dehaze_internal(uwip_ctx *ctx, const uwip_batch_u8 *in, const uwip_batch_u8 *out, int w, int flags,
                         const double *d_B_inject, double *d_refined_t, double *d_float_out, uint32_t *d_hist_out)
{
    const int full = (flags & UWIP_DEHAZE_FULL) != 0, guard = (flags & UWIP_DEHAZE_GUARD_S) != 0;
    int rc = check_in(ctx, in, w);
    if (rc) return rc;
    if (out) {
        rc = uwip_check_batch(ctx, out, 3);
        if (rc) return rc;
        UWIP_REQUIRE(ctx, in->rows == out->rows && in->cols == out->cols && in->frames == out->frames, "in/out shape mismatch");
    }
    if (in->frames == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, !uwip_batch_empty(in), "empty image");
    const int F = in->frames, H = in->rows, W = in->cols;
    const size_t n = (size_t)H * W;
    const int r = 40;
    const double eps = 1e-3, tmin = 0.2;
    DzBufs b;
    rc = alloc_bufs(ctx, F, H, W, &b);
    if (rc) return rc;
    const uint8_t *img = (const uint8_t *)in->data;
    rc = run_bglight(ctx, in, w, b, d_B_inject, w == 15);
    if (rc) return rc;
    bool fused_recover = false;
    if (b.u8min && uwip_gf_pu8_ok(img, in->step, in->frame_stride, b.u8min, 2, W, r)) {
        // the transmission is a per-frame function of the 8-bit window-minimum planes: the guided filter derives it on
        // the fly (uwip_gf_pu8) and the float64 P planes of k_transmission are never written
        uwip_gf_pu8 pu8;
        pu8.planes = b.u8min; pu8.nplanes = 3;
        pu8.sc = b.sc; pu8.sc_stride = SC_COUNT; pu8.b_off = SC_B0;
        pu8.tmin = tmin; pu8.w = 15; pu8.pad = 7;                     // refined_t drops w (BGDehaze.py:52, B-10)
        // ... and, unless the caller wants the refined t itself, recovers the scene in its second kernel
        uwip_gf_recover rec;
        rec.sc = b.sc; rec.sc_stride = SC_COUNT; rec.b_off = SC_B0;
        fused_recover = !d_refined_t && b.redsum;
        rc = guided_filter_u8(ctx, img, in->step, in->frame_stride, b.si, SI_COUNT, nullptr, b.Q, b.AB, F, 2, H, W, r, eps, &pu8,
                              fused_recover ? &rec : nullptr);
        if (rc) return rc;
        if (fused_recover) {
            uwip_kscope ks(ctx, "k_recover");
            k_recover_final2<<<F, 64, 0, ctx->stream>>>(rec.part, rec.nb, b.redsum, b.si, b.sc, (double)n);
        }
    } else {
        rc = run_transmission(ctx, in, b, tmin, nullptr);
        if (rc) return rc;
        rc = guided_filter_u8(ctx, img, in->step, in->frame_stride, b.si, SI_COUNT, b.P, b.Q, b.AB, F, 2, H, W, r, eps);
        if (rc) return rc;
    }
    if (d_refined_t)
        UWIP_HIP(ctx, hipMemcpyAsync(d_refined_t, b.Q, sizeof(double) * 2 * n * F, hipMemcpyDeviceToDevice, ctx->stream));
    if (!fused_recover) {
        uwip_kscope ks(ctx, "k_recover");
        k_recover<<<dim3(RED_BLOCKS, F), 256, 0, ctx->stream>>>(img, in->step, in->frame_stride, b.si, b.sc, b.Q, H, W, b.part);
        k_recover_final<<<F, 64, 0, ctx->stream>>>(b.part, RED_BLOCKS, b.sc, (double)n);
    }
    if (!fused_recover) {
        uwip_kscope ks(ctx, "k_normJ");
        k_normJ<<<dim3(RED_BLOCKS, F), 256, 0, ctx->stream>>>(b.Q, b.sc, H, W, b.part);
        k_normJ_final<<<F, 64, 0, ctx->stream>>>(b.part, RED_BLOCKS, b.si, b.sc, (double)n);
    }
    uint8_t *o = out ? (uint8_t *)out->data : nullptr;
    const size_t ostep = out ? out->step : 0, ofs = out ? out->frame_stride : 0;
    if (!full) {
        uwip_kscope ks(ctx, "k_rc_out");
        k_rc_out<<<dim3(512, F), 256, 0, ctx->stream>>>(img, in->step, in->frame_stride, b.si, b.sc, b.Q, H, W, o, ostep, ofs, d_float_out);
        UWIP_HIP(ctx, hipGetLastError());
        return UWIP_OK;
    }
    // adaptive exposure map: third guided filter, guide = normalised YCrCb of the input
    uint8_t *YI = (uint8_t *)uwip_ws(ctx, "dz.YI", (size_t)3 * n * F);
    double *S = b.P;                       // p planes are free again
    double *RS = b.P + n * F;
    if (!YI) return UWIP_ERR_NOMEM;
    uint8_t *YJ = b.u8planes;              // window planes are free again
    {
        uwip_kscope ks(ctx, "k_exp_prep");
        k_exp_prep<<<dim3(RED_BLOCKS, F), 256, 0, ctx->stream>>>(img, in->step, in->frame_stride, b.si, b.sc, b.Q, H, W, YI, YJ);
    }
    {
        uwip_kscope ks(ctx, "k_exp_S");
        k_exp_S<<<dim3(512, F), 256, 0, ctx->stream>>>(YI, YJ, b.si, n, S, guard);
    }
    rc = guided_filter_u8(ctx, YI, (size_t)W * 3, n * 3, b.si + SI_YIMN, SI_COUNT, S, RS, b.AB, F, 1, H, W, r, eps);
    if (rc) return rc;
    {
        uwip_kscope ks(ctx, "k_exp_out");
        k_exp_out<0, false><<<dim3(RED_BLOCKS, F), 256, 0, ctx->stream>>>(img, in->step, in->frame_stride, b.si, b.sc, b.Q, RS, H, W,
                                                                       b.part, nullptr, 0, 0, nullptr, nullptr);
        k_exp_out_final<<<F, 64, 0, ctx->stream>>>(b.part, RED_BLOCKS, b.sc);
        if (d_hist_out && o)
            k_exp_out<1, true><<<dim3(512, F), 256, 0, ctx->stream>>>(img, in->step, in->frame_stride, b.si, b.sc, b.Q, RS, H, W,
                                                                   nullptr, o, ostep, ofs, d_float_out, d_hist_out);
        else
            k_exp_out<1, false><<<dim3(512, F), 256, 0, ctx->stream>>>(img, in->step, in->frame_stride, b.si, b.sc, b.Q, RS, H, W,
                                                                    nullptr, o, ostep, ofs, d_float_out, nullptr);
    }
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

UWIP_API int uwip_dehaze(uwip_ctx *ctx, const uwip_batch_u8 *in, const uwip_batch_u8 *out, int w, int flags,
                         const double *d_B_inject, double *d_refined_t, double *d_float_out)
{
    return uwip_dehaze_internal(ctx, in, out, w, flags, d_B_inject, d_refined_t, d_float_out, nullptr);
}

int uwip_histretch_internal(uwip_ctx *ctx, const uwip_batch_u8 *img, const char *letters, int lo, int hi, unsigned flags,
                            bool hist_is_fresh);   // histretch.hip
uint32_t *uwip_histretch_hist_ws(uwip_ctx *ctx, const uwip_batch_u8 *img);

UWIP_API int uwip_dehaze_histretch(uwip_ctx *ctx, const uwip_batch_u8 *in, const uwip_batch_u8 *out, int w, int dehaze_flags,
                                   const char *letters, int lo, int hi, unsigned histretch_flags)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    UWIP_REQUIRE(ctx, out != nullptr, "the chained form needs an output batch");
    UWIP_REQUIRE(ctx, letters != nullptr, "null letters");
    const bool fuse = (dehaze_flags & UWIP_DEHAZE_FULL) != 0 && in && in->frames > 0 && !uwip_batch_empty(in);
    uint32_t *d_hist = nullptr;
    if (fuse) {
        int rc = uwip_check_batch(ctx, out, 3);
        if (rc) return rc;
        d_hist = uwip_histretch_hist_ws(ctx, out);
        if (!d_hist) return UWIP_ERR_NOMEM;
        UWIP_HIP(ctx, hipMemsetAsync(d_hist, 0, sizeof(uint32_t) * 768 * (size_t)out->frames, ctx->stream));
    }
    int rc = uwip_dehaze_internal(ctx, in, out, w, dehaze_flags, nullptr, nullptr, nullptr, d_hist);
    if (rc) return rc;
    return uwip_histretch_internal(ctx, out, letters, lo, hi, histretch_flags, d_hist != nullptr);
}
