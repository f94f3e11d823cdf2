// Guided filter (guidedfilter.py:6-110 of the reference, colour guide, radius r) as two streaming
// "wave-strip" kernels for gfx950.
//
// One wavefront owns a strip of 256 image columns -- four adjacent columns per lane, TS = 256 - 2r of them
// outputs and r columns of halo on each side -- and walks down a chunk of rows:
//   * the vertical box sums are sliding-window accumulators in registers (one add row, one subtract row per
//     step, both prefetched while the previous row is being finished);
//   * the horizontal box sums come from an inclusive prefix over the strip: a 4-column serial prefix per lane,
//     a DPP scan of the lane totals across the wave (no LDS, no barrier), the prefix row parked in LDS once,
//     and every output column taking  G[x+r] - G[x-r-1];
//   * k_gf_ws_solve inverts the 3x3 (cov(I) + eps) per pixel and writes a (3 planes) and b; k_gf_ws_final
//     box-filters a, b the same way and writes q = mean_a . I + mean_b.
// By default a block is a single wave, so there is no workgroup barrier anywhere; the LDS row is private to the wave and
// LDS operations of one wave execute in order.  (Optional forms share a strip of 512 / 1024 columns between 2 / 4 waves:
// template parameter NW of both kernels, selected by UWIP_GF_NW / UWIP_GF_FINAL_NW; DESIGN.md section 5 has the measurements.)  Columns outside the image contribute zeros and every mean
// divides by the analytic in-image window size, as guidedfilter.py:39-41,67 does.
//
// The guide statistics are exact: with I = (v - mn) / (mx - mn) for 8-bit v, the window sums of I and I*I' are
// uint32 sums of (v - mn) and (v - mn)(v' - mn) scaled once by 1/(mx-mn) and 1/(mx-mn)^2.  The sums that involve
// p are float64 sums of (v - mn) * p scaled by 1/(mx-mn) afterwards.  Float64 throughout; FMA contraction is
// allowed in this file (the result is checked against the reference at 1e-9, tests/test_dehaze_gpu.py).
#include "uwip_internal.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace {

template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp0(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, true);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp0(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = dpp0<CTRL, ROWMASK>((uint32_t)b), hi = dpp0<CTRL, ROWMASK>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// inclusive prefix over the 64 lanes: Hillis-Steele inside each row of 16 lanes (row_shr 1,2,4,8), then
// row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3; lanes without a source read 0
template <class T>
__device__ __forceinline__ T wave_incl(T v)
{
    v += dpp0<0x111, 0xf>(v);
    v += dpp0<0x112, 0xf>(v);
    v += dpp0<0x114, 0xf>(v);
    v += dpp0<0x118, 0xf>(v);
    v += dpp0<0x142, 0xa>(v);
    v += dpp0<0x143, 0xc>(v);
    return v;
}
// exclusive prefix: shift the wave by one lane (wave_shr:1), then the inclusive scan
template <class T>
__device__ __forceinline__ T wave_excl(T v) { return wave_incl(dpp0<0x138, 0xf>(v)); }

__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int byte_of(const uint32_t (&g)[3], int k) { return (int)((g[k >> 2] >> ((k & 3) * 8)) & 0xffu); }
// three dwords as ONE value (a 96-bit register tuple): what a dwordx3 load produces, so a loop-carried row buffer of
// this type needs no per-element copies behind the load
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ int byte_of(const u32x3 &g, int k) { return (int)((g[k >> 2] >> ((k & 3) * 8)) & 0xffu); }

// Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share an L2).  Give every XCD a contiguous range of
// the logical (strip, y, z) index space instead, so that neighbouring strips -- which share 2r halo columns -- run
// on the same XCD at about the same time and the second reader hits in L2.  Returns false for the padding blocks.
__device__ __forceinline__ bool xcd_decode(unsigned nx, unsigned ny, unsigned nz, unsigned &bx, unsigned &by, unsigned &bz)
{
    const unsigned total = nx * ny * nz, per = (total + 7u) / 8u;
    const unsigned m = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (m >= total) return false;
    bx = m % nx;
    const unsigned t = m / nx;
    by = t % ny;
    bz = t / ny;
    return true;
}

struct StripGeom {
    int l, x0;           // lane, image column of the lane's first column
    bool in[4];          // column inside the image
    bool act[4];         // column is an output column of this strip
    int ah[4], al[4];    // LDS slots (j * 64 + lane) of G[c + r] and G[c - r - 1] inside the row of the wave that holds them
    int ahw[4], alw[4];  // ... and that wave (a strip of `nw` waves is 256 nw columns wide; nw = 1: always 0)
    bool lo_ok[4];       // c - r - 1 >= 0 (else G = 0)
    __device__ __forceinline__ void init(int bx, int TS, int r, int W, int nw = 1)
    {
        l = threadIdx.x & 63;
        const int L = nw > 1 ? (int)threadIdx.x : l;          // lane index inside the strip
        x0 = bx * TS - r + 4 * L;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 4 * L + j, x = x0 + j;
            in[j] = x >= 0 && x < W;
            act[j] = c >= r && c < r + TS && x < W;
            const int h = min(c + r, 256 * nw - 1), lo = c - r - 1;
            lo_ok[j] = lo >= 0;
            const int lc = max(lo, 0);
            ahw[j] = h >> 8; alw[j] = lc >> 8;
            ah[j] = (h & 3) * 64 + ((h & 255) >> 2);
            al[j] = (lc & 3) * 64 + ((lc & 255) >> 2);
        }
    }
};

// 1/d for normal, finite d: v_rcp_f64 seed and two Newton steps (error ~1 ulp; the IEEE division sequence is
// four times as many instructions and the result is only compared at 1e-9)
__device__ __forceinline__ double fast_rcp(double d)
{
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    x = fma(fma(-d, x, 1.0), x, x);
    return x;
}

__device__ __forceinline__ double count_of(int lo, int hi, int n) { return (double)(min(hi, n - 1) - max(lo, 0) + 1); }

// ---- a, b ---------------------------------------------------------------------------------------------
template <int NP, bool PU8>
struct SolveRow {
    u32x3 g;            // 4 columns x 3 bytes of the guide
    double p[NP][4];
};
template <int NP>
struct SolveRow<NP, true> {
    u32x3 g;
    uint32_t m[NP];     // 4 columns x 1 byte per p plane: p = table[byte]
};

// PU8 (bgdehaze's transmission, needs VEC): p is not read as float64 planes but derived on the fly -- plane ip of frame f
// is  p = max(1 - normv(m) / B_ip, tmin)  of the 8-bit window-minimum plane m (BGDehaze.py:28-37, :52), i.e. a 256-entry
// per-frame table, with p = 1 where the w x w window leaves the image (zero padding, :32).  Same values as
// k_transmission writes, 1 byte instead of 8 per sample and no P planes in HBM at all.
// C3 (TS <= 192, i.e. r >= 32: bgdehaze's r = 40 gives TS = 176): the SOLVE half of a row -- the 3x3 inversion and the
// running column sums, about half of the kernel's float64 work -- runs on a second lane -> column mapping, three
// adjacent OUTPUT columns per lane (lane l: strip columns r + 3l .. r + 3l + 2), so 59 of 64 lanes solve columns that are
// written, where the accumulation mapping (four columns per lane, halo included) leaves the 20 halo lanes solving
// nothing.  The window sums come out of the LDS prefix rows either way, so any lane can solve any column; the column
// sums `cs` live with the solving lane.  Three column solves per wave-row instead of four: -25 % of the solve.
// NW = 2: TWO waves share a strip of 512 columns (432 outputs + the same 2 x 40 halo columns), each accumulating and
// scanning its own 256; the scans meet in LDS -- a window that straddles the two halves adds the left wave's row total --
// behind one workgroup barrier per phase.  Useful columns per lane-column rise from 176 / 256 to 432 / 512: at 1920
// columns 5 strips x 512 instead of 11 x 256 (-9 % of all work), at 3840 columns 9 x 512 instead of 22 x 256 (-18 %).
// PTAB (PU8 only): p through a 256-entry float64 table per plane in LDS (4 KB at NP = 2) -- or, PTAB = false, computed per
// sample as max(1 - (m - mn) * c_ip, tmin) with c_ip = 1 / ((mx - mn) B_ip): a conversion, an fma and a max instead of an
// LDS gather, within 2 ulp of the table's two divisions (the filter is compared at 1e-9), and 25.6 instead of 29.7 KB of
// LDS per wave: four solve waves then leave a CU 56 KB, room for one 52 KB block of k_clahe_sweep (with the table they
// left 41 KB and the VALU-bound sweep could not join the latency-bound solve on a CU; DESIGN.md section 5, co-residency).
template <int NP, bool VEC, bool PU8, bool C3, int NW, bool PTAB = true>
__global__ __launch_bounds__(64 * NW) void k_gf_ws_solve(const uint8_t *__restrict__ guide, size_t step, size_t fs,
                                                    const int *__restrict__ gnorm, int gnorm_stride,
                                                    const double *__restrict__ P /*[F][NP][H][W]*/,
                                                    double *__restrict__ AB /*[F*NP][4][H][W]: column prefix sums of a, b*/,
                                                    int H, int W, int r, double eps, int TS, int rpc, int fdiv, uint3 nb,
                                                    uwip_gf_pu8 pu8)
{
#pragma clang fp contract(fast)
    static_assert(!PU8 || VEC, "the 8-bit p source needs the aligned path");
    static_assert(NW == 1 || (NW == 2 && !C3), "one or two waves per strip; the 3-column solve mapping is the one-wave form");
    __shared__ uint4 s_u4[NW * 2 * 4 * 64];
    __shared__ uint32_t s_u1[NW * 4 * 64];
    __shared__ double2 s_d2[NW * NP * 2 * 4 * 64];
    __shared__ uint32_t s_tu[NW][12];          // row totals of a wave's nine integer planes
    __shared__ double s_td[NW][NP * 4];        // ... and of its 4 NP float64 planes
    __shared__ double s_ptab[PU8 && PTAB ? NP * 256 : 1];
    unsigned bx, by, bz;
    if (!xcd_decode(nb.x, nb.y, nb.z, bx, by, bz)) return;
    StripGeom sg;
    sg.init(bx, TS, r, W, NW);
    const int wv = NW > 1 ? (int)(threadIdx.x >> 6) : 0;
    // hand-over between the phases of a row: the LDS rows are private to the wave (NW = 1) or shared by the block
    auto sync = [&]() { if constexpr (NW > 1) __syncthreads(); else wave_lds_fence(); };
    // bz counts groups of NP p-planes; fdiv of them share a frame (fdiv = np / NP)
    const int l = sg.l, zg = bz, f = zg / fdiv;
    const size_t n = (size_t)H * W;
    const int mn = gnorm[(size_t)f * gnorm_stride], mx = gnorm[(size_t)f * gnorm_stride + 1];
    const uint32_t fillw = (uint32_t)mn * 0x01010101u;
    const uint8_t *gf = guide + (size_t)f * fs;
    const double *pin = P + (size_t)zg * NP * n;
    double p_out = 0.0;         // PU8: p where the window leaves the image
    bool cin[4] = {true, true, true, true};
    const int ip0 = (zg - f * fdiv) * NP;     // first of this block's p planes inside the frame
    double pc[NP];              // !PTAB: c_ip
#pragma unroll
    for (int ip = 0; ip < NP; ++ip) pc[ip] = 0.0;
    if (PU8) {
        if constexpr (PTAB) {
            for (int idx = (int)threadIdx.x; idx < NP * 256; idx += 64 * NW) {
                const int ip = idx >> 8, v = idx & 255;
                const double B = pu8.sc[(size_t)f * pu8.sc_stride + pu8.b_off + ip0 + ip];
                const double q = ((double)(v - mn) / (double)(mx - mn)) / B;
                s_ptab[idx] = fmax(1.0 - q, pu8.tmin);
            }
        } else {
#pragma unroll
            for (int ip = 0; ip < NP; ++ip)
                pc[ip] = 1.0 / ((double)(mx - mn) * pu8.sc[(size_t)f * pu8.sc_stride + pu8.b_off + ip0 + ip]);
        }
        p_out = fmax(1.0 - 0.0, pu8.tmin);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = sg.x0 + j;
            cin[j] = (x - pu8.pad >= 0) && (x - pu8.pad + pu8.w <= W);
        }
        sync();
    }

    uint32_t gi[4][9];
    double pf[4][NP][4];
    constexpr int NSC = C3 ? 3 : 4;      // columns solved per lane
    double cs[NSC][NP][4];               // running column sums of a0, a1, a2, b over this block's rows
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int k = 0; k < 9; ++k) gi[j][k] = 0u;
#pragma unroll
        for (int ip = 0; ip < NP; ++ip) pf[j][ip][0] = pf[j][ip][1] = pf[j][ip][2] = pf[j][ip][3] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < NSC; ++j)
#pragma unroll
        for (int ip = 0; ip < NP; ++ip) cs[j][ip][0] = cs[j][ip][1] = cs[j][ip][2] = cs[j][ip][3] = 0.0;
    // the solve mapping: column j of this lane is image column sx0 + j, strip column sc0 + j
    const int sc0 = C3 ? r + 3 * l : 4 * l, sx0 = C3 ? (int)bx * TS + 3 * l : sg.x0;
    bool s_act[NSC], s_lo_ok[NSC];
    int s_ah[NSC], s_al[NSC], s_ahw[NSC], s_alw[NSC];
#pragma unroll
    for (int j = 0; j < NSC; ++j) {
        if constexpr (C3) {
            const int c = sc0 + j;
            s_act[j] = 3 * l + j < TS && sx0 + j < W;
            const int hh = min(c + r, 255), lo = c - r - 1;
            s_lo_ok[j] = lo >= 0;
            const int lc = max(lo, 0);
            s_ah[j] = (hh & 3) * 64 + (hh >> 2);
            s_al[j] = (lc & 3) * 64 + (lc >> 2);
            s_ahw[j] = 0; s_alw[j] = 0;
        } else {
            s_act[j] = sg.act[j]; s_lo_ok[j] = sg.lo_ok[j]; s_ah[j] = sg.ah[j]; s_al[j] = sg.al[j];
            s_ahw[j] = sg.ahw[j]; s_alw[j] = sg.alw[j];
        }
    }

    // A row buffer is cleared ONCE (guide bytes = mn, p = 0: contributes nothing); load_row then only overwrites the
    // lanes / columns that are inside the image, so out-of-image columns stay neutral without a per-row refill.
    auto clear_row = [&](SolveRow<NP, PU8> &R) {
        R.g = u32x3{fillw, fillw, fillw};
        if constexpr (PU8) {
#pragma unroll
            for (int ip = 0; ip < NP; ++ip) R.m[ip] = 0u;
        } else {
#pragma unroll
            for (int ip = 0; ip < NP; ++ip) R.p[ip][0] = R.p[ip][1] = R.p[ip][2] = R.p[ip][3] = 0.0;
        }
    };
    // per-lane bases; the row offsets below are wave-uniform (scalar) products
    const uint8_t *g_lane = gf + (size_t)(sg.in[0] ? sg.x0 : 0) * 3;
    const double *p_lane = pin + (sg.in[0] ? sg.x0 : 0);
    const uint8_t *m_lane = PU8 ? pu8.planes + (size_t)f * pu8.nplanes * n + (sg.in[0] ? sg.x0 : 0) : nullptr;
    // VEC: every lane loads unconditionally (lanes outside the image read the row's first columns and are neutralised in
    // accum by mn_l / pm): a load under a lane mask has to be merged with the register's old contents, and that merge
    // sits right behind the load -- it made every row wait for its own prefetch.
    const int mn_l = (VEC && !sg.in[0]) ? 255 : mn;
    const double pm = (VEC && !sg.in[0]) ? 0.0 : 1.0;
    auto load_row = [&](int yy, SolveRow<NP, PU8> &R) {   // yy must be a row of the image
        if constexpr (PU8) {
            R.g = *reinterpret_cast<const u32x3 *>(g_lane + (size_t)yy * step);
#pragma unroll
            for (int ip = 0; ip < NP; ++ip) R.m[ip] = *reinterpret_cast<const uint32_t *>(m_lane + ((size_t)(ip0 + ip) * H + yy) * W);
        } else if constexpr (VEC) {
            {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(g_lane + (size_t)yy * step);
                R.g = *reinterpret_cast<const u32x3 *>(q);
#pragma unroll
                for (int ip = 0; ip < NP; ++ip) {
                    const double2 *pp = reinterpret_cast<const double2 *>(p_lane + ((size_t)ip * H + yy) * W);
                    const double2 u = pp[0], v = pp[1];
                    R.p[ip][0] = u.x; R.p[ip][1] = u.y; R.p[ip][2] = v.x; R.p[ip][3] = v.y;
                }
            }
        } else {
            uint32_t b[12];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint8_t *q = gf + (size_t)yy * step + (size_t)(sg.in[j] ? sg.x0 + j : 0) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) b[3 * j + c] = sg.in[j] ? (uint32_t)q[c] : (uint32_t)mn;
#pragma unroll
                for (int ip = 0; ip < NP; ++ip)
                    R.p[ip][j] = sg.in[j] ? pin[(size_t)ip * n + (size_t)yy * W + sg.x0 + j] : 0.0;
            }
#pragma unroll
            for (int w = 0; w < 3; ++w) R.g[w] = b[4 * w] | (b[4 * w + 1] << 8) | (b[4 * w + 2] << 16) | (b[4 * w + 3] << 24);
        }
    };
    auto accum = [&](const SolveRow<NP, PU8> &R, auto add_tag, int yy) {
        constexpr bool ADD = decltype(add_tag)::value;
        const bool rin = PU8 && (yy - pu8.pad >= 0) && (yy - pu8.pad + pu8.w <= H);   // uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t a = (uint32_t)max(byte_of(R.g, 3 * j) - mn_l, 0), b = (uint32_t)max(byte_of(R.g, 3 * j + 1) - mn_l, 0),
                           c = (uint32_t)max(byte_of(R.g, 3 * j + 2) - mn_l, 0);
            // a, b, c < 256: v_mul_u32_u24 (full rate) instead of the quarter-rate 32-bit multiply
            const uint32_t v[9] = {a, b, c, (uint32_t)__umul24(a, a), (uint32_t)__umul24(a, b), (uint32_t)__umul24(a, c),
                                   (uint32_t)__umul24(b, b), (uint32_t)__umul24(b, c), (uint32_t)__umul24(c, c)};
#pragma unroll
            for (int k = 0; k < 9; ++k) gi[j][k] = ADD ? gi[j][k] + v[k] : gi[j][k] - v[k];
            const double da = (double)a, db = (double)b, dc = (double)c;
#pragma unroll
            for (int ip = 0; ip < NP; ++ip) {
                double pv;
                if constexpr (PU8 && PTAB) pv = (rin && cin[j]) ? s_ptab[ip * 256 + ((R.m[ip] >> (8 * j)) & 255u)] : p_out;
                else if constexpr (PU8) {
                    const double pt = fmax(fma(-(double)((int)((R.m[ip] >> (8 * j)) & 255u) - mn), pc[ip], 1.0), pu8.tmin);
                    pv = (rin && cin[j]) ? pt : p_out;
                }
                else pv = R.p[ip][j];
                pv *= (ADD ? pm : -pm);
                pf[j][ip][0] += pv; pf[j][ip][1] += da * pv; pf[j][ip][2] += db * pv; pf[j][ip][3] += dc * pv;
            }
        }
    };
    const std::true_type ADD{};
    const std::false_type SUB{};

    const int y0 = by * rpc, y1 = min(H, y0 + rpc);
    const double rdd = 1.0 / (double)(mx - mn);
    const double rDD = 1.0 / ((double)(2 * r + 1) * (double)(2 * r + 1));
    bool lane_full = true;
#pragma unroll
    for (int j = 0; j < NSC; ++j)
        if (s_act[j]) lane_full = lane_full && (sx0 + j - r >= 0) && (sx0 + j + r < W);
    const bool wave_full = __all(lane_full) != 0;      // every solved column's window lies inside the image's columns
    // warm-up: rows [max(0, y0 - r), y0 + r) in batches of four loads
    int v = max(y0 - r, 0);
    const int wend = min(y0 + r, H);   // first row that belongs to the steady loop
    {
        SolveRow<NP, PU8> R0, R1, R2, R3;
        clear_row(R0); clear_row(R1); clear_row(R2); clear_row(R3);
        for (; v < wend; v += 4) {
            load_row(v, R0);
            if (v + 1 < wend) load_row(v + 1, R1);
            if (v + 2 < wend) load_row(v + 2, R2);
            if (v + 3 < wend) load_row(v + 3, R3);
            accum(R0, ADD, v);
            if (v + 1 < wend) accum(R1, ADD, v + 1);
            if (v + 2 < wend) accum(R2, ADD, v + 2);
            if (v + 3 < wend) accum(R3, ADD, v + 3);
        }
    }
    SolveRow<NP, PU8> Ra, Rs;
    clear_row(Ra); clear_row(Rs);
    if (y0 + r < H) load_row(y0 + r, Ra);
    for (int y = y0; y < y1; ++y) {
        // rows [y - r, y + r]: add y + r, drop y - r - 1 (only rows this block added itself); all conditions uniform
        if (y + r < H) accum(Ra, ADD, y + r);
        if (y > y0 && y - r - 1 >= 0) accum(Rs, SUB, y - r - 1);
        // always issued (row index clamped into the image; an unused row is simply not accumulated): under a branch the
        // loaded registers would be copied at its end, i.e. waited for at once
        load_row(min(y + 1 + r, H - 1), Ra);
        load_row(max(y - r, 0), Rs);

        // ---- horizontal prefix of the nine guide planes
        {
            uint32_t G[9][4];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const uint32_t s0 = gi[0][k], s1 = s0 + gi[1][k], s2 = s1 + gi[2][k], s3 = s2 + gi[3][k];
                const uint32_t e = wave_incl(s3) - s3;
                G[k][0] = e + s0; G[k][1] = e + s1; G[k][2] = e + s2; G[k][3] = e + s3;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s_u4[wv * 512 + (0 * 4 + j) * 64 + l] = make_uint4(G[0][j], G[1][j], G[2][j], G[3][j]);
                s_u4[wv * 512 + (1 * 4 + j) * 64 + l] = make_uint4(G[4][j], G[5][j], G[6][j], G[7][j]);
                s_u1[wv * 256 + j * 64 + l] = G[8][j];
            }
            if (NW > 1 && l == 63) {
#pragma unroll
                for (int k = 0; k < 9; ++k) s_tu[wv][k] = G[k][3];
            }
        }
        // ---- and of the 4 NP float64 planes
#pragma unroll
        for (int ip = 0; ip < NP; ++ip) {
            double D[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double s0 = pf[0][ip][k], s1 = s0 + pf[1][ip][k], s2 = s1 + pf[2][ip][k], s3 = s2 + pf[3][ip][k];
                const double e = wave_excl(s3);
                D[k][0] = e + s0; D[k][1] = e + s1; D[k][2] = e + s2; D[k][3] = e + s3;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s_d2[wv * (NP * 512) + ((ip * 2 + 0) * 4 + j) * 64 + l] = make_double2(D[0][j], D[1][j]);
                s_d2[wv * (NP * 512) + ((ip * 2 + 1) * 4 + j) * 64 + l] = make_double2(D[2][j], D[3][j]);
            }
            if (NW > 1 && l == 63) {
#pragma unroll
                for (int k = 0; k < 4; ++k) s_td[wv][ip * 4 + k] = D[k][3];
            }
        }
        sync();

        const int cyi = min(y + r, H - 1) - max(y - r, 0) + 1;
        const double cy = (double)cyi;
        // 1 / window size: one reciprocal per row where every window of the wave spans 2r + 1 columns, none where the rows do too
        double rrow = 0.0;
        if (wave_full) rrow = cyi == 2 * r + 1 ? rDD : fast_rcp(cy * (double)(2 * r + 1));
        auto solve_col = [&](int j) {
            const int x = sx0 + j;
            const double rbase = wave_full ? rrow : fast_rcp(cy * count_of(x - r, x + r, W));
            const double r1 = rdd * rbase, r2 = (rdd * rdd) * rbase;
            uint32_t w9[9];
            {
                const uint4 h0 = s_u4[s_ahw[j] * 512 + 0 * 256 + s_ah[j]], h1 = s_u4[s_ahw[j] * 512 + 1 * 256 + s_ah[j]];
                const uint32_t h2 = s_u1[s_ahw[j] * 256 + s_ah[j]];
                uint4 l0 = s_u4[s_alw[j] * 512 + 0 * 256 + s_al[j]], l1 = s_u4[s_alw[j] * 512 + 1 * 256 + s_al[j]];
                uint32_t l2 = s_u1[s_alw[j] * 256 + s_al[j]];
                if (!s_lo_ok[j]) { l0 = make_uint4(0, 0, 0, 0); l1 = l0; l2 = 0u; }
                if (NW > 1 && s_ahw[j] != s_alw[j]) {      // the window straddles the two halves: + the left wave's row totals
                    l0.x -= s_tu[0][0]; l0.y -= s_tu[0][1]; l0.z -= s_tu[0][2]; l0.w -= s_tu[0][3];
                    l1.x -= s_tu[0][4]; l1.y -= s_tu[0][5]; l1.z -= s_tu[0][6]; l1.w -= s_tu[0][7];
                    l2 -= s_tu[0][8];
                }
                w9[0] = h0.x - l0.x; w9[1] = h0.y - l0.y; w9[2] = h0.z - l0.z; w9[3] = h0.w - l0.w;
                w9[4] = h1.x - l1.x; w9[5] = h1.y - l1.y; w9[6] = h1.z - l1.z; w9[7] = h1.w - l1.w;
                w9[8] = h2 - l2;
            }
            const double m0 = (double)w9[0] * r1, m1 = (double)w9[1] * r1, m2 = (double)w9[2] * r1;
            const double s00 = (double)w9[3] * r2 - m0 * m0 + eps, s01 = (double)w9[4] * r2 - m0 * m1,
                         s02 = (double)w9[5] * r2 - m0 * m2, s11 = (double)w9[6] * r2 - m1 * m1 + eps,
                         s12 = (double)w9[7] * r2 - m1 * m2, s22 = (double)w9[8] * r2 - m2 * m2 + eps;
            const double k00 = s11 * s22 - s12 * s12, k01 = s02 * s12 - s01 * s22, k02 = s01 * s12 - s02 * s11;
            const double k11 = s00 * s22 - s02 * s02, k12 = s01 * s02 - s00 * s12, k22 = s00 * s11 - s01 * s01;
            const double rdet = fast_rcp(s00 * k00 + s01 * k01 + s02 * k02);
#pragma unroll
            for (int ip = 0; ip < NP; ++ip) {
                const double2 hA = s_d2[s_ahw[j] * (NP * 512) + (ip * 2 + 0) * 256 + s_ah[j]],
                              hB = s_d2[s_ahw[j] * (NP * 512) + (ip * 2 + 1) * 256 + s_ah[j]];
                double2 lA = s_d2[s_alw[j] * (NP * 512) + (ip * 2 + 0) * 256 + s_al[j]],
                        lB = s_d2[s_alw[j] * (NP * 512) + (ip * 2 + 1) * 256 + s_al[j]];
                if (!s_lo_ok[j]) { lA = make_double2(0.0, 0.0); lB = lA; }
                if (NW > 1 && s_ahw[j] != s_alw[j]) {
                    lA.x -= s_td[0][ip * 4 + 0]; lA.y -= s_td[0][ip * 4 + 1]; lB.x -= s_td[0][ip * 4 + 2]; lB.y -= s_td[0][ip * 4 + 3];
                }
                const double mp = (hA.x - lA.x) * rbase;
                const double c0 = (hA.y - lA.y) * r1 - m0 * mp, c1 = (hB.x - lB.x) * r1 - m1 * mp,
                             c2 = (hB.y - lB.y) * r1 - m2 * mp;
                const double a0 = (c0 * k00 + c1 * k01 + c2 * k02) * rdet;
                const double a1 = (c0 * k01 + c1 * k11 + c2 * k12) * rdet;
                const double a2 = (c0 * k02 + c1 * k12 + c2 * k22) * rdet;
                cs[j][ip][0] += a0; cs[j][ip][1] += a1; cs[j][ip][2] += a2;
                cs[j][ip][3] += mp - a0 * m0 - a1 * m1 - a2 * m2;
            }
        };
        if constexpr (C3) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (s_act[j]) solve_col(j);
                const size_t i = (size_t)y * W + sx0 + j;
#pragma unroll
                for (int ip = 0; ip < NP; ++ip) {
                    double *o = AB + ((size_t)zg * NP + ip) * 4 * n + i;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (s_act[j]) o[(size_t)q * n] = cs[j][ip][q];
                }
            }
        } else {
#pragma unroll
            for (int jj = 0; jj < 4; jj += 2) {
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2)
                    if (s_act[jj + j2]) solve_col(jj + j2);
                const size_t i = (size_t)y * W + sg.x0 + jj;
#pragma unroll
                for (int ip = 0; ip < NP; ++ip) {
                    double *o = AB + ((size_t)zg * NP + ip) * 4 * n + i;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (VEC) {
                            if (sg.act[jj]) *reinterpret_cast<double2 *>(o + (size_t)q * n) = make_double2(cs[jj][ip][q], cs[jj + 1][ip][q]);
                        } else {
                            if (sg.act[jj]) o[(size_t)q * n] = cs[jj][ip][q];
                            if (sg.act[jj + 1]) o[(size_t)q * n + 1] = cs[jj + 1][ip][q];
                        }
                    }
                }
            }
        }
        sync();
    }
}

// ---- q = box(a) . I + box(b) ----------------------------------------------------------------------------
// k_gf_ws_solve leaves S = the running column sums of a, b (restarting at each of its row chunks), so the vertical
// box sum of row y is  S[y + r] - S[y - r - 1]  (plus the previous chunk's last row when the two straddle a chunk
// start): no sliding window, no warm-up rows.  A wave walks "chains" of rows 2r+1 apart,
//     y = s, s + (2r+1), s + 2(2r+1), ...
// because S[y + r] of one link is S[y' - r - 1] of the next: every row of S is read once per chain set instead of
// twice.  Chains are independent, so any number of waves can share a strip at no extra traffic.
struct FinalRow {
    double v[4][4];   // [plane][column]
};

// REC (bgdehaze's scene recovery fused into the first filter, BGDehaze.py:50-52): instead of q = refined t the kernel
// writes  J_ip = (normv(I_ip) - B_ip) / q + B_ip  and leaves each wave's min / max / sum of J in `rec.part` -- the separate
// k_recover pass (read 19 B + write 16 B per pixel) disappears.  The division by q = X / N is a multiplication by N / X with a
// Newton reciprocal (within 2 ulp of k_recover's division; compared at 1e-9).
template <bool VEC, bool REC, int NW>
__global__ __launch_bounds__(64 * NW) void k_gf_ws_final(const double *__restrict__ S /*[Z][4][H][W]*/,
                                                    const uint8_t *__restrict__ guide, size_t step, size_t fs,
                                                    const int *__restrict__ gnorm, int gnorm_stride, int NP,
                                                    double *__restrict__ Q /*[Z][H][W]*/, int H, int W, int r, int TS,
                                                    int rpc /*rows per chunk of S*/, int spw /*chains per wave*/, uint3 nb,
                                                    uwip_gf_recover rec)
{
#pragma clang fp contract(fast)
    static_assert(NW == 1 || VEC, "several waves per strip: aligned path only");
    __shared__ double2 s_d2[NW * 2 * 4 * 64];
    __shared__ double s_tot[NW][4];            // row totals of a wave's four planes
    __shared__ double s_nt[REC ? 256 : 1];
    __shared__ double s_j[REC && NW > 1 ? NW * 3 : 1];
    unsigned bx, by, bz;
    if (!xcd_decode(nb.x, nb.y, nb.z, bx, by, bz)) return;
    StripGeom sg;
    sg.init(bx, TS, r, W, NW);
    const int wv = NW > 1 ? (int)(threadIdx.x >> 6) : 0;
    auto sync = [&]() { if constexpr (NW > 1) __syncthreads(); else wave_lds_fence(); };
    const int l = sg.l, z = bz, f = z / NP;
    const size_t n = (size_t)H * W;
    const int mn = gnorm[(size_t)f * gnorm_stride], mx = gnorm[(size_t)f * gnorm_stride + 1];
    double jmin = 1e300, jmax = -1e300, jsum = 0.0, Bc = 0.0;
    const int ipc = z - f * NP;          // the p plane = the guide channel this block recovers
    double *jpart = REC ? rec.part + ((size_t)z * nb.x * nb.y + (size_t)by * nb.x + bx) * 3 : nullptr;
    if (REC) {
        Bc = rec.sc[(size_t)f * rec.sc_stride + rec.b_off + ipc];
        for (int v = (int)threadIdx.x; v < 256; v += 64 * NW) s_nt[v] = (double)(v - mn) / (double)(mx - mn);
        sync();
    }
    const uint8_t *gf = guide + (size_t)f * fs;
    const double *sp = S + (size_t)z * 4 * n;
    const int D = 2 * r + 1;
    const int s_end = min(min(D, H), (int)(by + 1) * spw);
    int s = by * spw, y = s;
    if (s >= s_end) {       // uniform over the block
        if (REC && threadIdx.x == 0) { jpart[0] = jmin; jpart[1] = jmax; jpart[2] = 0.0; }
        return;
    }

    // row buffers are zeroed once; load_row only overwrites in-image columns (yy must be a row of the image)
    auto clear_row = [&](FinalRow &R) {
#pragma unroll
        for (int k = 0; k < 4; ++k) R.v[k][0] = R.v[k][1] = R.v[k][2] = R.v[k][3] = 0.0;
    };
    const double *s_lane = sp + (sg.in[0] ? sg.x0 : 0);
    auto load_row = [&](int yy, FinalRow &R) {
        if (VEC) {
            if (sg.in[0]) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double2 *pp = reinterpret_cast<const double2 *>(s_lane + ((size_t)k * H + yy) * W);
                    const double2 u = pp[0], v = pp[1];
                    R.v[k][0] = u.x; R.v[k][1] = u.y; R.v[k][2] = v.x; R.v[k][3] = v.y;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    R.v[k][j] = sg.in[j] ? sp[(size_t)k * n + (size_t)yy * W + sg.x0 + j] : 0.0;
        }
    };
    const double rdd = 1.0 / (double)(mx - mn);
    const double rDD = 1.0 / ((double)D * (double)D);
    // every output column of this wave has its whole window inside the image (all strips but the first and the last)
    bool lane_full = true;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (sg.act[j]) lane_full = lane_full && (sg.x0 + j - r >= 0) && (sg.x0 + j + r < W);
    const bool wave_full = __all(lane_full) != 0;

    // one link of a chain: `cur` holds S[min(y + r, H - 1)], `prev` holds S[y - r - 1] (zeros above the image).
    // Leaves the next link's rows in (prev := next cur, cur := next prev) -- the caller swaps the two buffers.
    auto link = [&](FinalRow &prev, FinalRow &cur) -> bool {
        const int hi = min(y + r, H - 1), lo = y - r - 1;
        double d[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[k][j] = cur.v[k][j] - prev.v[k][j];
        if (lo >= 0 && hi / rpc != lo / rpc) {   // uniform: the window straddles a chunk start of S
            FinalRow E;
            clear_row(E);
            load_row((hi / rpc) * rpc - 1, E);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[k][j] += E.v[k][j];
        }
        // next link (prefetched under the horizontal pass of this one)
        int yn = y + D, sn = s;
        bool chain_start = false;
        if (yn >= H) { sn = s + 1; yn = sn; chain_start = true; }
        const bool more = sn < s_end;
        if (more) {
            load_row(min(yn + r, H - 1), prev);                    // next cur
            if (chain_start) {                                     // next prev (else: this cur)
                if (yn - r - 1 >= 0) load_row(yn - r - 1, cur);
                else clear_row(cur);
            }
        }
        uint32_t gw[3] = {0u, 0u, 0u};
        if (VEC) {
            if (sg.in[0]) {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(gf + (size_t)y * step + (size_t)sg.x0 * 3);
                gw[0] = q[0]; gw[1] = q[1]; gw[2] = q[2];
            }
        } else {
            uint32_t b[12];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint8_t *q = gf + (size_t)y * step + (size_t)(sg.in[j] ? sg.x0 + j : 0) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) b[3 * j + c] = sg.in[j] ? (uint32_t)q[c] : 0u;
            }
#pragma unroll
            for (int w = 0; w < 3; ++w) gw[w] = b[4 * w] | (b[4 * w + 1] << 8) | (b[4 * w + 2] << 16) | (b[4 * w + 3] << 24);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double s1 = d[k][0] + d[k][1], s2 = s1 + d[k][2], s3 = s2 + d[k][3];
            const double e = wave_excl(s3);
            d[k][0] += e; d[k][1] = e + s1; d[k][2] = e + s2; d[k][3] = e + s3;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s_d2[wv * 512 + (0 * 4 + j) * 64 + l] = make_double2(d[0][j], d[1][j]);
            s_d2[wv * 512 + (1 * 4 + j) * 64 + l] = make_double2(d[2][j], d[3][j]);
        }
        if (NW > 1 && l == 63) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s_tot[wv][k] = d[k][3];
        }
        sync();
        // window size N = cy * cx: one reciprocal per row where every window of the wave spans 2r + 1 columns, none where the
        // rows do too; the recovery divides by q = X / N, i.e. multiplies by N / X, and needs no 1 / N at all
        const int cyi = min(y + r, H - 1) - max(y - r, 0) + 1;
        const double cy = (double)cyi;
        double nrow = 0.0;
        if (REC) nrow = cy * (double)D;
        else if (wave_full) nrow = cyi == D ? rDD : fast_rcp(cy * (double)D);
        double qv[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (sg.act[j]) {
                const int x = sg.x0 + j;
                const double2 hA = s_d2[sg.ahw[j] * 512 + 0 * 256 + sg.ah[j]], hB = s_d2[sg.ahw[j] * 512 + 1 * 256 + sg.ah[j]];
                double2 lA = s_d2[sg.alw[j] * 512 + 0 * 256 + sg.al[j]], lB = s_d2[sg.alw[j] * 512 + 1 * 256 + sg.al[j]];
                if (!sg.lo_ok[j]) { lA = make_double2(0.0, 0.0); lB = lA; }
                if (NW > 1 && sg.ahw[j] != sg.alw[j]) {    // the window straddles two waves' columns (2r + 1 <= 256: never three)
                    const int t = sg.alw[j];
                    lA.x -= s_tot[t][0]; lA.y -= s_tot[t][1]; lB.x -= s_tot[t][2]; lB.y -= s_tot[t][3];
                }
                // X = sum(a) . I + sum(b) with I = u / (mx - mn), u the guide byte above the frame minimum
                const double u0 = (double)(byte_of(gw, 3 * j) - mn), u1 = (double)(byte_of(gw, 3 * j + 1) - mn),
                             u2 = (double)(byte_of(gw, 3 * j + 2) - mn);
                const double X = ((hA.x - lA.x) * u0 + (hA.y - lA.y) * u1 + (hB.x - lB.x) * u2) * rdd + (hB.y - lB.y);
                if (REC) {
                    const double N = wave_full ? nrow : cy * count_of(x - r, x + r, W);
                    const double jv = ((s_nt[byte_of(gw, 3 * j + ipc)] - Bc) * N) * fast_rcp(X) + Bc;
                    qv[j] = jv;
                    jmin = fmin(jmin, jv); jmax = fmax(jmax, jv); jsum += jv;
                } else {
                    qv[j] = X * (wave_full ? nrow : fast_rcp(cy * count_of(x - r, x + r, W)));
                }
            }
        }
        double *o = Q + (size_t)z * n + (size_t)y * W + sg.x0;
        if (VEC) {
            if (sg.act[0]) {
                reinterpret_cast<double2 *>(o)[0] = make_double2(qv[0], qv[1]);
                reinterpret_cast<double2 *>(o)[1] = make_double2(qv[2], qv[3]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (sg.act[j]) o[j] = qv[j];
        }
        sync();
        y = yn; s = sn;
        return more;
    };

    FinalRow A, B;
    clear_row(A); clear_row(B);
    if (y - r - 1 >= 0) load_row(y - r - 1, A);
    load_row(min(y + r, H - 1), B);
    for (;;) {
        if (!link(A, B)) break;
        if (!link(B, A)) break;
    }
    if (REC) {
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            jmin = fmin(jmin, __shfl_xor(jmin, sft, 64));
            jmax = fmax(jmax, __shfl_xor(jmax, sft, 64));
            jsum += __shfl_xor(jsum, sft, 64);
        }
        if constexpr (NW > 1) {
            if (l == 0) { s_j[wv * 3 + 0] = jmin; s_j[wv * 3 + 1] = jmax; s_j[wv * 3 + 2] = jsum; }
            __syncthreads();
            if (threadIdx.x == 0) {
#pragma unroll
                for (int t = 1; t < NW; ++t) {
                    jmin = fmin(jmin, s_j[t * 3 + 0]); jmax = fmax(jmax, s_j[t * 3 + 1]); jsum += s_j[t * 3 + 2];
                }
                jpart[0] = jmin; jpart[1] = jmax; jpart[2] = jsum;
            }
        } else if (l == 0) { jpart[0] = jmin; jpart[1] = jmax; jpart[2] = jsum; }
    }
}

}  // namespace

// Launches the two kernels.  guide: 3-channel u8 frames; gnorm[f*gstride + {0,1}] = the frame's min / max guide
// value; P [F][np][H][W] -> Q [F][np][H][W]; AB [F*np][4][H][W] scratch (column prefix sums of a, b).
bool uwip_gf_pu8_ok(const uint8_t *guide, size_t step, size_t fs, const uint8_t *planes, int np, int W, int r)
{
    return np == 2 && planes && (W % 4 == 0) && (r % 4 == 0) && (step % 4 == 0) && (fs % 4 == 0) && (((uintptr_t)guide) % 4 == 0) &&
           (((uintptr_t)planes) % 4 == 0);
}

int uwip_gf_wave_strip(uwip_ctx *ctx, const uint8_t *guide, size_t step, size_t fs, const int *gnorm, int gstride,
                       const double *P, double *Q, double *AB, int F, int np, int H, int W, int r, double eps,
                       const uwip_gf_pu8 *pu8, uwip_gf_recover *rec)
{
    UWIP_REQUIRE(ctx, np == 1 || np == 2, "np must be 1 or 2");
    UWIP_REQUIRE(ctx, r >= 1 && 2 * r <= 192, "radius out of range for the 256-column strip");
    UWIP_REQUIRE(ctx, H >= 2 * r + 1 && W >= 2 * r + 1, "guided filter needs rows, cols >= 2r+1");
    UWIP_REQUIRE(ctx, (uint64_t)uwip_cdiv(W, 256 - 2 * r) * 16 * F * np < (1ull << 31) && (uint64_t)uwip_cdiv(W, 256 - 2 * r) * (2 * r + 1) * F * np < (1ull << 31), "too many blocks for one launch");
    UWIP_REQUIRE(ctx, (uint64_t)256 * (2 * r + 1) * 65025ull < (1ull << 32), "window too large for the exact integer guide sums");
    const int D = 2 * r + 1;
    // UWIP_DIAG_GF_ONLY=solve | final (read at every call, in a process started with UWIP_TEST_HOOKS=1 only): launch only that kernel of the pair, the other one's output being
    // whatever the workspace holds from an earlier complete call.  A MEASUREMENT hook (tools/corun_matrix.py runs the two
    // kernels against each other and against the sweep on separate streams); results of such a call are meaningless.
    int diag_only = 0;
    if (const char *e = uwip_test_hooks() ? getenv("UWIP_DIAG_GF_ONLY") : nullptr) diag_only = e[0] == 's' ? 1 : (e[0] == 'f' ? 2 : 0);
    // four adjacent columns of a lane are one aligned vector access when everything is a multiple of 4
    const bool vec = (W % 4 == 0) && (r % 4 == 0) && (step % 4 == 0) && (fs % 4 == 0) && (((uintptr_t)guide) % 4 == 0) &&
                     (((uintptr_t)P | (uintptr_t)Q | (uintptr_t)AB) % 16 == 0);
    if (pu8) UWIP_REQUIRE(ctx, vec && uwip_gf_pu8_ok(guide, step, fs, pu8->planes, np, W, r) && pu8->sc, "8-bit p source: unsupported geometry");
    UWIP_REQUIRE(ctx, pu8 || P, "null p planes");
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus < 1) cus = 256;
    auto slots_of = [&](const void *kernel, int threads = 64) {   // resident blocks on the whole chip
        int per_cu = 8;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) per_cu = 8;
        return (double)per_cu * cus;
    };
    const char *env_split = getenv("UWIP_GF_SPLIT");
    const bool split = np == 2 && env_split && atoi(env_split) > 0;   // two one-plane solves instead of a fused one
    const int knp = split ? 1 : np;
    const unsigned zs = (unsigned)F * (np / knp);
    // (occupancy query only: the 3-column and 4-column solve mappings differ by a few registers, both one wave per SIMD)
    const void *ksolve = pu8 ? (knp == 2 ? (const void *)k_gf_ws_solve<2, true, true, true, 1> : (const void *)k_gf_ws_solve<1, true, true, true, 1>)
                       : knp == 2 ? (vec ? (const void *)k_gf_ws_solve<2, true, false, true, 1> : (const void *)k_gf_ws_solve<2, false, false, true, 1>)
                                  : (vec ? (const void *)k_gf_ws_solve<1, true, false, true, 1> : (const void *)k_gf_ws_solve<1, false, false, true, 1>);
    // Two waves per strip (512 columns; aligned path only) cover the width with fewer lane-columns -- 1920 columns, r = 40:
    // 5 x 512 against 11 x 256 (-9 %), 3840 columns: 9 x 512 against 22 x 256 (-18 %) -- and measure within 1 % of the
    // one-wave form at 1080p and -4.5 % (solve only) at 4K: the two waves meet at a barrier twice per row and the kernel is
    // latency-bound per wave, so the saved wave-rows come back as waiting.  Default from 3072 columns up; UWIP_GF_NW = 1 | 2.
    int nw = 1;
    if (vec && 512 - 2 * r > 0) {
        if (W >= 3072) nw = 2;         // 3840 columns: full pipe 703 -> 707.5 frames/s (same box, twice each)
        const char *e = getenv("UWIP_GF_NW");
        if (e && (atoi(e) == 1 || atoi(e) == 2)) nw = atoi(e);
    }
    const int TSs = 256 * nw - 2 * r;                       // outputs per strip of the solve kernel
    const unsigned strips_s = uwip_cdiv(W, TSs);
    // solve: every row chunk re-reads 2r warm-up rows, so use as few chunks as keep the chip full, preferring a whole
    // number of "rounds" of resident waves.  A chunk is at least 2r+1 rows (k_gf_ws_final relies on it).
    int c = 1;
    {
        const int cmax = std::max(1, H / D);
        const char *e = getenv("UWIP_GF_CHUNKS");
        if (e && atoi(e) > 0) c = std::min(atoi(e), cmax);
        else {
            const double slots = slots_of(ksolve);
            double best_cost = 1e300;
            for (int t = 1; t <= 16 && t <= cmax; ++t) {
                const double waves = (double)strips_s * nw * zs * t;
                const double rounds = std::max(1.0, std::ceil(waves / slots));
                const double cost = rounds * ((double)((H + t - 1) / t) + 2.0 * r);
                if (cost < best_cost * 0.97) { best_cost = cost; c = t; }
            }
        }
    }
    int rpc = (H + c - 1) / c;
    if (rpc < D) rpc = D;
    {
        const uint3 nb = make_uint3(strips_s, uwip_cdiv(H, rpc), zs);
        const unsigned grid = 8u * ((nb.x * nb.y * nb.z + 7u) / 8u);
        uwip_kscope ks(ctx, "k_gf_ws_solve");
        const int fdiv = np / knp;
        const uwip_gf_pu8 none{};
        const uwip_gf_pu8 &pa = pu8 ? *pu8 : none;
#define UWIP_GF_SOLVE(NPV, VECV, PU8V, C3V, NWV)                                                                                     \
    k_gf_ws_solve<NPV, VECV, PU8V, C3V, NWV><<<grid, 64 * NWV, 0, ctx->stream>>>(guide, step, fs, gnorm, gstride, P, AB, H, W, r, eps, \
                                                                               TSs, rpc, fdiv, nb, pa)
#define UWIP_GF_SOLVE_NT(NPV, C3V, NWV)                                                                                              \
    k_gf_ws_solve<NPV, true, true, C3V, NWV, false><<<grid, 64 * NWV, 0, ctx->stream>>>(guide, step, fs, gnorm, gstride, P, AB, H, W, r, \
                                                                                      eps, TSs, rpc, fdiv, nb, pa)
        const bool c3 = nw == 1 && TSs <= 192;
        // the 8-bit p source without its LDS table (the default; UWIP_GF_PTAB=1 keeps the table: A/B)
        static const bool ptab = [] { const char *e = getenv("UWIP_GF_PTAB"); return e && atoi(e) > 0; }();
        if (diag_only == 2) {
            // diagnostic: the second kernel alone (tools/corun_matrix.py)
        } else if (pu8 && knp == 1 && !ptab) {
            if (nw == 2) UWIP_GF_SOLVE_NT(1, false, 2); else if (c3) UWIP_GF_SOLVE_NT(1, true, 1); else UWIP_GF_SOLVE_NT(1, false, 1);
        } else if (pu8 && !ptab) {
            if (nw == 2) UWIP_GF_SOLVE_NT(2, false, 2); else if (c3) UWIP_GF_SOLVE_NT(2, true, 1); else UWIP_GF_SOLVE_NT(2, false, 1);
        } else if (pu8 && knp == 1) {
            if (nw == 2) UWIP_GF_SOLVE(1, true, true, false, 2); else if (c3) UWIP_GF_SOLVE(1, true, true, true, 1); else UWIP_GF_SOLVE(1, true, true, false, 1);
        } else if (pu8) {
            if (nw == 2) UWIP_GF_SOLVE(2, true, true, false, 2); else if (c3) UWIP_GF_SOLVE(2, true, true, true, 1); else UWIP_GF_SOLVE(2, true, true, false, 1);
        } else if (knp == 2) {
            if (vec) { if (nw == 2) UWIP_GF_SOLVE(2, true, false, false, 2); else if (c3) UWIP_GF_SOLVE(2, true, false, true, 1); else UWIP_GF_SOLVE(2, true, false, false, 1); }
            else { if (c3) UWIP_GF_SOLVE(2, false, false, true, 1); else UWIP_GF_SOLVE(2, false, false, false, 1); }
        } else {
            if (vec) { if (nw == 2) UWIP_GF_SOLVE(1, true, false, false, 2); else if (c3) UWIP_GF_SOLVE(1, true, false, true, 1); else UWIP_GF_SOLVE(1, true, false, false, 1); }
            else { if (c3) UWIP_GF_SOLVE(1, false, false, true, 1); else UWIP_GF_SOLVE(1, false, false, false, 1); }
        }
#undef UWIP_GF_SOLVE_NT
#undef UWIP_GF_SOLVE
    }
    {
        // final: chains are independent, so split the 2r+1 chain starts over enough waves for ~4 rounds.
        // NWF waves share a strip of 256 NWF columns (aligned path): the kernel runs at the HBM rate and every strip re-reads
        // 2r halo columns of S, so wider strips are fewer bytes -- 1920 columns, r = 40: 11 strips read 2720 columns of every
        // row, 3 strips of 4 waves 2080 (-24 %); lanes beyond the image load nothing.  UWIP_GF_FINAL_NW = 1 | 2 | 4.
        const unsigned Z = (unsigned)F * np;
        int nwf = 1;
        if (vec) {
            const char *e = getenv("UWIP_GF_FINAL_NW");
            if (e && (atoi(e) == 1 || atoi(e) == 2 || atoi(e) == 4)) nwf = atoi(e);
        }
        const int TSf = 256 * nwf - 2 * r;
        const unsigned strips_f = uwip_cdiv(W, TSf);
        const void *kfinal = rec ? (nwf == 4 ? (const void *)k_gf_ws_final<true, true, 4> : nwf == 2 ? (const void *)k_gf_ws_final<true, true, 2> : (const void *)k_gf_ws_final<true, true, 1>)
                           : vec ? (nwf == 4 ? (const void *)k_gf_ws_final<true, false, 4> : nwf == 2 ? (const void *)k_gf_ws_final<true, false, 2> : (const void *)k_gf_ws_final<true, false, 1>)
                                 : (const void *)k_gf_ws_final<false, false, 1>;
        if (rec) UWIP_REQUIRE(ctx, vec && rec->sc, "fused recovery needs the aligned path");
        const int nchain = std::min(D, H);
        int groups = (int)std::ceil(4.0 * slots_of(kfinal, 64 * nwf) / ((double)strips_f * Z));
        const char *e = getenv("UWIP_GF_GROUPS");
        if (e && atoi(e) > 0) groups = atoi(e);
        groups = std::max(1, std::min(groups, nchain));
        const int spw = (nchain + groups - 1) / groups;
        const uint3 nb = make_uint3(strips_f, uwip_cdiv(nchain, spw), Z);
        const unsigned grid = 8u * ((nb.x * nb.y * nb.z + 7u) / 8u);
        uwip_kscope ks(ctx, "k_gf_ws_final");
        uwip_gf_recover ra{};
        if (rec) {
            // one (min, max, sum) triple per block of the launch; the caller reduces them
            rec->nb = (int)(nb.x * nb.y);
            rec->part = (double *)uwip_ws(ctx, "gf.jpart", sizeof(double) * 3 * (size_t)rec->nb * Z);
            if (!rec->part) return UWIP_ERR_NOMEM;
            ra = *rec;
        }
#define UWIP_GF_FINAL(VECV, RECV, NWV) \
    k_gf_ws_final<VECV, RECV, NWV><<<grid, 64 * NWV, 0, ctx->stream>>>(AB, guide, step, fs, gnorm, gstride, np, Q, H, W, r, TSf, rpc, spw, nb, ra)
        if (diag_only == 1) {
            // diagnostic: the first kernel alone
        } else if (rec) { if (nwf == 4) UWIP_GF_FINAL(true, true, 4); else if (nwf == 2) UWIP_GF_FINAL(true, true, 2); else UWIP_GF_FINAL(true, true, 1); }
        else if (vec) { if (nwf == 4) UWIP_GF_FINAL(true, false, 4); else if (nwf == 2) UWIP_GF_FINAL(true, false, 2); else UWIP_GF_FINAL(true, false, 1); }
        else UWIP_GF_FINAL(false, false, 1);
#undef UWIP_GF_FINAL
    }
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}
