// aclahe hot path for gfx950.
//
// cv::CLAHE::apply as driven by modules/aclahe/src/aclahe.cpp:175-187 becomes
//   k_clahe_tilehist  per-tile 256-bin histograms (per-wave LDS copies,
//                     BORDER_REFLECT_101 padding folded into the indexing)
//   k_clahe_lut       clip / redistribute / cumulative LUT, one block per
//                     (tile, clip limit, frame)
//   k_clahe_apply     bilinear blend of the 4 neighbouring tile LUTs; a block
//                     owns a strip of rows that share (ty1, ty2), stages the
//                     (gx+1) column cells' four LUTs packed as one uint32 per
//                     grey level in LDS, so each pixel costs one ds_read_b32
// and the 5 x 51 sweep of aclahe.cpp:160-193 becomes
//   k_clahe_sweep     a block owns interpolation cells, keeps 17 clip limits'
//                     packed LUTs and 17 output histograms in LDS, and never
//                     writes the 255 intermediate images
//   k_entropy         aclaheEntropy (aclahe.cpp:228-248) on 256-bin counts.
// All float32 arithmetic keeps OpenCV's operation order (file is compiled with
// -ffp-contract=off).
#include "uwip_internal.hpp"
#include "device_utils.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>

int uwip_launch_hist_internal(uwip_ctx *ctx, const uwip_batch_u8 *img, uint32_t *d_hist);

namespace {

struct ClaheGeom {
    int rows, cols, gx, gy, tw, th, pc, pr, area;
    float inv_tw, inv_th, lutScale;
};

ClaheGeom make_geom(int rows, int cols, int gx, int gy)
{
    ClaheGeom g{};
    g.rows = rows; g.cols = cols; g.gx = gx; g.gy = gy;
    g.pc = cols; g.pr = rows;
    if (!(cols % gx == 0 && rows % gy == 0)) {      // both pads, as cv::CLAHE does
        g.pr = rows + (gy - (rows % gy));
        g.pc = cols + (gx - (cols % gx));
    }
    g.tw = g.pc / gx; g.th = g.pr / gy;
    g.area = g.tw * g.th;
    g.inv_tw = 1.0f / (float)g.tw;
    g.inv_th = 1.0f / (float)g.th;
    g.lutScale = (float)255 / (float)g.area;
    return g;
}

int clip_from_limit(double clipLimit, int area)
{
    int clip = 0;
    if (clipLimit > 0.0) {
        clip = (int)(clipLimit * area / 256);
        clip = std::max(clip, 1);
    }
    return clip;
}

// first coordinate p in [0, n] whose cell index floor(p*inv - 0.5f) + 1 is >= c
// (cell index is non-decreasing in p).  Same float32 expression as the kernels.
inline int cell_of(int p, float inv) { return (int)floorf((float)p * inv - 0.5f) + 1; }

void cell_starts(int n, int g, float inv, std::vector<int> &starts)
{
    starts.assign(g + 2, n);
    int p = 0;
    for (int c = 0; c <= g; ++c) {
        while (p < n && cell_of(p, inv) < c) ++p;
        starts[c] = p;
    }
    starts[g + 1] = n;
    // cells beyond the last occupied one are empty: starts stay at n
    for (int c = g; c >= 0; --c) starts[c] = std::min(starts[c], starts[c + 1]);
}

struct ClipList {
    int n;
    int clip[51];
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- BGR -> V (max) -------------------------------------------------------
__global__ __launch_bounds__(256) void k_bgr_to_v(const uint8_t *__restrict__ src, size_t sstep,
                                                  size_t sfs, uint8_t *__restrict__ dst,
                                                  size_t dstep, size_t dfs, int rows, int cols,
                                                  int vec)
{
    const int f = blockIdx.z;
    const int y = blockIdx.y;
    const uint8_t *s = src + (size_t)f * sfs + (size_t)y * sstep;
    uint8_t *d = dst + (size_t)f * dfs + (size_t)y * dstep;
    const int groups = (cols + 15) / 16;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < groups; g += gridDim.x * 256) {
        const int x0 = g * 16;
        if (vec && x0 + 16 <= cols) {
            const uint4 a = *reinterpret_cast<const uint4 *>(s + (size_t)x0 * 3);
            const uint4 b = *reinterpret_cast<const uint4 *>(s + (size_t)x0 * 3 + 16);
            const uint4 c = *reinterpret_cast<const uint4 *>(s + (size_t)x0 * 3 + 32);
            const uint32_t w[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
            uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int b0 = 3 * i, b1 = 3 * i + 1, b2 = 3 * i + 2;
                const uint32_t B = (w[b0 >> 2] >> ((b0 & 3) * 8)) & 255u;
                const uint32_t G = (w[b1 >> 2] >> ((b1 & 3) * 8)) & 255u;
                const uint32_t R = (w[b2 >> 2] >> ((b2 & 3) * 8)) & 255u;
                o[i >> 2] |= max(max(B, G), R) << ((i & 3) * 8);
            }
            *reinterpret_cast<uint4 *>(d + x0) = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
            for (int x = x0; x < min(x0 + 16, cols); ++x) {
                const uint8_t B = s[3 * x], G = s[3 * x + 1], R = s[3 * x + 2];
                d[x] = max(max(B, G), R);
            }
        }
    }
}

// ---- cv2.GaussianBlur(img, (3,3), 0) on an 8-bit plane (ACLAHE.py:15) ------------------------------------
// ksize 3 with sigma <= 0 takes OpenCV's fixed table [0.25, 0.5, 0.25]; BORDER_DEFAULT = REFLECT_101.  Both passes are
// exact in fixed point, so the result is (sum of the 3x3 window weighted 1 2 1 / 2 4 2 / 1 2 1) / 16 rounded:
//   rule 0 (OpenCV 3.4.x, bit-exact 8-bit path: ufixedpoint16 -> uchar adds one half and truncates): round half UP
//   rule 1 (OpenCV 3.2, float rows/columns + cvRound): round half to EVEN.   parity unpinned (OpenCV-internal).
constexpr int GS3_ROWS = 8;      // rows per block row of the aligned path
template <bool VEC>
__global__ __launch_bounds__(256) void k_gauss3_u8(const uint8_t *__restrict__ src, size_t sstep, size_t sfs,
                                                   uint8_t *__restrict__ dst, size_t dstep, size_t dfs, int rows, int cols, int rule)
{
    const int f = blockIdx.z, y = blockIdx.y;
    const int ym = rows == 1 ? 0 : (y == 0 ? 1 : y - 1), yp = rows == 1 ? 0 : (y == rows - 1 ? rows - 2 : y + 1);
    const uint8_t *r0 = src + (size_t)f * sfs + (size_t)ym * sstep, *r1 = src + (size_t)f * sfs + (size_t)y * sstep,
                  *r2 = src + (size_t)f * sfs + (size_t)yp * sstep;
    uint8_t *d = dst + (size_t)f * dfs + (size_t)y * dstep;
    auto finish = [&](int s) -> uint32_t {                     // s = window sum, <= 16 * 255
        int q = (s + 8) >> 4;                                  // half up
        if (rule == 1 && (s & 15) == 8) q = ((s >> 4) & 1) ? (s >> 4) + 1 : (s >> 4);      // tie -> even
        return (uint32_t)q;
    };
    if (VEC) {
        // four pixels per thread and GS3_ROWS rows per block row: the horizontal 1 2 1 sums of a row (three dwords: the
        // thread's own and its two neighbours') are formed once and serve the three output rows they touch -- the window
        // sum is an exact integer, so horizontal-then-vertical equals vertical-then-horizontal
        const int n4 = cols >> 2;
        const int y0 = blockIdx.y * GS3_ROWS, y1 = min(y0 + GS3_ROWS, rows);
        const uint8_t *base = src + (size_t)f * sfs;
        for (int g = blockIdx.x * 256 + threadIdx.x; g < n4; g += gridDim.x * 256) {
            const int gl = g == 0 ? 0 : g - 1, gr = g == n4 - 1 ? g : g + 1;
            // h[k] = s[x-1] + 2 s[x] + s[x+1] of the row's pixels x = 4g + k, reflect-101 at the row ends
            auto hrow = [&](int yy, uint32_t (&hh)[4]) {
                const uint32_t *p = reinterpret_cast<const uint32_t *>(base + (size_t)yy * sstep);
                const uint32_t a = p[g], l = p[gl], q = p[gr];
                const uint32_t c0 = a & 255u, c1 = (a >> 8) & 255u, c2 = (a >> 16) & 255u, c3 = a >> 24;
                const uint32_t cm = g == 0 ? c1 : l >> 24, cp = g == n4 - 1 ? c2 : q & 255u;
                hh[0] = cm + 2u * c0 + c1; hh[1] = c0 + 2u * c1 + c2; hh[2] = c1 + 2u * c2 + c3; hh[3] = c2 + 2u * c3 + cp;
            };
            auto refl = [&](int yy) { return rows == 1 ? 0 : (yy < 0 ? 1 : (yy >= rows ? rows - 2 : yy)); };
            uint32_t ha[4], hb[4], hc[4];
            hrow(refl(y0 - 1), ha);
            hrow(y0, hb);
            for (int yy = y0; yy < y1; ++yy) {
                hrow(refl(yy + 1), hc);
                uint32_t o = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) o |= finish((int)(ha[k] + 2u * hb[k] + hc[k])) << (8 * k);
                reinterpret_cast<uint32_t *>(dst + (size_t)f * dfs + (size_t)yy * dstep)[g] = o;
#pragma unroll
                for (int k = 0; k < 4; ++k) { ha[k] = hb[k]; hb[k] = hc[k]; }
            }
        }
        return;
    }
    for (int x = blockIdx.x * 256 + threadIdx.x; x < cols; x += gridDim.x * 256) {
        const int xm = cols == 1 ? 0 : (x == 0 ? 1 : x - 1), xp = cols == 1 ? 0 : (x == cols - 1 ? cols - 2 : x + 1);
        const int v0 = r0[xm] + 2 * r0[x] + r0[xp], v1 = r1[xm] + 2 * r1[x] + r1[xp], v2 = r2[xm] + 2 * r2[x] + r2[xp];
        d[x] = (uint8_t)finish(v0 + 2 * v1 + v2);
    }
}

// ---- C1a: tile histograms ---------------------------------------------------
// One WAVE per (tile, row part); a block is four independent waves, no workgroup barrier.  The wave reads its rows as
// ALIGNED dwords (4 pixels per lane and load); a tile narrower than 64 dwords is covered several rows at a time (lane ->
// (row, dword) of a patch), so a 61-pixel tile of the 32 x 32 grid still keeps 51 of the 64 lanes busy.  Bytes of an
// edge dword that lie outside the tile's columns are masked by a per-lane byte mask computed once.  Same-bin atomics of
// one LDS instruction serialise (tools/ubench/lds_rand.hip: 1.7 ns with distinct banks, 53 ns with one word), and
// neighbouring pixels of a smooth underwater frame fall into few bins: the wave's counters are replicated TH_REP times,
// keyed by the lane index modulo TH_REP, with a replica stride of 256 + 8 words (equal bins of different replicas sit in
// different banks).  The reflect-101 padding of a grid that does not divide the image (columns >= cols) is a per-pixel
// loop over the few padded columns; padded rows are index math.
constexpr int TH_REP = 4;
constexpr int TH_RSTRIDE = 256 + 8;
// BP (tiles of >= TH_BP_MIN pixels whose rows are 16-byte aligned runs: the 2x2 ... 8x8 grids of a 1080p or 4K frame): a
// slot-keyed layout -- TH_BP_SLOTS slots per PAIR of bins, two 16-bit counters per word, lane l counts in slot l mod 16 (a
// slot sees the pixels of four lanes: < 65536 for any part below 1 M pixels), 8 KB per wave -- read with 16 pixels per
// lane and load, four row patches in flight.  Measured per 64 frames (tools/tilehist_only.py): replica form 56-62 us;
// 32 slots (one lane pair per bank, 16 KB per wave, 8 waves per CU) 51 us; 16 slots 40 us; 8 slots 59 us; 4 slots 74 us;
// 2 / 4 / 8 row patches in flight 42-45 / 40 / 40 us.  The flush adds the slots of a bin with a lane-rotated slot index.
constexpr int TH_BP_MIN = 8192;
constexpr int TH_BP_SLOTS = 16;
constexpr int TH_BP_WORDS = 128 * TH_BP_SLOTS;
// FORM 2 (round 4): the slot-keyed layout for ANY tile -- the padded 121 x 68 / 61 x 34 tiles of the 16 x 16 / 32 x 32 grids of a
// 1080p frame, unaligned or strided images.  A tile row is read as 16-byte units starting at the tile's own first column
// (unaligned global_load_dwordx4: the hardware splits what crosses a line); the last unit of a row that is not a multiple of
// 16 is loaded ENDING at the tile's last in-image column and its leading bytes -- pixels of the previous unit -- add zero
// (a per-lane mask bit shifted into the increment: no per-pixel branch); reflect-101 padding columns and rows as in the
// replica form.  The replica form spent 90 / 112 us per 64 frames on those grids (4 pixels per load, 16 lanes per replica),
// against 40 us of the slot-keyed form on the aligned grids.
// Measured per 64 frames of 1080p (tools/tilehist_only.py, UWIP_TILEHIST_GENERAL=0|1): 16 x 16 grid (121 x 68 tiles) 86 -> 67 us;
// 32 x 32 grid (61 x 34 tiles = 2074 pixels: 33 pixels per lane against an 8 KB zero + 2048-slot flush per wave, three row
// passes of 16 rows for 34 rows) 107 -> 157 us: those stay with the replica form.
constexpr int TH_G_MIN = 4096;            // pixels per tile from which the 8 KB zero + flush of the slot-keyed layout pays
template <int FORM>
__global__ __launch_bounds__(256) void k_clahe_tilehist(const uint8_t *__restrict__ src,
                                                        size_t step, size_t fstride, int rows,
                                                        int cols, int gx, int tw, int th, int split,
                                                        int rows_per_part,
                                                        const int *__restrict__ frame_map,
                                                        uint32_t *__restrict__ hists, int tiles,
                                                        const int *__restrict__ nf_dev /*optional: frames that take part*/)
{
    if (nf_dev && (int)blockIdx.y >= *nf_dev) return;
    constexpr bool BP = FORM != 0;
    constexpr int HWORDS = BP ? TH_BP_WORDS : TH_REP * TH_RSTRIDE;
    __shared__ __attribute__((aligned(16))) uint32_t sh[4][HWORDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wave;
    if (item >= tiles * split) return;
    uint32_t *H = sh[wave];
    if (BP) for (int i = lane; i < HWORDS / 4; i += 64) reinterpret_cast<uint4 *>(H)[i] = make_uint4(0, 0, 0, 0);
    else for (int i = lane; i < HWORDS; i += 64) H[i] = 0;
    const int t = item / split, part = item - t * split;
    const int ty = t / gx, tx = t - ty * gx;
    const int f = blockIdx.y;
    const int fr = frame_map ? frame_map[f] : f;
    const uint8_t *base = src + (size_t)fr * fstride;
    const int j0 = part * rows_per_part, j1 = min(th, j0 + rows_per_part);
    uint32_t *my = BP ? H + (lane & (TH_BP_SLOTS - 1)) : H + (lane % TH_REP) * TH_RSTRIDE;
    // count one pixel of value v
    auto add = [&](uint32_t v) {
        if (BP) atomicAdd(&my[(v >> 1) * TH_BP_SLOTS], (v & 1u) ? 65536u : 1u);
        else atomicAdd(&my[v], 1u);
    };
    const int xs = tx * tw, xe = max(xs, min(xs + tw, cols));   // in-image columns [xs, xe); [xe, xs + tw) is reflected padding
    const bool vec = ((reinterpret_cast<uintptr_t>(base) | step) & 3u) == 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if constexpr (FORM == 2) {
        struct __attribute__((packed, aligned(1))) U16 { uint32_t x, y, z, w; };      // 16 bytes at any address
        const int n_in = xe - xs;
        if (n_in > 0) {
            const int nu = (n_in + 15) >> 4, CW = min(nu, 64), RW = 64 / CW;
            const int r = lane / CW, c = lane - r * CW;
            if (r < RW) {
                for (int cb = c; cb < nu; cb += 64) {
                    const int x0 = xs + cb * 16;
                    int xl = x0;
                    uint32_t vm = 0xffffu;                 // bit k: byte k of the unit is a pixel of this unit
                    if (x0 + 16 > xe) { xl = xe - 16; vm = (0xffffu << (x0 - xl)) & 0xffffu; }     // host: cols >= 16
                    const uint8_t *colp = base + xl;
                    constexpr int U = 4;
                    for (int j = j0 + r; j < j1; j += RW * U) {
                        U16 w[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int jj = min(j + u * RW, j1 - 1);           // clamped: a duplicate row is loaded, not counted
                            w[u] = *reinterpret_cast<const U16 *>(colp + (size_t)reflect101(ty * th + jj, rows) * step);
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            if (j + u * RW >= j1) break;
                            const uint32_t q[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
                            for (int k = 0; k < 4; ++k)
#pragma unroll
                                for (int b = 0; b < 4; ++b) {
                                    const uint32_t v = (q[k] >> (8 * b)) & 255u;
                                    atomicAdd(&my[(v >> 1) * TH_BP_SLOTS], ((vm >> (4 * k + b)) & 1u) << ((v & 1u) << 4));
                                }
                        }
                    }
                }
            }
        }
    } else if constexpr (FORM == 1) {
        // fast path (host guarantees: rows 16-byte aligned, tw a multiple of 16, no padded columns): 16 pixels per lane
        // and load, the wave covers RW rows x CW 16-byte units, four such patches in flight (4 KB per wave: at 8 waves
        // per CU that is what the HBM latency needs)
        const int n16 = tw >> 4, CW = min(n16, 64), RW = 64 / CW;
        const int r = lane / CW, c = lane - r * CW;
        if (r < RW) {
            for (int cb = c; cb < n16; cb += 64) {
                const uint8_t *colp = base + xs + cb * 16;
                constexpr int U = 4;
                for (int j = j0 + r; j < j1; j += RW * U) {
                    uint4 w[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int jj = min(j + u * RW, j1 - 1);           // clamped: a duplicate row is loaded, not counted
                        w[u] = *reinterpret_cast<const uint4 *>(colp + (size_t)reflect101(ty * th + jj, rows) * step);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        if (j + u * RW >= j1) break;
                        const uint32_t q[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) { add(q[k] & 255u); add((q[k] >> 8) & 255u); add((q[k] >> 16) & 255u); add(q[k] >> 24); }
                    }
                }
            }
        }
    } else if (xe > xs) {
        // units along a row: aligned dwords (vec) or single pixels; the wave covers a patch of RW rows x CW units
        const int u0 = vec ? (xs >> 2) : xs, nu = vec ? ((xe + 3) >> 2) - u0 : xe - xs;
        const int CW = min(nu, 64), RW = 64 / CW;
        const int r = lane / CW, c = lane - r * CW;
        if (r < RW) {
            for (int cb = c; cb < nu; cb += 64) {
                const int u = u0 + cb;
                uint32_t bmask = 0xfu;          // which bytes of the dword are columns of this tile
                bool whole = true;              // the dword may be loaded as one (it ends inside the image row)
                if (vec) {
                    bmask = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) bmask |= (4 * u + k >= xs && 4 * u + k < xe) ? (1u << k) : 0u;
                    whole = 4 * u + 4 <= cols;
                }
                for (int j = j0 + r; j < j1; j += RW) {
                    const int y = reflect101(ty * th + j, rows);
                    const uint8_t *row = base + (size_t)y * step;
                    if (!vec) { add(row[u]); continue; }
                    if (bmask == 0xfu) {
                        const uint32_t w = reinterpret_cast<const uint32_t *>(row)[u];
                        add(w & 255u); add((w >> 8) & 255u); add((w >> 16) & 255u); add(w >> 24);
                    } else {
                        uint32_t w = 0;
                        if (whole) w = reinterpret_cast<const uint32_t *>(row)[u];
                        else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) if (bmask & (1u << k)) w |= (uint32_t)row[4 * u + k] << (8 * k);
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (bmask & (1u << k)) add((w >> (8 * k)) & 255u);
                    }
                }
            }
        }
    }
    // reflected padding columns [xe, xs + tw): a handful per row
    const int npad = FORM == 1 ? 0 : xs + tw - xe;
    if (npad > 0) {
        for (int i = lane; i < npad * (j1 - j0); i += 64) {
            const int jr = i / npad, x = xe + (i - jr * npad);
            const int y = reflect101(ty * th + j0 + jr, rows);
            add(base[(size_t)y * step + reflect101(x, cols)]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t *out = hists + ((size_t)f * tiles + t) * 256;
    if (BP) {
        for (int bp = lane; bp < 128; bp += 64) {
            uint32_t lo = 0, hi = 0;
#pragma unroll 8
            for (int q = 0; q < TH_BP_SLOTS; ++q) {
                const uint32_t w = H[bp * TH_BP_SLOTS + ((q + lane) & (TH_BP_SLOTS - 1))];      // lane-rotated slot: spreads the banks
                lo += w & 0xffffu; hi += w >> 16;
            }
            if (split == 1) { out[2 * bp] = lo; out[2 * bp + 1] = hi; }
            else { if (lo) atomicAdd(&out[2 * bp], lo); if (hi) atomicAdd(&out[2 * bp + 1], hi); }
        }
        return;
    }
    for (int bn = lane; bn < 256; bn += 64) {
        uint32_t sum = 0;
#pragma unroll
        for (int q = 0; q < TH_REP; ++q) sum += H[q * TH_RSTRIDE + bn];
        if (split == 1) out[bn] = sum;                  // the only writer: no memset, no atomic
        else if (sum) atomicAdd(&out[bn], sum);
    }
}

// Histograms of a g x g grid from those of the 2g x 2g grid when neither is padded (a tile is then exactly four
// tiles of the finer grid): no second pass over the image.
__global__ __launch_bounds__(256) void k_clahe_tilehist_merge(const uint32_t *__restrict__ child, int gx, int tiles,
                                                              uint32_t *__restrict__ hists)
{
    const int t = blockIdx.x, f = blockIdx.y, v = threadIdx.x;
    const int ty = t / gx, tx = t - ty * gx, gx2 = 2 * gx;
    const uint32_t *c = child + (size_t)f * tiles * 4 * 256;
    const size_t c00 = (size_t)(2 * ty) * gx2 + 2 * tx;
    hists[((size_t)f * tiles + t) * 256 + v] = c[c00 * 256 + v] + c[(c00 + 1) * 256 + v] + c[(c00 + gx2) * 256 + v] + c[(c00 + gx2 + 1) * 256 + v];
}

// residual -> stepr | magic << 9 (k_clahe_lut)
struct SteprTab { uint32_t v[256]; };
constexpr SteprTab make_stepr_tab()
{
    SteprTab t{};
    for (int r = 0; r < 256; ++r) {
        const uint32_t stepr = r ? (256u / (uint32_t)r > 1u ? 256u / (uint32_t)r : 1u) : 1u;
        t.v[r] = stepr | ((65536u / stepr + 1u) << 9);
    }
    return t;
}
static __device__ const SteprTab D_STEPR = make_stepr_tab();

// ---- C1b: clip, redistribute, cumulative LUT --------------------------------
// One wave per (tile, frame); lane l owns bins 4l..4l+3 and the wave walks all clip limits with shuffle-only
// reductions and scans (no barriers).  Arithmetic is cv::CLAHE's: integer clip / redistribute, then
// lut = sat_u8(rne(float(cumsum) * lutScale)).
// the wave-level body: h0 = this lane's four bins of the tile's histogram; writes the tile's ncl LUT rows (256 B each, lane l
// the bytes 4l .. 4l+3) from `out` on and, optionally, the tallest bin
__device__ __forceinline__ void clahe_lut_rows(const int (&h0)[4], int lane, float lutScale, const ClipList &cl, int frame_clip /*< 0: none*/,
                                               int rule, uint8_t *__restrict__ out, uint32_t *__restrict__ tile_max_out)
{
    const int ncl = cl.n;
    // the tile's tallest bin: a clip limit at or above it clips nothing (cv::CLAHE clips bins > limit only), so its LUT is
    // the unclipped one -- computed once here, and k_clahe_sweep never evaluates such limits
    int m = max(max(h0[0], h0[1]), max(h0[2], h0[3]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d));
    if (tile_max_out && lane == 0) *tile_max_out = (uint32_t)m;
    auto lut_word = [&](const int h[4]) {
        const int p0 = h[0], p1 = p0 + h[1], p2 = p1 + h[2], p3 = p2 + h[3];
        const int off = (int)wave_incl_scan_u32((uint32_t)p3) - p3;
        // sum * lutScale lies in [0, 255.0001]: v_cvt_pk_u8_f32 (round to nearest even, clamp, byte insert) is sat_u8_rne there
        uint32_t wv = __builtin_amdgcn_cvt_pk_u8_f32((float)(off + p0) * lutScale, 0, 0u);
        wv = __builtin_amdgcn_cvt_pk_u8_f32((float)(off + p1) * lutScale, 1, wv);
        wv = __builtin_amdgcn_cvt_pk_u8_f32((float)(off + p2) * lutScale, 2, wv);
        return __builtin_amdgcn_cvt_pk_u8_f32((float)(off + p3) * lutScale, 3, wv);
    };
    uint32_t w_unclipped = 0;
    bool have_unclipped = false;                 // computed when the first limit that clips nothing asks for it
    for (int c = 0; c < ncl; ++c) {
        const int clip = frame_clip >= 0 ? frame_clip : cl.clip[c];
        uint32_t w;
        if (!(clip > 0 && clip < m)) {                                 // wave-uniform
            if (!have_unclipped) { w_unclipped = lut_word(h0); have_unclipped = true; }
            w = w_unclipped;
        } else {
            int h[4] = {h0[0], h0[1], h0[2], h0[3]};
            int excess = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { excess += max(h[k] - clip, 0); h[k] = min(h[k], clip); }
            excess = (int)wave_sum_u32((uint32_t)excess);
            const int batch = excess >> 8;
            const int residual = excess & 255;
            // stepr = max(256 / residual, 1) and, for v / stepr with v < 256 without a per-lane integer division, the
            // multiplier m = floor(2^16 / stepr) + 1 (q = (v * m) >> 16 is exact here: v * (m * stepr - 2^16) <= 255 * 256 <
            // 2^16) -- both from a 256-entry table indexed by the wave-uniform residual instead of two division sequences
            // per clip limit
            const uint32_t sm = D_STEPR.v[__builtin_amdgcn_readfirstlane(residual)];
            const int stepr = (int)(sm & 511u);
            const uint32_t magic = sm >> 9;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int v = lane * 4 + k;
                h[k] += batch;
                if (residual != 0) {
                    if (rule == 0) {                                                           // OpenCV 3.4.x
                        const uint32_t q = __umul24((uint32_t)v, magic) >> 16;
                        if (__umul24(q, (uint32_t)stepr) == (uint32_t)v && (int)q < residual) h[k]++;
                    } else if (v < residual) h[k]++;                                           // OpenCV 3.2
                }
            }
            w = lut_word(h);
        }
        *reinterpret_cast<uint32_t *>(out + (size_t)c * 256 + lane * 4) = w;
    }
}

__global__ __launch_bounds__(256) void k_clahe_lut(const uint32_t *__restrict__ hists, int tiles, int nf,
                                                   float lutScale, ClipList cl,
                                                   const int *__restrict__ frame_clip, int rule,
                                                   uint8_t *__restrict__ luts, uint32_t *__restrict__ tile_max /*optional*/,
                                                   const int *__restrict__ nf_dev /*optional*/)
{
    const int lane = threadIdx.x & 63;
    const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= (long long)tiles * nf) return;
    const int f = (int)(wv / tiles), t = (int)(wv - (long long)f * tiles);
    if (nf_dev && f >= *nf_dev) return;
    const uint4 hv = *reinterpret_cast<const uint4 *>(hists + ((size_t)f * tiles + t) * 256 + lane * 4);
    const int h0[4] = {(int)hv.x, (int)hv.y, (int)hv.z, (int)hv.w};
    // [frame][tile][clip limit][256]: the wave's rows are one contiguous run (ncl = 1: the plain [frame][tile][256] table)
    clahe_lut_rows(h0, lane, lutScale, cl, frame_clip ? frame_clip[f] : -1, rule, luts + (((size_t)f * tiles + t) * cl.n) * 256,
                   tile_max ? tile_max + (size_t)f * tiles + t : nullptr);
}

// ---- C1a + C1b fused for small or padded tiles: one block per ROW OF TILES (round 4) -------------------------------------------
// The 61 x 34 tiles of the 32 x 32 grid of a 1080p frame (and the 121 x 68 ones of 16 x 16) are too small for a per-wave
// histogram image (8 KB of zeroing and flushing per 2074 pixels) and not aligned to anything, which left the replica form
// at 107 us per 64 frames against 40 us on the big aligned tiles.  Here a 512-thread block owns one row of tiles of one
// frame: gx histograms in LDS -- two 16-bit counters per word, BAND_SLOTS lane-keyed slots per pair of bins, tiles skewed by
// four banks -- filled from WHOLE IMAGE ROWS read as 16-byte units (a unit straddles at most two tiles: the counter base of
// each of its 16 pixels is precomputed per lane, the units a lane owns sit in the same columns of every row), the
// reflect-101 padding columns on a per-pixel path, padding rows by index; then the block's waves turn the histograms into
// LUT rows themselves (clahe_lut_rows) -- the 1 KB per tile of histogram never travels to HBM and back, and there is no
// second launch.
constexpr int BAND_SLOTS = 4;
constexpr int BAND_TSTRIDE = 128 * BAND_SLOTS + 4;          // words per tile: + 4 = the bank skew between neighbouring tiles
constexpr int BAND_THREADS = 512;
// A block may own a PART of the row of tiles (tpb tiles: the 32 tiles of a 32 x 32 grid as two blocks of 16, 33 KB of LDS
// each, four blocks per CU like the 16 x 16 grid instead of two at 66 KB): units at a part's edge straddle a neighbouring
// part's tile, whose pixels add zero here.  (Measured and dropped: the same image as 32-bit counters with two slots per bin --
// 3 instead of 7 vector instructions per pixel, twice the lanes per slot: 16 x 16 grid 48 -> 65 us, 32 x 32 64 -> 62: the
// kernel is bound by LDS collisions, not by its instruction count.)
template <int CHUNKS>       // 64-unit chunks per part of an image row (1, 2, 4, 8); BAND_THREADS / 64 / CHUNKS waves share a chunk
__global__ __launch_bounds__(BAND_THREADS) void k_clahe_band(const uint8_t *__restrict__ src, size_t step, size_t fstride, int rows, int cols,
                                                             int gx, int tw, uint32_t tw_magic, int th, int tpb, int nparts,
                                                             const int *__restrict__ frame_map, float lutScale,
                                                             ClipList cl, const int *__restrict__ frame_clip, int rule,
                                                             uint8_t *__restrict__ luts, uint32_t *__restrict__ tile_max, int tiles,
                                                             const int *__restrict__ nf_dev /*optional*/)
{
    if (nf_dev && (int)blockIdx.y >= *nf_dev) return;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_band[];      // [tpb][BAND_TSTRIDE]
    struct __attribute__((packed, aligned(1))) U16 { uint32_t x, y, z, w; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ty = (int)blockIdx.x / nparts, part = (int)blockIdx.x - ty * nparts, f = blockIdx.y;
    const int tx0 = part * tpb, ntx = min(tpb, gx - tx0);
    const int xA = tx0 * tw, xB = min((tx0 + ntx) * tw, cols);          // this part's in-image columns [xA, xB)
    const int fr = frame_map ? frame_map[f] : f;
    const uint8_t *base = src + (size_t)fr * fstride;
    for (int i = tid; i < ntx * BAND_TSTRIDE; i += BAND_THREADS) s_band[i] = 0;
    __syncthreads();
    constexpr int WPC = BAND_THREADS / 64 / CHUNKS;          // waves per chunk
    const int chunk = wave / WPC, wrow = wave - chunk * WPC;
    const int ua = xA >> 4, ub = (xB + 15) >> 4;             // the 16-byte units (counted from the row start) that touch [xA, xB)
    const int u = ua + chunk * 64 + lane;
    if (u < ub && xB > xA) {
        // this lane's unit: pixels [x0, x0 + 16) of every row -- the last unit of a row whose width is not a multiple of 16
        // is loaded ENDING at the last column and its leading bytes, pixels of the previous unit, add zero
        const int x0 = u * 16;
        int xl = x0;
        uint32_t vm = 0xffffu;
        if (x0 + 16 > cols) { xl = cols - 16; vm = (0xffffu << (x0 - xl)) & 0xffffu; }
        // LDS byte address of the counter image of the tile each of the 16 loaded pixels falls in (+ this lane's slot);
        // pixels left or right of this part's columns belong to another block: their mask bit goes
        uint32_t cbase[16];
        const uint32_t slot = (uint32_t)(lane & (BAND_SLOTS - 1)) * 4u;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int x = xl + k;
            const int tx = (int)__umulhi((unsigned)x, tw_magic);       // x / tw: tw_magic = floor(2^32 / tw) + 1, exact while x * tw < 2^32
            const bool mine = x >= xA && x < xB;
            if (!mine) vm &= ~(1u << k);
            cbase[k] = (uint32_t)(mine ? tx - tx0 : 0) * (BAND_TSTRIDE * 4u) + slot;
        }
        const uint8_t *colp = base + xl;
        constexpr int U = 4;
        for (int j = wrow; j < th; j += WPC * U) {
            U16 w[U];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int jj = min(j + q * WPC, th - 1);           // clamped: a duplicate row is loaded, not counted
                w[q] = *reinterpret_cast<const U16 *>(colp + (size_t)reflect101(ty * th + jj, rows) * step);
            }
#pragma unroll
            for (int q = 0; q < U; ++q) {
                if (j + q * WPC >= th) break;
                const uint32_t d[4] = {w[q].x, w[q].y, w[q].z, w[q].w};
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint32_t v = (d[k >> 2] >> (8 * (k & 3))) & 255u;
                    const uint32_t inc = ((vm >> k) & 1u) << ((v & 1u) << 4);
                    atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(s_band) + cbase[k] + (v >> 1) * (BAND_SLOTS * 4u)), inc);
                }
            }
        }
    }
    // reflected padding columns [cols, gx * tw): every part counts those that fall into ITS tiles (with a grid that is not a
    // power of two the padding can cover more tiles than the last part owns: cols = 1915, gx = 49 -> tw = 40, padding over
    // tiles 47 and 48, the last part holding tile 48 alone)
    {
        const int pa = max(cols, tx0 * tw), pb = (tx0 + ntx) * tw;
        const int npad = pb - pa;
        for (int i = tid; i < npad * th; i += BAND_THREADS) {
            const int jr = i / npad, x = pa + (i - jr * npad);
            const uint32_t v = base[(size_t)reflect101(ty * th + jr, rows) * step + reflect101(x, cols)];
            const int tx = x / tw - tx0;
            atomicAdd(&s_band[tx * BAND_TSTRIDE + (v >> 1) * BAND_SLOTS + (tid & (BAND_SLOTS - 1))], (v & 1u) ? 65536u : 1u);
        }
    }
    __syncthreads();
    // histograms -> LUT rows: a wave per tile, lane l owns bins 4l .. 4l+3 = pairs 2l, 2l + 1
    for (int tl = wave; tl < ntx; tl += BAND_THREADS / 64) {
        const uint4 a = *reinterpret_cast<const uint4 *>(&s_band[tl * BAND_TSTRIDE + (2 * lane) * BAND_SLOTS]);
        const uint4 b = *reinterpret_cast<const uint4 *>(&s_band[tl * BAND_TSTRIDE + (2 * lane + 1) * BAND_SLOTS]);
        static_assert(BAND_SLOTS == 4, "one 16-byte read per pair of bins");
        // a tile holds < 65536 pixels (host: checked), so the packed halves can be added as whole words
        const uint32_t sa = a.x + a.y + a.z + a.w, sb = b.x + b.y + b.z + b.w;
        const int h0[4] = {(int)(sa & 0xffffu), (int)(sa >> 16), (int)(sb & 0xffffu), (int)(sb >> 16)};
        const int t = ty * gx + tx0 + tl;
        clahe_lut_rows(h0, lane, lutScale, cl, frame_clip ? frame_clip[f] : -1, rule, luts + (((size_t)f * tiles + t) * cl.n) * 256,
                       tile_max ? tile_max + (size_t)f * tiles + t : nullptr);
    }
}

// ---- C1c: bilinear LUT interpolation, strip per block -------------------------
// strips[s] = (cy, r0, r1, unused): rows [r0,r1) all have floor(y*inv_th-0.5)+1 == cy.
// One launch may mix tile grids (the per-frame parameters of the aclahe stage): a per-frame descriptor then
// replaces the launch-wide geometry, so that 64 frames stay ONE long launch instead of one short launch per grid size.
struct ApplyFrame {
    const int4 *strips;     // this frame's strip list
    const uint8_t *luts;    // its tile LUTs [gy * gx][256]
    int fr;                 // frame index in src / dst
    int nstrips, gx, gy, TX, xs;    // xs: column parts per strip
    float inv_tw, inv_th;
};
template <bool VEC>
__global__ __launch_bounds__(256) void k_clahe_apply(const uint8_t *__restrict__ src, size_t sstep,
                                                     size_t sfs, uint8_t *__restrict__ dst,
                                                     size_t dstep, size_t dfs, int cols, int gx,
                                                     int gy, float inv_tw, float inv_th,
                                                     const uint8_t *__restrict__ luts,
                                                     size_t lut_fs, const int4 *__restrict__ strips,
                                                     const int *__restrict__ frame_map, int TX, int xs,
                                                     const ApplyFrame *__restrict__ desc)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_pack[];   // [cells of this block's columns][256]
    const int tid = threadIdx.x;
    const int f = blockIdx.y;
    int fr = frame_map ? frame_map[f] : f;
    const uint8_t *lbase = luts + (size_t)f * lut_fs;
    int nstrips = (int)gridDim.x / max(xs, 1);
    if (desc) {                                     // block-uniform
        const ApplyFrame d = desc[f];
        strips = d.strips; lbase = d.luts; fr = d.fr; gx = d.gx; gy = d.gy; TX = d.TX; xs = d.xs; inv_tw = d.inv_tw; inv_th = d.inv_th;
        nstrips = d.nstrips;
    }
    // A block = one strip of rows that share (ty1, ty2) x one of xs column parts: the LUT row it stages (1 KB per
    // interpolation cell) shrinks with the part, and tall strips amortise it over more rows (at 32 x 32 tiles and 16-row
    // strips the LUT row was as many bytes as the pixels).
    const int strip = (int)blockIdx.x / xs, part = (int)blockIdx.x - strip * xs;
    if (strip >= nstrips) return;
    const int4 sd = strips[strip];
    const int cy = sd.x, r0 = sd.y, r1 = sd.z;
    const int groups = (cols + 7) / 8;
    const int g_lo = (int)(((long long)groups * part) / xs), g_hi = (int)(((long long)groups * (part + 1)) / xs);
    if (g_lo >= g_hi) return;
    // cells touched by columns [8 g_lo, 8 g_hi): cell(x) = floor(x * inv_tw - 0.5) + 1, non-decreasing in x
    const int c_lo = (int)floorf((float)(g_lo * 8) * inv_tw - 0.5f) + 1;
    const int c_hi = (int)floorf((float)(min(g_hi * 8, cols) - 1) * inv_tw - 0.5f) + 1;
    // the packed LUT row of this strip's cells, built while it is staged: entry v of interpolation cell c holds the four
    // neighbour tiles' LUT bytes TL | TR << 8 | BL << 16 | BR << 24.  A thread takes four grey levels of a cell: one dword
    // from each of the four tile LUTs (L2 hits: a tile serves four cells and every strip of its cell rows), byte-
    // transposed by v_perm, one 16-byte LDS store.  (Round 2 packed the rows in a kernel of its own and staged copies.)
    {
        const uint32_t *L4 = reinterpret_cast<const uint32_t *>(lbase);
        const int ty1 = max(cy - 1, 0), ty2 = min(cy, gy - 1);
        for (int idx = tid; idx < (c_hi - c_lo + 1) * 64; idx += 256) {
            const int c = c_lo + (idx >> 6), v4 = idx & 63;
            const int tx1 = max(c - 1, 0), tx2 = min(c, gx - 1);
            const uint32_t a = L4[((size_t)ty1 * gx + tx1) * 64 + v4], b = L4[((size_t)ty1 * gx + tx2) * 64 + v4];
            const uint32_t cc = L4[((size_t)ty2 * gx + tx1) * 64 + v4], d = L4[((size_t)ty2 * gx + tx2) * 64 + v4];
            const uint32_t t0 = __builtin_amdgcn_perm(b, a, 0x05010400u), t1 = __builtin_amdgcn_perm(b, a, 0x07030602u);
            const uint32_t u0 = __builtin_amdgcn_perm(d, cc, 0x05010400u), u1 = __builtin_amdgcn_perm(d, cc, 0x07030602u);
            reinterpret_cast<uint4 *>(s_pack)[idx] =
                make_uint4(__builtin_amdgcn_perm(u0, t0, 0x05040100u), __builtin_amdgcn_perm(u0, t0, 0x07060302u),
                           __builtin_amdgcn_perm(u1, t1, 0x05040100u), __builtin_amdgcn_perm(u1, t1, 0x07060302u));
        }
    }
    __syncthreads();
    const uint8_t *sb = src + (size_t)fr * sfs;
    uint8_t *db = dst + (size_t)fr * dfs;
    const int tx = tid % TX, ty = tid / TX, TY = 256 / TX;
    for (int g = g_lo + tx; g < g_hi; g += TX) {
        const int x0 = g * 8;
        uint32_t base[8];
        float xa[8], xa1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float txf = (float)(x0 + i) * inv_tw - 0.5f;
            const float fl = floorf(txf);
            xa[i] = txf - fl;
            xa1[i] = 1.0f - xa[i];
            base[i] = (uint32_t)(((int)fl + 1 - c_lo) << 8);
        }
        const bool full = VEC && (x0 + 8 <= cols);
        constexpr int RU = 4;                       // rows in flight per thread (memory-level parallelism)
        for (int yb = r0 + ty; yb < r1; yb += TY * RU) {
            uint32_t w[RU][2];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int y = yb + u * TY;
                w[u][0] = w[u][1] = 0;
                if (y < r1) {
                    const uint8_t *sp = sb + (size_t)y * sstep + x0;
                    if (full) {
                        const uint2 q = *reinterpret_cast<const uint2 *>(sp);
                        w[u][0] = q.x; w[u][1] = q.y;
                    } else {
                        for (int i = 0; i < min(8, cols - x0); ++i) w[u][i >> 2] |= (uint32_t)sp[i] << ((i & 3) * 8);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int y = yb + u * TY;
                if (y >= r1) break;
                const float tyf = (float)y * inv_th - 0.5f;
                const float ya = tyf - floorf(tyf), ya1 = 1.0f - ya;
                uint8_t *dp = db + (size_t)y * dstep + x0;
                uint32_t o[2] = {0, 0};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t v = (w[u][i >> 2] >> ((i & 3) * 8)) & 255u;
                    const uint32_t p = s_pack[base[i] + v];
                    // (TL*xa1 + TR*xa)*ya1 + (BL*xa1 + BR*xa)*ya, OpenCV's products, sums and order (no FMA: -ffp-contract=off)
                    // plain f32 operations: on gfx950 a v_pk_mul/add_f32 costs 2.6x a v_mul/add_f32 (tools/ubench/valu_rate.hip)
                    const float top = (float)(p & 255u) * xa1[i] + (float)((p >> 8) & 255u) * xa[i];
                    const float bot = (float)((p >> 16) & 255u) * xa1[i] + (float)(p >> 24) * xa[i];
                    const float res = top * ya1 + bot * ya;
                    // v_cvt_pk_u8_f32: round-to-nearest-even + clamp + byte insert in one instruction
                    o[i >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(res, i & 3, o[i >> 2]);
                }
                if (full) {
                    *reinterpret_cast<uint2 *>(dp) = make_uint2(o[0], o[1]);
                } else {
                    for (int i = 0; i < min(8, cols - x0); ++i) dp[i] = (uint8_t)(o[i >> 2] >> ((i & 3) * 8));
                }
            }
        }
    }
}

// ---- C3: the sweep, one interpolation cell (chunk) at a time ------------------
constexpr int SWEEP_NCL = 51;      // clip limits 0, 0.5, ..., 25  (aclahe.cpp:181)
constexpr int SWEEP_GROUP = 17;    // clip limits per block (51 = 3 x 17)

struct CellItem {
    int cx, cy;       // cell indices in [0,gx] x [0,gy]
    int x0, x1;       // pixel columns [x0,x1)
    int r0, r1;       // pixel rows    [r0,r1)
    int pad0, pad1;
};

// Same-bin LDS atomics from one wave serialise, and neighbouring pixels of a smooth underwater frame land in
// few bins: the output histograms are therefore replicated SWEEP_REP times, keyed by the thread index modulo
// SWEEP_REP.  A block is 512 threads; packed LUTs (17 KB) + 3 replicas of 16-bit counters (27 KB) + the tail
// histograms (8 KB, below) = 52 KB of LDS, three blocks per CU (measured: 2 replicas x 4 blocks and 4 replicas x 2
// blocks are both slower).
constexpr int SWEEP_THREADS = 512;
constexpr int SWEEP_REP = 3;
constexpr int SWEEP_HROWS = (SWEEP_GROUP + 1) / 2;   // two clip limits share a word: 16-bit counters (a block sees < 65536 pixels)
constexpr int SWEEP_SPREAD = 8;    // multiple of SWEEP_THREADS / 64
constexpr int SWEEP_RSTRIDE = SWEEP_HROWS * 256 + 8;   // +8 words: equal bins of different replicas fall in different LDS banks
// A clip limit at or above the tallest bin of a cell's four tiles clips nothing: the 17 limits of a group therefore give
// `nd` different LUTs followed by 17 - nd repeats of the last one (clip limits grow with their index).  The repeats are
// never evaluated: the pixel's output under limit nd-1 is counted ONCE, in a tail histogram T[nd-1], and the flush adds
// T[0..c] to H[c].  (T[16] does not exist: nd = 17 has no repeats.)  One copy, 16-bit counter pairs like H.
constexpr int SWEEP_TROWS = (SWEEP_GROUP - 1) / 2;
// Whole groups of repeats are not even walked: when limit 16 (33) already clips nothing in a cell, every limit of group 1
// (2) gives the outputs of limit 0, the unclipped LUT.  The block of group 0 counts those once more in G and adds G to
// the rows of groups 1 / 2 at flush time; the blocks of groups 1 / 2 skip the cell.  G lives in the one counter slot H
// leaves free (the high half of row 8: 17 limits in 18 slots): replica 0 for cells where groups 1 and 2 repeat, replica 1
// for cells where only group 2 does.
constexpr size_t SWEEP_LDS_WORDS = (size_t)SWEEP_GROUP * 256 + (size_t)SWEEP_REP * SWEEP_RSTRIDE + (size_t)SWEEP_TROWS * 256;
static_assert(SWEEP_REP >= 2 && (SWEEP_GROUP & 1) == 1 && SWEEP_LDS_WORDS * 4 + 128 <= 54528, "three blocks per CU (tools/ubench/lds_occ.hip: 54528 B is the most LDS a block of three may hold)");
// (TL*xa1 + TR*xa)*ya1 + (BL*xa1 + BR*xa)*ya -> RNE, clamped byte; pk = TL | TR << 8 | BL << 16 | BR << 24.
// Plain f32 multiplies and adds in OpenCV's order (no FMA): on gfx950 a v_pk_mul/add_f32 costs 2.6x a v_mul/add_f32
// (tools/ubench/valu_rate.hip: 2.97 vs 1.14 ns per wave-instruction), so the two-rows-per-packed-pair form lost.
__device__ __forceinline__ uint32_t sweep_eval(uint32_t pk, float xa1, float xa, float ya1, float ya)
{
    const float top = (float)(pk & 255u) * xa1 + (float)((pk >> 8) & 255u) * xa;
    const float bot = (float)((pk >> 16) & 255u) * xa1 + (float)(pk >> 24) * xa;
    return __builtin_amdgcn_cvt_pk_u8_f32(top * ya1 + bot * ya, 0, 0u);     // RNE + clamp
}

// clip limits K0 .. K0+N-1 of one pixel: the N LUT reads go out together, their evaluations interleave
template <int K0, int N>
__device__ __forceinline__ void sweep_run(const uint32_t *pack_v, uint32_t *my_hist, float xa1, float xa, float ya1, float ya)
{
    uint32_t pk[N];
#pragma unroll
    for (int i = 0; i < N; ++i) pk[i] = pack_v[(K0 + i) * 256];
    // Limits 2j and 2j + 1 share a counter word (low / high half).  Where a pixel's two outputs agree -- neighbouring limits
    // often blend the same four LUT entries -- ONE atomic adds to both halves, and the second one runs only for the lanes that
    // differ (fewer active lanes = fewer same-bank collisions in the LDS pipe, the kernel's other limit).
    uint32_t o[N];
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = sweep_eval(pk[i], xa1, xa, ya1, ya);
    constexpr int FIRST = K0 & 1;            // an odd first limit is the high half of a word on its own
    if constexpr (FIRST) atomicAdd(&my_hist[(K0 >> 1) * 256 + o[0]], 65536u);
#pragma unroll
    for (int i = FIRST; i + 1 < N; i += 2) {
        const bool same = o[i] == o[i + 1];
        atomicAdd(&my_hist[((K0 + i) >> 1) * 256 + o[i]], same ? 0x10001u : 1u);
        if (!same) atomicAdd(&my_hist[((K0 + i) >> 1) * 256 + o[i + 1]], 65536u);
    }
    if constexpr (((N - FIRST) & 1) != 0) atomicAdd(&my_hist[((K0 + N - 1) >> 1) * 256 + o[N - 1]], 1u);
}

__global__ __launch_bounds__(SWEEP_THREADS) void k_clahe_sweep(const uint8_t *__restrict__ src, size_t step,
                                                     size_t fstride, int gx, int gy, float inv_tw,
                                                     float inv_th,
                                                     const uint8_t *__restrict__ luts /*[F][tiles][51][256]*/,
                                                     const CellItem *__restrict__ items, int nitems,
                                                     int items_per_block,
                                                     uint32_t *__restrict__ out_hist /*[F][51][256]*/,
                                                     size_t out_fs, const uint32_t *__restrict__ tile_max /*[F][tiles]*/,
                                                     ClipList cl, int rem_mode)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_sweep[];
    uint32_t *s_pack = s_sweep;                                   // [SWEEP_GROUP][256]
    uint32_t *s_hist = s_sweep + SWEEP_GROUP * 256;               // [SWEEP_REP][SWEEP_HROWS][256], two 16-bit counters per word
    uint32_t *s_tail = s_hist + SWEEP_REP * SWEEP_RSTRIDE;        // [SWEEP_TROWS][256], same packing
    const int tid = threadIdx.x;
    const int cg = blockIdx.y, f = blockIdx.z;
    const int clip_g1 = cl.clip[SWEEP_GROUP - 1], clip_g2 = cl.clip[2 * SWEEP_GROUP - 1];
    const int tiles = gx * gy;
    const uint32_t *tmax = tile_max + (size_t)f * tiles;
    const int i0 = blockIdx.x * items_per_block, i1 = min(nitems, i0 + items_per_block);
    if (cg != 0) {
        // nothing to do when every cell of this block repeats limit 0 throughout the group (block-uniform)
        const uint32_t clip_prev = (uint32_t)(cg == 1 ? clip_g1 : clip_g2);
        bool any = false;
        for (int it = i0; it < i1; ++it) {
            const CellItem ci = items[it];
            const int tx1 = max(ci.cx - 1, 0), tx2 = min(ci.cx, gx - 1), ty1 = max(ci.cy - 1, 0), ty2 = min(ci.cy, gy - 1);
            any = any || clip_prev < max(max(tmax[ty1 * gx + tx1], tmax[ty1 * gx + tx2]), max(tmax[ty2 * gx + tx1], tmax[ty2 * gx + tx2]));
        }
        if (!any) return;
    }
    for (int i = tid; i < SWEEP_REP * SWEEP_RSTRIDE + SWEEP_TROWS * 256; i += SWEEP_THREADS) s_hist[i] = 0;
    uint32_t *my_hist = s_hist + (tid % SWEEP_REP) * SWEEP_RSTRIDE;
    const uint8_t *fb = src + (size_t)f * fstride;
    const uint8_t *L = luts + ((size_t)f * tiles * SWEEP_NCL + (size_t)cg * SWEEP_GROUP) * 256;      // + (tile * 51 + c) * 256
    for (int it = i0; it < i1; ++it) {
        const CellItem ci = items[it];
        const int tx1 = max(ci.cx - 1, 0), tx2 = min(ci.cx, gx - 1);
        const int ty1 = max(ci.cy - 1, 0), ty2 = min(ci.cy, gy - 1);
        const uint32_t cellmax = max(max(tmax[ty1 * gx + tx1], tmax[ty1 * gx + tx2]), max(tmax[ty2 * gx + tx1], tmax[ty2 * gx + tx2]));
        const bool rep1 = (uint32_t)clip_g1 >= cellmax, rep2 = (uint32_t)clip_g2 >= cellmax;   // rep1 implies rep2
        if ((cg == 1 && rep1) || (cg == 2 && rep2)) continue;            // counted by the block of group 0 (block-uniform)
        const bool g_any = cg == 0 && rep2, g_both = cg == 0 && rep1;
        uint32_t *s_g = s_hist + (g_both ? 0 : SWEEP_RSTRIDE) + (SWEEP_HROWS - 1) * 256;
        __syncthreads();
        // nd = how many of the group's 17 limits have LUTs of their own in this cell: limit c repeats limit c-1 once
        // limit c-1 is at or above the tallest bin of the four tiles (both are then the unclipped LUT), and the limits
        // grow with c, so the distinct ones come first.  (Block-uniform; small tiles, where two limits that still clip
        // round to one integer, evaluate such a pair twice: harmless.)
        int nd = 1;
        for (int c = 1; c < SWEEP_GROUP; ++c) nd += (uint32_t)cl.clip[cg * SWEEP_GROUP + c - 1] < cellmax ? 1 : 0;
        const int ns = nd - 1;                                     // limits 0 .. ns-1 go to H, limit ns to T[ns]
        // four grey levels per thread: one dword from each of the four tiles' LUTs, byte-transposed by v_perm into
        // four packed entries (TL | TR << 8 | BL << 16 | BR << 24) and stored as one 16-byte LDS write; rows >= nd are
        // never read
        for (int idx = tid; idx < nd * 64; idx += SWEEP_THREADS) {
            const int c = idx >> 6, v4 = idx & 63;
            const uint32_t *Lc = reinterpret_cast<const uint32_t *>(L + (size_t)c * 256);
            const uint32_t a = Lc[((size_t)ty1 * gx + tx1) * (SWEEP_NCL * 64) + v4];
            const uint32_t b = Lc[((size_t)ty1 * gx + tx2) * (SWEEP_NCL * 64) + v4];
            const uint32_t cc = Lc[((size_t)ty2 * gx + tx1) * (SWEEP_NCL * 64) + v4];
            const uint32_t d = Lc[((size_t)ty2 * gx + tx2) * (SWEEP_NCL * 64) + v4];
            const uint32_t t0 = __builtin_amdgcn_perm(b, a, 0x05010400u), t1 = __builtin_amdgcn_perm(b, a, 0x07030602u);
            const uint32_t u0 = __builtin_amdgcn_perm(d, cc, 0x05010400u), u1 = __builtin_amdgcn_perm(d, cc, 0x07030602u);
            reinterpret_cast<uint4 *>(s_pack)[idx] =
                make_uint4(__builtin_amdgcn_perm(u0, t0, 0x05040100u), __builtin_amdgcn_perm(u0, t0, 0x07060302u),
                           __builtin_amdgcn_perm(u1, t1, 0x05040100u), __builtin_amdgcn_perm(u1, t1, 0x07060302u));
        }
        __syncthreads();
        // with rep1 the last distinct limit is the unclipped LUT itself: its output is reused for G
        const bool g_last = g_both && nd < SWEEP_GROUP;
        const bool g_sep = g_any && !g_last;
        const int w = ci.x1 - ci.x0;
        const int npix = w * (ci.r1 - ci.r0);
        const float inv_w = 1.0f / (float)w;
        // pixel p of the cell -> (x, y); the byte for the NEXT iteration is requested before this one's 17
        // evaluations so its latency hides behind them
        // (p < 2^16, w < 2^16, rows * step < 2^31: 24-bit multiplies and a 32-bit byte offset are exact and full rate,
        // where the 32 x 32 and 64-bit forms are quarter rate)
        auto locate = [&](int p, int &x, int &y) {
            int q = (int)(((float)p + 0.5f) * inv_w);
            int r = p - (int)__umul24((unsigned)q, (unsigned)w);
            if (r < 0) { q--; r += w; }
            if (r >= w) { q++; r -= w; }
            x = ci.x0 + r; y = ci.r0 + q;
        };
        const uint32_t step24 = (uint32_t)step;
        auto pix_at = [&](int x, int y) { return (uint32_t)fb[__umul24((unsigned)y, step24) + (unsigned)x]; };
        int xn = 0, yn = 0;
        uint32_t vnext = 0;
        // The 64 pixels of one LDS-atomic instruction are SWEEP_SPREAD apart (lane i of wave w takes pixel
        // SWEEP_SPREAD*i + w + 8m of every 64*SWEEP_SPREAD), so fewer of them fall into the same output bin than 64
        // neighbours of a smooth frame would.
        constexpr int MS = SWEEP_SPREAD / (SWEEP_THREADS / 64);
        auto pix_of = [&](int t) { return (t / MS) * (64 * SWEEP_SPREAD) + (tid & 63) * SWEEP_SPREAD + (tid >> 6) + (SWEEP_THREADS / 64) * (t % MS); };
        static_assert(MS == 1, "consecutive pixels of a thread are SWEEP_THREADS apart");
        // Whole groups of SWEEP_THREADS pixels go by the spread mapping (a permutation of the group); the remainder of the
        // cell (npix mod 512 pixels) is taken contiguously, one pixel per thread from thread 0 on, so that only
        // ceil(rem / 64) waves run the last round instead of all eight with a few lanes each (a 61 x 34 cell of the 32 x 32
        // grid has 4 whole groups + 26 pixels: 5 rounds for every wave became 4 + one wave's).
        const int nfull = npix / SWEEP_THREADS, rem = npix - nfull * SWEEP_THREADS;
        // A SMALL remainder is not given a round of its own at all (the block would wait a whole round for one wave): its
        // rem x nl evaluations -- nl per pixel: the distinct limits, + the separate G evaluation -- are dealt one per thread,
        // a thread = (pixel, evaluation).  Same counts in the same counters.
        const int nl = (nd == SWEEP_GROUP ? SWEEP_GROUP : nd) + (g_sep ? 1 : 0);
        const bool rem_spread = rem_mode == 2 && rem > 0 && rem * nl <= 4 * SWEEP_THREADS;
        if (rem_spread) {
            for (int idx = tid; idx < rem * nl; idx += SWEEP_THREADS) {
                const int pp = idx / nl, e = idx - pp * nl;
                int x, y;
                locate(nfull * SWEEP_THREADS + pp, x, y);
                const uint32_t v = pix_at(x, y);
                const float txf = (float)x * inv_tw - 0.5f;
                const float xa = txf - floorf(txf), xa1 = 1.0f - xa;
                const float tyf = (float)y * inv_th - 0.5f;
                const float ya = tyf - floorf(tyf), ya1 = 1.0f - ya;
                const uint32_t *pack_v = s_pack + v;
                if (g_sep && e == nl - 1) { atomicAdd(&s_g[sweep_eval(pack_v[0], xa1, xa, ya1, ya)], 65536u); continue; }
                const uint32_t o = sweep_eval(pack_v[e * 256], xa1, xa, ya1, ya);
                if (nd == SWEEP_GROUP || e < ns) atomicAdd(&my_hist[(e >> 1) * 256 + o], (e & 1) ? 65536u : 1u);
                else {                                              // e == ns: the last distinct limit -> tail histogram
                    atomicAdd(&s_tail[(ns >> 1) * 256 + o], (ns & 1) ? 65536u : 1u);
                    if (g_last) atomicAdd(&s_g[o], 65536u);
                }
            }
        }
        // rem_mode 0 (diagnostic): the remainder by the spread mapping too, i.e. every wave runs the last round with a few lanes
        const int rem_pos = rem_mode == 0 ? pix_of(0) : tid;
        const int ntot = nfull + ((!rem_spread && rem_pos < rem) ? 1 : 0);
        int t = 0;
        if (ntot > 0) { locate(nfull > 0 ? pix_of(0) : rem_pos, xn, yn); vnext = pix_at(xn, yn); }
        // a thread's next pixel is SWEEP_THREADS further along the cell: step (x, y) instead of dividing again
        const int dq = SWEEP_THREADS / w, dr = SWEEP_THREADS - dq * w;     // wave-uniform
        for (; t < ntot;) {
            const int x = xn, y = yn;
            const uint32_t v = vnext;
            ++t;
            if (t < ntot) {
                if (t < nfull) {
                    xn += dr; yn += dq;
                    if (xn >= ci.x1) { xn -= w; yn++; }
                } else {
                    locate(nfull * SWEEP_THREADS + rem_pos, xn, yn);      // the remainder pixel
                }
                vnext = pix_at(xn, yn);
            }
            const float txf = (float)x * inv_tw - 0.5f;
            const float xa = txf - floorf(txf), xa1 = 1.0f - xa;
            const float tyf = (float)y * inv_th - 0.5f;
            const float ya = tyf - floorf(tyf), ya1 = 1.0f - ya;
            const uint32_t *pack_v = s_pack + v;
            if (g_sep) atomicAdd(&s_g[sweep_eval(pack_v[0], xa1, xa, ya1, ya)], 65536u);
            if (nd == SWEEP_GROUP) {
                sweep_run<0, SWEEP_GROUP>(pack_v, my_hist, xa1, xa, ya1, ya);      // every clip limit has its own LUTs
                continue;
            }
            {
                // ns evaluations in straight-line runs of 8 / 4 / 2 / 1 (ns < 16), then the last distinct limit into the tail
                const uint32_t pk_last = pack_v[ns * 256];
                if (ns & 8) sweep_run<0, 8>(pack_v, my_hist, xa1, xa, ya1, ya);
                if (ns & 4) {
                    if (ns & 8) sweep_run<8, 4>(pack_v, my_hist, xa1, xa, ya1, ya);
                    else sweep_run<0, 4>(pack_v, my_hist, xa1, xa, ya1, ya);
                }
                if (ns & 2) {
                    switch (ns & 12) {
                    case 0: sweep_run<0, 2>(pack_v, my_hist, xa1, xa, ya1, ya); break;
                    case 4: sweep_run<4, 2>(pack_v, my_hist, xa1, xa, ya1, ya); break;
                    case 8: sweep_run<8, 2>(pack_v, my_hist, xa1, xa, ya1, ya); break;
                    default: sweep_run<12, 2>(pack_v, my_hist, xa1, xa, ya1, ya); break;
                    }
                }
                if (ns & 1) {
                    switch (ns & 14) {
                    case 0: sweep_run<0, 1>(pack_v, my_hist, xa1, xa, ya1, ya); break;
                    case 2: sweep_run<2, 1>(pack_v, my_hist, xa1, xa, ya1, ya); break;
                    case 4: sweep_run<4, 1>(pack_v, my_hist, xa1, xa, ya1, ya); break;
                    case 6: sweep_run<6, 1>(pack_v, my_hist, xa1, xa, ya1, ya); break;
                    case 8: sweep_run<8, 1>(pack_v, my_hist, xa1, xa, ya1, ya); break;
                    case 10: sweep_run<10, 1>(pack_v, my_hist, xa1, xa, ya1, ya); break;
                    case 12: sweep_run<12, 1>(pack_v, my_hist, xa1, xa, ya1, ya); break;
                    default: sweep_run<14, 1>(pack_v, my_hist, xa1, xa, ya1, ya); break;
                    }
                }
                const uint32_t o_last = sweep_eval(pk_last, xa1, xa, ya1, ya);
                atomicAdd(&s_tail[(ns >> 1) * 256 + o_last], (ns & 1) ? 65536u : 1u);
                if (g_last) atomicAdd(&s_g[o_last], 65536u);
                continue;
            }
        }
    }
    __syncthreads();
    uint32_t *out = out_hist + (size_t)f * out_fs + (size_t)cg * SWEEP_GROUP * 256;
    // H[c] + T[0] + ... + T[min(c, 15)], one thread per grey level walking up the clip limits
    if (tid < 256) {
        uint32_t run = 0;
#pragma unroll
        for (int c = 0; c < SWEEP_GROUP; ++c) {
            const int sh = (c & 1) * 16;
            if (c < SWEEP_GROUP - 1) run += (s_tail[(c >> 1) * 256 + tid] >> sh) & 0xffffu;
            uint32_t sum = run;
#pragma unroll
            for (int r = 0; r < SWEEP_REP; ++r) sum += (s_hist[r * SWEEP_RSTRIDE + (c >> 1) * 256 + tid] >> sh) & 0xffffu;
            if (sum) atomicAdd(&out[c * 256 + tid], sum);
        }
        if (cg == 0) {
            const uint32_t g1 = s_hist[(SWEEP_HROWS - 1) * 256 + tid] >> 16;
            const uint32_t g2 = g1 + (s_hist[SWEEP_RSTRIDE + (SWEEP_HROWS - 1) * 256 + tid] >> 16);
            if (g1) for (int c = SWEEP_GROUP; c < 2 * SWEEP_GROUP; ++c) atomicAdd(&out[c * 256 + tid], g1);
            if (g2) for (int c = 2 * SWEEP_GROUP; c < 3 * SWEEP_GROUP; ++c) atomicAdd(&out[c * 256 + tid], g2);
        }
    }
}

// ---- C2: entropy of 256-bin counts ---------------------------------------------
__global__ __launch_bounds__(256) void k_entropy(const uint32_t *__restrict__ hist, int rows,
                                                 int cols, float *__restrict__ out)
{
    __shared__ double s_term[256];
    const int v = threadIdx.x;
    const size_t h = blockIdx.x;
    const float p = (float)hist[h * 256 + v] / (float)(cols * rows);
    s_term[v] = (double)p * log2((double)p + 0.00001);
    __syncthreads();
    if (v == 0) {
        float e = 0.0f;
        for (int i = 0; i < 256; ++i) e = (float)((double)e + s_term[i]);
        out[h] = -e;
    }
}

// -----------------------------------------------------------------------------
bool aligned_for(const uwip_batch_u8 *b, size_t a)
{
    return ((uintptr_t)b->data % a == 0) && (b->step % a == 0) && (b->frames <= 1 || b->frame_stride % a == 0);
}

int launch_tilehist(uwip_ctx *ctx, const uwip_batch_u8 *src, const ClaheGeom &g, const int *d_frame_map,
                    int nf, uint32_t *d_hists, const int *d_nf = nullptr)
{
    const int tiles = g.gx * g.gy;
    // one wave per (tile, row part).  Large aligned tiles take the slot-keyed form: parts of >= TH_BP_MIN pixels, enough of
    // them for ~4096 waves; small or unaligned tiles the replica form: enough parts for >= 16384 waves, at
    // least 8 rows each.
    // The slot-keyed form counts in 16-bit halves: a slot is shared by the lanes l = s mod 16 of a wave, at most 4 of at
    // least 33 active ones, so it sees at most 1/8 of its part's pixels -- a part must stay below 2^19 pixels or a constant
    // (saturated / black) tile carries the even bin's counter into the odd bin's.  TH_BP_PART_MAX keeps a margin for
    // the rounding of rows_per_part (one more row of < 2^17 columns); wider tiles take the 32-bit replica form.
    constexpr long long TH_BP_PART_MAX = 3ll << 17;                   // 393 216 pixels
    const bool bp = (long long)g.tw * g.th >= TH_BP_MIN && (g.tw & 15) == 0 && g.tw * g.gx == g.cols && g.tw < (1 << 17) &&
                    ((reinterpret_cast<uintptr_t>(src->data) | src->step | src->frame_stride) & 15u) == 0;
    // FORM 2: any other tile of >= TH_G_MIN pixels in an image at least 16 columns wide (the tail unit of a row is loaded
    // ending at the tile's last in-image column)
    const bool bpg = !bp && (long long)g.tw * g.th >= TH_G_MIN && g.cols >= 16 && g.tw >= 16 && g.tw < (1 << 17);
    int split;
    if (bp || bpg) {
        split = (int)(((bp ? 4096 : 16384) + (size_t)tiles * nf - 1) / ((size_t)tiles * nf));
        split = std::max(1, std::min(split, (int)(((long long)g.tw * g.th) / (bp ? TH_BP_MIN : TH_G_MIN))));
        split = std::max(split, (int)(((long long)g.tw * g.th + TH_BP_PART_MAX - 1) / TH_BP_PART_MAX));
        split = std::min(split, g.th);
    } else {
        split = (int)((16384 + (size_t)tiles * nf - 1) / ((size_t)tiles * nf));
        split = std::max(1, std::min(split, std::max(1, g.th / 8)));
    }
    const int rpp = (g.th + split - 1) / split;
    split = (g.th + rpp - 1) / rpp;                       // no empty parts
    if (split > 1) UWIP_HIP(ctx, hipMemsetAsync(d_hists, 0, sizeof(uint32_t) * 256 * (size_t)tiles * nf, ctx->stream));
    dim3 grid((unsigned)((tiles * split + 3) / 4), (unsigned)nf);
    uwip_kscope ks(ctx, "k_clahe_tilehist");
    static const bool no_general = [] { const char *e = std::getenv("UWIP_TILEHIST_GENERAL"); return e && *e == '0'; }();    // A/B
    if (bp)
        k_clahe_tilehist<1><<<grid, 256, 0, ctx->stream>>>((const uint8_t *)src->data, src->step, src->frame_stride, g.rows, g.cols,
                                                           g.gx, g.tw, g.th, split, rpp, d_frame_map, d_hists, tiles, d_nf);
    else if (bpg && !no_general)
        k_clahe_tilehist<2><<<grid, 256, 0, ctx->stream>>>((const uint8_t *)src->data, src->step, src->frame_stride, g.rows, g.cols,
                                                           g.gx, g.tw, g.th, split, rpp, d_frame_map, d_hists, tiles, d_nf);
    else
        k_clahe_tilehist<0><<<grid, 256, 0, ctx->stream>>>((const uint8_t *)src->data, src->step, src->frame_stride, g.rows, g.cols,
                                                           g.gx, g.tw, g.th, split, rpp, d_frame_map, d_hists, tiles, d_nf);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

int launch_lut(uwip_ctx *ctx, const ClaheGeom &g, const uint32_t *d_hists, const ClipList &cl,
               const int *d_frame_clip, int nf, int rule, uint8_t *d_luts, uint32_t *d_tile_max = nullptr, const int *d_nf = nullptr)
{
    const int tiles = g.gx * g.gy;
    uwip_kscope ks(ctx, "k_clahe_lut");
    k_clahe_lut<<<uwip_cdiv((size_t)tiles * nf, 4), 256, 0, ctx->stream>>>(d_hists, tiles, nf, g.lutScale, cl, d_frame_clip, rule, d_luts, d_tile_max, d_nf);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

// tile histograms + LUT rows in one launch (k_clahe_band) where the geometry allows and the big-tile form does not apply
bool band_ok(const uwip_batch_u8 *src, const ClaheGeom &g)
{
    static const bool off = [] { const char *e = std::getenv("UWIP_CLAHE_BAND"); return e && *e == '0'; }();      // A/B
    const bool bp = (long long)g.tw * g.th >= TH_BP_MIN && (g.tw & 15) == 0 && g.tw * g.gx == g.cols && g.tw < (1 << 17) &&
                    ((reinterpret_cast<uintptr_t>(src->data) | src->step | src->frame_stride) & 15u) == 0;
    // tw >= 2: the column -> tile division is a multiply-high by floor(2^32 / tw) + 1, which does not exist for tw = 1
    return !off && !bp && g.tw >= 2 && g.cols >= 16 && g.cols <= 8192 && (long long)g.tw * g.th < 65536 && g.gy <= 4096;
}

int launch_band(uwip_ctx *ctx, const uwip_batch_u8 *src, const ClaheGeom &g, const int *d_frame_map, int nf, const ClipList &cl,
                const int *d_frame_clip, int rule, uint8_t *d_luts, uint32_t *d_tile_max, const int *d_nf = nullptr)
{
    // tiles per block: as many as fill one 64-unit chunk (1024 columns: every lane of the block's eight waves busy), at most
    // 16 (33 KB of LDS: four blocks per CU).  Measured per 64 frames of 1080p (tools/tilehist_only.py, UWIP_BAND_TPB): 16 x 16
    // grid (121-pixel tiles) 8 tiles per block 43.9 us, 16 (the whole row) 48.5; 32 x 32 grid (61-pixel tiles) 16 per block
    // 63.1, 32 (the whole row, 66 KB) 66.5, 8 (half the lanes idle) 99.5.
    static const int env_tpb = [] { const char *e = std::getenv("UWIP_BAND_TPB"); return e && *e ? std::atoi(e) : 0; }();
    int tpb = std::min(g.gx, std::min(16, std::max(1, 1024 / g.tw)));
    if (env_tpb > 0) tpb = std::min(env_tpb, g.gx);
    const int nparts = (g.gx + tpb - 1) / tpb;
    const int part_cols = std::min(tpb * g.tw, g.cols);
    const int nu = (part_cols + 15) / 16 + 1;                          // a part's columns need not start on a unit boundary
    const int chunks = nu <= 64 ? 1 : (nu <= 128 ? 2 : (nu <= 256 ? 4 : 8));
    const size_t lds = (size_t)tpb * BAND_TSTRIDE * 4;
    const uint32_t magic = (uint32_t)(4294967296ull / (uint64_t)g.tw) + 1u;
    const int tiles = g.gx * g.gy;
    dim3 grid((unsigned)(g.gy * nparts), (unsigned)nf);
    uwip_kscope ks(ctx, "k_clahe_band");
#define UWIP_BAND(C)                                                                                                                   \
    do {                                                                                                                               \
        int rc_l = uwip_lds_optin(ctx, "k_clahe_band" #C, (const void *)k_clahe_band<C>, lds);                                          \
        if (rc_l) return rc_l;                                                                                                         \
        k_clahe_band<C><<<grid, BAND_THREADS, lds, ctx->stream>>>((const uint8_t *)src->data, src->step, src->frame_stride, g.rows, g.cols, \
                                                               g.gx, g.tw, magic, g.th, tpb, nparts, d_frame_map, g.lutScale, cl, d_frame_clip, rule, \
                                                               d_luts, d_tile_max, tiles, d_nf);                                        \
    } while (0)
    switch (chunks) {
    case 1: UWIP_BAND(1); break;
    case 2: UWIP_BAND(2); break;
    case 4: UWIP_BAND(4); break;
    default: UWIP_BAND(8); break;
    }
#undef UWIP_BAND
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

// strip table for (rows, gy, th, max_rows): built once per geometry and cached
int build_strips(uwip_ctx *ctx, const ClaheGeom &g, int max_rows, const int4 **d_strips, int *nstrips)
{
    char key[96];
    snprintf(key, sizeof key, "strips:%d:%d:%d:%d", g.rows, g.gy, g.th, max_rows);
    size_t bytes = 0;
    const void *d = uwip_table_find(ctx, key, &bytes);
    if (!d) {
        std::vector<int> ys;
        cell_starts(g.rows, g.gy, g.inv_th, ys);
        std::vector<int4> strips;
        for (int cy = 0; cy <= g.gy; ++cy) {
            const int L = ys[cy + 1] - ys[cy];
            if (L <= 0) continue;
            const int nch = (L + max_rows - 1) / max_rows, rows = (L + nch - 1) / nch;     // balanced chunks
            for (int r = ys[cy]; r < ys[cy + 1]; r += rows)
                strips.push_back(make_int4(cy, r, std::min(r + rows, ys[cy + 1]), 0));
        }
        bytes = strips.size() * sizeof(int4);
        d = uwip_table_put(ctx, key, strips.data(), bytes);
        if (!d) return UWIP_ERR_NOMEM;
    }
    *d_strips = (const int4 *)d;
    *nstrips = (int)(bytes / sizeof(int4));
    return UWIP_OK;
}

// launch shape of the interpolation for one geometry: column parts per strip, threads along x, rows per strip and the
// cells (KB of LDS) a block can touch
struct ApplyShape { int xs, TX, max_rows, lds_cells; };
ApplyShape apply_shape(const ClaheGeom &g)
{
    ApplyShape a;
    const int groups = (g.cols + 7) / 8;
    a.xs = groups >= 128 ? 4 : (groups >= 32 ? 2 : 1);
    const int pg = (groups + a.xs - 1) / a.xs;          // groups per part
    a.TX = 256;
    while (a.TX > 32 && a.TX / 2 >= pg) a.TX /= 2;
    // strips as tall as a cell row (capped): the staged LUT bytes per pixel fall with the strip height
    a.max_rows = std::min(std::max(g.th, 16), 64);
    // a part of pg groups spans at most (8 pg) / tw + 2 cells
    a.lds_cells = std::min(g.gx + 1, (8 * pg + g.tw - 1) / g.tw + 2);
    return a;
}

int launch_apply(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst, const ClaheGeom &g,
                 const uint8_t *d_luts, size_t lut_fs, const int *d_frame_map, int nf)
{
    const ApplyShape sh = apply_shape(g);
    const int4 *d_strips = nullptr;
    int nstrips = 0;
    int rc = build_strips(ctx, g, sh.max_rows, &d_strips, &nstrips);
    if (rc) return rc;
    if (nstrips == 0) return UWIP_OK;
    const size_t lds = (size_t)sh.lds_cells * 256 * sizeof(uint32_t);
    const bool vec = aligned_for(src, 8) && aligned_for(dst, 8);
    dim3 grid((unsigned)(nstrips * sh.xs), (unsigned)nf);
    uwip_kscope ks(ctx, "k_clahe_apply");
    if (vec)
        k_clahe_apply<true><<<grid, 256, lds, ctx->stream>>>((const uint8_t *)src->data, src->step, src->frame_stride,
                                                             (uint8_t *)dst->data, dst->step, dst->frame_stride, g.cols,
                                                             g.gx, g.gy, g.inv_tw, g.inv_th, d_luts, lut_fs, d_strips,
                                                             d_frame_map, sh.TX, sh.xs, nullptr);
    else
        k_clahe_apply<false><<<grid, 256, lds, ctx->stream>>>((const uint8_t *)src->data, src->step, src->frame_stride,
                                                              (uint8_t *)dst->data, dst->step, dst->frame_stride, g.cols,
                                                              g.gx, g.gy, g.inv_tw, g.inv_th, d_luts, lut_fs, d_strips,
                                                              d_frame_map, sh.TX, sh.xs, nullptr);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

// all frames of a batch in ONE launch, each with its own geometry (desc[f], device memory)
int launch_apply_mixed(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst, const ApplyFrame *d_desc, int nf,
                       int max_blocks, int max_cells)
{
    if (max_blocks == 0 || nf == 0) return UWIP_OK;
    const size_t lds = (size_t)max_cells * 256 * sizeof(uint32_t);
    const bool vec = aligned_for(src, 8) && aligned_for(dst, 8);
    dim3 grid((unsigned)max_blocks, (unsigned)nf);
    uwip_kscope ks(ctx, "k_clahe_apply");
    if (vec)
        k_clahe_apply<true><<<grid, 256, lds, ctx->stream>>>((const uint8_t *)src->data, src->step, src->frame_stride,
                                                             (uint8_t *)dst->data, dst->step, dst->frame_stride, src->cols, 0, 0,
                                                             0.f, 0.f, nullptr, 0, nullptr, nullptr, 0, 1, d_desc);
    else
        k_clahe_apply<false><<<grid, 256, lds, ctx->stream>>>((const uint8_t *)src->data, src->step, src->frame_stride,
                                                              (uint8_t *)dst->data, dst->step, dst->frame_stride, src->cols, 0, 0,
                                                              0.f, 0.f, nullptr, 0, nullptr, nullptr, 0, 1, d_desc);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

// ---- the block-size search at a clip limit OUTSIDE the swept grid, on the device (ACLAHE.py:102-112) -----------------------
// A knee index d >= 26 makes d itself (the reference uses the index as a clip limit) leave the sweep's 0 .. 25: the five
// entropies at that clip limit are then evaluated for the frame.  It takes a degenerate fit to get here, so
// this is written for correctness, not speed: one block per (flagged frame, grid) -- tile histograms by global atomics into
// the (frame, grid) slice of the tile-histogram workspace, LUT rows by clahe_lut_rows, the CLAHE output of every pixel
// (cv::CLAHE's float32 blend, the operations of k_clahe_apply) counted into a 256-bin LDS histogram without being stored,
// aclaheEntropy (k_entropy's operations); the last of a frame's five blocks to finish (an arrival counter in the spare field
// of the frame's parameter record) rewrites the frame's BS.  Blocks of unflagged frames exit at once.  Worst case (every
// frame of a batch flagged: tests/test_clahe_gpu.py measures it) ~13 ms at 1080p -- round 4 walked the five grids in ONE
// block per frame: 63 ms.
// tiles of the grids 2, 4, 8, 16, 32 one after the other: offsets 0, 4, 20, 84, 340 = (4^(k+1) - 4) / 3, 1364 per frame
constexpr int EXACT_TILES = 1364;
__device__ __forceinline__ int exact_tile_off(int k) { return ((4 << (2 * k)) - 4) / 3; }
struct ExactGrids {
    int g[5], tw[5], th[5], pc[5], pr[5], area[5];
    float inv_tw[5], inv_th[5], lutScale[5];
};
__global__ __launch_bounds__(512) void k_aclahe_exact_bs(const uint8_t *__restrict__ src, size_t step, size_t fstride, int rows, int cols,
                                                        ExactGrids G, int rule, int32_t *__restrict__ par /*[F][4]*/,
                                                        uint32_t *__restrict__ hist_ws /*[F][1364][256]*/, uint8_t *__restrict__ lut_ws /*[F][1364][256]*/,
                                                        float *__restrict__ ent_ws /*[F][5]*/)
{
    const int f = blockIdx.x;
    if (par[4 * f + 2] != 1) return;             // (the last block of a flagged frame writes 2 only after all five have read this)
    __shared__ uint32_t s_out[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = par[4 * f + 1];
    const uint8_t *plane = src + (size_t)f * fstride;
    {
        const int k = blockIdx.y;
        uint32_t *hist = hist_ws + ((size_t)f * EXACT_TILES + exact_tile_off(k)) * 256;
        uint8_t *lut = lut_ws + ((size_t)f * EXACT_TILES + exact_tile_off(k)) * 256;
        const int gx = G.g[k], gy = G.g[k], tiles = gx * gy, tw = G.tw[k], th = G.th[k], pc = G.pc[k], pr = G.pr[k];
        for (int i = tid; i < tiles * 256; i += 512) hist[i] = 0;
        if (tid < 256) s_out[tid] = 0;
        __threadfence();
        __syncthreads();
        for (int i = tid; i < pc * pr; i += 512) {
            const int y = i / pc, x = i - y * pc;
            const uint32_t v = plane[(size_t)reflect101(y, rows) * step + reflect101(x, cols)];
            atomicAdd(&hist[((y / th) * gx + x / tw) * 256 + v], 1u);
        }
        __threadfence();
        __syncthreads();
        int clip = 0;                                  // clip_from_limit((double)d, area)
        if (d > 0) { clip = (int)((double)d * G.area[k] / 256); clip = max(clip, 1); }
        ClipList cl;
        cl.n = 1;
        for (int t = wave; t < tiles; t += 8) {
            const uint4 hv = *reinterpret_cast<const uint4 *>(hist + (size_t)t * 256 + lane * 4);
            const int h0[4] = {(int)hv.x, (int)hv.y, (int)hv.z, (int)hv.w};
            clahe_lut_rows(h0, lane, G.lutScale[k], cl, clip, rule, lut + (size_t)t * 256, nullptr);
        }
        __threadfence();
        __syncthreads();
        const float inv_tw = G.inv_tw[k], inv_th = G.inv_th[k];
        for (int i = tid; i < rows * cols; i += 512) {
            const int y = i / cols, x = i - y * cols;
            const uint32_t v = plane[(size_t)y * step + x];
            const float txf = (float)x * inv_tw - 0.5f, tyf = (float)y * inv_th - 0.5f;
            const float flx = floorf(txf), fly = floorf(tyf);
            const float xa = txf - flx, xa1 = 1.0f - xa, ya = tyf - fly, ya1 = 1.0f - ya;
            const int tx1 = max((int)flx, 0), tx2 = min((int)flx + 1, gx - 1), ty1 = max((int)fly, 0), ty2 = min((int)fly + 1, gy - 1);
            const float TL = (float)lut[(size_t)(ty1 * gx + tx1) * 256 + v], TR = (float)lut[(size_t)(ty1 * gx + tx2) * 256 + v];
            const float BL = (float)lut[(size_t)(ty2 * gx + tx1) * 256 + v], BR = (float)lut[(size_t)(ty2 * gx + tx2) * 256 + v];
            const float res = (TL * xa1 + TR * xa) * ya1 + (BL * xa1 + BR * xa) * ya;
            atomicAdd(&s_out[__builtin_amdgcn_cvt_pk_u8_f32(res, 0, 0u)], 1u);
        }
        __syncthreads();
        if (tid == 0) {                                // aclaheEntropy, aclahe.cpp:241-247
            float e = 0.0f;
            for (int i = 0; i < 256; ++i) {
                const float p = (float)s_out[i] / (float)(cols * rows);
                e = (float)((double)e + (double)p * log2((double)p + 0.00001));
            }
            ent_ws[(size_t)f * 5 + k] = -e;
            __threadfence();
            if (atomicAdd(&par[4 * f + 3], 1) == 4) {              // the last of the frame's five blocks
                __threadfence();
                int w = 0;
                float best = 0.f;
                for (int q = 0; q < 5; ++q) {
                    const float h = (float)(_Float16)__hip_atomic_load(&ent_ws[(size_t)f * 5 + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (q == 0 || h >= best) { best = h; w = q; }  // last maximum wins (ACLAHE.py:118-124)
                }
                par[4 * f + 0] = G.g[w];
                par[4 * f + 3] = 0;
                par[4 * f + 2] = 2;                                  // evaluated
            }
        }
    }
}

// ---- the final CLAHE of the aclahe stage launched from DEVICE-side parameters (round 4) -------------------------------------
// The parameter choice is made on the device (aclahe_device.hip); to launch the per-frame CLAHE without bringing {BS, CL}
// back, everything the host used to derive from them is derived here: the frames are grouped by block size (group k =
// block size 2, 4, 8, 16, 32, frame order kept), every group gets its frame map, clip limits and frame count, every frame
// its interpolation descriptor.  The host then launches the tile-histogram / LUT kernels of ALL five grids over the whole
// batch -- a block whose frame index is beyond its group's count exits at once -- and the one mixed interpolation launch.
struct PfGrids {
    const int4 *strips[5];
    uint8_t *luts[5];
    int *map[5], *clip[5];
    int nstrips[5], g[5], TX[5], xs[5], area[5], tiles[5];
    float inv_tw[5], inv_th[5];
};
__global__ void k_pf_prepare(const int32_t *__restrict__ par /*[F][4] = BS, CL, ..*/, int F, PfGrids G, int *__restrict__ count /*[5]*/,
                             ApplyFrame *__restrict__ desc)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int cnt[5] = {0, 0, 0, 0, 0};
    for (int f = 0; f < F; ++f) {                   // a sub-batch: a serial walk keeps every group in frame order
        const int bs = par[4 * f], cl = par[4 * f + 1];
        const int k = bs == 2 ? 0 : (bs == 4 ? 1 : (bs == 8 ? 2 : (bs == 16 ? 3 : 4)));
        int i = 0;
#pragma unroll
        for (int q = 0; q < 5; ++q) if (q == k) i = cnt[q]++;
        G.map[k][i] = f;
        int clip = 0;                                // clip_from_limit((double)cl, area)
        if (cl > 0) { clip = (int)((double)cl * G.area[k] / 256); clip = max(clip, 1); }
        G.clip[k][i] = clip;
        ApplyFrame a;
        a.strips = G.strips[k]; a.luts = G.luts[k] + (size_t)i * G.tiles[k] * 256; a.fr = f;
        a.nstrips = G.nstrips[k]; a.gx = G.g[k]; a.gy = G.g[k]; a.TX = G.TX[k]; a.xs = G.xs[k];
        a.inv_tw = G.inv_tw[k]; a.inv_th = G.inv_th[k];
        desc[f] = a;
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) count[q] = cnt[q];
}

int check_pair(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst)
{
    int rc = uwip_check_batch(ctx, src, 1);
    if (rc) return rc;
    rc = uwip_check_batch(ctx, dst, 1);
    if (rc) return rc;
    UWIP_REQUIRE(ctx, src->rows == dst->rows && src->cols == dst->cols && src->frames == dst->frames,
                 "src/dst shape mismatch");
    return UWIP_OK;
}

int check_grid(uwip_ctx *ctx, int gx, int gy)
{
    UWIP_REQUIRE(ctx, gx >= 1 && gy >= 1 && gx <= 62 && gy <= 128, "tile grid must be in [1,62] x [1,128]");
    return UWIP_OK;
}

}  // namespace

// ---- exported entry points --------------------------------------------------

UWIP_API int uwip_bgr_to_v(uwip_ctx *ctx, const uwip_batch_u8 *bgr, const uwip_batch_u8 *v)
{
    int rc = uwip_check_batch(ctx, bgr, 3);
    if (rc) return rc;
    rc = uwip_check_batch(ctx, v, 1);
    if (rc) return rc;
    UWIP_REQUIRE(ctx, bgr->rows == v->rows && bgr->cols == v->cols && bgr->frames == v->frames, "shape mismatch");
    if (uwip_batch_empty(bgr)) return UWIP_OK;
    const int groups = (bgr->cols + 15) / 16;
    dim3 grid(uwip_cdiv(groups, 256), (unsigned)bgr->rows, (unsigned)bgr->frames);
    UWIP_REQUIRE(ctx, bgr->rows <= 65535 && bgr->frames <= 65535, "too many rows/frames for one launch");
    const int vec = aligned_for(bgr, 16) && aligned_for(v, 16);
    uwip_kscope ks(ctx, "k_bgr_to_v");
    k_bgr_to_v<<<grid, 256, 0, ctx->stream>>>((const uint8_t *)bgr->data, bgr->step, bgr->frame_stride,
                                              (uint8_t *)v->data, v->step, v->frame_stride, bgr->rows, bgr->cols, vec);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

UWIP_API int uwip_GaussianBlur3(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst, int rounding_rule)
{
    int rc = check_pair(ctx, src, dst);
    if (rc) return rc;
    if (uwip_batch_empty(src)) return UWIP_OK;
    UWIP_REQUIRE(ctx, src->data != dst->data, "GaussianBlur3 cannot run in place");
    UWIP_REQUIRE(ctx, src->frames <= 65535 && src->rows <= 65535, "batch too large for one launch");
    uwip_kscope ks(ctx, "k_gauss3_u8");
    const bool vec = src->cols >= 8 && src->cols % 4 == 0 && aligned_for(src, 4) && aligned_for(dst, 4);
    const dim3 grid(std::min(uwip_cdiv(vec ? src->cols / 4 : src->cols, 256), 64u), vec ? uwip_cdiv(src->rows, GS3_ROWS) : (unsigned)src->rows, (unsigned)src->frames);
    if (vec)
        k_gauss3_u8<true><<<grid, 256, 0, ctx->stream>>>((const uint8_t *)src->data, src->step, src->frame_stride, (uint8_t *)dst->data, dst->step,
                                                        dst->frame_stride, src->rows, src->cols, rounding_rule);
    else
        k_gauss3_u8<false><<<grid, 256, 0, ctx->stream>>>((const uint8_t *)src->data, src->step, src->frame_stride, (uint8_t *)dst->data, dst->step,
                                                         dst->frame_stride, src->rows, src->cols, rounding_rule);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

UWIP_API int uwip_clahe_luts(uwip_ctx *ctx, const uwip_batch_u8 *src, double clipLimit, int gx, int gy,
                             int residual_rule, uint8_t *d_luts)
{
    int rc = uwip_check_batch(ctx, src, 1);
    if (rc) return rc;
    rc = check_grid(ctx, gx, gy);
    if (rc) return rc;
    if (uwip_batch_empty(src)) return UWIP_OK;
    UWIP_REQUIRE(ctx, d_luts != nullptr, "null LUT buffer");
    const ClaheGeom g = make_geom(src->rows, src->cols, gx, gy);
    const int tiles = gx * gy;
    ClipList cl{};
    cl.n = 1;
    cl.clip[0] = clip_from_limit(clipLimit, g.area);
    if (band_ok(src, g)) return launch_band(ctx, src, g, nullptr, src->frames, cl, nullptr, residual_rule, d_luts, nullptr);
    uint32_t *d_hists = (uint32_t *)uwip_ws(ctx, "clahe.tilehist", sizeof(uint32_t) * 256 * (size_t)tiles * src->frames);
    if (!d_hists) return UWIP_ERR_NOMEM;
    rc = launch_tilehist(ctx, src, g, nullptr, src->frames, d_hists);
    if (rc) return rc;
    return launch_lut(ctx, g, d_hists, cl, nullptr, src->frames, residual_rule, d_luts);
}

UWIP_API int uwip_clahe(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst, double clipLimit,
                        int gx, int gy, int residual_rule)
{
    int rc = check_pair(ctx, src, dst);
    if (rc) return rc;
    rc = check_grid(ctx, gx, gy);
    if (rc) return rc;
    if (uwip_batch_empty(src)) return UWIP_OK;
    const ClaheGeom g = make_geom(src->rows, src->cols, gx, gy);
    const int tiles = gx * gy;
    uint8_t *d_luts = (uint8_t *)uwip_ws(ctx, "clahe.luts", (size_t)256 * tiles * src->frames);
    if (!d_luts) return UWIP_ERR_NOMEM;
    rc = uwip_clahe_luts(ctx, src, clipLimit, gx, gy, residual_rule, d_luts);
    if (rc) return rc;
    return launch_apply(ctx, src, dst, g, d_luts, (size_t)tiles * 256, nullptr, src->frames);
}

UWIP_API int uwip_clahe_per_frame(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst,
                                  const double *h_clipLimit, const int32_t *h_grid, int residual_rule)
{
    int rc = check_pair(ctx, src, dst);
    if (rc) return rc;
    if (uwip_batch_empty(src)) return UWIP_OK;
    UWIP_REQUIRE(ctx, h_clipLimit && h_grid, "null parameter arrays");
    const int F = src->frames;
    for (int f = 0; f < F; ++f) {
        rc = check_grid(ctx, h_grid[f], h_grid[f]);
        if (rc) return rc;
    }
    // group frames by grid size; one (tilehist, lut, apply) triple per group
    std::vector<int> order(F);
    for (int f = 0; f < F; ++f) order[f] = f;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return h_grid[a] < h_grid[b]; });
    int *h_map = (int *)uwip_host_ws(ctx, "clahe.pf.map", sizeof(int) * 2 * (size_t)F);
    int *d_map = (int *)uwip_ws(ctx, "clahe.pf.map", sizeof(int) * 2 * (size_t)F);
    ApplyFrame *h_desc = (ApplyFrame *)uwip_host_ws(ctx, "clahe.pf.desc", sizeof(ApplyFrame) * (size_t)F);
    ApplyFrame *d_desc = (ApplyFrame *)uwip_ws(ctx, "clahe.pf.desc", sizeof(ApplyFrame) * (size_t)F);
    if (!h_map || !d_map || !h_desc || !d_desc) return UWIP_ERR_NOMEM;
    UWIP_HIP(ctx, uwip_stream_wait(ctx));
    int *h_clip = h_map + F, *d_clip = d_map + F;
    for (int i = 0; i < F; ++i) {
        const int f = order[i];
        h_map[i] = f;
        const ClaheGeom g = make_geom(src->rows, src->cols, h_grid[f], h_grid[f]);
        h_clip[i] = clip_from_limit(h_clipLimit[f], g.area);
    }
    UWIP_HIP(ctx, hipMemcpyAsync(d_map, h_map, sizeof(int) * 2 * (size_t)F, hipMemcpyHostToDevice, ctx->stream));
    size_t max_tiles = 0;
    for (int f = 0; f < F; ++f) max_tiles = std::max(max_tiles, (size_t)h_grid[f] * h_grid[f]);
    uint32_t *d_hists = (uint32_t *)uwip_ws(ctx, "clahe.tilehist", sizeof(uint32_t) * 256 * max_tiles * F);
    uint8_t *d_luts = (uint8_t *)uwip_ws(ctx, "clahe.luts", (size_t)256 * max_tiles * F);
    if (!d_hists || !d_luts) return UWIP_ERR_NOMEM;
    // per group of equal grid size: tile histograms and LUTs (small launches; every group keeps its own LUT range, the
    // interpolation reads them all); the interpolation itself
    // then runs ONCE over all frames with a per-frame descriptor (a launch per group would be too short to reach the
    // HBM rate: at 4K 16 frames split three ways ran at 29 % of peak, the single launch at 41 %)
    int i = 0, max_blocks = 0, max_cells = 0;
    size_t loff = 0;
    while (i < F) {
        int j = i;
        const int gsz = h_grid[order[i]];
        while (j < F && h_grid[order[j]] == gsz) ++j;
        const int nf = j - i;
        const ClaheGeom g = make_geom(src->rows, src->cols, gsz, gsz);
        const int tiles = gsz * gsz;
        ClipList cl{};
        cl.n = 1;
        if (band_ok(src, g)) {
            rc = launch_band(ctx, src, g, d_map + i, nf, cl, d_clip + i, residual_rule, d_luts + loff, nullptr);
            if (rc) return rc;
        } else {
            rc = launch_tilehist(ctx, src, g, d_map + i, nf, d_hists);
            if (rc) return rc;
            rc = launch_lut(ctx, g, d_hists, cl, d_clip + i, nf, residual_rule, d_luts + loff);
            if (rc) return rc;
        }
        // d_hists is reused by the next group: stream order keeps that safe
        const ApplyShape sh = apply_shape(g);
        const int4 *d_strips = nullptr;
        int nstrips = 0;
        rc = build_strips(ctx, g, sh.max_rows, &d_strips, &nstrips);
        if (rc) return rc;
        for (int k = 0; k < nf; ++k) {
            ApplyFrame &a = h_desc[i + k];
            a.strips = d_strips; a.luts = d_luts + loff + (size_t)k * tiles * 256; a.fr = order[i + k];
            a.nstrips = nstrips; a.gx = g.gx; a.gy = g.gy; a.TX = sh.TX; a.xs = sh.xs; a.inv_tw = g.inv_tw; a.inv_th = g.inv_th;
        }
        max_blocks = std::max(max_blocks, nstrips * sh.xs);
        max_cells = std::max(max_cells, sh.lds_cells);
        loff += (size_t)tiles * 256 * nf;
        i = j;
    }
    UWIP_HIP(ctx, hipMemcpyAsync(d_desc, h_desc, sizeof(ApplyFrame) * (size_t)F, hipMemcpyHostToDevice, ctx->stream));
    return launch_apply_mixed(ctx, src, dst, d_desc, F, max_blocks, max_cells);
}

// the exact block-size search for the frames uwip_aclahe_select_device flagged (one predicated launch; see k_aclahe_exact_bs)
static int aclahe_exact_bs_device(uwip_ctx *ctx, const uwip_batch_u8 *src, int32_t *d_par, int residual_rule)
{
    static const int BlockSize[5] = {2, 4, 8, 16, 32};
    const int F = src->frames;
    ExactGrids G{};
    for (int k = 0; k < 5; ++k) {
        const ClaheGeom g = make_geom(src->rows, src->cols, BlockSize[k], BlockSize[k]);
        G.g[k] = BlockSize[k]; G.tw[k] = g.tw; G.th[k] = g.th; G.pc[k] = g.pc; G.pr[k] = g.pr; G.area[k] = g.area;
        G.inv_tw[k] = g.inv_tw; G.inv_th[k] = g.inv_th; G.lutScale[k] = g.lutScale;
    }
    uint32_t *d_hists = (uint32_t *)uwip_ws(ctx, "clahe.tilehist", sizeof(uint32_t) * 256 * (size_t)EXACT_TILES * F);
    uint8_t *d_luts = (uint8_t *)uwip_ws(ctx, "sweep.luts", (size_t)256 * 1024 * SWEEP_NCL * F);      // >= 256 * EXACT_TILES * F
    float *d_ent = (float *)uwip_ws(ctx, "auto.exact_ent", sizeof(float) * 5 * (size_t)F);
    if (!d_hists || !d_luts || !d_ent) return UWIP_ERR_NOMEM;
    uwip_kscope ks(ctx, "k_aclahe_exact_bs");
    k_aclahe_exact_bs<<<dim3(F, 5), 512, 0, ctx->stream>>>((const uint8_t *)src->data, src->step, src->frame_stride, src->rows, src->cols, G,
                                                           residual_rule, d_par, d_hists, d_luts, d_ent);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

// createCLAHE(CL, (BS, BS)).apply per frame with {BS, CL} read from device memory (d_par [F][4]): no host wait, no copy
static int clahe_per_frame_device(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst, const int32_t *d_par, int residual_rule)
{
    static const int BlockSize[5] = {2, 4, 8, 16, 32};
    int rc = check_pair(ctx, src, dst);
    if (rc) return rc;
    if (uwip_batch_empty(src)) return UWIP_OK;
    const int F = src->frames;
    ClaheGeom g[5];
    ApplyShape sh[5];
    PfGrids G{};
    size_t lut_bytes = 0;
    for (int k = 0; k < 5; ++k) {
        rc = check_grid(ctx, BlockSize[k], BlockSize[k]);
        if (rc) return rc;
        g[k] = make_geom(src->rows, src->cols, BlockSize[k], BlockSize[k]);
        sh[k] = apply_shape(g[k]);
        lut_bytes += (size_t)BlockSize[k] * BlockSize[k] * 256 * F;
    }
    uint32_t *d_hists = (uint32_t *)uwip_ws(ctx, "clahe.tilehist", sizeof(uint32_t) * 256 * (size_t)1024 * F);
    uint8_t *d_luts = (uint8_t *)uwip_ws(ctx, "clahe.pfd.luts", lut_bytes);
    int *d_ints = (int *)uwip_ws(ctx, "clahe.pfd.ints", sizeof(int) * (10 * (size_t)F + 8));
    ApplyFrame *d_desc = (ApplyFrame *)uwip_ws(ctx, "clahe.pf.desc", sizeof(ApplyFrame) * (size_t)F);
    if (!d_hists || !d_luts || !d_ints || !d_desc) return UWIP_ERR_NOMEM;
    int *d_count = d_ints + 10 * (size_t)F;
    int max_blocks = 0, max_cells = 0;
    size_t loff = 0;
    for (int k = 0; k < 5; ++k) {
        const int4 *d_strips = nullptr;
        int nstrips = 0;
        rc = build_strips(ctx, g[k], sh[k].max_rows, &d_strips, &nstrips);
        if (rc) return rc;
        G.strips[k] = d_strips; G.nstrips[k] = nstrips; G.g[k] = BlockSize[k]; G.TX[k] = sh[k].TX; G.xs[k] = sh[k].xs;
        G.area[k] = g[k].area; G.tiles[k] = BlockSize[k] * BlockSize[k]; G.inv_tw[k] = g[k].inv_tw; G.inv_th[k] = g[k].inv_th;
        G.luts[k] = d_luts + loff;
        loff += (size_t)G.tiles[k] * 256 * F;
        G.map[k] = d_ints + (size_t)k * F;
        G.clip[k] = d_ints + (size_t)(5 + k) * F;
        max_blocks = std::max(max_blocks, nstrips * sh[k].xs);
        max_cells = std::max(max_cells, sh[k].lds_cells);
    }
    {
        uwip_kscope ks(ctx, "k_pf_prepare");
        k_pf_prepare<<<1, 64, 0, ctx->stream>>>(d_par, F, G, d_count, d_desc);
        UWIP_HIP(ctx, hipGetLastError());
    }
    ClipList cl{};
    cl.n = 1;
    for (int k = 0; k < 5; ++k) {
        if (band_ok(src, g[k])) {
            rc = launch_band(ctx, src, g[k], G.map[k], F, cl, G.clip[k], residual_rule, G.luts[k], nullptr, d_count + k);
            if (rc) return rc;
        } else {
            rc = launch_tilehist(ctx, src, g[k], G.map[k], F, d_hists, d_count + k);
            if (rc) return rc;
            rc = launch_lut(ctx, g[k], d_hists, cl, G.clip[k], F, residual_rule, G.luts[k], nullptr, d_count + k);
            if (rc) return rc;
        }
    }
    return launch_apply_mixed(ctx, src, dst, d_desc, F, max_blocks, max_cells);
}

UWIP_API int uwip_entropy(uwip_ctx *ctx, const uwip_batch_u8 *src, float *d_entropy)
{
    int rc = uwip_check_batch(ctx, src, 1);
    if (rc) return rc;
    if (src->frames == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, d_entropy != nullptr, "null output");
    UWIP_REQUIRE(ctx, !uwip_batch_empty(src), "entropy of an empty image");
    uint32_t *d_hist = (uint32_t *)uwip_ws(ctx, "entropy.hist", sizeof(uint32_t) * 256 * (size_t)src->frames);
    if (!d_hist) return UWIP_ERR_NOMEM;
    rc = uwip_launch_hist_internal(ctx, src, d_hist);
    if (rc) return rc;
    uwip_kscope ks(ctx, "k_entropy");
    k_entropy<<<src->frames, 256, 0, ctx->stream>>>(d_hist, src->rows, src->cols, d_entropy);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

UWIP_API int uwip_aclahe_sweep_hist(uwip_ctx *ctx, const uwip_batch_u8 *src, int residual_rule, float *d_entropy, uint32_t *d_hist_tap)
{
    static const int BlockSize[5] = {2, 4, 8, 16, 32};          // aclahe.cpp:161
    int rc = uwip_check_batch(ctx, src, 1);
    if (rc) return rc;
    if (src->frames == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, d_entropy != nullptr, "null output");
    UWIP_REQUIRE(ctx, !uwip_batch_empty(src), "sweep of an empty image");
    UWIP_REQUIRE(ctx, src->frames <= 65535, "too many frames for one launch");
    // k_clahe_sweep counts in 16-bit LDS counters: a block must see < 65536 pixels, and its smallest work item is one
    // row of an interpolation cell (at most a tile wide)
    UWIP_REQUIRE(ctx, src->cols <= 65535, "image too wide for the sweep");
    // ... and addresses a pixel of a frame by a 32-bit byte offset formed with a 24-bit multiply
    UWIP_REQUIRE(ctx, src->step < (1u << 24) && (uint64_t)src->rows * src->step < (1ull << 32), "frame too large for the sweep");
    const int F = src->frames;
    const size_t out_fs = (size_t)5 * SWEEP_NCL * 256;
    uint32_t *d_out = (uint32_t *)uwip_ws(ctx, "sweep.outhist", sizeof(uint32_t) * out_fs * F);
    uint32_t *hbuf[2] = {(uint32_t *)uwip_ws(ctx, "clahe.tilehist", sizeof(uint32_t) * 256 * (size_t)1024 * F),
                         (uint32_t *)uwip_ws(ctx, "clahe.tilehist2", sizeof(uint32_t) * 256 * (size_t)1024 * F)};
    uint8_t *d_luts = (uint8_t *)uwip_ws(ctx, "sweep.luts", (size_t)256 * 1024 * SWEEP_NCL * F);
    uint32_t *d_tmax = (uint32_t *)uwip_ws(ctx, "sweep.tilemax", sizeof(uint32_t) * (size_t)1024 * F);
    if (!d_out || !hbuf[0] || !hbuf[1] || !d_luts || !d_tmax) return UWIP_ERR_NOMEM;
    UWIP_HIP(ctx, hipMemsetAsync(d_out, 0, sizeof(uint32_t) * out_fs * F, ctx->stream));
    // finest grid first: a coarser unpadded grid sums the tile histograms of the grid twice as fine
    ClaheGeom finer{};
    bool finer_has_hists = false;
    for (int gi = 4; gi >= 0; --gi) {
        const int gsz = BlockSize[gi];
        const ClaheGeom g = make_geom(src->rows, src->cols, gsz, gsz);
        uint32_t *d_hists = hbuf[gi & 1];
        const bool nested = gi < 4 && finer.gx == 2 * g.gx && finer.gy == 2 * g.gy && finer.pc == finer.cols && finer.pr == finer.rows &&
                            g.pc == g.cols && g.pr == g.rows && finer.tw * 2 == g.tw && finer.th * 2 == g.th;
        ClipList cl{};
        cl.n = 0;
        for (float c = 0.0f; c <= 25.0f; c += 0.5f) cl.clip[cl.n++] = clip_from_limit((double)c, g.area);
        if (nested && finer_has_hists) {
            uwip_kscope ks(ctx, "k_clahe_tilehist");
            k_clahe_tilehist_merge<<<dim3((unsigned)(g.gx * g.gy), (unsigned)F), 256, 0, ctx->stream>>>(hbuf[(gi + 1) & 1], g.gx, g.gx * g.gy, d_hists);
            UWIP_HIP(ctx, hipGetLastError());
            finer_has_hists = true;
        } else if (band_ok(src, g)) {
            // histograms and the 51 LUT rows of every tile in one launch; the histograms stay in LDS
            rc = launch_band(ctx, src, g, nullptr, F, cl, nullptr, residual_rule, d_luts, d_tmax);
            if (rc) return rc;
            finer_has_hists = false;
        } else {
            rc = launch_tilehist(ctx, src, g, nullptr, F, d_hists);
            if (rc) return rc;
            finer_has_hists = true;
        }
        finer = g;
        if (finer_has_hists) {
            rc = launch_lut(ctx, g, d_hists, cl, nullptr, F, residual_rule, d_luts, d_tmax);
            if (rc) return rc;
        }
        // work items: interpolation cells cut into row chunks of <= ~16K pixels (cached per geometry)
        char key[96];
        snprintf(key, sizeof key, "cells:%d:%d:%d", g.rows, g.cols, gsz);
        size_t bytes = 0;
        const void *d_tab = uwip_table_find(ctx, key, &bytes);
        if (!d_tab) {
            std::vector<int> xs, ys;
            cell_starts(g.cols, g.gx, g.inv_tw, xs);
            cell_starts(g.rows, g.gy, g.inv_th, ys);
            std::vector<CellItem> items;
            for (int cy = 0; cy <= g.gy; ++cy) {
                for (int cx = 0; cx <= g.gx; ++cx) {
                    const int w = xs[cx + 1] - xs[cx], h = ys[cy + 1] - ys[cy];
                    if (w <= 0 || h <= 0) continue;
                    const int rows_per = std::max(1, 16384 / w);
                    for (int r = ys[cy]; r < ys[cy + 1]; r += rows_per) {
                        CellItem ci{};
                        ci.cx = cx; ci.cy = cy; ci.x0 = xs[cx]; ci.x1 = xs[cx + 1];
                        ci.r0 = r; ci.r1 = std::min(r + rows_per, ys[cy + 1]);
                        items.push_back(ci);
                    }
                }
            }
            bytes = items.size() * sizeof(CellItem);
            d_tab = uwip_table_put(ctx, key, items.data(), bytes);
            if (!d_tab) return UWIP_ERR_NOMEM;
        }
        const CellItem *d_items = (const CellItem *)d_tab;
        const int nitems = (int)(bytes / sizeof(CellItem));
        const int cell_px = std::max(1, g.tw * g.th);
        const int ipb = std::max(1, std::min(32, 32768 / cell_px));   // <= 32768 pixels per block: the 16-bit LDS counters cannot overflow
        dim3 grid(uwip_cdiv(nitems, ipb), SWEEP_NCL / SWEEP_GROUP, (unsigned)F);
        uwip_kscope ks(ctx, "k_clahe_sweep");
        const size_t sweep_lds = sizeof(uint32_t) * SWEEP_LDS_WORDS;
        // how a cell's npix mod 512 pixels are walked (UWIP_SWEEP_REM; same box, 64-frame sweeps, tools/sweep_ab.py): 0 = round 2's
        // spread round, every wave with a few lanes: 7.93 ms; 1 = one contiguous round, one wave's worth: 7.67 ms (default);
        // 2 = small remainders as (pixel, evaluation) threads: 7.78 ms
        static const int rem_mode = [] { const char *e = getenv("UWIP_SWEEP_REM"); return e && *e >= '0' && *e <= '2' ? *e - '0' : 1; }();
        rc = uwip_lds_optin(ctx, "k_clahe_sweep", (const void *)k_clahe_sweep, sweep_lds);
        if (rc) return rc;
        k_clahe_sweep<<<grid, SWEEP_THREADS, sweep_lds, ctx->stream>>>((const uint8_t *)src->data, src->step, src->frame_stride, g.gx,
                                                     g.gy, g.inv_tw, g.inv_th, d_luts, d_items, nitems, ipb,
                                                     d_out + (size_t)gi * SWEEP_NCL * 256, out_fs, d_tmax, cl, rem_mode);
        UWIP_HIP(ctx, hipGetLastError());
    }
    {
        uwip_kscope ks(ctx, "k_entropy");
        k_entropy<<<F * 5 * SWEEP_NCL, 256, 0, ctx->stream>>>(d_out, src->rows, src->cols, d_entropy);
        UWIP_HIP(ctx, hipGetLastError());
    }
    if (d_hist_tap) UWIP_HIP(ctx, hipMemcpyAsync(d_hist_tap, d_out, sizeof(uint32_t) * out_fs * F, hipMemcpyDeviceToDevice, ctx->stream));
    return UWIP_OK;
}

UWIP_API int uwip_aclahe_sweep(uwip_ctx *ctx, const uwip_batch_u8 *src, int residual_rule, float *d_entropy)
{
    return uwip_aclahe_sweep_hist(ctx, src, residual_rule, d_entropy, nullptr);
}

int uwip_aclahe_select_internal(const float *h_entropy, int frames, int32_t *h_bs, int32_t *h_cl, int32_t *h_knee,
                                int32_t *h_need_eval, const float *h_extra, const int32_t *h_extra_valid);
extern "C" int uwip_aclahe_select_device(uwip_ctx *ctx, const float *d_entropy, int frames, int32_t *d_par, int32_t *d_knee);

// C3 + C4 + the final apply in one call: sweep -> (host) parameter choice -> per-frame CLAHE.
// This is the whole "aclahe" stage of the pipe.  h_bs / h_cl receive the chosen parameters.
UWIP_API int uwip_aclahe_auto_ex(uwip_ctx *ctx, const uwip_batch_u8 *img, const uwip_batch_u8 *dst, int residual_rule, unsigned flags,
                                 int32_t *h_bs, int32_t *h_cl)
{
    int rc = check_pair(ctx, img, dst);
    if (rc) return rc;
    if (img->frames == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, !uwip_batch_empty(img), "aclahe of an empty image");
    UWIP_REQUIRE(ctx, (flags & ~(unsigned)(UWIP_ACLAHE_PREFILTER | UWIP_ACLAHE_HOST_SELECT | UWIP_ACLAHE_ASYNC)) == 0, "unknown flag");
    UWIP_REQUIRE(ctx, !(flags & UWIP_ACLAHE_ASYNC) || (!h_bs && !h_cl), "UWIP_ACLAHE_ASYNC: the parameters are fetched with uwip_aclahe_last_params");
    const int F = img->frames;
    // a call that fails leaves "no parameters recorded" (not the previous call's): recorded only after the last launch succeeded
    ctx->aclahe_last_n = 0;
    ctx->aclahe_last_on_device = false;
    ctx->aclahe_last_host.clear();
    // Where the choice is made.  Default: on the device (no table copy, no host computation: 0.25 CPU-seconds per 512-frame
    // step and rank otherwise) -- except for batches of a few frames, where latency is what matters (the paced 4K@60 stream
    // runs one frame per call): a curve takes one wavefront 1.3 ms on the device (a serial dependent chain of float64
    // divisions and square roots at one instruction per ~9 cycles) and a host core 0.15 ms.  UWIP_ACLAHE_SELECT=host|device
    // or the flag force one form; both give the same parameters bit for bit.
    static const int env_sel = [] { const char *e = getenv("UWIP_ACLAHE_SELECT"); return !e ? 0 : ((*e == 'h' || *e == 'H') ? 1 : ((*e == 'd' || *e == 'D') ? 2 : 0)); }();
    const bool host_select = (flags & UWIP_ACLAHE_HOST_SELECT) || env_sel == 1 || (env_sel == 0 && F <= 4);
    // ParametrosACLAHE searches its parameters on imgfilt = GaussianBlur(img, (3,3), 0) (ACLAHE.py:15: the sweep :40-47 and
    // the block-size search :102-112 both run on it); the final createCLAHE(CL,(BS,BS)).apply takes the unfiltered image
    // (python/main.py:19-20).  The C++ driver (aclahe.cpp:152-187) sweeps the unfiltered plane: flags = 0.
    uwip_batch_u8 filt = *img;
    if (flags & UWIP_ACLAHE_PREFILTER) {
        uint8_t *fb = (uint8_t *)uwip_ws(ctx, "auto.blur", (size_t)img->rows * img->cols * F);
        if (!fb) return UWIP_ERR_NOMEM;
        filt.data = fb; filt.step = (size_t)img->cols; filt.frame_stride = (size_t)img->rows * img->cols;
        rc = uwip_GaussianBlur3(ctx, img, &filt, residual_rule);
        if (rc) return rc;
    }
    const uwip_batch_u8 *src = &filt;
    float *d_ent = (float *)uwip_ws(ctx, "auto.entropy", sizeof(float) * 255 * F);
    float *h_ent = (float *)uwip_host_ws(ctx, "auto.entropy", sizeof(float) * 255 * F);
    int32_t *h_par = (int32_t *)uwip_host_ws(ctx, "auto.params", sizeof(int32_t) * 4 * F + sizeof(float) * 5 * F + sizeof(double) * F);
    if (!d_ent || !h_ent || !h_par) return UWIP_ERR_NOMEM;
    rc = uwip_aclahe_sweep(ctx, src, residual_rule, d_ent);
    if (rc) return rc;
    int32_t *bs = h_par, *cl = h_par + F, *need = h_par + 2 * F, *valid = h_par + 3 * F;
    float *extra = (float *)(h_par + 4 * F);
    double *clip = (double *)(extra + 5 * F);
    bool any = false;
    if (!host_select && (flags & UWIP_ACLAHE_ASYNC)) {
        // nothing comes back to the host: the choice (aclahe_device.hip) and the launch of the final CLAHE from the
        // device-side parameters (clahe_per_frame_device) are queued behind the sweep, and the call returns
        int32_t *d_par = (int32_t *)uwip_ws(ctx, "auto.par", sizeof(int32_t) * 4 * (size_t)F);
        if (!d_par) return UWIP_ERR_NOMEM;
        rc = uwip_aclahe_select_device(ctx, d_ent, F, d_par, nullptr);
        if (rc) return rc;
        rc = aclahe_exact_bs_device(ctx, src, d_par, residual_rule);
        if (rc) return rc;
        rc = clahe_per_frame_device(ctx, img, dst, d_par, residual_rule);
        if (rc) return rc;
        ctx->aclahe_last_n = F;
        ctx->aclahe_last_on_device = true;
        return UWIP_OK;
    }
    if (!host_select) {
        // the choice on the device (aclahe_device.hip): only {BS, CL, need} per frame come back -- the launch geometry of
        // the final CLAHE depends on them
        int32_t *d_par = (int32_t *)uwip_ws(ctx, "auto.par", sizeof(int32_t) * 4 * (size_t)F);
        int32_t *h_dpar = (int32_t *)uwip_host_ws(ctx, "auto.dpar", sizeof(int32_t) * 4 * (size_t)F);
        if (!d_par || !h_dpar) return UWIP_ERR_NOMEM;
        rc = uwip_aclahe_select_device(ctx, d_ent, F, d_par, nullptr);
        if (rc) return rc;
        UWIP_HIP(ctx, hipMemcpyAsync(h_dpar, d_par, sizeof(int32_t) * 4 * (size_t)F, hipMemcpyDeviceToHost, ctx->stream));
        UWIP_HIP(ctx, uwip_stream_wait(ctx));
        for (int f = 0; f < F; ++f) {
            bs[f] = h_dpar[4 * f]; cl[f] = h_dpar[4 * f + 1]; need[f] = h_dpar[4 * f + 2]; valid[f] = 0;
            any = any || need[f];
        }
        if (any) {               // the rare frames whose clip limit leaves the swept grid go through the host form below
            UWIP_HIP(ctx, hipMemcpyAsync(h_ent, d_ent, sizeof(float) * 255 * F, hipMemcpyDeviceToHost, ctx->stream));
            UWIP_HIP(ctx, uwip_stream_wait(ctx));
        }
    } else {
        UWIP_HIP(ctx, hipMemcpyAsync(h_ent, d_ent, sizeof(float) * 255 * F, hipMemcpyDeviceToHost, ctx->stream));
        UWIP_HIP(ctx, uwip_stream_wait(ctx));            // host decision point (ACLAHE.py:66-129)
        rc = uwip_aclahe_select_internal(h_ent, F, bs, cl, nullptr, need, nullptr, nullptr);
        if (rc) return ctx->fail(rc, "aclahe select");
        for (int f = 0; f < F; ++f) { valid[f] = 0; any = any || need[f]; }
    }
    if (any) {
        // clip limit outside the swept grid: evaluate the five block sizes at that clip limit (ACLAHE.py:102-112)
        static const int BlockSize[5] = {2, 4, 8, 16, 32};
        uint8_t *tmp = (uint8_t *)uwip_ws(ctx, "auto.tmp", (size_t)src->rows * src->cols);
        float *d_e1 = (float *)uwip_ws(ctx, "auto.e1", sizeof(float) * 8);
        if (!tmp || !d_e1) return UWIP_ERR_NOMEM;
        for (int f = 0; f < F; ++f) {
            if (!need[f]) continue;
            uwip_batch_u8 one = *src;
            one.data = (uint8_t *)src->data + (size_t)f * src->frame_stride;
            one.frames = 1;
            uwip_batch_u8 t1 = one;
            t1.data = tmp; t1.step = (size_t)src->cols; t1.frame_stride = (size_t)src->rows * src->cols;
            for (int g = 0; g < 5; ++g) {
                rc = uwip_clahe(ctx, &one, &t1, (double)cl[f], BlockSize[g], BlockSize[g], residual_rule);
                if (rc) return rc;
                rc = uwip_entropy(ctx, &t1, d_e1 + g);
                if (rc) return rc;
            }
            UWIP_HIP(ctx, hipMemcpyAsync(extra + (size_t)f * 5, d_e1, sizeof(float) * 5, hipMemcpyDeviceToHost, ctx->stream));
            UWIP_HIP(ctx, uwip_stream_wait(ctx));
            valid[f] = 1;
        }
        rc = uwip_aclahe_select_internal(h_ent, F, bs, cl, nullptr, need, extra, valid);
        if (rc) return ctx->fail(rc, "aclahe select");
    }
    for (int f = 0; f < F; ++f) clip[f] = (double)cl[f];
    if (h_bs) for (int f = 0; f < F; ++f) h_bs[f] = bs[f];
    if (h_cl) for (int f = 0; f < F; ++f) h_cl[f] = cl[f];
    rc = uwip_clahe_per_frame(ctx, img, dst, clip, bs, residual_rule);
    if (rc) return rc;
    ctx->aclahe_last_host.resize(2 * (size_t)F);
    for (int f = 0; f < F; ++f) { ctx->aclahe_last_host[2 * f] = bs[f]; ctx->aclahe_last_host[2 * f + 1] = cl[f]; }
    ctx->aclahe_last_n = F;
    return UWIP_OK;
}

// the parameters of the most recent uwip_aclahe_auto_ex on this context (waits for the stream when they are still on the device)
UWIP_API int uwip_aclahe_last_params(uwip_ctx *ctx, int32_t *h_bs, int32_t *h_cl, int frames)
{
    if (int rc_e = uwip_enter(ctx)) return rc_e;
    if (frames == 0 && ctx->aclahe_last_n == 0) return UWIP_OK;
    UWIP_REQUIRE(ctx, ctx->aclahe_last_n > 0, "no parameters recorded (no uwip_aclahe_auto_ex yet, or the last one failed)");
    UWIP_REQUIRE(ctx, frames == ctx->aclahe_last_n, "frame count differs from the last uwip_aclahe_auto_ex");
    UWIP_REQUIRE(ctx, h_bs && h_cl, "null output");
    if (ctx->aclahe_last_on_device) {
        const int32_t *d_par = (const int32_t *)uwip_ws(ctx, "auto.par", sizeof(int32_t) * 4 * (size_t)frames);
        int32_t *h = (int32_t *)uwip_host_ws(ctx, "auto.dpar", sizeof(int32_t) * 4 * (size_t)frames);
        if (!d_par || !h) return UWIP_ERR_NOMEM;
        UWIP_HIP(ctx, hipMemcpyAsync(h, d_par, sizeof(int32_t) * 4 * (size_t)frames, hipMemcpyDeviceToHost, ctx->stream));
        UWIP_HIP(ctx, uwip_stream_wait(ctx));
        for (int f = 0; f < frames; ++f) { h_bs[f] = h[4 * f]; h_cl[f] = h[4 * f + 1]; }
        return UWIP_OK;
    }
    UWIP_REQUIRE(ctx, ctx->aclahe_last_host.size() == 2 * (size_t)frames, "no parameters recorded");
    for (int f = 0; f < frames; ++f) { h_bs[f] = ctx->aclahe_last_host[2 * f]; h_cl[f] = ctx->aclahe_last_host[2 * f + 1]; }
    return UWIP_OK;
}

UWIP_API int uwip_aclahe_auto(uwip_ctx *ctx, const uwip_batch_u8 *src, const uwip_batch_u8 *dst, int residual_rule,
                              int32_t *h_bs, int32_t *h_cl)
{
    return uwip_aclahe_auto_ex(ctx, src, dst, residual_rule, 0u, h_bs, h_cl);
}

// ---- "transform back image" (aclahe.cpp:216): BGR -> HSV, V := CLAHE(V), HSV -> BGR -------------------
// cvtColor(BGR2HSV) / cvtColor(HSV2BGR) for 8-bit images restated from OpenCV 3.4 color.cpp
// (integer forward tables with hsv_shift = 12; float inverse with hscale = 6/180).  parity unpinned.
namespace {
// one pixel: BGR -> (H, S) by the integer forward tables, then HSV -> BGR with the new V
__device__ __forceinline__ void hsv_replace_px(int b, int g, int r, int vnew, const int *__restrict__ sdiv,
                                               const int *__restrict__ hdiv, uint32_t &ob8, uint32_t &og8, uint32_t &or8)
{
    const int v = max(b, max(g, r)), vmin = min(b, min(g, r));
    if (vnew < 0) vnew = v;      // plain BGR -> HSV -> BGR round trip
    const int diff = v - vmin;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    const int sat = (diff * sdiv[v] + (1 << 11)) >> 12;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * hdiv[diff] + (1 << 11)) >> 12;
    h += h < 0 ? 180 : 0;
    // inverse with the new V
    float hf = (float)(uint8_t)h;
    const float sf = (float)(uint8_t)sat * (1.f / 255.f), vf = (float)vnew * (1.f / 255.f);
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
    // the three values lie in [0, 1]: v_cvt_pk_u8_f32 (round to nearest even + clamp) is saturate_cast<uchar> there
    ob8 = __builtin_amdgcn_cvt_pk_u8_f32(ob * 255.f, 0, 0u);
    og8 = __builtin_amdgcn_cvt_pk_u8_f32(og * 255.f, 0, 0u);
    or8 = __builtin_amdgcn_cvt_pk_u8_f32(orr * 255.f, 0, 0u);
}

constexpr int HSV_ROWS_PER_BLOCK = 8;
// VEC: rows are 4-byte aligned and cols % 4 == 0 -> a thread takes 4 pixels as 3 + 1 dword loads and 3 dword stores
template <bool VEC>
__global__ __launch_bounds__(256) void k_hsv_replace_v(const uint8_t *__restrict__ src, size_t sstep, size_t sfs,
                                                      const uint8_t *__restrict__ vnew, size_t vstep, size_t vfs,
                                                      uint8_t *__restrict__ dst, size_t dstep, size_t dfs, int rows,
                                                      int cols, const int *__restrict__ sdiv_g, const int *__restrict__ hdiv_g)
{
    // the two division tables (2 KB) live in LDS: as global gathers they, not the pixels, set the kernel's pace
    __shared__ int s_tab[512];
    for (int i = threadIdx.x; i < 256; i += 256) { s_tab[i] = sdiv_g[i]; s_tab[256 + i] = hdiv_g[i]; }
    __syncthreads();
    const int *sdiv = s_tab, *hdiv = s_tab + 256;
    const int f = blockIdx.z;
    for (int y = blockIdx.y * HSV_ROWS_PER_BLOCK; y < min(rows, (int)(blockIdx.y + 1) * HSV_ROWS_PER_BLOCK); ++y) {
    const uint8_t *s = src + (size_t)f * sfs + (size_t)y * sstep;
    const uint8_t *vn = vnew ? vnew + (size_t)f * vfs + (size_t)y * vstep : nullptr;   // null: keep the pixel's own V
    uint8_t *d = dst + (size_t)f * dfs + (size_t)y * dstep;
    if (VEC) {
        for (int x4 = blockIdx.x * 256 + threadIdx.x; x4 < cols / 4; x4 += gridDim.x * 256) {
            const uint32_t *sp = reinterpret_cast<const uint32_t *>(s) + 3 * x4;
            const uint32_t w[3] = {sp[0], sp[1], sp[2]};
            const uint32_t vv = vn ? reinterpret_cast<const uint32_t *>(vn)[x4] : 0u;
            uint32_t o[3] = {0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 3 * j;
                const int b = (w[k >> 2] >> ((k & 3) * 8)) & 255, g = (w[(k + 1) >> 2] >> (((k + 1) & 3) * 8)) & 255,
                          r = (w[(k + 2) >> 2] >> (((k + 2) & 3) * 8)) & 255;
                uint32_t ob, og, orr;
                hsv_replace_px(b, g, r, vn ? (int)((vv >> (8 * j)) & 255u) : -1, sdiv, hdiv, ob, og, orr);
                o[k >> 2] |= ob << ((k & 3) * 8);
                o[(k + 1) >> 2] |= og << (((k + 1) & 3) * 8);
                o[(k + 2) >> 2] |= orr << (((k + 2) & 3) * 8);
            }
            uint32_t *dp = reinterpret_cast<uint32_t *>(d) + 3 * x4;
            dp[0] = o[0]; dp[1] = o[1]; dp[2] = o[2];
        }
    } else {
        for (int x = blockIdx.x * 256 + threadIdx.x; x < cols; x += gridDim.x * 256) {
            uint32_t ob, og, orr;
            hsv_replace_px(s[3 * x], s[3 * x + 1], s[3 * x + 2], vn ? (int)vn[x] : -1, sdiv, hdiv, ob, og, orr);
            d[3 * x] = (uint8_t)ob; d[3 * x + 1] = (uint8_t)og; d[3 * x + 2] = (uint8_t)orr;
        }
    }
    }
}

const int *hsv_tables(uwip_ctx *ctx)
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

UWIP_API int uwip_hsv_replace_v(uwip_ctx *ctx, const uwip_batch_u8 *bgr, const uwip_batch_u8 *v_new, const uwip_batch_u8 *bgr_out)
{
    int rc = uwip_check_batch(ctx, bgr, 3);
    if (rc) return rc;
    rc = uwip_check_batch(ctx, v_new, 1);
    if (rc) return rc;
    rc = uwip_check_batch(ctx, bgr_out, 3);
    if (rc) return rc;
    UWIP_REQUIRE(ctx, bgr->rows == v_new->rows && bgr->cols == v_new->cols && bgr->frames == v_new->frames &&
                          bgr->rows == bgr_out->rows && bgr->cols == bgr_out->cols && bgr->frames == bgr_out->frames, "shape mismatch");
    if (uwip_batch_empty(bgr)) return UWIP_OK;
    UWIP_REQUIRE(ctx, bgr->rows <= 65535 && bgr->frames <= 65535, "too many rows/frames for one launch");
    const int *tabs = hsv_tables(ctx);
    if (!tabs) return UWIP_ERR_NOMEM;
    uwip_kscope ks(ctx, "k_hsv_replace_v");
    auto al4 = [](const uwip_batch_u8 *b) { return ((uintptr_t)b->data | b->step | b->frame_stride) % 4 == 0; };
    const bool vec = bgr->cols % 4 == 0 && al4(bgr) && al4(v_new) && al4(bgr_out);
    const dim3 grid(uwip_cdiv(vec ? bgr->cols / 4 : bgr->cols, 256), (unsigned)uwip_cdiv(bgr->rows, HSV_ROWS_PER_BLOCK), (unsigned)bgr->frames);
    if (vec)
        k_hsv_replace_v<true><<<grid, 256, 0, ctx->stream>>>(
            (const uint8_t *)bgr->data, bgr->step, bgr->frame_stride, (const uint8_t *)v_new->data, v_new->step, v_new->frame_stride,
            (uint8_t *)bgr_out->data, bgr_out->step, bgr_out->frame_stride, bgr->rows, bgr->cols, tabs, tabs + 256);
    else
        k_hsv_replace_v<false><<<grid, 256, 0, ctx->stream>>>(
            (const uint8_t *)bgr->data, bgr->step, bgr->frame_stride, (const uint8_t *)v_new->data, v_new->step, v_new->frame_stride,
            (uint8_t *)bgr_out->data, bgr_out->step, bgr_out->frame_stride, bgr->rows, bgr->cols, tabs, tabs + 256);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}

// cvtColor(BGR2HSV) followed by cvtColor(HSV2BGR), 8-bit, in place: what an HSV letter of histretch leaves in the image
// (histretch.cpp:232-238).
int uwip_hsv_roundtrip(uwip_ctx *ctx, const uwip_batch_u8 *img)
{
    if (uwip_batch_empty(img)) return UWIP_OK;
    UWIP_REQUIRE(ctx, img->rows <= 65535 && img->frames <= 65535, "too many rows/frames for one launch");
    const int *tabs = hsv_tables(ctx);
    if (!tabs) return UWIP_ERR_NOMEM;
    uwip_kscope ks(ctx, "k_hsv_replace_v");
    const bool vec = img->cols % 4 == 0 && ((uintptr_t)img->data | img->step | img->frame_stride) % 4 == 0;
    const dim3 grid(uwip_cdiv(vec ? img->cols / 4 : img->cols, 256), (unsigned)uwip_cdiv(img->rows, HSV_ROWS_PER_BLOCK), (unsigned)img->frames);
    uint8_t *d = (uint8_t *)img->data;
    if (vec)
        k_hsv_replace_v<true><<<grid, 256, 0, ctx->stream>>>(d, img->step, img->frame_stride, nullptr, 0, 0, d, img->step, img->frame_stride,
                                                             img->rows, img->cols, tabs, tabs + 256);
    else
        k_hsv_replace_v<false><<<grid, 256, 0, ctx->stream>>>(d, img->step, img->frame_stride, nullptr, 0, 0, d, img->step, img->frame_stride,
                                                              img->rows, img->cols, tabs, tabs + 256);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}
