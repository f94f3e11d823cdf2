// What would a VALUE-SORTED sweep cost?  (VERDICT r3 #3 asks for the form or for a measurement that shows why not.)
//
// The sweep (csrc/clahe.hip, k_clahe_sweep) evaluates, per pixel and clip limit, the bilinear blend of four LUT bytes and
// counts the result: ~18 vector instructions + one LDS read + 0.8 LDS atomics per evaluation, 121 evaluations per pixel
// over the five grids.  A form organised by (interpolation cell, grey level) would make the 4-tuple wave-uniform (no
// per-lane byte conversions, no LUT gather) and would evaluate a tuple once for the whole run of limits it holds for (76
// distinct tuples per pixel instead of 121 limits) -- but it needs the pixels of a cell counting-sorted by grey level, and
// it has to COUNT differently: all 64 lanes of an instruction hold the same grey level, so their outputs fall into the
// span of four LUT bytes, and a tuple that holds for the limits [c_j, c_j+1) is counted at both ends (+1 at c_j, -1 at
// c_j+1; prefix over c at flush time).
//
// This program measures the three unknowns on synthetic 1080p planes and the 2 x 2 grid (3 x 3 cells: the grid where the
// (cell, grey level) groups are largest):
//   sort      counting sort of every cell's pixel positions by grey level (histogram, scan, scatter)
//   eval A    sorted evaluation, outputs counted by LDS atomics into the block's histogram rows (two per tuple)
//   eval B    sorted evaluation, outputs counted into per-lane private LDS bins ((o - lo) * 64 + lane: conflict-free),
//             reduced per (group, tuple) with a wave reduction and two atomics per bin
//   eval 0    the same blend + one conflict-free counter per evaluation (the floor of the arithmetic alone)
// and prints ns per 64-pixel tuple evaluation per SIMD beside the sweep's own figure (26 - 36 ns per evaluated limit,
// DESIGN.md section 5) and the milliseconds per 64 frames each piece would add up to.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o sweep_sorted.bin sweep_sorted.hip && ./sweep_sorted.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

constexpr int W = 1920, H = 1080, NCELL = 9, NCL = 51;
__host__ __device__ inline void cell_rect(int cell, int &x0, int &x1, int &y0, int &y1)
{
    const int cx = cell % 3, cy = cell / 3;
    const int xs[4] = {0, 480, 1440, 1920}, ys[4] = {0, 270, 810, 1080};
    x0 = xs[cx]; x1 = xs[cx + 1]; y0 = ys[cy]; y1 = ys[cy + 1];
}

// ---- counting sort of a cell's positions by grey level ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_count(const uint8_t *v, unsigned *hist /*[F][9][256]*/)
{
    __shared__ unsigned s[256];
    const int f = blockIdx.z, cell = blockIdx.y;
    int x0, x1, y0, y1;
    cell_rect(cell, x0, x1, y0, y1);
    s[threadIdx.x] = 0;
    __syncthreads();
    const int w = x1 - x0, n = w * (y1 - y0);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int y = y0 + i / w, x = x0 + i % w;
        atomicAdd(&s[v[(size_t)f * W * H + (size_t)y * W + x]], 1u);
    }
    __syncthreads();
    if (s[threadIdx.x]) atomicAdd(&hist[((size_t)f * NCELL + cell) * 256 + threadIdx.x], s[threadIdx.x]);
}
__global__ void k_scan(const unsigned *hist, unsigned *start /*[F][9][257]*/, unsigned *cursor, int F)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;          // one thread per (f, cell): tiny
    if (g >= F * NCELL) return;
    int x0, x1, y0, y1;
    unsigned base = 0;
    for (int c = 0; c < g % NCELL; ++c) { cell_rect(c, x0, x1, y0, y1); base += (unsigned)((x1 - x0) * (y1 - y0)); }
    unsigned run = base;
    for (int b = 0; b < 256; ++b) { start[(size_t)g * 257 + b] = run; cursor[(size_t)g * 256 + b] = run; run += hist[(size_t)g * 256 + b]; }
    start[(size_t)g * 257 + 256] = run;
}
// a block takes a chunk of a cell: local histogram, one range reservation per grey level, scatter
__global__ __launch_bounds__(256) void k_scatter(const uint8_t *v, unsigned *cursor, unsigned *pos /*[F][W*H]*/, int chunk)
{
    __shared__ unsigned s_cnt[256], s_base[256];
    const int f = blockIdx.z, cell = blockIdx.y;
    int x0, x1, y0, y1;
    cell_rect(cell, x0, x1, y0, y1);
    const int w = x1 - x0, n = w * (y1 - y0);
    const int i0 = blockIdx.x * chunk, i1 = min(n, i0 + chunk);
    if (i0 >= n) return;
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const int y = y0 + i / w, x = x0 + i % w;
        atomicAdd(&s_cnt[v[(size_t)f * W * H + (size_t)y * W + x]], 1u);
    }
    __syncthreads();
    const unsigned c = s_cnt[threadIdx.x];
    s_base[threadIdx.x] = c ? atomicAdd(&cursor[((size_t)f * NCELL + cell) * 256 + threadIdx.x], c) : 0u;
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const int y = y0 + i / w, x = x0 + i % w;
        const unsigned b = v[(size_t)f * W * H + (size_t)y * W + x];
        const unsigned k = atomicAdd(&s_cnt[b], 1u);
        pos[(size_t)f * W * H + s_base[b] + k] = ((unsigned)y << 16) | (unsigned)x;
    }
}

// ---- sorted evaluation ----------------------------------------------------------------------------------------------------------
// tuples[cell][v][c] = four LUT bytes of limit c (synthetic: a monotone map of v whose slope shrinks with c, perturbed per
// tile, changing between consecutive limits about as often as the bench's LUTs do: 76 distinct of 51 x 5 over the grids)
__device__ __forceinline__ unsigned eval_px(unsigned pk, float xa1, float xa, float ya1, float ya)
{
    const float top = (float)(pk & 255u) * xa1 + (float)((pk >> 8) & 255u) * xa;
    const float bot = (float)((pk >> 16) & 255u) * xa1 + (float)(pk >> 24) * xa;
    return __builtin_amdgcn_cvt_pk_u8_f32(top * ya1 + bot * ya, 0, 0u);
}
// MODE 0: one conflict-free counter per evaluation; 1: LDS atomics into shared rows (+1 / -1); 2: per-lane private bins
template <int MODE>
__global__ __launch_bounds__(256) void k_eval_sorted(const unsigned *start, const unsigned *pos, const unsigned *tuples /*[9][256][51]*/,
                                                    unsigned *out /*[51][256]*/, unsigned long long *nevals)
{
    __shared__ unsigned s_rows[NCL * 256];                 // MODE 1: the block's 51 histogram rows (32-bit here; deltas)
    __shared__ unsigned s_priv[4][33 * 64];                // MODE 2: per wave, (o - lo) * 64 + lane, span <= 32
    const int f = blockIdx.z, cell = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < NCL * 256; i += 256) s_rows[i] = 0;
    for (int i = lane; i < 33 * 64; i += 64) s_priv[wave][i] = 0;
    __syncthreads();
    int x0, x1, y0, y1;
    cell_rect(cell, x0, x1, y0, y1);
    const float inv_tw = 1.0f / 960.0f, inv_th = 1.0f / 540.0f;
    unsigned long long my_evals = 0;
    // the block's four waves take grey levels round robin
    for (int v = blockIdx.x * 4 + wave; v < 256; v += gridDim.x * 4) {
        const unsigned s0 = start[((size_t)f * NCELL + cell) * 257 + v], s1 = start[((size_t)f * NCELL + cell) * 257 + v + 1];
        if (s1 == s0) continue;
        const unsigned *T = tuples + ((size_t)cell * 256 + v) * NCL;
        // distinct consecutive tuples: lane c holds tuple c; a lane is a "head" when its tuple differs from the previous limit's
        const unsigned tc = lane < NCL ? T[lane] : 0u;
        const unsigned prev = __shfl_up(tc, 1, 64);
        const unsigned long long heads = __ballot(lane < NCL && (lane == 0 || tc != prev));
        for (unsigned p0 = s0; p0 < s1; p0 += 64) {
            const bool act = p0 + lane < s1;
            const unsigned pp = act ? pos[(size_t)f * W * H + p0 + lane] : 0u;
            const int x = (int)(pp & 0xffffu), y = (int)(pp >> 16);
            const float txf = (float)x * inv_tw - 0.5f, tyf = (float)y * inv_th - 0.5f;
            const float xa = txf - floorf(txf), xa1 = 1.0f - xa, ya = tyf - floorf(tyf), ya1 = 1.0f - ya;
            unsigned long long hm = heads;
            unsigned oprev = 0;
            int cj = 0;
            while (hm) {
                const int c = __ffsll((long long)hm) - 1;
                hm &= hm - 1;
                const unsigned pk = (unsigned)__builtin_amdgcn_readlane((int)tc, c);     // wave-uniform tuple
                const unsigned o = eval_px(pk, xa1, xa, ya1, ya);
                my_evals += act ? 1 : 0;
                if (MODE == 0) {
                    if (act) atomicAdd(&s_priv[wave][lane], o & 1u);                        // one private word per lane
                } else if (MODE == 1) {
                    if (act) {
                        atomicAdd(&s_rows[c * 256 + o], 1u);                                // +1 where the tuple starts to hold
                        if (c != 0) atomicAdd(&s_rows[c * 256 + oprev], 0xffffffffu);       // -1 for what held before
                    }
                } else {
                    // private bins over the span of the tuple's bytes
                    const unsigned b0 = pk & 255u, b1 = (pk >> 8) & 255u, b2 = (pk >> 16) & 255u, b3 = pk >> 24;
                    const unsigned lo = min(min(b0, b1), min(b2, b3)), hi = max(max(b0, b1), max(b2, b3));
                    const unsigned span = min(hi - lo, 32u);
                    if (act) atomicAdd(&s_priv[wave][min(o - lo, 32u) * 64 + lane], 1u);
                    // reduce the bins of this (chunk, tuple) -- a real kernel would do it once per group; per chunk is the upper bound
                    if (p0 + 64 >= s1) {
                        for (unsigned k = 0; k <= span; ++k) {
                            unsigned t = s_priv[wave][k * 64 + lane];
                            s_priv[wave][k * 64 + lane] = 0;
                            for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d, 64);
                            if (lane == 0 && t) { atomicAdd(&s_rows[c * 256 + lo + k], t); if (cj) atomicAdd(&s_rows[cj * 256 + lo + k], 0u - t); }
                        }
                    }
                }
                oprev = o;
                cj = c;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NCL * 256; i += 256) if (s_rows[i]) atomicAdd(&out[i], s_rows[i]);
    if (MODE == 0 && lane == 0) atomicAdd(&out[0], s_priv[wave][0]);
    for (int d = 32; d >= 1; d >>= 1) my_evals += __shfl_xor((long long)my_evals, d, 64);
    if (lane == 0) atomicAdd(nevals, my_evals);
}

int main()
{
    const int F = 16;
    std::vector<uint8_t> hv((size_t)F * W * H);
    unsigned rs = 12345;
    for (int f = 0; f < F; ++f)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                rs = rs * 1664525u + 1013904223u;
                const float s = 128.f + 70.f * std::sin((x + 13 * f) / 97.f) * std::cos(y / 71.f) + 25.f * std::sin(x / 11.f + y / 7.f) + (float)((rs >> 24) % 9) - 4.f;
                hv[(size_t)f * W * H + (size_t)y * W + x] = (uint8_t)std::fmin(255.f, std::fmax(0.f, s));
            }
    // synthetic tuples: tuple changes between consecutive limits with probability ~0.6 while c < 20, never after
    std::vector<unsigned> ht((size_t)NCELL * 256 * NCL);
    for (int cell = 0; cell < NCELL; ++cell)
        for (int v = 0; v < 256; ++v) {
            unsigned pk = 0;
            for (int c = 0; c < NCL; ++c) {
                rs = rs * 1664525u + 1013904223u;
                if (c == 0 || (c < 20 && (rs >> 24) % 10 < 6)) {
                    const int base = std::min(255, std::max(0, (int)(v * (0.6 + 0.02 * c)) + 20));
                    auto b = [&](int k) { rs = rs * 1664525u + 1013904223u; return (unsigned)std::min(255, std::max(0, base + (int)((rs >> 24) % 13) - 6 + 2 * k)); };
                    pk = b(0) | (b(1) << 8) | (b(2) << 16) | (b(3) << 24);
                }
                ht[((size_t)cell * 256 + v) * NCL + c] = pk;
            }
        }
    uint8_t *dv; unsigned *dhist, *dstart, *dcur, *dpos, *dt, *dout; unsigned long long *dne;
    CK(hipMalloc(&dv, hv.size())); CK(hipMemcpy(dv, hv.data(), hv.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&dhist, sizeof(unsigned) * F * NCELL * 256)); CK(hipMalloc(&dstart, sizeof(unsigned) * F * NCELL * 257));
    CK(hipMalloc(&dcur, sizeof(unsigned) * F * NCELL * 256)); CK(hipMalloc(&dpos, sizeof(unsigned) * (size_t)F * W * H));
    CK(hipMalloc(&dt, sizeof(unsigned) * ht.size())); CK(hipMemcpy(dt, ht.data(), sizeof(unsigned) * ht.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, sizeof(unsigned) * NCL * 256)); CK(hipMalloc(&dne, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto fn, int reps) {
        fn();
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) fn();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-44s %8.3f ms per %d frames = %7.3f ms per 64 frames\n", name, ms / reps, F, ms / reps * 64.0 / F);
        return ms / reps;
    };
    const int chunk = 16384;
    timeit("sort: histogram + scan + scatter (2x2 grid)", [&] {
        CK(hipMemsetAsync(dhist, 0, sizeof(unsigned) * F * NCELL * 256));
        k_count<<<dim3(32, NCELL, F), 256>>>(dv, dhist);
        k_scan<<<(F * NCELL + 63) / 64, 64>>>(dhist, dstart, dcur, F);
        k_scatter<<<dim3((960 * 540 + chunk - 1) / chunk, NCELL, F), 256>>>(dv, dcur, dpos, chunk);
    }, 5);
    unsigned long long ne = 0;
    auto run_eval = [&](int mode, const char *name) {
        CK(hipMemset(dne, 0, 8));
        const float ms = timeit(name, [&] {
            CK(hipMemsetAsync(dout, 0, sizeof(unsigned) * NCL * 256));
            if (mode == 0) k_eval_sorted<0><<<dim3(64, NCELL, F), 256>>>(dstart, dpos, dt, dout, dne);
            if (mode == 1) k_eval_sorted<1><<<dim3(64, NCELL, F), 256>>>(dstart, dpos, dt, dout, dne);
            if (mode == 2) k_eval_sorted<2><<<dim3(64, NCELL, F), 256>>>(dstart, dpos, dt, dout, dne);
        }, 5);
        CK(hipMemcpy(&ne, dne, 8, hipMemcpyDeviceToHost));
        const double evals = (double)ne / 6.0;                         // 1 warm-up + 5 timed runs accumulate
        const double wave_evals = evals / 64.0;
        std::printf("    %.1f M evaluations per run = %.2f per pixel; %.1f ns per 64-pixel evaluation per SIMD (1024 SIMDs)\n",
                    evals / 1e6, evals / ((double)F * W * H), ms * 1e6 / (wave_evals / 1024.0));
    };
    run_eval(0, "eval 0: blend + a private counter");
    run_eval(1, "eval A: LDS atomics into shared rows (+1/-1)");
    run_eval(2, "eval B: per-lane private bins, reduced per tuple");
    std::printf("for comparison: k_clahe_sweep evaluates 121 limits per pixel over five grids at 26 - 36 ns per evaluated limit and\n"
                "SIMD (DESIGN.md section 5); its 2 x 2 launch takes ~0.68 ms per 32 frames = 1.36 ms per 64 frames on the bench stream\n");
    return 0;
}
