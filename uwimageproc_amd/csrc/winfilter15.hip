// 15x15 window max and min of an interleaved 3-channel u8 frame (bgdehaze D1/D2: BGDehaze.py:14-37 take the
// per-channel maximum / minimum over the w = 15 neighbourhood of every pixel), both in ONE streaming pass.
//
// One wave owns 256 image columns (4 adjacent pixels per lane: 12 bytes = 3 aligned dwords per row) and walks
// down a chunk of rows.  Samples are held as u16 pairs so that every v_pk_max_u16 handles two of them; the minimum
// is the maximum of the complemented bytes, so both filters run the same code and out-of-image samples are simply 0.
//   horizontal: window doubling 1 -> 2 -> 4 -> 8 -> 15 pixels; the neighbours come from the next lanes by DPP
//               (wave_shl:1) and v_alignbit, no LDS;
//   vertical:   the same doubling over rows, with the 2 / 4 / 7 rows of history of each level kept byte-packed in
//               registers.
// Rows are loaded four at a time, one group ahead of the arithmetic.  Output: planar [F][3][H][W] u8.
// Only w = 15 and 4-byte aligned rows with W % 4 == 0 take this kernel (dehaze.hip keeps the general one).
#include "uwip_internal.hpp"
#include <algorithm>

namespace {

constexpr int WF15_TS = 240;   // output columns per wave (256 - 15 - 1, a multiple of 4)

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pkmax(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
// lane i <- lane i + 1 (0 into lane 63)
__device__ __forceinline__ uint32_t shl1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }
// (x.byte[xb], y.byte[yb]) as a u16 pair
__device__ __forceinline__ uint32_t pick2(uint32_t x, int xb, uint32_t y, int yb)
{
    return __builtin_amdgcn_perm(y, x, 0x0c000c00u | ((uint32_t)(4 + yb) << 16) | (uint32_t)xb);
}
// u16 pairs (a0,a1), (b0,b1) -> bytes a0 a1 b0 b1
__device__ __forceinline__ uint32_t pack4(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x06040200u); }
__device__ __forceinline__ uint32_t lo2(uint32_t p) { return __builtin_amdgcn_perm(p, p, 0x0c010c00u); }   // bytes 0,1 -> pair
__device__ __forceinline__ uint32_t hi2(uint32_t p) { return __builtin_amdgcn_perm(p, p, 0x0c030c02u); }   // bytes 2,3 -> pair

struct U6 {
    uint32_t v[6];   // [channel][pair]: (p0,p1), (p2,p3) of the lane's four pixels
};

// r[x] = max(s[x+1 .. x+15]) along the wave's 256 positions (valid for x <= 240)
__device__ __forceinline__ U6 hmax15(const U6 &s)
{
    U6 r;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t A = s.v[2 * c], B = s.v[2 * c + 1];
        const uint32_t nA = shl1(A);
        const uint32_t m2A = pkmax(A, __builtin_amdgcn_alignbit(B, A, 16)), m2B = pkmax(B, __builtin_amdgcn_alignbit(nA, B, 16));
        const uint32_t m4A = pkmax(m2A, m2B), m4B = pkmax(m2B, shl1(m2A));
        const uint32_t m8A = pkmax(m4A, shl1(m4A)), m8B = pkmax(m4B, shl1(m4B));
        const uint32_t nm8A = shl1(m8A), nnm8A = shl1(nm8A), nnm8B = shl1(shl1(m8B));
        r.v[2 * c] = pkmax(__builtin_amdgcn_alignbit(m8B, m8A, 16), nnm8A);
        r.v[2 * c + 1] = pkmax(__builtin_amdgcn_alignbit(nm8A, m8B, 16), nnm8B);
    }
    return r;
}

struct P3 {
    uint32_t v[3];   // byte-packed: one dword per channel
};
__device__ __forceinline__ P3 pack(const U6 &u)
{
    P3 p;
#pragma unroll
    for (int c = 0; c < 3; ++c) p.v[c] = pack4(u.v[2 * c], u.v[2 * c + 1]);
    return p;
}
__device__ __forceinline__ U6 unpack(const P3 &p)
{
    U6 u;
#pragma unroll
    for (int c = 0; c < 3; ++c) { u.v[2 * c] = lo2(p.v[c]); u.v[2 * c + 1] = hi2(p.v[c]); }
    return u;
}
__device__ __forceinline__ U6 max6(const U6 &a, const U6 &b)
{
    U6 r;
#pragma unroll
    for (int k = 0; k < 6; ++k) r.v[k] = pkmax(a.v[k], b.v[k]);
    return r;
}

// running maximum over the last 15 rows
struct VState {
    U6 hp;
    P3 r2[2], r4[4], r8[7];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int k = 0; k < 6; ++k) hp.v[k] = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            r2[0].v[c] = r2[1].v[c] = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) r4[i].v[c] = 0;
#pragma unroll
            for (int i = 0; i < 7; ++i) r8[i].v[c] = 0;
        }
    }
    // feeds row y, returns max over rows y-14 .. y (planar bytes per channel)
    __device__ __forceinline__ P3 push(const U6 &h)
    {
        const U6 v2 = max6(h, hp);
        hp = h;
        const U6 v4 = max6(v2, unpack(r2[1]));
        r2[1] = r2[0]; r2[0] = pack(v2);
        const U6 v8 = max6(v4, unpack(r4[3]));
#pragma unroll
        for (int i = 3; i > 0; --i) r4[i] = r4[i - 1];
        r4[0] = pack(v4);
        const U6 v15 = max6(v8, unpack(r8[6]));
#pragma unroll
        for (int i = 6; i > 0; --i) r8[i] = r8[i - 1];
        r8[0] = pack(v8);
        return pack(v15);
    }
};

// stats (optional, needs both filters): stats[f * stats_stride + {0,1,2,3}] = min, max over all channels and min, max of
// the red channel (bgdehaze D0), by atomicMin / atomicMax -- the caller initialises them to 255 / 0.
template <bool DO_MAX, bool DO_MIN>
__global__ __launch_bounds__(64) void k_winfilter15(const uint8_t *__restrict__ img, size_t step, size_t fs, int H, int W,
                                                    uint8_t *__restrict__ out_max, uint8_t *__restrict__ out_min, int rpc,
                                                    int *__restrict__ stats, int stats_stride,
                                                    unsigned long long *__restrict__ redsum /*[F] or null: exact sum of channel 2*/)
{
    constexpr bool BOTH = DO_MAX && DO_MIN;
    uint32_t st_max = 0, st_imax = 0, st_rmax = 0, st_rimax = 0, st_rsum = 0;
    const int l = threadIdx.x, f = blockIdx.z;
    const int col0 = (int)blockIdx.x * WF15_TS - 8 + 4 * l;      // image column of the lane's first sample
    const bool in = col0 >= 0 && col0 < W;                         // W % 4 == 0: all four or none
    const int X0 = (int)blockIdx.x * WF15_TS + 4 * l;             // image column of the lane's first output
    const bool outl = l < WF15_TS / 4 && X0 < W;
    const uint32_t xorv = in ? 0x00ff00ffu : 0u;
    const uint8_t *src = img + (size_t)f * fs + (size_t)(in ? col0 : 0) * 3;
    const size_t n = (size_t)H * W;
    const int y0 = blockIdx.y * rpc, y1 = min(H, y0 + rpc);

    VState smax, smin;
    if (DO_MAX) smax.clear();
    if (DO_MIN) smin.clear();

    struct Raw { uint32_t d[3]; };
    auto load = [&](int y, Raw &r) {
        r.d[0] = r.d[1] = r.d[2] = 0u;
        if (y >= 0 && y < H && in) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(src + (size_t)y * step);
            r.d[0] = q[0]; r.d[1] = q[1]; r.d[2] = q[2];
        }
    };
    auto process = [&](const Raw &r, int y) {
        const int Y = y - 7;                  // the output row completed by input row y
        const bool rowin = y >= 0 && y < H;   // uniform
        U6 u;
        // byte 3j + c of the 12 is pixel j, channel c
        u.v[0] = pick2(r.d[0], 0, r.d[0], 3); u.v[1] = pick2(r.d[1], 2, r.d[2], 1);
        u.v[2] = pick2(r.d[0], 1, r.d[1], 0); u.v[3] = pick2(r.d[1], 3, r.d[2], 2);
        u.v[4] = pick2(r.d[0], 2, r.d[1], 1); u.v[5] = pick2(r.d[2], 0, r.d[2], 3);
        if (DO_MAX) {
            const P3 o = smax.push(hmax15(u));
            if (Y >= y0 && outl) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    *reinterpret_cast<uint32_t *>(out_max + ((size_t)f * 3 + c) * n + (size_t)Y * W + X0) = o.v[c];
            }
        }
        if (DO_MIN) {
            U6 ui;
            const uint32_t xr = rowin ? xorv : 0u;
#pragma unroll
            for (int k = 0; k < 6; ++k) ui.v[k] = u.v[k] ^ xr;     // 255 - v inside the image, 0 outside
            if (BOTH) {
                // every pixel belongs to exactly one (strip, row chunk): lanes 2..61 hold the strip's own 240 columns
                if (y >= y0 && y < y1 && l >= 2 && l < 62)
                    st_rsum += (u.v[4] & 0xffffu) + (u.v[4] >> 16) + (u.v[5] & 0xffffu) + (u.v[5] >> 16);
                const uint32_t r = pkmax(u.v[4], u.v[5]), ri = pkmax(ui.v[4], ui.v[5]);
                st_rmax = pkmax(st_rmax, r); st_rimax = pkmax(st_rimax, ri);
                st_max = pkmax(st_max, pkmax(pkmax(pkmax(u.v[0], u.v[1]), pkmax(u.v[2], u.v[3])), r));
                st_imax = pkmax(st_imax, pkmax(pkmax(pkmax(ui.v[0], ui.v[1]), pkmax(ui.v[2], ui.v[3])), ri));
            }
            const P3 o = smin.push(hmax15(ui));
            if (Y >= y0 && outl) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    *reinterpret_cast<uint32_t *>(out_min + ((size_t)f * 3 + c) * n + (size_t)Y * W + X0) = ~o.v[c];
            }
        }
    };

    // input rows y0 - 7 .. y1 - 1 + 7, four at a time, the next group in flight while this one is processed
    const int ybeg = y0 - 7, yend = y1 + 7;
    Raw cur[4], nxt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) load(ybeg + k < yend ? ybeg + k : -1, cur[k]);
    for (int y = ybeg; y < yend; y += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) load(y + 4 + k < yend ? y + 4 + k : -1, nxt[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (y + k < yend) process(cur[k], y + k);
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
    }
    if (BOTH && stats) {
        int a = (int)max(st_max & 0xffffu, st_max >> 16), b = (int)max(st_imax & 0xffffu, st_imax >> 16);
        int c = (int)max(st_rmax & 0xffffu, st_rmax >> 16), d = (int)max(st_rimax & 0xffffu, st_rimax >> 16);
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            a = max(a, __shfl_xor(a, sft, 64)); b = max(b, __shfl_xor(b, sft, 64));
            c = max(c, __shfl_xor(c, sft, 64)); d = max(d, __shfl_xor(d, sft, 64));
        }
        if (l == 0) {
            int *sp = stats + (size_t)f * stats_stride;
            atomicMin(&sp[0], 255 - b); atomicMax(&sp[1], a); atomicMin(&sp[2], 255 - d); atomicMax(&sp[3], c);
        }
        if (redsum) {
            uint32_t rs = st_rsum;
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) rs += (uint32_t)__shfl_xor((int)rs, sft, 64);
            if (l == 0) atomicAdd(&redsum[f], (unsigned long long)rs);
        }
    }
}

}  // namespace

bool uwip_winfilter15_ok(const uint8_t *img, size_t step, size_t fs, int H, int W, int w)
{
    return w == 15 && W % 4 == 0 && step % 4 == 0 && fs % 4 == 0 && ((uintptr_t)img) % 4 == 0 && H >= 1 && H <= 65535 * 16;
}

// out_max / out_min: planar [F][3][H][W] (either may be null)
int uwip_winfilter15(uwip_ctx *ctx, const uint8_t *img, size_t step, size_t fs, int F, int H, int W, uint8_t *out_max,
                     uint8_t *out_min, int *stats, int stats_stride, unsigned long long *redsum)
{
    UWIP_REQUIRE(ctx, !redsum || stats, "the red sum comes with the fused statistics");
    UWIP_REQUIRE(ctx, !stats || (out_max && out_min), "the fused statistics need both filters");
    UWIP_REQUIRE(ctx, uwip_winfilter15_ok(img, step, fs, H, W, 15), "k_winfilter15: unsupported geometry");
    UWIP_REQUIRE(ctx, out_max || out_min, "no output");
    UWIP_REQUIRE(ctx, (((uintptr_t)out_max | (uintptr_t)out_min) & 3) == 0, "unaligned output planes");
    const unsigned strips = uwip_cdiv(W, WF15_TS);
    // row chunks: 14 warm-up rows each; enough waves for ~3 per SIMD on 256 CUs
    int chunks = (int)((3072 + (size_t)strips * F - 1) / ((size_t)strips * F));
    chunks = std::max(1, std::min(chunks, std::max(1, H / 56)));
    const int rpc = (H + chunks - 1) / chunks;
    const dim3 grid(strips, uwip_cdiv(H, rpc), (unsigned)F);
    uwip_kscope ks(ctx, "k_winfilter15");
    if (redsum) UWIP_HIP(ctx, hipMemsetAsync(redsum, 0, sizeof(unsigned long long) * F, ctx->stream));
    if (out_max && out_min) k_winfilter15<true, true><<<grid, 64, 0, ctx->stream>>>(img, step, fs, H, W, out_max, out_min, rpc, stats, stats_stride, redsum);
    else if (out_max) k_winfilter15<true, false><<<grid, 64, 0, ctx->stream>>>(img, step, fs, H, W, out_max, out_min, rpc, nullptr, 0, nullptr);
    else k_winfilter15<false, true><<<grid, 64, 0, ctx->stream>>>(img, step, fs, H, W, out_max, out_min, rpc, nullptr, 0, nullptr);
    UWIP_HIP(ctx, hipGetLastError());
    return UWIP_OK;
}
