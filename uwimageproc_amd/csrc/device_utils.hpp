// Device helpers shared by the kernels (gfx950: wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

// cv::saturate_cast<uchar>(float): cvRound (round-half-even) then saturate.
// x86's cvtss2si yields INT_MIN for NaN / out-of-range, which saturates to 0;
// gfx950's v_cvt_i32_f32 saturates +inf to INT_MAX instead, so handle it here.
__device__ __forceinline__ uint32_t sat_u8_rne(float v)
{
    if (!(v < 2147483648.0f)) return 0u;  // NaN, +inf, >= 2^31  -> INT_MIN -> 0
    int r = __float2int_rn(v);
    return (uint32_t)min(max(r, 0), 255);
}

// For values already known to lie in [0, 255.5): plain RNE conversion.
__device__ __forceinline__ uint32_t rne_u8_inrange(float v) { return (uint32_t)__float2int_rn(v); }

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// Inclusive scan / sum over a 256-thread block of uint32 (4 waves).
// `scratch` must hold >= 8 uint32 in LDS.  All 256 threads must call.
// Inclusive prefix over the 64 lanes of a wave by DPP (no LDS traffic, unlike __shfl_up): Hillis-Steele inside each
// row of 16 lanes (row_shr 1,2,4,8), then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3; lanes without a
// source read 0.  All 64 lanes must be active.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t uwip_dpp0_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xf, true);
}
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    v += uwip_dpp0_u32<0x111, 0xf>(v);
    v += uwip_dpp0_u32<0x112, 0xf>(v);
    v += uwip_dpp0_u32<0x114, 0xf>(v);
    v += uwip_dpp0_u32<0x118, 0xf>(v);
    v += uwip_dpp0_u32<0x142, 0xa>(v);
    v += uwip_dpp0_u32<0x143, 0xc>(v);
    return v;
}
// sum over the wave, returned in every lane
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(v), 63);
}

__device__ __forceinline__ uint32_t block256_incl_scan_u32(uint32_t v, uint32_t *scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t s = wave_incl_scan_u32(v);
    __syncthreads();                      // protect scratch reuse
    if (lane == 63) scratch[wave] = s;
    __syncthreads();
    uint32_t off = 0;
#pragma unroll
    for (int w = 0; w < 3; ++w) off += (w < wave) ? scratch[w] : 0u;
    return s + off;
}

__device__ __forceinline__ uint32_t block256_sum_u32(uint32_t v, uint32_t *scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t s = wave_sum_u32(v);
    __syncthreads();
    if (lane == 0) scratch[wave] = s;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
