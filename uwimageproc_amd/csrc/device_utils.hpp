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
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
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

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
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
