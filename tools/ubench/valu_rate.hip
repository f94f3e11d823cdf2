// Throughput of single VALU / LDS instructions on gfx950: cycles per wave-instruction per SIMD at W waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o gpurun_out/valu_rate && gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters, long long *cyc)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    float a0 = in[tid], a1 = in[tid + 1], a2 = in[tid + 2], a3 = in[tid + 3], a4 = in[tid + 4], a5 = in[tid + 5], a6 = in[tid + 6], a7 = in[tid + 7];
    float b = in[tid + 8], c = in[tid + 9];
    unsigned ua = __float_as_uint(in[tid + 10]);
    unsigned addr = (unsigned)((OP == 21) ? 0 : (OP == 22 ? (tid & 63) * 4 * 64 : tid * 4)) & 0x3fff;   // 21: same word, 22: same bank, else linear
    for (int i = tid; i < 4096; i += 256) lds[i] = 0.f;
    __syncthreads();
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
        if (OP == 1) { REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 2) { REP8(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b));) }
        if (OP == 3) { REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b));) }
        if (OP == 4) { REP8(asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n v_cvt_f32_ubyte0 %4, %8\n v_cvt_f32_ubyte1 %5, %8\n v_cvt_f32_ubyte2 %6, %8\n v_cvt_f32_ubyte3 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ua));) }
        if (OP == 5) { REP8(asm volatile("v_fma_mix_f32 %0, %8, %9, 0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %9, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %8, %9, 0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %9, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %4, %8, %9, 0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %9, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %6, %8, %9, 0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %9, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ua), "v"(b));) }
        if (OP == 6) { REP8(asm volatile("v_cvt_pk_u8_f32 %0, %8, 0, %0\n v_cvt_pk_u8_f32 %1, %8, 0, %1\n v_cvt_pk_u8_f32 %2, %8, 0, %2\n v_cvt_pk_u8_f32 %3, %8, 0, %3\n v_cvt_pk_u8_f32 %4, %8, 0, %4\n v_cvt_pk_u8_f32 %5, %8, 0, %5\n v_cvt_pk_u8_f32 %6, %8, 0, %6\n v_cvt_pk_u8_f32 %7, %8, 0, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 7) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 2, %8\n v_lshl_add_u32 %1, %1, 2, %8\n v_lshl_add_u32 %2, %2, 2, %8\n v_lshl_add_u32 %3, %3, 2, %8\n v_lshl_add_u32 %4, %4, 2, %8\n v_lshl_add_u32 %5, %5, 2, %8\n v_lshl_add_u32 %6, %6, 2, %8\n v_lshl_add_u32 %7, %7, 2, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ua));) }
        if (OP == 8) { REP8(asm volatile("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ua), "v"(b));) }
        if (OP == 9) { REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        if (OP == 10) { REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ua));) }
        if (OP == 11) { REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b));) }
        if (OP == 12) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ua), "v"(b));) }
        if (OP == 13) { REP8(asm volatile("v_cvt_f16_u16_sdwa %0, %8 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\n v_cvt_f16_u16_sdwa %1, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\n v_cvt_f16_u16_sdwa %2, %8 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n v_cvt_f16_u16_sdwa %3, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_cvt_f16_u16_sdwa %4, %8 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0\n v_cvt_f16_u16_sdwa %5, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\n v_cvt_f16_u16_sdwa %6, %8 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2\n v_cvt_f16_u16_sdwa %7, %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ua));) }
        if (OP == 30) { REP8(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b));) }
        if (OP == 31) { REP8(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b));) }
        if (OP == 32) { REP8(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b));) }
        if (OP == 33) { REP8(asm volatile("v_rcp_f64 %0, %4\n v_rcp_f64 %1, %4\n v_rcp_f64 %2, %4\n v_rcp_f64 %3, %4\n v_rcp_f64 %0, %4\n v_rcp_f64 %1, %4\n v_rcp_f64 %2, %4\n v_rcp_f64 %3, %4" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b));) }
        if (OP == 34) { REP8(asm volatile("v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %4\n v_cvt_f64_u32 %2, %4\n v_cvt_f64_u32 %3, %4\n v_cvt_f64_u32 %0, %4\n v_cvt_f64_u32 %1, %4\n v_cvt_f64_u32 %2, %4\n v_cvt_f64_u32 %3, %4" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(ua));) }
        if (OP == 35) { REP8(asm volatile("v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %1, %8 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %2, %8 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %3, %8 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %4, %8 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_mov_b32_dpp %5, %8 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_mov_b32_dpp %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %7, %8 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ua));) }
        if (OP == 36) { REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ua));) }
        if (OP == 37) { REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ua));) }
        if (OP == 38) { REP8(asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ua));) }
        if (OP == 39) { REP8(asm volatile("v_add_u32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_u32_dpp %1, %8, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_u32_dpp %2, %8, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_u32_dpp %3, %8, %3 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_u32_dpp %4, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_u32_dpp %5, %8, %5 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_u32_dpp %6, %8, %6 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_u32_dpp %7, %8, %7 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ua));) }
        // round 5: the integer-weight form of the sweep's evaluation (priced in DESIGN.md section 5)
        if (OP == 40) { REP8(asm volatile("v_dot4_u32_u8 %0, %8, %9, %0\n v_dot4_u32_u8 %1, %8, %9, %1\n v_dot4_u32_u8 %2, %8, %9, %2\n v_dot4_u32_u8 %3, %8, %9, %3\n v_dot4_u32_u8 %4, %8, %9, %4\n v_dot4_u32_u8 %5, %8, %9, %5\n v_dot4_u32_u8 %6, %8, %9, %6\n v_dot4_u32_u8 %7, %8, %9, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(ua), "v"(b));) }
        if (OP == 41) { REP8(asm volatile("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (OP == 42) { REP8(asm volatile("v_cmp_eq_u32 vcc, %0, %8\n v_cmp_eq_u32 vcc, %1, %8\n v_cmp_eq_u32 vcc, %2, %8\n v_cmp_eq_u32 vcc, %3, %8\n v_cmp_eq_u32 vcc, %4, %8\n v_cmp_eq_u32 vcc, %5, %8\n v_cmp_eq_u32 vcc, %6, %8\n v_cmp_eq_u32 vcc, %7, %8" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(ua) : "vcc");) }
        if (OP == 43) { REP8(asm volatile("v_cvt_u32_f32 %0, %8\n v_cvt_u32_f32 %1, %8\n v_cvt_u32_f32 %2, %8\n v_cvt_u32_f32 %3, %8\n v_cvt_u32_f32 %4, %8\n v_cvt_u32_f32 %5, %8\n v_cvt_u32_f32 %6, %8\n v_cvt_u32_f32 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
        // LDS: 20 ds_read_b32, 23 ds_read_b64, 24 ds_read_b128, 21/22/25 ds_add_u32 (same word / same bank / linear)
        if (OP == 20) { REP8(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(addr));) }
        if (OP == 23) { REP8(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:2048\n ds_read_b64 %2, %4 offset:4096\n ds_read_b64 %3, %4 offset:6144\n ds_read_b64 %0, %4 offset:8192\n ds_read_b64 %1, %4 offset:10240\n ds_read_b64 %2, %4 offset:12288\n ds_read_b64 %3, %4 offset:14336\n s_waitcnt lgkmcnt(0)" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(addr * 2));) }
        if (OP == 21 || OP == 22 || OP == 25) { REP8(asm volatile("ds_add_u32 %0, %1\n ds_add_u32 %0, %1 offset:4\n ds_add_u32 %0, %1 offset:8\n ds_add_u32 %0, %1 offset:12\n ds_add_u32 %0, %1 offset:16\n ds_add_u32 %0, %1 offset:20\n ds_add_u32 %0, %1 offset:24\n ds_add_u32 %0, %1 offset:28\n s_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(ua));) }
    }
    long long t1 = clock64();
    out[blockIdx.x * 256 + tid] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + lds[tid];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char *name, int waves_per_simd)
{
    // one block of 256 threads = 4 waves = 1 per SIMD; `waves_per_simd` blocks per CU
    const int blocks = 256 * waves_per_simd, iters = 200;
    float *out, *in; long long *cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&in, 4096 * 4); hipMalloc(&cyc, blocks * 8);
    std::vector<float> h(4096, 1.0001f);
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256, 16384 * 2>>>(out, in, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256, 16384 * 2>>>(out, in, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> hc(blocks);
    hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto c : hc) mean += c; mean /= blocks;
    const double n_inst = (double)iters * 64;          // per wave
    // clock64 = s_memtime-like counter (100 MHz?) -> use wall time instead: instr per SIMD = n_inst * waves_per_simd
    const double ns_per_inst_per_simd = ms * 1e6 / (n_inst * waves_per_simd);
    printf("%-22s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)  clock64 delta/instr %.2f\n", name, waves_per_simd, ms,
           ns_per_inst_per_simd, ns_per_inst_per_simd * 2.4, mean / n_inst);
    hipFree(out); hipFree(in); hipFree(cyc);
}

int main()
{
    for (int w : {1, 2}) {
        run<30>("v_fma_f64", w); run<31>("v_add_f64", w); run<32>("v_mul_f64", w); run<33>("v_rcp_f64", w); run<34>("v_cvt_f64_u32", w);
        run<35>("v_mov_b32_dpp", w); run<39>("v_add_u32_dpp", w); run<36>("v_add_u32", w); run<37>("v_mov_b32", w); run<38>("v_mul_u32_u24", w);
        printf("\n");
    }
    if (getenv("UB_INT"))
    for (int w : {2, 4, 6}) {
        run<40>("v_dot4_u32_u8", w); run<41>("v_lshrrev_b32", w); run<42>("v_cmp_eq_u32", w); run<43>("v_cvt_u32_f32", w); run<7>("v_lshl_add_u32", w);
        run<10>("v_and_b32", w); run<1>("v_mul_f32", w); run<4>("v_cvt_f32_ubyteN", w); run<8>("v_perm_b32", w);
        printf("\n");
    }
    if (getenv("UB_ALL"))
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w); run<1>("v_mul_f32", w); run<9>("v_add_f32", w); run<2>("v_pk_mul_f32", w); run<3>("v_pk_add_f32", w);
        run<11>("v_pk_fma_f32", w); run<4>("v_cvt_f32_ubyteN", w); run<5>("v_fma_mix_f32", w); run<6>("v_cvt_pk_u8_f32", w);
        run<7>("v_lshl_add_u32", w); run<8>("v_perm_b32", w); run<10>("v_and_b32", w); run<12>("v_mad_u32_u24", w); run<13>("v_cvt_f16_u16_sdwa", w);
        run<20>("ds_read_b32", w); run<23>("ds_read_b64", w); run<25>("ds_add_u32 linear", w); run<22>("ds_add_u32 same bank", w); run<21>("ds_add_u32 same word", w);
        printf("\n");
    }
    return 0;
}
