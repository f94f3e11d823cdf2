// LDS throughput with data-dependent addresses (256-entry tables indexed by a random byte per lane):
// ds_read_b32 @ 4v, ds_read_b64 @ 8v, ds_read_b128 @ 16v, ds_add_u32 @ 4o with o drawn from `spread` distinct values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, const unsigned *in, int iters)
{
    extern __shared__ unsigned lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) lds[i] = 0;
    unsigned v = in[blockIdx.x * 256 + tid] & 255u;
    unsigned a32 = v * 4, a64 = v * 8, a128 = v * 16;
    unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    unsigned long long q0 = 0, q1 = 0;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP8(asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:1024\n ds_read_b32 %2, %4 offset:2048\n ds_read_b32 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a32));) }
        if (OP == 1) { REP8(asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:2048\n ds_read_b64 %0, %2 offset:4096\n ds_read_b64 %1, %2 offset:6144\n s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1) : "v"(a64));) }
        if (OP == 2) { REP8(asm volatile("ds_add_u32 %0, %1\n ds_add_u32 %0, %1 offset:1024\n ds_add_u32 %0, %1 offset:2048\n ds_add_u32 %0, %1 offset:3072\n s_waitcnt lgkmcnt(0)" :: "v"(a32), "v"(v));) }
    }
    out[blockIdx.x * 256 + tid] = r0 + r1 + r2 + r3 + (unsigned)q0 + (unsigned)q1 + lds[tid];
}
template <int OP>
void run(const char *name, int spread, int wps)
{
    const int blocks = 256 * wps, iters = 200;
    unsigned *out, *in;
    (void)hipMalloc(&out, blocks * 256 * 4); (void)hipMalloc(&in, blocks * 256 * 4);
    std::vector<unsigned> h(blocks * 256);
    for (auto &x : h) x = 100 + (rand() % spread);
    (void)hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<blocks, 256, 32768>>>(out, in, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<OP><<<blocks, 256, 32768>>>(out, in, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 32 * wps * 4;       // wave-instructions per CU
    printf("%-14s spread=%3d waves/SIMD=%d  %.3f ns per wave-instr per CU\n", name, spread, wps, ms * 1e6 / n);
    (void)hipFree(out); (void)hipFree(in);
}
int main()
{
    for (int sp : {1, 4, 8, 16, 32, 64, 156})
        { run<0>("ds_read_b32", sp, 4); run<1>("ds_read_b64", sp, 4); run<2>("ds_add_u32", sp, 4); }
    return 0;
}
